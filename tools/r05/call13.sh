#!/bin/bash
# round 5: what the driver runs at round end -- smoke(), then `python bench.py` with no flags (N = 1) -- on the committed state
O=gpurun_out/r05c13; mkdir -p $O
( time timeout 600 python __graft_entry__.py smoke ) > $O/smoke.txt 2>&1; tail -5 $O/smoke.txt
( time timeout 900 python bench.py ) > $O/bench_noflags.json 2> $O/bench_noflags.err; tail -c 1500 $O/bench_noflags.json | head -c 600; echo; tail -4 $O/bench_noflags.err
