#!/bin/bash
# round 5: one-pass GroupNorm with the first data round issued ahead of the statistics phase -- op tests, forward A/B
O=gpurun_out/r05c11; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "group_norm" > $O/pytest_gn.txt 2>&1; echo "gn tests rc $?" >> $O/pytest_gn.txt
grep -E "passed|failed|FAILED" $O/pytest_gn.txt | head
timeout 400 python tools/r05/forward_ab.py --modes "early:;late:LADI_GN_EARLY=0;e128:LADI_GN_PPB=128;e512:LADI_GN_PPB=512" --rounds 4 > $O/gn_ab.txt 2>&1
tail -2 $O/gn_ab.txt
