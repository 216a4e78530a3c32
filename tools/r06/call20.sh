#!/bin/bash
# round 6, GPU call 20: per-shape launch table of one CFG UNet forward (LADI_PROF_DUMP=1: HIP events around every GEMM-family launch, grouped by (P, Q, K, ksize, cfg))
O=gpurun_out/r06c20; mkdir -p $O
LADI_PROF_DUMP=1 timeout 600 python bench.py --roofline-only --no-cpu-baseline --roofline-iters 8 > /dev/null 2> $O/prof_dump.txt
grep "igemm-prof" $O/prof_dump.txt | sort -t= -k7 -n -r | head -80
