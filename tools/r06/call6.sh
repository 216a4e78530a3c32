#!/bin/bash
# round 6, GPU call 6: ablation of the small-GEMM launch: igemm_kernel<2,2,2,1,64,3,2,0> (128x64 tile, 4 waves, 2 workgroups / CU)
# on the 16x12-level X -> X + residual projection (P 3072, K 1280, Q 1280: 480 workgroups, 20 K steps, 10.1 GFLOP), and on K = 5120 for the slope.
# masks: 1 no DMA in the loop | 2 no fragment reads | 4 no MFMA | 8 no wait + barrier | 16 no epilogue | 32 empty kernel | 64 no prologue DMA
O=gpurun_out/r06c6; mkdir -p $O
out=$O/ring_ablate.txt; : > $out
for rep in 1 2; do
  for m in 0 32 16 87 23 7 3 1 2 4 8 17 19; do echo -n "ABL $m  " >> $out; timeout 60 tools/r06/bin/ring9_abl$m 3072 1280 1280 | sed 's/rel-L2.*//' >> $out 2>&1; done
  for m in 0 16 1 2 3; do echo -n "ABL $m  " >> $out; timeout 60 tools/r06/bin/ring9_abl$m 3072 5120 1280 | sed 's/rel-L2.*//' >> $out 2>&1; done
done
cat $out
