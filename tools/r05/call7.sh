#!/bin/bash
# round 5, seventh GPU call: ablation of igemm_halo_kernel<2,2,1,3,2,24> on conv3x3 640 -> 640 @ 32x24, n = 16 (tools/r05/halo_ablate.hip)
O=gpurun_out/r05c7; mkdir -p $O
for m in 0 1 2 3 4 8 16 5 12 20 7 28 0; do timeout 60 tools/r05/bin/halo_abl_$m >> $O/halo_ablate.txt 2>&1; done
echo "# 16x12 1280 -> 1280 (no split-K here: 240 workgroups)" >> $O/halo_ablate.txt
for m in 0 3 4 8; do timeout 60 tools/r05/bin/halo_abl_$m 16 12 1280 1280 >> $O/halo_ablate.txt 2>&1; done
cat $O/halo_ablate.txt
