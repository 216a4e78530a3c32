#!/bin/bash
# (run on the tree with tools/experiments/halo_weights_direct.patch applied: `git apply tools/experiments/halo_weights_direct.patch`, rebuild, build the harness binaries as the patch header of wd_sched.hip says)
# round 6, GPU call 28: ablation of the weights-direct halo kernel on conv 640 -> 640 @ 32x24 (n = 16) and 1920 -> 640 (three times the K loop)
O=gpurun_out/r06c28; mkdir -p $O
for m in 0 1 2 4 8 7 15; do timeout 120 tools/r06/bin/wd_abl_$m 32 24 640 640; timeout 120 tools/r06/bin/wd_abl_$m 32 24 1920 640; done > $O/wd_ablate.txt 2>&1
cat $O/wd_ablate.txt
