#!/bin/bash
# (run on the tree with tools/experiments/halo_weights_direct.patch applied: `git apply tools/experiments/halo_weights_direct.patch`, rebuild, build the harness binaries as the patch header of wd_sched.hip says)
# round 6, GPU call 29: schedule variants of the weights-direct step (0 fenced double buffer, 1 compiler, 2 sched_group_barrier pipeline) beside the ring-halo form
O=gpurun_out/r06c29; mkdir -p $O
for b in wd_s0 wd_s1 wd_s2 wd_s3 wd_s4; do [ -x tools/r06/bin/$b ] && { timeout 120 tools/r06/bin/$b 32 24 640 640; timeout 120 tools/r06/bin/$b 32 24 1920 640; }; done > $O/wd_sched.txt 2>&1
timeout 120 tools/r06/bin/halo_sched_lib 32 24 640 640 >> $O/wd_sched.txt 2>&1; timeout 120 tools/r06/bin/halo_sched_lib 32 24 1920 640 >> $O/wd_sched.txt 2>&1
cat $O/wd_sched.txt
