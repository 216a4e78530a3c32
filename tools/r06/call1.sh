#!/bin/bash
# round 6, GPU call 1: instruction-schedule variants of the dominant halo kernel (tools/r06/halo_sched.hip): 0 = library schedule,
# 1 = sched_barrier-pinned double buffer inside a step, 2 = software pipeline across steps (rendezvous in the middle of a step)
mkdir -p gpurun_out
out=gpurun_out/r06_halo_sched.txt
: > $out
for rep in 1 2; do
  for v in 0 1 2; do
    timeout 120 tools/r06/bin/halo_sched_$v 32 24 640 640 >> $out 2>&1
  done
done
for v in 0 2; do
  timeout 120 tools/r06/bin/halo_sched_$v 16 12 1280 1280 >> $out 2>&1
  timeout 120 tools/r06/bin/halo_sched_$v 16 12 1280 1280 2 >> $out 2>&1
  timeout 120 tools/r06/bin/halo_sched_$v 32 24 1280 640 >> $out 2>&1
  timeout 120 tools/r06/bin/halo_sched_$v 32 24 320 640 >> $out 2>&1
done
cat $out
