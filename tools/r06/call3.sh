#!/bin/bash
# round 6, GPU call 3: the ablation of tools/r05/halo_ablate.hip repeated under the PINNED schedule (SCHED = 1): what would a loader / consumer split buy?
O=gpurun_out/r06c3; mkdir -p $O
out=$O/halo_s1_ablate.txt; : > $out
for rep in 1 2; do
  timeout 60 tools/r06/bin/halo_sched_1 32 24 640 640 >> $out 2>&1
  for a in 3 4 16 7 19; do echo -n "ABL $a: " >> $out; timeout 60 tools/r06/bin/halo_s1_abl$a 32 24 640 640 >> $out 2>&1; done
done
cat $out
