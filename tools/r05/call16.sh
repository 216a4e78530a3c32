#!/bin/bash
# round 5: fragment prefetch depth of the halo kernel (all sixteen reads of a K step behind its barrier vs one sub-step ahead), ablation harness, same box
O=gpurun_out/r05c16; mkdir -p $O
for r in 1 2 3; do for v in base pfall; do echo -n "$v " >> $O/ab.txt; timeout 60 tools/r05/bin/halo_$v >> $O/ab.txt 2>&1; done; done
for v in base pfall; do echo -n "$v " >> $O/ab.txt; timeout 60 tools/r05/bin/halo_$v 16 12 1280 1280 >> $O/ab.txt 2>&1; done
cat $O/ab.txt
