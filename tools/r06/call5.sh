#!/bin/bash
# round 6, GPU call 5: pinned vs compiler schedule on the other halo forms of the forward
O=gpurun_out/r06c5; mkdir -p $O
out=$O/halo_pin_forms.txt; : > $out
for rep in 1 2; do
  for s in 0 1; do
    timeout 60 tools/r06/bin/h_c2212w2_s$s 64 48 320 320 >> $out 2>&1
    timeout 60 tools/r06/bin/h_c2212w2_s$s 64 48 640 320 >> $out 2>&1
    timeout 60 tools/r06/bin/h_c214w4_s$s 8 6 1280 1280 2 >> $out 2>&1
    timeout 60 tools/r06/bin/h_c214w4_s$s 8 6 1280 1280 4 >> $out 2>&1
    timeout 60 tools/r06/bin/h_c214w4_s$s 16 12 1280 1280 >> $out 2>&1
    timeout 60 tools/r06/bin/h_c512w6_s$s 64 48 320 320 >> $out 2>&1
  done
done
cat $out
