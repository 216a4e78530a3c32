#!/bin/bash
# round 6, GPU call 4: ring kernel, pinned vs compiler-scheduled fragment double buffer, on the X -> X + residual projections
O=gpurun_out/r06c4; mkdir -p $O
out=$O/ring_sched.txt; : > $out
for rep in 1 2; do
  for c in 9 48; do for v in nopin pin; do timeout 60 tools/r06/bin/ring${c}_$v 3072 1280 1280 >> $out 2>&1; done; done
  for v in nopin pin; do timeout 60 tools/r06/bin/ring7_$v 12288 640 640 >> $out 2>&1; done
  for v in nopin pin; do timeout 60 tools/r06/bin/ring7_$v 12288 2560 640 >> $out 2>&1; done
  for v in nopin pin; do timeout 60 tools/r06/bin/ring9_$v 3072 5120 1280 >> $out 2>&1; done
done
cat $out
