#!/bin/bash
# round 6, GPU call 7: epilogue operands (bias row, residual rows) requested in the prologue (EpiPF) vs loaded in the epilogue, ring kernel
O=gpurun_out/r06c7; mkdir -p $O
out=$O/ring_epf.txt; : > $out
for rep in 1 2 3; do
  for v in noepf epf; do echo -n "$v " >> $out; timeout 60 tools/r06/bin/ring9_$v 3072 1280 1280 >> $out 2>&1; done
  for v in noepf epf; do echo -n "$v " >> $out; timeout 60 tools/r06/bin/ring7_$v 12288 640 640 >> $out 2>&1; done
  for v in noepf epf; do echo -n "$v " >> $out; timeout 60 tools/r06/bin/ring7_$v 12288 2560 640 >> $out 2>&1; done
  for v in noepf epf; do echo -n "$v " >> $out; timeout 60 tools/r06/bin/ring9_$v 3072 5120 1280 >> $out 2>&1; done
  for v in noepf epf; do echo -n "$v " >> $out; timeout 60 tools/r06/bin/ring9_$v 768 1280 1280 >> $out 2>&1; done
done
cat $out
