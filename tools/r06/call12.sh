#!/bin/bash
# round 6, GPU call 12: weight-slice-major tile map (tile_map 3) of the halo kernel on the 8x6 / 16x12 convolutions, LADI_HALO_MAP3=0 (round-5 maps) vs 1,
# with the fabric traffic of both (FETCH_SIZE / WRITE_SIZE, separate passes)
O=gpurun_out/r06c12; mkdir -p $O
out=$O/halo_map3.txt; : > $out
for rep in 1 2 3; do for m in 0 1; do
  echo -n "MAP3=$m " >> $out; LADI_HALO_MAP3=$m timeout 60 tools/r06/bin/m3_8x6 8 6 1280 1280 4 >> $out 2>&1
  echo -n "MAP3=$m " >> $out; LADI_HALO_MAP3=$m timeout 60 tools/r06/bin/m3_8x6 8 6 2560 1280 4 >> $out 2>&1
  echo -n "MAP3=$m " >> $out; LADI_HALO_MAP3=$m timeout 60 tools/r06/bin/m3_16x12 16 12 1280 1280 2 >> $out 2>&1
  echo -n "MAP3=$m " >> $out; LADI_HALO_MAP3=$m timeout 60 tools/r06/bin/m3_16x12 16 12 2560 1280 2 >> $out 2>&1
  echo -n "MAP3=$m " >> $out; LADI_HALO_MAP3=$m timeout 60 tools/r06/bin/m3_16x12 16 12 640 1280 2 >> $out 2>&1
done; done
cat $out
R=$PWD; cd /tmp; export TMPDIR=/tmp
for m in 0 1; do for c in FETCH_SIZE WRITE_SIZE; do
  LADI_HALO_MAP3=$m timeout 120 rocprofv3 --pmc $c --kernel-trace -d $R/$O/pmc_${m}_$c -- $R/tools/r06/bin/m3_8x6 8 6 1280 1280 4 > /dev/null 2>&1
  python $R/tools/rocpd_pmc.py $(find $R/$O/pmc_${m}_$c -name "*.db" | head -1) $R/$O/pmc_8x6_map${m}_$c.txt > /dev/null; rm -rf $R/$O/pmc_${m}_$c
  echo "== 8x6 MAP3=$m $c"; head -2 $R/$O/pmc_8x6_map${m}_$c.txt | cut -c1-200
  LADI_HALO_MAP3=$m timeout 120 rocprofv3 --pmc $c --kernel-trace -d $R/$O/pmc_${m}_$c -- $R/tools/r06/bin/m3_16x12 16 12 1280 1280 2 > /dev/null 2>&1
  python $R/tools/rocpd_pmc.py $(find $R/$O/pmc_${m}_$c -name "*.db" | head -1) $R/$O/pmc_16x12_map${m}_$c.txt > /dev/null; rm -rf $R/$O/pmc_${m}_$c
  echo "== 16x12 MAP3=$m $c"; head -2 $R/$O/pmc_16x12_map${m}_$c.txt | cut -c1-200
done; done
