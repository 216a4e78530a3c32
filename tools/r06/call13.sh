#!/bin/bash
# round 6, GPU call 13: folded-upsample form of the halo kernel (UPS = 1) on the three Upsample2D convolutions of the UNet (n = 16): correctness against a
# CPU reference + timing.  Round-5 reference for the same layers (ring kernels, per-lane gather): 64x48 640->640 382 us, 32x24 1280->1280 379 us.
O=gpurun_out/r06c13; mkdir -p $O
out=$O/halo_ups.txt; : > $out
for rep in 1 2; do
  for f in 128x192 320x192 128x128; do timeout 60 tools/r06/bin/ups_$f 64 48 640 640 >> $out 2>&1; done
  for f in 128x192n 128x192 320x192 128x128; do timeout 60 tools/r06/bin/ups_$f 32 24 1280 1280 >> $out 2>&1; done
  for f in 128x192n 128x192; do timeout 60 tools/r06/bin/ups_$f 16 12 1280 1280 >> $out 2>&1; timeout 60 tools/r06/bin/ups_$f 16 12 1280 1280 2 >> $out 2>&1; done
done
cat $out
