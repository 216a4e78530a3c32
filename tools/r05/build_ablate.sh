#!/bin/bash
# cross-compile the ablation binaries of tools/r05/halo_ablate.hip (one per mask) into tools/r05/bin/ (git-ignored, travels with gpurun)

mkdir -p tools/r05/bin
for m in ${MASKS:-0 1 2 3 4 8 16 5 12 20 7 28}; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function -DLADI_HALO_ABL=$m -x hip tools/r05/halo_ablate.hip -o tools/r05/bin/halo_abl_$m 2> tools/r05/bin/build_$m.log && echo built $m ) &
  if (( $(jobs -r | wc -l) >= 6 )); then wait -n; fi
done
wait
ls -la tools/r05/bin | head -20
