#!/bin/bash
# cross-compile the ablation binaries of tools/r05/halo_ablate.hip (one per mask) into tools/r05/bin/ (git-ignored, travels with gpurun)

mkdir -p tools/r05/bin
for m in ${MASKS:-0 1 2 3 4 8 16 5 12 20 7 28}; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function -DLADI_HALO_ABL=$m -x hip tools/r05/halo_ablate.hip -o tools/r05/bin/halo_abl_$m 2> tools/r05/bin/build_$m.log && echo built $m ) &
  if (( $(jobs -r | wc -l) >= 6 )); then wait -n; fi
done
wait
ls -la tools/r05/bin | head -20
# The A/B binaries of GPU calls 15 / 16 (profiles/r05_halo_zero_rows_ab.txt) are the same harness at mask 0 built from a tree with the experiment's patch applied:
#   git apply tools/experiments/halo_two_zero_rows.patch
#   hipcc ... -DLADI_HALO_ABL=0                          tools/r05/halo_ablate.hip -o tools/r05/bin/halo_new
#   hipcc ... -DLADI_HALO_ABL=0 -DLADI_HALO_ONE_ZERO_ROW tools/r05/halo_ablate.hip -o tools/r05/bin/halo_old     (then: git checkout ladi_vton_amd/csrc/igemm_halo.hip)
#   git apply tools/experiments/halo_prefetch_all.patch
#   hipcc ... -DLADI_HALO_ABL=0                          tools/r05/halo_ablate.hip -o tools/r05/bin/halo_base
#   hipcc ... -DLADI_HALO_ABL=0 -DLADI_HALO_PFALL=1      tools/r05/halo_ablate.hip -o tools/r05/bin/halo_pfall   (then: git checkout ...)
