#!/bin/bash
# The three stand-alone prototypes in one short GPU call (compile ~5 s each on the box, run ~5 s each; every run is bounded):
#   /usr/local/graft/bin/gpurun --timeout 240 -- 'bash tools/experiments/next/first_call.sh'
# Output: gpurun_out/next_<kernel>.txt (error against the fp32 CPU reference, us per launch, bit-equality of repeats).
mkdir -p gpurun_out
for k in xattn_q xattn_full ff_fused; do
    if timeout 120 /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/experiments/next/$k.hip -o /tmp/$k 2> gpurun_out/next_${k}_build.txt; then
        timeout 60 /tmp/$k > gpurun_out/next_$k.txt 2>&1
        echo "== $k (rc $?)"; cat gpurun_out/next_$k.txt
    else
        echo "== $k: build failed"; tail -5 gpurun_out/next_${k}_build.txt
    fi
done
