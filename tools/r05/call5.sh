#!/bin/bash
# round 5, fifth GPU call: the four-wave one-per-SIMD halo 320x192 tile (cfg 96): op tests, then against the other level-0 forms
O=gpurun_out/r05c5; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "halo_resident or fused_output_statistics" > $O/pytest_halo.txt 2>&1; echo "halo tests rc $?" >> $O/pytest_halo.txt
grep -E "passed|failed|FAILED" $O/pytest_halo.txt | head
timeout 300 python tools/bench_shapes.py --filter "64x48 conv3" --cfgs 22,32,39,76,85,92,96 --iters 20 > $O/shapes_l0.txt 2>&1
cat $O/shapes_l0.txt | cut -c1-200
