#!/bin/bash
# (run on the tree with tools/experiments/halo_weights_direct.patch applied: `git apply tools/experiments/halo_weights_direct.patch`, rebuild, build the harness binaries as the patch header of wd_sched.hip says)
# round 6, GPU call 27: the weights-direct halo kernel (cfg 110..117): parity (bit-equal to its halo twin) and the 3x3 shapes of the forward
O=gpurun_out/r06c27; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "weights_direct or halo_resident" > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
timeout 900 python tools/bench_shapes.py --n 16 --iters 20 --filter conv3 --cfgs 84,85,86,88,90,91,92,83,109,110,111,112,113,114,115,116,117 > $O/shapes_conv3.txt 2>&1
cat $O/shapes_conv3.txt | cut -c1-200
