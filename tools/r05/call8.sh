#!/bin/bash
# round 5, eighth GPU call: halo 128x256 four-wave forms (cfg 97-99) -- op tests, then against the other forms on the 32x24 / 16x12 convolutions
O=gpurun_out/r05c8; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "halo_resident" > $O/pytest_halo.txt 2>&1; echo "halo tests rc $?" >> $O/pytest_halo.txt
grep -E "passed|failed|FAILED" $O/pytest_halo.txt | head
timeout 300 python tools/bench_shapes.py --filter "32x24 conv3" --cfgs 7,33,74,75,84,88,89,90,97,98,99 --iters 20 > $O/shapes_l1.txt 2>&1
timeout 300 python tools/bench_shapes.py --filter "16x12 conv3" --cfgs 14,37,79,86,87,88,90,91,97,98,99 --iters 20 > $O/shapes_l2.txt 2>&1
cat $O/shapes_l1.txt $O/shapes_l2.txt | cut -c1-200
