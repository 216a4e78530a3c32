cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "halo or eight_wave or split_k or fused_output" 2>&1 | tail -8
timeout 600 python tools/bench_shapes.py --n 16 --iters 10 --filter conv3 --cfgs 7,14,22,32,33,37,62,65,70,74,75,76,77,78,79,80,81,82,83,84,85,86,87,12,73 2>&1 | grep -v amdgpu.ids
