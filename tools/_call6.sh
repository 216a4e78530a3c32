cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "eight_wave or mfma_layout or split_k or staggered or race_free or fused_output" 2>&1 | tail -4
timeout 600 python tools/bench_shapes.py --n 16 --iters 10 --json gpurun_out/shapes_n16_lc.json 2>&1 | grep -v amdgpu.ids
