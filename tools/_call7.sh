cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
timeout 900 python -m pytest tests/test_gpu_modules.py -x -q -k "refinement or tps or warp" 2>&1 | tail -15
cat gpurun_out/parity_r03.json 2>/dev/null | head -60
