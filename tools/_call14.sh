cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
timeout 300 python tools/bench_attn.py 2>&1 | grep -E "self|cross"
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "attn or attention" 2>&1 | tail -3
