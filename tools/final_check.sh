# round-end check on the GPU box: full GPU suite, smoke(), default bench line (run from the repo root)
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 1700 python -m pytest tests/ -x -q -m gpu ${LADI_PYTEST_K:+-k "$LADI_PYTEST_K"} --durations=12 2>&1 | tail -22 > gpurun_out/final_pytest.txt
cat gpurun_out/final_pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee gpurun_out/final_smoke.txt
