# round-end check on the GPU box: full GPU suite, smoke(), default bench line (run from the repo root)
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -4 > gpurun_out/final_pytest.txt
cat gpurun_out/final_pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee gpurun_out/final_smoke.txt
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
head -c 700 gpurun_out/final_bench.json; echo
python - <<'PY'
import json
d = json.loads(open("gpurun_out/final_bench.json").read().strip().split("\n")[-1])
print({k: d["roofline"].get(k) for k in ("kernel", "achieved", "frac", "traffic")})
PY
