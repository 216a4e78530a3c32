cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "igemm or conv or mfma or linear or fused_output" > gpurun_out/c1_ops.log 2>&1; echo "ops rc=$?" 
tail -5 gpurun_out/c1_ops.log
timeout 600 python tools/bench_shapes.py --n 16 --iters 10 --json gpurun_out/shapes_n16.json > gpurun_out/c1_shapes.log 2>&1; echo "shapes rc=$?"
tail -3 gpurun_out/c1_shapes.log
LADI_TUNE_NO_SHIPPED=1 LADI_TUNE_CACHE=gpurun_out/tune_c1.txt LADI_PROF_DUMP=1 timeout 900 python bench.py --roofline-only --no-cpu-baseline > gpurun_out/c1_roofline.json 2> gpurun_out/c1_roofline.err; echo "roofline rc=$?"
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/c1_roofline.json').read().strip().splitlines()[-1])
    r=d['roofline']; print('unet_forward_ms', r['unet_forward_ms'], 'dominant', r['kernel'], r['achieved'])
    for k,v in r['per_symbol'].items(): print(k, v)
except Exception as e: print('parse fail', e)
PY
