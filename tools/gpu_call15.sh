set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=$PWD/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "linear_" 2>&1 | tail -4
timeout 300 python tools/bench_shapes.py --filter "64x48 lin1 320" --cfgs 23,24,25,26,27,93,94,95,22 --iters 30 2>&1 | tail -6
export LADI_TUNE_NO_SHIPPED=1
export LADI_TUNE_CACHE=$O/r04_tune_c15.txt
rm -f $LADI_TUNE_CACHE
for i in 1 2; do
timeout 400 python bench.py --roofline-only --no-cpu-baseline > $O/c15_roofline_$i.json 2> $O/c15_err.txt
done
python - <<'PY'
import json
for t in ("1","2"):
    d=json.loads(open("gpurun_out/c15_roofline_%s.json"%t).read().strip().split("\n")[-1])["roofline"]
    print(t, d["unet_forward_ms"], d.get("unet_forward_lanes_ms"), {k:(v["avg_ms"],v["launches"]) for k,v in d["per_symbol"].items() if "xs" in k})
PY
grep -c " 9[345]$" $LADI_TUNE_CACHE
