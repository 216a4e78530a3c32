set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=$PWD/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_modules.py -x -q -k "graph_equals_eager or tryon_pipeline_tiny or lanes or graph_survives or fused_and_modular" 2>&1 | tail -4
for i in 1 2; do
timeout 300 python bench.py --roofline-only --no-cpu-baseline > $O/c11_roofline_$i.json 2> $O/c11_err.txt
LADI_UNET_SIDE_STREAM=0 timeout 300 python bench.py --roofline-only --no-cpu-baseline > $O/c11_roofline_noside_$i.json 2>> $O/c11_err.txt
done
python - <<'PY'
import json
for t in ("1","noside_1","2","noside_2"):
    d=json.loads(open("gpurun_out/c11_roofline_%s.json"%t).read().strip().split("\n")[-1])["roofline"]
    print(t, "eager", d["unet_forward_ms"], "graph(+side)", d.get("unet_forward_lanes_ms"))
PY
timeout 600 python -m pytest tests/test_gpu_e2e_golden.py -x -q -k "unet_forward_at or baseline_batch8" 2>&1 | tail -4
