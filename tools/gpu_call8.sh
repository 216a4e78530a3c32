set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=$PWD/gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -k "fused_group_norm_affine or linear_" 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_e2e_golden.py -x -q 2>&1 | tail -6
cp gpurun_out/r04_tune.txt ladi_vton_amd/tune_gfx950.txt 2>/dev/null
for i in 1 2; do
timeout 600 python bench.py --roofline-only --no-cpu-baseline > $O/c8_roofline_$i.json 2> $O/c8_err.txt
LADI_GN_FUSE=0 timeout 600 python bench.py --roofline-only --no-cpu-baseline > $O/c8_roofline_nogn_$i.json 2>> $O/c8_err.txt
done
python - <<'PY'
import json
for t in ("1","nogn_1","2","nogn_2"):
    d=json.loads(open("gpurun_out/c8_roofline_%s.json"%t).read().strip().split("\n")[-1])["roofline"]
    print(t, d["unet_forward_ms"], d.get("unet_forward_lanes"), d.get("unet_forward_lanes_ms"), d["igemm_all_tflops"], {k:v["avg_ms"] for k,v in d["per_symbol"].items() if "xs" in k and k.endswith("2>")})
PY
