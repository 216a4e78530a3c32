# round-4 GPU call 2: in-launch split-K combine + new halo forms: op tests, per-shape A/B, forward with re-tuned tiles
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "halo_resident or split_k or fused_output_statistics or igemm8_staggered" 2>&1 | tail -15 > $O/c2_tests_ops.txt
cat $O/c2_tests_ops.txt
CF=22,32,39,76,85,84,74,75,77,86,87,88,89,90,91,92,62,63,7,14,33,12,66,73,83,79,80,35,37
timeout 600 python tools/bench_shapes.py --filter conv3 --cfgs $CF --iters 20 --json $O/c2_shapes_inline.json > $O/c2_shapes_inline.txt 2>&1
LADI_SPLITK_TWO_PASS=1 timeout 600 python tools/bench_shapes.py --filter conv3 --cfgs 86,87,90,91,12,14,73,83,79,80,35,37 --iters 20 > $O/c2_shapes_twopass.txt 2>&1
export LADI_TUNE_NO_SHIPPED=1
export LADI_TUNE_CACHE=$PWD/$O/r04_tune_c2.txt
rm -f $LADI_TUNE_CACHE
timeout 600 python bench.py --roofline-only --no-cpu-baseline > $O/c2_roofline_inline.json 2> $O/c2_roofline_inline.err
LADI_SPLITK_TWO_PASS=1 timeout 600 python bench.py --roofline-only --no-cpu-baseline > $O/c2_roofline_twopass.json 2> $O/c2_roofline_twopass.err
tail -25 $O/c2_shapes_inline.txt; tail -25 $O/c2_shapes_twopass.txt
python - <<'PY'
import json
for t in ("inline","twopass"):
    try:
        d=json.loads(open("gpurun_out/c2_roofline_%s.json"%t).read().strip().split("\n")[-1])["roofline"]
        print(t, d["unet_forward_ms"], d["igemm_all_tflops"], d["kernel"], d["achieved"], d.get("clock"))
    except Exception as e: print(t, "ERR", e)
PY
wc -l $LADI_TUNE_CACHE
