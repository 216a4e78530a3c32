#!/bin/bash
# round 6, GPU call 14: folded-upsample halo forms in the library -- op tests, then the forward with the shipped table (old choices) vs a fresh measurement
O=gpurun_out/r06c14; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "conv or halo or split_k or linear or group_norm or layer_norm" > $O/pytest_ops.txt 2>&1; echo "rc $?" >> $O/pytest_ops.txt; tail -4 $O/pytest_ops.txt
timeout 600 python bench.py --roofline-only --no-cpu-baseline --roofline-iters 8 2> /dev/null | tail -1 > $O/roofline_shipped.json
export LADI_TUNE_NO_SHIPPED=1 LADI_TUNE_CACHE=$PWD/$O/tune_new.txt; rm -f $LADI_TUNE_CACHE
timeout 900 python bench.py --roofline-only --no-cpu-baseline --roofline-iters 8 2> /dev/null | tail -1 > $O/roofline_retuned.json
unset LADI_TUNE_NO_SHIPPED LADI_TUNE_CACHE
for arm in shipped retuned; do
python - $arm <<'PY'
import json,sys
arm=sys.argv[1]
d=json.loads(open("gpurun_out/r06c14/roofline_%s.json"%arm).read().strip().splitlines()[-1])
r=d["roofline"]
print(arm, "unet_forward_ms", r["unet_forward_ms"], "lanes_ms", r["unet_forward_lanes_ms"], "igemm_all", r["igemm_all_tflops"], "dom", r["kernel"], r["achieved"], "clock", r["clock"].get("under_unet_forward_mhz"))
for k,v in list(r["per_symbol"].items())[:30]: print("  %-56s %s" % (k, v))
PY
done
grep -c . $O/tune_new.txt; grep -E " 10[4-8]$" $O/tune_new.txt
