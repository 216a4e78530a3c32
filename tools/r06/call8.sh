#!/bin/bash
# round 6, GPU call 8: first full-library run of the round -- op + module tests, then the CFG UNet forward (roofline-only) and VAE stages with
# the A/B switches of this round's changes
O=gpurun_out/r06c8; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_modules.py -q -m gpu -x > $O/pytest_ops_modules.txt 2>&1; echo "rc $?" >> $O/pytest_ops_modules.txt
tail -5 $O/pytest_ops_modules.txt
timeout 600 python bench.py --roofline-only --no-cpu-baseline --roofline-iters 8 > $O/roofline_only.json 2> $O/roofline_only.err; tail -c 600 $O/roofline_only.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06c8/roofline_only.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("unet_forward_ms", r["unet_forward_ms"], "lanes_ms", r["unet_forward_lanes_ms"], "igemm_all", r["igemm_all_tflops"], "dom", r["kernel"], r["achieved"], "clock", r["clock"])
for k,v in list(r["per_symbol"].items())[:24]: print("  %-52s %s" % (k, v))
PY
timeout 300 python tools/bench_vae.py > $O/vae_stages.txt 2>&1; tail -12 $O/vae_stages.txt
LADI_GN_REDUCE=0 timeout 300 python tools/bench_vae.py > $O/vae_stages_noreduce.txt 2>&1; tail -12 $O/vae_stages_noreduce.txt
