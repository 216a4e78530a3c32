#!/bin/bash
# round 6, GPU call 26: bench.py's roofline.vendor_gemm calibration field (live hipBLASLt 8192^3 fp16 beside the dominant kernel's rate)
O=gpurun_out/r06c26; mkdir -p $O
timeout 600 python bench.py --roofline-only --no-cpu-baseline > $O/roofline_only.json 2> $O/err.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06c26/roofline_only.json").read().strip().split("\n")[-1])
r = d["roofline"]
print({k: r.get(k) for k in ("achieved", "frac", "unet_forward_ms", "vendor_gemm", "clock")})
PY
timeout 600 python -m pytest tests/test_gpu_modules.py -x -q -m gpu -k "bench_line" 2>&1 | tail -3
