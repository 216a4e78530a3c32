#!/bin/bash
# round 6, GPU call 30: the default bench line once more on the final tree (bench.py now carries roofline.vendor_gemm; library digest unchanged)
O=gpurun_out/r06c30; mkdir -p $O
timeout 1500 python bench.py --steps 10 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err
tail -c 1500 $O/bench_default.json | head -c 400; echo; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06c30/bench_default.json").read().strip().split("\n")[-1])
r = d["roofline"]
print(d["value"], d["ms_per_step"], d["stage_ms_rank0"], {k: r.get(k) for k in ("achieved", "frac", "unet_forward_ms", "vendor_gemm", "clock", "traffic")})
print(d["cpu_baseline"])
PY
