#!/bin/bash
# round 6, GPU call 18: same-box A/B of the WHOLE bench step (not only the forward): this tree vs the round-5 library (tools/r06/base_r05), arms interleaved
O=gpurun_out/r06c18; mkdir -p $O
for rep in 1 2; do for arm in base new; do
  if [ $arm = base ]; then D=tools/r06/base_r05; else D=.; fi
  ( cd $D && timeout 900 python bench.py --no-cpu-baseline --no-roofline --steps 6 --warmup 2 2> /dev/null | tail -1 ) > $O/bench_${arm}_$rep.json
  python - $arm $rep <<'PY'
import json,sys
arm,rep=sys.argv[1],sys.argv[2]
try:
    d=json.loads(open("gpurun_out/r06c18/bench_%s_%s.json"%(arm,rep)).read().strip().splitlines()[-1])
    print(arm, rep, "images/s", d["value"], "ms_per_step", d["ms_per_step"], "stages", [round(x,2) for x in d["stage_ms_rank0"]], "with tail", d["with_d2h_pil_images_per_s"])
except Exception as e: print(arm, rep, "ERR", e)
PY
done; done
