#!/bin/bash
# round 6, GPU call 10: re-run of what call 9 failed (gn_reduce race fixed) + attention tests with the pipelined fragment reads, micro-benchmarks,
# pin A/B on the VAE's 2-D blocked halo forms, and the same-box A/B of the forward / VAE stages / attention against the round-5 library (tools/r06/base_r05)
O=gpurun_out/r06c10; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_full.py tests/test_gpu_modules.py -q -m gpu -x > $O/pytest_a.txt 2>&1; echo "rc $?" >> $O/pytest_a.txt; tail -4 $O/pytest_a.txt
timeout 2400 python -m pytest tests/test_gpu_e2e.py -q -m gpu -x -k "single_forward or batched_launch" > $O/pytest_b.txt 2>&1; echo "rc $?" >> $O/pytest_b.txt; tail -4 $O/pytest_b.txt
tools/r06/bin/mfma_dep > $O/mfma_dep.txt 2>&1; cat $O/mfma_dep.txt
out=$O/halo_g2d_pin.txt; : > $out
for rep in 1 2; do for p in 0 1; do
  timeout 60 tools/r06/bin/g_g128x128_p$p 256 192 128 128 1 8 >> $out 2>&1
  timeout 60 tools/r06/bin/g_g128x256_p$p 256 192 256 128 1 8 >> $out 2>&1
  timeout 60 tools/r06/bin/g_g128x128_p$p 128 96 256 256 1 8 >> $out 2>&1
done; done; cat $out
for arm in new base; do
  if [ $arm = base ]; then D=tools/r06/base_r05; else D=.; fi
  ( cd $D && timeout 600 python bench.py --roofline-only --no-cpu-baseline --roofline-iters 8 2> /dev/null | tail -1 ) > $O/roofline_$arm.json
  ( cd $D && timeout 300 python tools/bench_vae.py 2> /dev/null | tail -3 ) > $O/vae_$arm.txt
  ( cd $D && timeout 300 python tools/bench_attn.py 2> /dev/null | head -8 ) > $O/attn_$arm.txt
done
for arm in new base; do
python - $arm <<'PY'
import json,sys
arm=sys.argv[1]
d=json.loads(open("gpurun_out/r06c10/roofline_%s.json"%arm).read().strip().splitlines()[-1])
r=d["roofline"]
print(arm, "unet_forward_ms", r["unet_forward_ms"], "lanes_ms", r["unet_forward_lanes_ms"], "igemm_all", r["igemm_all_tflops"], "dom", r["kernel"], r["achieved"], "clock", r["clock"].get("under_unet_forward_mhz"))
PY
cat $O/vae_$arm.txt $O/attn_$arm.txt
done
