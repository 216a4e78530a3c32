#!/bin/bash
# round 6, GPU call 11: same-box A/B of the forward / VAE stages / attention: this tree against the round-5 library (tools/r06/base_r05), arms
# interleaved twice; pin A/B on the VAE's 2-D blocked halo forms
O=gpurun_out/r06c11; mkdir -p $O
out=$O/halo_g2d_pin.txt; : > $out
for rep in 1 2; do for p in 0 1; do
  timeout 60 tools/r06/bin/g_g128x128_p$p 256 192 128 128 1 8 >> $out 2>&1
  timeout 60 tools/r06/bin/g_g128x256_p$p 256 192 256 128 1 8 >> $out 2>&1
  timeout 60 tools/r06/bin/g_g128x128_p$p 128 96 256 256 1 8 >> $out 2>&1
done; done; cat $out
for rep in 1 2; do for arm in base new; do
  if [ $arm = base ]; then D=tools/r06/base_r05; else D=.; fi
  ( cd $D && timeout 600 python bench.py --roofline-only --no-cpu-baseline --roofline-iters 8 2> $OLDPWD/$O/err_$arm.txt | tail -1 ) > $O/roofline_${arm}_$rep.json
  ( cd $D && timeout 300 python tools/bench_vae.py 2> /dev/null | tail -3 ) > $O/vae_${arm}_$rep.txt
  ( cd $D && timeout 300 python tools/bench_attn.py 2> /dev/null | head -8 ) > $O/attn_${arm}_$rep.txt
python - $arm $rep <<'PY'
import json,sys
arm,rep=sys.argv[1],sys.argv[2]
try:
    d=json.loads(open("gpurun_out/r06c11/roofline_%s_%s.json"%(arm,rep)).read().strip().splitlines()[-1])
    r=d["roofline"]
    print(arm, rep, "unet_forward_ms", r["unet_forward_ms"], "lanes_ms", r["unet_forward_lanes_ms"], "igemm_all", r["igemm_all_tflops"], "dom", r["kernel"], r["achieved"], "clock", r["clock"].get("under_unet_forward_mhz"))
    xs={k:(v["avg_ms"],v["launches"]) for k,v in r["per_symbol"].items() if "xs_kernel" in k and ", 2, 1," in k or "halo" in k}
    print("   ", xs)
except Exception as e: print(arm, rep, "ERR", e)
PY
  cat $O/vae_${arm}_$rep.txt $O/attn_${arm}_$rep.txt
done; done
