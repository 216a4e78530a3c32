set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=$PWD/gpurun_out; mkdir -p $O
export LADI_TUNE_NO_SHIPPED=1
export LADI_TUNE_CACHE=$O/r04_tune_c6.txt
rm -f $LADI_TUNE_CACHE
for m in 0 1 2 0 2 1; do
  LADI_EPI_STORE=$m timeout 600 python bench.py --roofline-only --no-cpu-baseline > $O/c6_roofline_$m.json 2> $O/c6_err.txt
  python - <<PY
import json
d=json.loads(open("$O/c6_roofline_$m.json").read().strip().split("\n")[-1])["roofline"]
print("epi_store=$m", d["unet_forward_ms"], d["igemm_all_tflops"], d["achieved"])
PY
done
