cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export LADI_TUNE_NO_SHIPPED=1
export LADI_TUNE_CACHE=$O/tune_c9.txt
rm -f $LADI_TUNE_CACHE
timeout 900 python bench.py --no-cpu-baseline --steps 5 --warmup 2 > $O/c9_bench.json 2> $O/c9_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c9_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'stage', d['stage_ms_rank0'], 'tail', d['with_d2h_pil_images_per_s'])
r=d['roofline']; print('unet_forward_ms', r['unet_forward_ms'], 'dominant', r['kernel'], r['achieved'], r['frac'])
for k,v in r['per_symbol'].items(): print(' ', k, v)
PY
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt9 -- python $R/bench.py --roofline-only --no-cpu-baseline > /dev/null 2>&1
cd $R
python tools/rocpd_stats.py $(find $O/kt9 -name "*.db" | head -1) $O/c9_unet_forward_kernel_stats.txt > /dev/null
rm -rf $O/kt9
head -40 $O/c9_unet_forward_kernel_stats.txt
