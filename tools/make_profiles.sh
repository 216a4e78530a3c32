# Regenerates every round-6 artifact under profiles/ in ONE GPU call (run from the repo root on the GPU box):
#   bash tools/make_profiles.sh            -> writes gpurun_out/r06_*; copy what is to be judged into profiles/
# Order matters: (1) tile selections are measured and installed as the shipped table, (2) the kernel traces and PMC passes of the
# roofline command are taken and installed under profiles/ (digest-stamped), (3) only then the headline bench line is produced, so that
# its roofline.traffic / roofline.hbm_kernels are the digest-checked committed values of THIS binary.
set -u
R=$PWD
O=$R/gpurun_out
mkdir -p $O
DIG=$(cat ladi_vton_amd/csrc/_obj/stamp)
# ---- 1. tile selections for the BASELINE batch sizes (shipped as ladi_vton_amd/tune_gfx950.txt); LADI_PROFILE_KEEP_TABLE=1 keeps the committed table
#         (a second profile call on another box must measure the table the parity record was taken with)
if [ -z "${LADI_PROFILE_KEEP_TABLE:-}" ]; then
export LADI_TUNE_NO_SHIPPED=1
export LADI_TUNE_CACHE=$O/r06_tune.txt
rm -f $LADI_TUNE_CACHE
timeout 600 python bench.py --no-cpu-baseline --no-roofline --no-tail --steps 1 --warmup 1 > /dev/null 2>&1
timeout 300 python bench.py --roofline-only --no-cpu-baseline > /dev/null 2>&1
timeout 400 python bench.py --config 2 --no-cpu-baseline --no-roofline --no-tail --steps 1 --warmup 1 > /dev/null 2>&1
timeout 400 python bench.py --config 4 --no-cpu-baseline --no-roofline --no-tail --steps 1 --warmup 0 > /dev/null 2>&1
unset LADI_TUNE_NO_SHIPPED; unset LADI_TUNE_CACHE
cp $O/r06_tune.txt ladi_vton_amd/tune_gfx950.txt
fi
wc -l ladi_vton_amd/tune_gfx950.txt
# ---- 2. kernel traces and PMC passes of the roofline command (each counter set in its own run)
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -- python $R/bench.py --roofline-only --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_f -- python $R/bench.py --roofline-only --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_w -- python $R/bench.py --roofline-only --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_m -- python $R/bench.py --roofline-only --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_w8 -- python $R/bench.py --roofline-only --no-cpu-baseline > /dev/null 2>&1
timeout 300 python $R/tools/bench_vae.py > $O/r06_vae_stages.txt 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats -d $O/ktv -- python $R/tools/bench_vae.py --iters 1 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fv -- python $R/tools/bench_vae.py --iters 1 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_wv -- python $R/tools/bench_vae.py --iters 1 > /dev/null 2>&1
cd $R
python tools/rocpd_stats.py $(find $O/kt -name "*.db" | head -1) $O/r06_unet_forward_kernel_stats.txt > /dev/null
python tools/rocpd_pmc.py $(find $O/pmc_f -name "*.db" | head -1) $O/r06_pmc_fetch_size.txt --digest $DIG > /dev/null
python tools/rocpd_pmc.py $(find $O/pmc_w -name "*.db" | head -1) $O/r06_pmc_write_size.txt --digest $DIG > /dev/null
python tools/rocpd_pmc.py $(find $O/pmc_m -name "*.db" | head -1) $O/r06_pmc_mfma_busy.txt --digest $DIG > /dev/null
python tools/rocpd_pmc.py $(find $O/pmc_w8 -name "*.db" | head -1) $O/r06_pmc_wave_cycles.txt --digest $DIG > /dev/null
python tools/rocpd_stats.py $(find $O/ktv -name "*.db" | head -1) $O/r06_vae_kernel_stats.txt > /dev/null
python tools/rocpd_pmc.py $(find $O/pmc_fv -name "*.db" | head -1) $O/r06_vae_pmc_fetch_size.txt --digest $DIG > /dev/null
python tools/rocpd_pmc.py $(find $O/pmc_wv -name "*.db" | head -1) $O/r06_vae_pmc_write_size.txt --digest $DIG > /dev/null
rm -rf $O/kt $O/pmc_f $O/pmc_w $O/pmc_m $O/pmc_w8 $O/ktv $O/pmc_fv $O/pmc_wv
cp $O/r06_unet_forward_kernel_stats.txt $O/r06_pmc_fetch_size.txt $O/r06_pmc_write_size.txt $O/r06_vae_kernel_stats.txt $O/r06_vae_pmc_fetch_size.txt $O/r06_vae_pmc_write_size.txt profiles/
# ---- 3. bench lines (shipped table, committed PMC files of this binary)
timeout 1200 python bench.py --steps 10 --warmup 3 > $O/r06_bench_default.json 2> $O/r06_bench_default.err
timeout 300 python bench.py --config 2 --no-cpu-baseline --no-roofline --steps 2 --warmup 1 > $O/r06_bench_config2.json 2>/dev/null
timeout 300 python bench.py --scheduler ddim --no-cpu-baseline --no-roofline --steps 3 --warmup 1 > $O/r06_bench_ddim.json 2>/dev/null
[ -z "${LADI_PROFILE_QUICK:-}" ] && timeout 300 python bench.py --scheduler lms --no-cpu-baseline --no-roofline --steps 3 --warmup 1 > $O/r06_bench_lms.json 2>/dev/null
timeout 400 python bench.py --config 4 --no-cpu-baseline --no-roofline --steps 1 --warmup 1 > $O/r06_bench_config4.json 2>/dev/null
timeout 300 python bench.py --roofline-only --no-cpu-baseline > $O/r06_bench_roofline_only.json 2>/dev/null
[ -z "${LADI_PROFILE_QUICK:-}" ] && timeout 400 python bench.py --config 2 --roofline-only --no-cpu-baseline > $O/r06_bench_config2_roofline_only.json 2>/dev/null
[ -z "${LADI_PROFILE_QUICK:-}" ] && timeout 400 python bench.py --config 4 --roofline-only --no-cpu-baseline > $O/r06_bench_config4_roofline_only.json 2>/dev/null
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/ktb -- python $R/bench.py --no-cpu-baseline --no-roofline --no-tail --steps 1 --warmup 1 > /dev/null 2>&1
cd $R
python tools/rocpd_stats.py $(find $O/ktb -name "*.db" | head -1) $O/r06_bench_kernel_stats.txt > /dev/null; rm -rf $O/ktb
timeout 300 python $R/tools/bench_attn.py > $O/r06_attn_bench.txt 2>/dev/null
timeout 500 python tools/bench_shapes.py --n 16 --iters ${LADI_SHAPES_ITERS:-10} --json $O/r06_shapes_n16.json > $O/r06_shapes_n16.txt 2>&1
head -c 500 $O/r06_bench_default.json; echo; tail -2 $O/r06_bench_default.err
head -14 $O/r06_unet_forward_kernel_stats.txt | cut -c1-150; head -6 $O/r06_pmc_mfma_busy.txt | cut -c1-200
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_bench_default.json").read().strip().split("\n")[-1])
r = d["roofline"]
print({k: r.get(k) for k in ("kernel", "achieved", "frac", "traffic", "unet_forward_ms", "igemm_all_tflops", "clock")})
print(json.dumps(r.get("hbm_kernels"))[:1500])
PY

# quick end-to-end parity of the final binary against the committed oracle outputs (the full suite is tools/final_check.sh)
timeout 600 python -m pytest tests/test_gpu_e2e_golden.py -x -q -k "unet_forward_at or baseline_batch8 or config2_chain" 2>&1 | tail -3
