# Regenerates every round-3 artifact under profiles/ in ONE GPU call (run from the repo root on the GPU box):
#   bash tools/make_profiles.sh            -> writes gpurun_out/r03_*; copy what is to be judged into profiles/
set -u
R=$PWD
O=$R/gpurun_out
mkdir -p $O
DIG=$(cat ladi_vton_amd/csrc/_obj/stamp)
# 1. tile selections for the BASELINE batch sizes (the table shipped as ladi_vton_amd/tune_gfx950.txt)
export LADI_TUNE_NO_SHIPPED=1
export LADI_TUNE_CACHE=$O/r03_tune.txt
rm -f $LADI_TUNE_CACHE
timeout 1200 python bench.py --steps 10 --warmup 3 > $O/r03_bench_default.json 2> $O/r03_bench_default.err
timeout 300 python bench.py --config 2 --no-cpu-baseline --no-roofline --steps 2 --warmup 1 > $O/r03_bench_config2.json 2>/dev/null
timeout 300 python bench.py --scheduler ddim --no-cpu-baseline --no-roofline --steps 3 --warmup 1 > $O/r03_bench_ddim.json 2>/dev/null
timeout 300 python bench.py --scheduler lms --no-cpu-baseline --no-roofline --steps 3 --warmup 1 > $O/r03_bench_lms.json 2>/dev/null
timeout 400 python bench.py --config 4 --no-cpu-baseline --no-roofline --steps 1 --warmup 1 > $O/r03_bench_config4.json 2>/dev/null
timeout 300 python bench.py --roofline-only --no-cpu-baseline > $O/r03_bench_roofline_only.json 2>/dev/null
# 2. kernel traces (tuned selections preloaded: no tuning launches in the traces)
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -- python $R/bench.py --roofline-only --no-cpu-baseline > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/ktb -- python $R/bench.py --no-cpu-baseline --no-roofline --steps 1 --warmup 1 > /dev/null 2>&1
# 3. PMC passes, each in its own run (FETCH_SIZE and WRITE_SIZE cannot share a pass)
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_f -- python $R/bench.py --roofline-only --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_w -- python $R/bench.py --roofline-only --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_m -- python $R/bench.py --roofline-only --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_l2 -- python $R/bench.py --roofline-only --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_w8 -- python $R/bench.py --roofline-only --no-cpu-baseline > /dev/null 2>&1
# VAE / EMASC stages alone (counter collection segfaults on the hipGraph-replaying full bench): time + HBM bytes per kernel
timeout 300 python $R/tools/bench_vae.py > $O/r03_vae_stages.txt 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats -d $O/ktv -- python $R/tools/bench_vae.py --iters 1 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fv -- python $R/tools/bench_vae.py --iters 1 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_wv -- python $R/tools/bench_vae.py --iters 1 > /dev/null 2>&1
# attention / norm micro-benchmark: rates, then three counter passes (instruction mix, LDS, MFMA busy) on the 3072-token self-attention alone
timeout 300 python $R/tools/bench_attn.py > $O/r03_attn_bench.txt 2>/dev/null
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace -d $O/pa$i -- python $R/tools/bench_attn.py --attn-only --only self_L0 > /dev/null 2>&1
done
cd $R
for i in 1 2 3; do python tools/rocpd_pmc.py $(find $O/pa$i -name "*.db" | head -1) $O/r03_attn_pmc_$i.txt --digest $DIG > /dev/null; rm -rf $O/pa$i; done
python tools/rocpd_stats.py $(find $O/kt -name "*.db" | head -1) $O/r03_unet_forward_kernel_stats.txt > /dev/null
python tools/rocpd_stats.py $(find $O/ktb -name "*.db" | head -1) $O/r03_bench_kernel_stats.txt > /dev/null
python tools/rocpd_pmc.py $(find $O/pmc_f -name "*.db" | head -1) $O/r03_pmc_fetch_size.txt --digest $DIG > /dev/null
python tools/rocpd_pmc.py $(find $O/pmc_w -name "*.db" | head -1) $O/r03_pmc_write_size.txt --digest $DIG > /dev/null
python tools/rocpd_pmc.py $(find $O/pmc_m -name "*.db" | head -1) $O/r03_pmc_mfma_busy.txt --digest $DIG > /dev/null
python tools/rocpd_pmc.py $(find $O/pmc_l2 -name "*.db" | head -1) $O/r03_pmc_l2.txt --digest $DIG > /dev/null
python tools/rocpd_pmc.py $(find $O/pmc_w8 -name "*.db" | head -1) $O/r03_pmc_wave_cycles.txt --digest $DIG > /dev/null
python tools/rocpd_stats.py $(find $O/ktv -name "*.db" | head -1) $O/r03_vae_kernel_stats.txt > /dev/null
python tools/rocpd_pmc.py $(find $O/pmc_fv -name "*.db" | head -1) $O/r03_vae_pmc_fetch_size.txt --digest $DIG > /dev/null
python tools/rocpd_pmc.py $(find $O/pmc_wv -name "*.db" | head -1) $O/r03_vae_pmc_write_size.txt --digest $DIG > /dev/null
rm -rf $O/kt $O/ktb $O/pmc_f $O/pmc_w $O/pmc_m $O/ktv $O/pmc_fv $O/pmc_wv $O/pmc_l2 $O/pmc_w8
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/dma_conv_pattern.hip -o /tmp/dma_conv 2>/dev/null && /tmp/dma_conv > $O/r03_dma_conv_pattern.txt 2>&1
timeout 600 python tools/bench_shapes.py --n 16 --iters 10 --json $O/r03_shapes_n16.json > $O/r03_shapes_n16.txt 2>&1
head -c 400 $O/r03_bench_default.json; echo; tail -2 $O/r03_bench_default.err
head -12 $O/r03_unet_forward_kernel_stats.txt; head -8 $O/r03_pmc_mfma_busy.txt; wc -l $O/r03_tune.txt
