set -u
R=$PWD
mkdir -p gpurun_out
cp profiles/r01_igemm_tune_cache_B8.txt gpurun_out/tune.txt
export LADI_TUNE_CACHE=$R/gpurun_out/tune.txt
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
timeout 300 python bench.py --roofline-only --no-cpu-baseline > gpurun_out/bench_roofline_only.json 2>/dev/null
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt -- python $R/bench.py --roofline-only --no-cpu-baseline > /dev/null 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_f -- python $R/bench.py --roofline-only --no-cpu-baseline > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc_w -- python $R/bench.py --roofline-only --no-cpu-baseline > /dev/null 2>&1
cd $R
python tools/rocpd_stats.py $(find gpurun_out/kt -name "*.db" | head -1) gpurun_out/kstats.txt > /dev/null
python tools/rocpd_pmc.py $(find gpurun_out/pmc_f -name "*.db" | head -1) gpurun_out/pmc_fetch.txt > /dev/null
python tools/rocpd_pmc.py $(find gpurun_out/pmc_w -name "*.db" | head -1) gpurun_out/pmc_write.txt > /dev/null
find gpurun_out -name "*.db" -delete
rm -rf gpurun_out/kt gpurun_out/pmc_f gpurun_out/pmc_w
head -c 600 gpurun_out/bench_default.json; echo; tail -3 gpurun_out/bench_default.err; head -8 gpurun_out/kstats.txt; head -5 gpurun_out/pmc_fetch.txt
# extra bench points quoted in BASELINE.md §4 / README.md
timeout 300 python bench.py --scheduler ddim --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > gpurun_out/bench_ddim.json
unset LADI_TUNE_CACHE
timeout 500 python bench.py --batch 32 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > gpurun_out/bench_b32.json
timeout 500 python bench.py --height 1024 --width 768 --batch 4 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > gpurun_out/bench_1024x768_b4.json
for f in bench_ddim bench_b32 bench_1024x768_b4; do head -c 150 gpurun_out/$f.json; echo; done
