cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 500 python bench.py --config 4 --roofline-only --no-cpu-baseline > $O/r03_bench_config4_roofline_only.json 2>/dev/null
timeout 500 python bench.py --config 2 --roofline-only --no-cpu-baseline > $O/r03_bench_config2_roofline_only.json 2>/dev/null
cd /tmp; export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats -d $O/kt4 -- python $R/bench.py --config 4 --roofline-only --no-cpu-baseline > /dev/null 2>&1
cd $R
python tools/rocpd_stats.py $(find $O/kt4 -name "*.db" | head -1) $O/r03_config4_unet_forward_kernel_stats.txt > /dev/null
rm -rf $O/kt4
head -c 600 $O/r03_bench_config4_roofline_only.json; echo; head -8 $O/r03_config4_unet_forward_kernel_stats.txt | cut -c1-150
