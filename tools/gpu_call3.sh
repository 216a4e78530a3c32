# round-4 GPU call 3: GroupNorm affine fused into proj_in, full-size parity on the new kernels, forward A/B, kernel trace
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "fused_group_norm_affine or linear_ or layernorm" 2>&1 | tail -6 > $O/c3_tests_ops.txt
cat $O/c3_tests_ops.txt
export LADI_TUNE_NO_SHIPPED=1
export LADI_TUNE_CACHE=$O/r04_tune_c3.txt
rm -f $LADI_TUNE_CACHE
timeout 600 python bench.py --roofline-only --no-cpu-baseline > $O/c3_roofline_a.json 2> $O/c3_roofline_a.err
LADI_GN_FUSE=0 timeout 600 python bench.py --roofline-only --no-cpu-baseline > $O/c3_roofline_b_nogn.json 2> $O/c3_roofline_b.err
timeout 600 python bench.py --roofline-only --no-cpu-baseline > $O/c3_roofline_a2.json 2> $O/c3_roofline_a2.err
timeout 700 python -m pytest tests/test_gpu_e2e_golden.py -x -q -k "unet_forward_at or config2_chain" 2>&1 | tail -8 > $O/c3_tests_golden.txt
cat $O/c3_tests_golden.txt
unset LADI_TUNE_NO_SHIPPED; unset LADI_TUNE_CACHE
LADI_GN_FUSE=0 LADI_SPLITK_TWO_PASS=1 timeout 600 python bench.py --roofline-only --no-cpu-baseline > $O/c3_roofline_c_r03like.json 2> $O/c3_roofline_c.err
export LADI_TUNE_NO_SHIPPED=1
export LADI_TUNE_CACHE=$O/r04_tune_c3.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt3 -- python $R/bench.py --roofline-only --no-cpu-baseline > /dev/null 2>&1
cd $R
python tools/rocpd_stats.py $(find $O/kt3 -name "*.db" | head -1) $O/c3_unet_forward_kernel_stats.txt > /dev/null; rm -rf $O/kt3
head -40 $O/c3_unet_forward_kernel_stats.txt | cut -c1-150
python - <<'PY'
import json
for t in ("a","b_nogn","a2","c_r03like"):
    try:
        d=json.loads(open("gpurun_out/c3_roofline_%s.json"%t).read().strip().split("\n")[-1])["roofline"]
        print(t, d["unet_forward_ms"], d["igemm_all_tflops"], d["kernel"], d["achieved"])
    except Exception as e: print(t, "ERR", e)
PY
timeout 200 python -X faulthandler tools/lanes_probe.py --lanes-list 4 --pipe-lanes "" > $O/c3_lanes4.json 2> $O/c3_lanes4.err; tail -30 $O/c3_lanes4.err; tail -2 $O/c3_lanes4.json
