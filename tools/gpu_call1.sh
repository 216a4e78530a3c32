# round-4 GPU call 1: new host paths (lanes, uint8, CLIP pre-processing, step counter), golden e2e cases, lanes A/B
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
export LADI_TUNE_CACHE=$PWD/gpurun_out/r04_tune_new.txt
rm -f $LADI_TUNE_CACHE
timeout 900 python -m pytest tests/test_gpu_modules.py -x -q -k "lanes or uint8 or clip_preprocess or graph_equals_eager or tryon_pipeline_tiny or graph_survives" 2>&1 | tail -15 > gpurun_out/c1_tests_modules.txt
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -k "sched or resize" 2>&1 | tail -5 > gpurun_out/c1_tests_ops.txt
timeout 700 python -m pytest tests/test_gpu_e2e_golden.py -x -q -k "unet_forward_at or config2_chain" 2>&1 | tail -15 > gpurun_out/c1_tests_golden.txt
timeout 700 python tools/lanes_probe.py > gpurun_out/c1_lanes.json 2> gpurun_out/c1_lanes.err
cat gpurun_out/c1_tests_modules.txt gpurun_out/c1_tests_ops.txt gpurun_out/c1_tests_golden.txt
tail -3 gpurun_out/c1_lanes.json; tail -5 gpurun_out/c1_lanes.err
wc -l $LADI_TUNE_CACHE
