set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_modules.py -x -q -k "text_encoder" 2>&1 | tail -5
timeout 300 python tools/bench_kslope.py --json $O/c4_kslope.json > $O/c4_kslope.txt 2>&1
cat $O/c4_kslope.txt
timeout 300 python -X faulthandler tools/lanes_probe.py --lanes-list 1,2,4,2,1 --pipe-lanes "" --iters 4 > $O/c4_lanes.json 2> $O/c4_lanes.err; tail -5 $O/c4_lanes.err; tail -1 $O/c4_lanes.json
