cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
for kr in 0 1; do
echo "== KROT=$kr"
LADI_KROT=$kr timeout 600 python tools/bench_shapes.py --n 16 --iters 20 --filter conv3 --cfgs 7,9,14,22,32,33,37,39,40,47,54,2 2>&1 | grep -v amdgpu.ids
done
LADI_KROT=1 timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "igemm8_staggered or race_free or eight_wave" 2>&1 | tail -3
