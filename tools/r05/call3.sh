#!/bin/bash
# round 5, third GPU call: GroupNorm op tests (all shapes, both forms), CLIP pre-processing, kernel trace of the forward
O=gpurun_out/r05c3; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "group_norm" > $O/pytest_gn.txt 2>&1; echo "gn tests rc $?" >> $O/pytest_gn.txt
grep -E "passed|failed|FAILED" $O/pytest_gn.txt | head -20
timeout 300 python -m pytest tests/test_gpu_modules.py -q -m gpu -k "clip_preprocess or text" > $O/pytest_clip.txt 2>&1; echo "clip tests rc $?" >> $O/pytest_clip.txt
tail -5 $O/pytest_clip.txt
R=$PWD
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/kt -- python $R/bench.py --roofline-only --no-cpu-baseline > $R/$O/roofline_only.json 2>/dev/null
cd $R
python tools/rocpd_stats.py $(find $O/kt -name "*.db" | head -1) $O/unet_forward_kernel_stats.txt > /dev/null; rm -rf $O/kt
head -40 $O/unet_forward_kernel_stats.txt | cut -c1-160
