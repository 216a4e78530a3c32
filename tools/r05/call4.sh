#!/bin/bash
# round 5, fourth GPU call: GroupNorm / CLIP tests again, one-pass GroupNorm with up to 96 partial rows (A/B against 32), config2 golden chain
O=gpurun_out/r05c4; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "group_norm" > $O/pytest_gn.txt 2>&1; echo "gn tests rc $?" >> $O/pytest_gn.txt
grep -E "passed|failed|FAILED" $O/pytest_gn.txt | head -20
timeout 300 python -m pytest tests/test_gpu_modules.py -q -m gpu -k "clip_preprocess" > $O/pytest_clip.txt 2>&1; echo "clip tests rc $?" >> $O/pytest_clip.txt
tail -4 $O/pytest_clip.txt
timeout 400 python tools/r05/forward_ab.py --modes "r96:;r32:LADI_GN_MAX_RPS=32;old:LADI_GN_ONEPASS=0" > $O/gn_ab.txt 2>&1
tail -2 $O/gn_ab.txt
timeout 900 python -m pytest tests/test_gpu_e2e_golden.py -x -q -m gpu -k "config2_chain or unet_forward" > $O/pytest_golden.txt 2>&1; echo "golden rc $?" >> $O/pytest_golden.txt
tail -6 $O/pytest_golden.txt
