#!/bin/bash
# round 5, first GPU call: the fused transformer sub-blocks (stand-alone prototypes, op tests, forward A/B, UNet parity with them on)
mkdir -p gpurun_out/r05c1
O=gpurun_out/r05c1
bash tools/experiments/next/first_call.sh > $O/first_call.txt 2>&1
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "fused_cross or fused_feed" > $O/pytest_fused.txt 2>&1
echo "fused tests rc $?" >> $O/pytest_fused.txt
tail -15 $O/pytest_fused.txt
timeout 300 python tools/r05/xf_forward_ab.py > $O/xf_ab.txt 2>&1
tail -8 $O/xf_ab.txt
timeout 600 python -m pytest tests/test_gpu_full.py tests/test_gpu_e2e_golden.py -x -q -m gpu -k "unet_forward or batch8" > $O/pytest_unet.txt 2>&1
echo "unet tests rc $?" >> $O/pytest_unet.txt
tail -15 $O/pytest_unet.txt
