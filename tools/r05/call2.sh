#!/bin/bash
# round 5, second GPU call: one-pass GroupNorm (op test, forward A/B with a block-size sweep, UNet / VAE parity)
O=gpurun_out/r05c2; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "group_norm" > $O/pytest_gn.txt 2>&1; echo "gn tests rc $?" >> $O/pytest_gn.txt
tail -5 $O/pytest_gn.txt
timeout 400 python tools/r05/forward_ab.py --modes "new:LADI_XF_FUSE=0;old:LADI_XF_FUSE=0,LADI_GN_ONEPASS=0;p64:LADI_XF_FUSE=0,LADI_GN_PPB=64;p128:LADI_XF_FUSE=0,LADI_GN_PPB=128;p256:LADI_XF_FUSE=0,LADI_GN_PPB=256;p512:LADI_XF_FUSE=0,LADI_GN_PPB=512;p1024:LADI_XF_FUSE=0,LADI_GN_PPB=1024" > $O/gn_ab.txt 2>&1
tail -3 $O/gn_ab.txt
LADI_XF_FUSE=0 timeout 900 python -m pytest tests/test_gpu_full.py tests/test_gpu_e2e_golden.py tests/test_gpu_modules.py -x -q -m gpu -k "unet_forward or batch8 or vae or graph" > $O/pytest_parity.txt 2>&1; echo "parity rc $?" >> $O/pytest_parity.txt
tail -8 $O/pytest_parity.txt
