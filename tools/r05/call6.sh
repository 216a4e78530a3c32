#!/bin/bash
# round 5, sixth GPU call: one-pass GroupNorm with direct statistics for tiny samples (8x6 level) -- op test, forward A/B, parity
O=gpurun_out/r05c6; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "group_norm" > $O/pytest_gn.txt 2>&1; echo "gn tests rc $?" >> $O/pytest_gn.txt
grep -E "passed|failed|FAILED" $O/pytest_gn.txt | head -20
timeout 400 python tools/r05/forward_ab.py --modes "direct:;partial:LADI_GN_DIRECT=0;old:LADI_GN_ONEPASS=0" > $O/gn_ab.txt 2>&1
tail -2 $O/gn_ab.txt
timeout 900 python -m pytest tests/test_gpu_full.py tests/test_gpu_e2e_golden.py tests/test_gpu_modules.py -x -q -m gpu -k "unet_forward or batch8 or graph or tiny" > $O/pytest_parity.txt 2>&1; echo "parity rc $?" >> $O/pytest_parity.txt
tail -5 $O/pytest_parity.txt
