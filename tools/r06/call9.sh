#!/bin/bash
# round 6, GPU call 9: the whole GPU suite on the current binary (pinned halo forms, complementary-erf GELU, GEGLU bias-init, folded GroupNorm rows, ...)
O=gpurun_out/r06c9; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "rc $?" >> $O/pytest_gpu.txt
tail -15 $O/pytest_gpu.txt
