#!/bin/bash
# round 6, GPU call 15: deferred range guard of the fused pipeline -- module tests (tiny models) + one full-size e2e
O=gpurun_out/r06c15; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_modules.py -q -m gpu -x > $O/pytest_modules.txt 2>&1; echo "rc $?" >> $O/pytest_modules.txt; tail -5 $O/pytest_modules.txt
timeout 1500 python -m pytest tests/test_gpu_e2e_golden.py -q -m gpu -x -k "baseline_batch8 or unet_forward_at" > $O/pytest_golden.txt 2>&1; echo "rc $?" >> $O/pytest_golden.txt; tail -3 $O/pytest_golden.txt
timeout 600 python bench.py --no-cpu-baseline --no-roofline --steps 3 --warmup 1 2> /dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','stage_ms_rank0','with_d2h_pil_images_per_s')})"
