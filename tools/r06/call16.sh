#!/bin/bash
O=gpurun_out/r06c16; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_modules.py -q -m gpu -x -k "range_guard or tryon_pipeline_tiny or graph" > $O/pytest_modules.txt 2>&1; echo "rc $?" >> $O/pytest_modules.txt; tail -25 $O/pytest_modules.txt
