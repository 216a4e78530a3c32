#!/bin/bash
# round 6, GPU call 21: the library path (torch-ROCm fp16: MIOpen / hipBLASLt / SDPA running the oracle's modules) beside the HIP path
O=gpurun_out/r06c21; mkdir -p $O
LADI_LIBRARY_TUNE=1 timeout 1500 python -m pytest tests/test_gpu_library_path.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -30 $O/pytest.txt
cat gpurun_out/r06_library_path.json
