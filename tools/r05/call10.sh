#!/bin/bash
# round 5: regenerate every profile artifact of the round on the final binary (tools/make_profiles.sh), then the full GPU suite (one parity record)
bash tools/make_profiles.sh > gpurun_out/r05_make_profiles.log 2>&1
tail -25 gpurun_out/r05_make_profiles.log
cp ladi_vton_amd/tune_gfx950.txt gpurun_out/r05_tune_gfx950.txt
timeout 1500 python -m pytest tests/ -q -m gpu > gpurun_out/r05_pytest_gpu.txt 2>&1; echo "gpu suite rc $?" >> gpurun_out/r05_pytest_gpu.txt
tail -6 gpurun_out/r05_pytest_gpu.txt
