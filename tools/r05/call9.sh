#!/bin/bash
# round 5, ninth GPU call: 2-D blocked halo tiles (cfg 100-103) -- op tests, VAE stages with the shipped selections vs a fresh tile measurement, parity
O=gpurun_out/r05c9; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "halo_2d or halo_resident" > $O/pytest_halo.txt 2>&1; echo "halo tests rc $?" >> $O/pytest_halo.txt
grep -E "passed|failed|FAILED|Error" $O/pytest_halo.txt | head
echo "== shipped selections" > $O/vae_stages.txt
timeout 300 python tools/bench_vae.py >> $O/vae_stages.txt 2>&1
echo "== fresh tile measurement (2-D halo forms offered)" >> $O/vae_stages.txt
LADI_TUNE_NO_SHIPPED=1 LADI_TUNE_CACHE=$PWD/$O/tune_vae.txt timeout 600 python tools/bench_vae.py >> $O/vae_stages.txt 2>&1
grep -v amdgpu.ids $O/vae_stages.txt
awk '$9>=100' $O/tune_vae.txt | head -40
LADI_TUNE_NO_SHIPPED=1 LADI_TUNE_CACHE=$PWD/$O/tune_vae.txt timeout 900 python -m pytest tests/test_gpu_full.py tests/test_gpu_modules.py -x -q -m gpu -k "vae or emasc" > $O/pytest_vae.txt 2>&1; echo "vae parity rc $?" >> $O/pytest_vae.txt
tail -4 $O/pytest_vae.txt
