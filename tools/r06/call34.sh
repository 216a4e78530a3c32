#!/bin/bash
# round 6, GPU call 34 (experiment build: attention.hip with LADI_ATTN_VSWZ): the V^T read's swizzle term f(row & 7) -- 0 identity (the library's), 1 x3, 2 x5, 3 bit reversal,
# 4 rotate -- on the 3 072-token self-attention: correctness (test_flash_attention*), time, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r06c34; mkdir -p $O
for m in 0 1 2 3 4; do
  echo "== LADI_ATTN_VSWZ=$m"
  LADI_ATTN_VSWZ=$m timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "flash_attention" 2>&1 | tail -1
  LADI_ATTN_VSWZ=$m timeout 120 python tools/bench_attn.py --attn-only --only self_L0 2>/dev/null | tail -1
  LADI_ATTN_VSWZ=$m timeout 120 python tools/bench_attn.py --attn-only --only self_L0 2>/dev/null | tail -1
  (cd /tmp; export TMPDIR=/tmp; LADI_ATTN_VSWZ=$m timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/p$m -- python $R/tools/bench_attn.py --attn-only --only self_L0 > /dev/null 2>&1)
  python tools/rocpd_pmc.py $(find $O/p$m -name "*.db" | head -1) | grep flash_attn64 | cut -c1-230
  rm -rf $O/p$m
done 2>&1 | tee $O/attn_vswz.txt
