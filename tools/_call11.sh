cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  timeout 200 rocprofv3 --pmc "$@" --kernel-trace -d $O/$name -- python $R/tools/run_one.py --shape 32,24,640,640,3 --cfgs 7,74,84,65 --iters 10 > /dev/null 2>&1
  python $R/tools/rocpd_pmc.py $(find $O/$name -name "*.db" | head -1) $O/$name.txt > /dev/null 2>&1
  rm -rf $O/$name
  grep -v "at::native" $O/$name.txt | cut -c1-56,79-
}
run h1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run h2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run h3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
