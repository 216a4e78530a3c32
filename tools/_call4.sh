cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  name=$1; shift
  timeout 200 rocprofv3 --pmc "$@" --kernel-trace -d $O/$name -- python $R/tools/run_one.py --shape 64,48,320,320,3 --cfgs 22,32,39,40 > /dev/null 2>&1
  python $R/tools/rocpd_pmc.py $(find $O/$name -name "*.db" | head -1) $O/$name.txt > /dev/null 2>&1
  rm -rf $O/$name
  cat $O/$name.txt | cut -c1-60,79-
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run b TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_BUFFER_READ_LDS_WAVEFRONTS_sum GRBM_GUI_ACTIVE
run c TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum GRBM_GUI_ACTIVE
run d TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum GRBM_GUI_ACTIVE
run e SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
