#!/bin/bash
# round 6, GPU call 2: within-step schedule variants of the dominant halo kernel + wave-state counters of variants 0 / 1 / 2
O=gpurun_out/r06c2; mkdir -p $O
out=$O/halo_sched.txt; : > $out
for rep in 1 2; do
  for v in 0 1 5 9 17 33 21 37 2; do
    timeout 120 tools/r06/bin/halo_sched_$v 32 24 640 640 >> $out 2>&1
  done
done
cat $out
R=$PWD; cd /tmp; export TMPDIR=/tmp
for v in 0 1 2; do
  timeout 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace -d $R/$O/pmc_$v -- $R/tools/r06/bin/halo_sched_$v > /dev/null 2>&1
  python $R/tools/rocpd_pmc.py $(find $R/$O/pmc_$v -name "*.db" | head -1) $R/$O/pmc_wave_$v.txt > /dev/null; rm -rf $R/$O/pmc_$v
  timeout 120 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace -d $R/$O/pmc2_$v -- $R/tools/r06/bin/halo_sched_$v > /dev/null 2>&1
  python $R/tools/rocpd_pmc.py $(find $R/$O/pmc2_$v -name "*.db" | head -1) $R/$O/pmc_inst_$v.txt > /dev/null; rm -rf $R/$O/pmc2_$v
  echo "== $v"; head -3 $R/$O/pmc_wave_$v.txt | cut -c1-400; head -3 $R/$O/pmc_inst_$v.txt | cut -c1-400
done
