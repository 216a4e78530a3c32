# round 6 (as rounds 4 and 5): L2 hit-rate pass of the roofline command and the three counter passes of the 3 072-token self-attention (separate GPU call, same binary)
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out; mkdir -p $O
DIG=$(cat ladi_vton_amd/csrc/_obj/stamp)
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_l2 -- python $R/bench.py --roofline-only --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_lds -- python $R/bench.py --roofline-only --no-cpu-baseline > /dev/null 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace -d $O/pa$i -- python $R/tools/bench_attn.py --attn-only --only self_L0 > /dev/null 2>&1
done
cd $R
python tools/rocpd_pmc.py $(find $O/pmc_l2 -name "*.db" | head -1) $O/r06_pmc_l2.txt --digest $DIG > /dev/null
python tools/rocpd_pmc.py $(find $O/pmc_lds -name "*.db" | head -1) $O/r06_pmc_lds.txt --digest $DIG > /dev/null
for i in 1 2 3; do python tools/rocpd_pmc.py $(find $O/pa$i -name "*.db" | head -1) $O/r06_attn_pmc_$i.txt --digest $DIG > /dev/null; rm -rf $O/pa$i; done
rm -rf $O/pmc_l2 $O/pmc_lds
head -8 $O/r06_pmc_l2.txt | cut -c1-200; head -5 $O/r06_attn_pmc_3.txt | cut -c1-220
