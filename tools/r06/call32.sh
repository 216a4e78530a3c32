#!/bin/bash
# round 6, GPU call 32: calibrate the LDS counters on the harness (tools/r06/stage_path.hip: modes whose LDS traffic is known by construction) so that the
# same counters on the library's kernels (profiles/r06_pmc_lds.txt) can be read as a fraction of the LDS port
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r06c32; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/p1 -- $R/tools/r06/bin/stage_path > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/p2 -- $R/tools/r06/bin/stage_path > /dev/null 2>&1
cd $R
python tools/rocpd_pmc.py $(find $O/p1 -name "*.db" | head -1) $O/stage_path_pmc_lds.txt > /dev/null
python tools/rocpd_pmc.py $(find $O/p2 -name "*.db" | head -1) $O/stage_path_pmc_mfma.txt > /dev/null
rm -rf $O/p1 $O/p2
cut -c1-250 $O/stage_path_pmc_lds.txt | head -50
cut -c1-250 $O/stage_path_pmc_mfma.txt | head -50
