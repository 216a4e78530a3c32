#!/bin/bash
# round 5: two zero rows in the halo kernel (out-of-image lanes keep their bank slot) -- op tests, same-box A/B of the old / new addressing with the
# ablation harness (time + LDS bank-conflict counters), on the 32x24 and 16x12 convolutions
O=gpurun_out/r05c15; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "halo_resident or halo_2d or fused_output_statistics" > $O/pytest_halo.txt 2>&1; echo "halo tests rc $?" >> $O/pytest_halo.txt
grep -E "passed|failed|FAILED" $O/pytest_halo.txt | head
for r in 1 2 3; do for v in old new; do echo -n "$v " >> $O/ab.txt; timeout 60 tools/r05/bin/halo_$v >> $O/ab.txt 2>&1; done; done
for v in old new; do echo -n "$v " >> $O/ab.txt; timeout 60 tools/r05/bin/halo_$v 16 12 1280 1280 >> $O/ab.txt 2>&1; done
for v in old new; do echo -n "$v " >> $O/ab.txt; timeout 60 tools/r05/bin/halo_$v 64 48 320 320 >> $O/ab.txt 2>&1; done
cat $O/ab.txt
R=$PWD; cd /tmp; export TMPDIR=/tmp
for v in old new; do
  timeout 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d $R/$O/pmc_$v -- $R/tools/r05/bin/halo_$v > /dev/null 2>&1
  python $R/tools/rocpd_pmc.py $(find $R/$O/pmc_$v -name "*.db" | head -1) $R/$O/pmc_lds_$v.txt > /dev/null; rm -rf $R/$O/pmc_$v
  echo "== $v"; head -4 $R/$O/pmc_lds_$v.txt | cut -c1-200
done
