#!/bin/bash
# round 6, GPU call 33: which clock does GRBM_GUI_ACTIVE count?  The harness prints shader cycles (s_memtime) and microseconds per K loop; the same run under
# rocprofv3 --pmc gives GRBM_GUI_ACTIVE and the duration per launch
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r06c33; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/p2 -- $R/tools/r06/bin/stage_path > $O/harness_under_rocprof.txt 2>/dev/null
cd $R
python - <<'PY'
import sqlite3, glob
db = glob.glob("gpurun_out/r06c33/p2/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(pmc_events)").fetchall()]
print("columns:", cols)
key_col = "dispatch_id" if "dispatch_id" in cols else ("id" if "id" in cols else cols[0])
rows = c.execute("select name, counter_name, counter_value, duration, %s from pmc_events order by %s" % (key_col, key_col)).fetchall()
# one line per dispatch: group consecutive rows of the same dispatch (4 counters each)
import collections
out = []
cur = {}
last = None
for n, cn, v, d, k_ in rows:
    key = (n, d, k_)
    if key != last and cur:
        out.append((last, cur)); cur = {}
    cur[cn] = v; last = key
if cur: out.append((last, cur))
print("dispatches:", len(out))
for (n, d, k_), cv in out:
    if "kloop<0, 128, 128, 2, 2, 2>" in n or "kloop<6, 128, 128, 2, 2, 4>" in n or "kloop<2, 128, 128, 2, 2, 2>" in n:
        g = cv.get("GRBM_GUI_ACTIVE", 0); m = cv.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
        print("%-40s dur %8.1f us  GRBM %10.0f (%.2f GHz)  MFMA_BUSY/GRBM %.2f  INSTS_MFMA %.0f" % (n[5:45], d / 1e3, g, g / d if d else 0, m / g if g else 0, cv.get("SQ_INSTS_MFMA", 0)))
PY
rm -rf $O/p2
echo "--- harness output under rocprof (cycles per step from s_memtime, us per K loop from HIP events):"
grep -E "fragments \+ MFMA only|LDS-DMA ring \+ MFMA|W direct-to-VGPR, X resident" $O/harness_under_rocprof.txt | cut -c1-170 | head -30
