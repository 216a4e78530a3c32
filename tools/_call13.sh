cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
for a in 0 1 16 17 8 2 4 6 7; do echo "== ABL=$a"; LADI_ATTN_ABL=$a timeout 300 python tools/bench_attn.py 2>&1 | grep -E "self L0|self L1" ; done
