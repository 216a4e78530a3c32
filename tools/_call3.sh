cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "^\s*(Name|name)\s*:\s*\S+|\b(TCP|TCC|TA|TD|SQ|GRBM)_[A-Z0-9_a-z]+" | sed 's/.*: *//' | sort -u > $GRAFT_REPO_ROOT/gpurun_out/pmc/counters.txt
wc -l $GRAFT_REPO_ROOT/gpurun_out/pmc/counters.txt
