cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
