cd $GRAFT_REPO_ROOT
timeout 300 python bench/r05_dbg.py 2>&1 | tail -60
