cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "deterministic" 2>&1 | tail -5
