cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "step_parity or cholesky or all_dampings or prior or sxb" 2>&1 | tail -3
for c in C4 C3 C2 C1; do timeout 300 python bench/quick.py $c; done
timeout 300 python bench/quick.py C3 -- --emulate-ranks 8
