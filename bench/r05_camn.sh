cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "step_parity or gradient or colnorm or residual_jacobian" 2>&1 | tail -3
for c in C3 C2 C4; do timeout 300 python bench/quick.py $c; done
bash bench/kt_quick.sh C4 2>&1 | head -5
