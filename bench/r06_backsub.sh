cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "signature_group or step_parity or synthetic_bundle or determin" 2>&1 | tail -3
for c in C1 C2 C3 C4; do python bench/quick.py $c; done
bash bench/r06_camn.sh C4 C2 2>&1 | grep backsub
