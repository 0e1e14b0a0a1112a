cd $GRAFT_REPO_ROOT
DBAT_HIP_SIG=2 timeout 600 python bench/r05_det.py small small+io C1 C1+io 2>&1 | grep -v amdgpu.ids
DBAT_HIP_SIG=0 timeout 600 python bench/r05_det.py small small+io C1 C1+io 2>&1 | grep -v amdgpu.ids
timeout 900 python bench/r05_det.py C2 C3 C4 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "deterministic or step_parity" 2>&1 | tail -5
for c in C3 C4; do timeout 300 python bench/quick.py $c -- --deterministic; done
