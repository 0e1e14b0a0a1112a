cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_hip_parity.py tests/test_fullsize_parity.py tests/test_multishard_gpu.py -x -q -m gpu -s -k "covariance" 2>&1 | grep -v "amdgpu.ids" | tail -12
