# k_chol_df with two products in flight: parity of the paths that factor, then the four configurations
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_pipe2; mkdir -p $O
{
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_fullsize_parity.py -q -x -k "step_parity or fullsize or rank or shard or two_process or covariance or C2 or C3" 2>&1 | grep -E "passed|failed|error" | tail -3
for c in C1 C2 C3 C4; do python bench/quick.py $c 2>&1 | grep "^$c"; done
for a in 0 2 3; do python bench/quick.py C4 DBAT_AMD_LIB=prof DBAT_HIP_DF_ABLATE=$a 2>&1 | grep "^C4"; done
python bench/quick.py C2 DBAT_HIP_DF_CHAIN=0 2>&1 | grep "^C2"
python bench/quick.py C3 DBAT_HIP_DF_CHAIN=0 2>&1 | grep "^C3"
} 2>&1 | tee $O/log.txt
