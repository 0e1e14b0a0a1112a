# k_chol_df<false> at two workgroups per CU (-DDBAT_DF_OCC2=1, dbat_amd/libdbat_hip_occ2.so) against the product build
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_occ2; mkdir -p $O
L=$GRAFT_REPO_ROOT/dbat_amd/libdbat_hip_occ2.so
{
python bench/quick.py C4
python bench/quick.py C4 DBAT_AMD_LIB=$L
python bench/quick.py C4 DBAT_AMD_LIB=$L DBAT_HIP_DF_GRID=384
python bench/quick.py C2 DBAT_HIP_DF_CHAIN=0
python bench/quick.py C2 DBAT_HIP_DF_CHAIN=0 DBAT_AMD_LIB=$L
python bench/quick.py C3 DBAT_HIP_DF_CHAIN=0
python bench/quick.py C3 DBAT_HIP_DF_CHAIN=0 DBAT_AMD_LIB=$L
DBAT_AMD_LIB=$L timeout 900 python -m pytest tests/test_hip_parity.py -q -x -k "step_parity or fullsize" 2>&1 | tail -3
} 2>&1 | grep -v "^\[dbat_amd\]" | tee $O/log.txt
