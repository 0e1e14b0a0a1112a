# C4 factorisation: is it the operand fetches?  (measurement build: DBAT_HIP_DF_ABLATE leaves them out -- wrong numbers, timing only)
cd $GRAFT_REPO_ROOT
export DBAT_AMD_LIB=$GRAFT_REPO_ROOT/dbat_amd/libdbat_hip_prof.so
for a in 0 1 2 3; do timeout 300 python bench/quick.py C4 DBAT_HIP_DF_ABLATE=$a | grep -v DBAT_AMD_LIB; done
for a in 0 3; do timeout 300 python bench/quick.py C3 DBAT_HIP_DF_ABLATE=$a | grep -v DBAT_AMD_LIB; done
