cd $GRAFT_REPO_ROOT
export DBAT_AMD_LIB=$GRAFT_REPO_ROOT/dbat_amd/libdbat_hip_prof.so
for c in C3 C4 C2; do DBAT_HIP_ABLATE=32 timeout 300 python bench/quick.py $c | grep -v DBAT_AMD_LIB | tail -2; done
