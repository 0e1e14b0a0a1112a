cd $GRAFT_REPO_ROOT
E=$GRAFT_REPO_ROOT/dbat_amd/libdbat_hip_exp1.so
python bench/quick.py C2 DBAT_AMD_LIB=$E | grep -v loading
