# experiment D1: the self-calibrating signature kernel at eight waves per workgroup
cd $GRAFT_REPO_ROOT
E=$GRAFT_REPO_ROOT/dbat_amd/libdbat_hip_exp1.so
for c in C2 C4; do
  python bench/quick.py $c
  python bench/quick.py $c DBAT_HIP_CMAX=16
  python bench/quick.py $c DBAT_AMD_LIB=$E | grep -v "DBAT_AMD_LIB: loading"
done
