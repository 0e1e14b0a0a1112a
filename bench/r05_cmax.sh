cd $GRAFT_REPO_ROOT
for v in 21 18 16; do timeout 300 python bench/quick.py C3 DBAT_HIP_CMAX=$v; done
