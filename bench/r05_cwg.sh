cd $GRAFT_REPO_ROOT
for v in 8 16 32 48; do timeout 300 python bench/quick.py C3 DBAT_HIP_DF_CHAIN_WG=$v; done
for v in 8 32; do timeout 300 python bench/quick.py C1 DBAT_HIP_DF_CHAIN_WG=$v; done
