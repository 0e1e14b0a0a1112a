# leaf size of the nested dissection against the factorisation's time (the chain role has changed the balance since round 3)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_ndleaf; mkdir -p $O
for c in C3 C1 C2; do for l in 8 16 24 32 48 64 96; do python bench/quick.py $c DBAT_HIP_ND_LEAF=$l 2>&1 | grep "^$c"; done; done | tee $O/log.txt
