cd $GRAFT_REPO_ROOT
for c in ${1:-C4}; do
DBAT_AMD_LIB=prof DBAT_HIP_PLAN_STATS=1 DBAT_HIP_ABLATE=32 python bench.py --config $c --no-cpu-baseline --no-solve --steps 3 --warmup 1 2>&1 | grep "sig prof\|\[plan\]" | tail -6
done
