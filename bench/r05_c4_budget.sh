cd $GRAFT_REPO_ROOT
export DBAT_AMD_LIB=$GRAFT_REPO_ROOT/dbat_amd/libdbat_hip_prof.so
for a in 0 3; do
DBAT_HIP_DF_ABLATE=$a DBAT_HIP_DF_TRACE=/tmp/df_trace.csv python bench.py --config C4 --steps 2 --warmup 1 --no-cpu-baseline --no-solve > /tmp/o.json 2>/dev/null
echo "ABLATE=$a"; python bench/chol_budget.py /tmp/df_trace.csv 256 116508
done
head -3 /tmp/df_trace.csv; wc -l /tmp/df_trace.csv
