# task order experiment: measured k_chol_df time for every candidate order (DBAT_HIP_DF_ORDER) next to the model's figure
cd $GRAFT_REPO_ROOT
C=${1:-C3}
DBAT_HIP_PLAN_STATS=1 python bench.py --config $C --steps 6 --warmup 2 --no-cpu-baseline --no-solve 2> /tmp/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$C default', d['value'], d['kernel_ms']['k_chol_df'])"
grep "^\[chol\].*simul" /tmp/err.txt | tail -1
for o in ${2:-0 1 2 3 4 5 6 7}; do
  DBAT_AMD_LIB=prof DBAT_HIP_DF_ORDER=$o python bench.py --config $C --steps 6 --warmup 2 --no-cpu-baseline --no-solve 2> /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$C candidate $o', d['value'], d['kernel_ms']['k_chol_df'])"
done
