# Dataflow Cholesky on the GPU box: bench line (kernel_ms) + per-task trace analysis.
# usage: bench/chol_round.sh [config] ; output under gpurun_out/
cd $GRAFT_REPO_ROOT
C=${1:-C3}
python bench.py --config $C --steps 10 --warmup 2 --no-cpu-baseline --no-solve 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['kernel_ms'])"
DBAT_AMD_LIB=prof DBAT_HIP_DF_TRACE=/tmp/df_trace.csv python bench.py --config $C --steps 2 --warmup 1 --no-cpu-baseline --no-solve > /dev/null 2>&1
python bench/chol_trace.py /tmp/df_trace.csv
