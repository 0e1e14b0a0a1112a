# leaf-size sweep of the nested dissection: bench line + schedule statistics
cd $GRAFT_REPO_ROOT
C=${1:-C3}
for leaf in 16 24 32 48 64 96; do
  echo "== ND_LEAF $leaf"
  DBAT_HIP_ND_LEAF=$leaf DBAT_HIP_PLAN_STATS=1 python bench.py --config $C --steps 10 --warmup 2 --no-cpu-baseline --no-solve 2> /tmp/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['kernel_ms'])"
  grep "^\[chol\]" /tmp/err.txt
done
