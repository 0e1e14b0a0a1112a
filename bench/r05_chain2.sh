cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_chain2; mkdir -p $O
for c in ${CFGS:-C3 C1}; do
DBAT_AMD_LIB=prof DBAT_HIP_DF_TRACE=$O/trace_$c.csv python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-solve > /dev/null 2>$O/err_$c.txt
python bench/chain_trace.py $O/trace_$c.csv all > $O/chain_$c.txt 2>&1
head -12 $O/chain_$c.txt
python bench/chol_path.py $O/trace_$c.csv 2>&1 | tail -22
done
