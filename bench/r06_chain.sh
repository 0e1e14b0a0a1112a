# the chain role clock by clock (C3), with the clocks inside df_potf2; C4: what the operand fetches cost today
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_chain; mkdir -p $O
for c in ${CFGS:-C3}; do
DBAT_AMD_LIB=prof DBAT_HIP_DF_TRACE_POTF2=1 DBAT_HIP_DF_TRACE=$O/trace_$c.csv python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-solve > /dev/null 2>$O/err_$c.txt
python bench/chain_trace.py $O/trace_$c.csv > $O/chain_$c.txt 2>&1
head -14 $O/chain_$c.txt
done
for a in 0 2 3; do python bench/quick.py C4 DBAT_AMD_LIB=prof DBAT_HIP_DF_ABLATE=$a 2>&1 | grep "^C4"; done | tee $O/c4_ablate.txt
