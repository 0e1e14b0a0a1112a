cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_pipe2_trace; mkdir -p $O
for l in prof $GRAFT_REPO_ROOT/dbat_amd/libdbat_hip_prof_old.so; do
n=$(basename $l .so)
DBAT_AMD_LIB=$l DBAT_HIP_DF_TRACE=$O/trace_$n.csv python bench.py --config C4 --steps 2 --warmup 1 --no-cpu-baseline --no-solve > /dev/null 2>$O/err_$n.txt
echo "== $n"; python bench/df_task_stats.py $O/trace_$n.csv
done 2>&1 | tee $O/log.txt
rm -f $O/*.csv
