# quick kernel-trace statistics of the bench step: bench/kt_quick.sh [config]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kt_quick; rm -rf $O; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt -- python3 bench.py --config ${1:-C3} --steps 10 --warmup 2 --no-cpu-baseline --no-solve > $O/out.json 2> $O/err.txt
python - <<'PY'
import csv,glob,os
f=glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/kt_quick/**/kt_kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:28]:
    print('%-60s %6s calls  avg %9.1f us  %5.1f %%' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage'])))
PY
find $O -name "*kernel_trace.csv" -size +20M -delete
