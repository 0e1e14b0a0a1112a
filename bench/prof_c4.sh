cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_c4
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o c4 -- python3 bench.py --config C4 --steps 3 --warmup 1 --no-cpu-baseline --no-solve > $O/bench.json 2> $O/err.txt
python3 - <<'PY'
import csv, glob
f = glob.glob('/root/repo/gpurun_out/prof_c4/kt/**/c4_kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r['Name'][:70], r['Calls'], round(int(r['TotalDurationNs'])/1e6,2), round(float(r['AverageNs'])/1e3,1))
PY
