# C4: HBM/fabric read traffic of k_chol_df with the tiles read by agent-scope loads (default) and through the L2 (DBAT_HIP_DF_L2=1)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_c4_fetch; rm -rf $O; mkdir -p $O; cd $R
for l2 in 0 1; do
  export DBAT_HIP_DF_L2=$l2
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f$l2 -o f -- python3 bench.py --config C4 --steps 2 --warmup 1 --no-cpu-baseline --no-solve > /dev/null 2> $O/err$l2.txt
  python3 - <<PY
import csv, glob
f = glob.glob('$O/f$l2/**/*counter_collection.csv', recursive=True)[0]
tot = {}; n = {}
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'][:40]
    tot[k] = tot.get(k, 0.0) + float(r['Counter_Value']); n[k] = n.get(k, 0) + 1
for k in sorted(tot, key=lambda k: -tot[k])[:6]:
    print('L2=$l2  %-42s launches %3d  FETCH_SIZE per launch %.1f MB (raw counter, KB units x 1024)' % (k, n[k], tot[k] / n[k] * 1024 / 1e6))
PY
done 2>&1 | tee $O/summary.txt
find $O -name "*.csv" -size +5M -delete
