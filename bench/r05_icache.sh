# instruction cache of the big kernels: requests / hits / misses and the mean instruction-fetch latency
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_icache; rm -rf $O; mkdir -p $O; cd $R
for c in C4 C3; do
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $O/ic_$c -o ic -- python3 bench.py --config $c --no-cpu-baseline --no-solve --steps 2 --warmup 1 > /dev/null 2> $O/ic_$c.err
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/if_$c -o if -- python3 bench.py --config $c --no-cpu-baseline --no-solve --steps 2 --warmup 1 > /dev/null 2> $O/if_$c.err
done
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ['GRAFT_REPO_ROOT']
for f in sorted(glob.glob(R+'/gpurun_out/r05_icache/**/*counter_collection.csv', recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:48]
        acc[k][r['Counter_Name']]+=float(r['Counter_Value']); 
    print(f.split('r05_icache/')[1].split('/')[0])
    for k,v in sorted(acc.items(), key=lambda kv:-sum(kv[1].values()))[:6]:
        print('   %-50s %s' % (k, ' '.join('%s=%.3g' % (a,b) for a,b in sorted(v.items()))))
PY
find $O -name "*.csv" -size +5M -delete
