# kernel-trace averages of the self-calibrating side kernels at C4 / C2
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_camn
for c in ${@:-C4 C2}; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06_camn/$c -o kt -- python3 bench.py --config $c --no-cpu-baseline --no-solve --steps 5 --warmup 2 > gpurun_out/r06_camn/$c.json 2>/dev/null
python3 - <<PY
import csv, glob
f = glob.glob('gpurun_out/r06_camn/$c/**/kt_kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r['Name']
    if any(k in n for k in ('k_cam_normal', 'k_backsub_sig', 'k_build_sig', 'k_chol_df', 'k_backsub<')):
        print('$c', n[:60], r['Calls'], 'avg us %.1f' % (float(r['AverageNs']) / 1e3))
PY
done
