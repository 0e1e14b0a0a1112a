cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_final; mkdir -p $O
for c in roma roma-selfcal camcal sxb; do timeout 600 python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err; echo "$c rc=$?"; done
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
python - <<'PY'
import json
for c in ('roma','roma-selfcal','camcal','sxb','default'):
    try:
        d=json.loads(open('gpurun_out/r05_final/bench_%s.json'%c).read().strip().splitlines()[-1])
    except Exception as e:
        print(c,'FAILED',e); continue
    sr=d.get('solve_reference_demo') or {}
    cb=d.get('cpu_baseline') or {}
    print(c, 'value %.1f it/s ms/step %.3f' % (d['value'], d['ms_per_step']), 'kernels', {k: round(v,4) for k,v in d['kernel_ms'].items()},
          '| demo solve:', {k: sr.get(k) for k in ('damping','iterations','time_s','sigma0','it_per_s')}, '| cpu', cb.get('value'), cb.get('cores'))
PY
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.txt
