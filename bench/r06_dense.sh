cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_dense; mkdir -p $O
python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "every_point or giant or mixed or camcal_known or determin" 2>&1 | tail -3
for c in ${@:-camcal dense:48x4096 dense:48x16384 dense:96x16384}; do
  python bench.py --config $c --no-cpu-baseline --no-solve --steps 5 --warmup 2 > $O/b.json 2> $O/b.err || tail -3 $O/b.err
  python - <<PY
import json
d = json.loads(open('$O/b.json').read().strip().splitlines()[-1])
r = d['roofline']
print('$c: step %.3f ms build %.3f factor %.3f | %s' % (d['ms_per_step'], d['ms_build_schur'], d['ms_factor_solve'], {k: round(v, 4) for k, v in d['kernel_ms'].items()}))
print('    roofline %s: %.2f TF algorithmic (%.3f of peak), executed %.3f of peak' % (r['kernel'], r['achieved'], r['frac'], r['executed_frac'] or 0))
PY
done
