# every rank's kernels on this round's code -- one GPU plays rank 0 of R (collectives as no-ops)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_emulate; mkdir -p $O
for c in C3 C4; do
  for r in 1 2 4 8; do
    if [ $r = 1 ]; then python bench.py --config $c --no-cpu-baseline --no-solve --steps 10 > $O/${c}_r$r.json 2> $O/${c}_r$r.err
    else python bench.py --config $c --no-cpu-baseline --no-solve --steps 10 --emulate-ranks $r > $O/${c}_r$r.json 2> $O/${c}_r$r.err; fi
    python - <<PY
import json
d = json.loads(open('$O/${c}_r$r.json').read().strip().splitlines()[-1])
m = d.get('multi_gpu') or {}
km = d['kernel_ms']
print('$c ranks $r: step %.3f ms  build %.3f  factor %.3f (domain %.3f top %.3f)  backsub %.3f  residual %.3f | tile kernel %.3f  obs %s  allreduce bytes %s' % (
    d['ms_per_step'], d['ms_build_schur'], d['ms_factor_solve'], m.get('ms_factor_domain', 0), m.get('ms_replicated', 0), d['ms_backsub'], d['ms_trial_residual'],
    list(km.values())[0], m.get('obs_this_rank'), m.get('allreduce_bytes_reduced_system')))
if m.get('per_rank_roofline'): print('   ', m['per_rank_roofline'][0])
PY
  done
done
python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "two_processes" 2>&1 | tail -3
python bench/plan_reuse.py C3 2>/dev/null | tail -1
