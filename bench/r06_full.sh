# the whole GPU suite, smoke, the default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06_full}; mkdir -p $O
( time python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$O/bench_default.json')); print({k:d[k] for k in ('value','ms_per_step','ms_build_schur','ms_factor_solve','ms_backsub','ms_trial_residual','kernel_ms')}); print(d['roofline']['frac'], d['cpu_baseline']['value'] if d['cpu_baseline'] else None, d['solve_it_s'])"
