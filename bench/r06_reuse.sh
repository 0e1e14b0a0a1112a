cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_reuse; mkdir -p $O
python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "plan_reuse or cached_handle" 2>&1 | tail -8
python -m pytest tests -m "not gpu" -x -q 2>&1 | tail -3
for c in C1 C3; do python bench/plan_reuse.py $c > $O/reuse_$c.json 2> $O/reuse_$c.err; tail -c 1800 $O/reuse_$c.json; echo; tail -3 $O/reuse_$c.err; done
