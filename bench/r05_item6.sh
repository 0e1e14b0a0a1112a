cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_item6; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_fullsize_parity.py -x -q -m gpu -k "lm_iteration_count or full_size_model_sample or bench_two_processes" > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt
tail -5 $O/pytest.txt
timeout 300 python bench.py --config C1 --no-cpu-baseline > $O/c1.json 2> $O/c1.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_item6/c1.json').read().strip().splitlines()[-1])
print('C1 value', d['value'], 'solve', d['solve'])
print('roofline_step', d['roofline_step'])
PY
