cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_reuse2; mkdir -p $O
python -m pytest tests/test_hip_parity.py -m gpu -x -q -n 4 -k "camcal or roma or sxb or prague or priors or report or plan_reuse or cached_handle or post or trace or histor or lm or loop or failure or rank or shard or c_driver" 2>&1 | grep -E "passed|failed|FAILED" | tail -5
for c in C3; do python bench/plan_reuse.py $c > $O/reuse_$c.json 2> $O/reuse_$c.err; tail -c 2200 $O/reuse_$c.json; echo; tail -3 $O/reuse_$c.err; done
python bench/r06_repeat.py C3 2>&1 | grep "^call"
