cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "step_parity or all_dampings or product_switches or deterministic" 2>&1 | tail -3
for c in C3 C1 C2 C4; do timeout 300 python bench/quick.py $c; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -E "icache|ifetch|SQC_INST|INST_CACHE" | head -20 > $GRAFT_REPO_ROOT/gpurun_out/icache_counters.txt
cd $GRAFT_REPO_ROOT
