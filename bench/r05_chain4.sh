cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_chain4; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_multishard_gpu.py -x -q -m gpu -k "step_parity or cholesky or product_switches or camcal_known or shard or rank" > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt
tail -3 $O/pytest.txt
for c in C1 C2 C3 C4; do
  timeout 300 python bench/quick.py $c DBAT_HIP_DF_CHAIN=0
  timeout 300 python bench/quick.py $c DBAT_HIP_PLAN_STATS=1
done 2>&1 | grep -v "^\[plan\]" | grep -v "cameras per\|chol all\|order " | tee $O/quick.txt
timeout 300 python bench/quick.py C4 DBAT_HIP_DF_CHAIN=1 DBAT_HIP_DF_CHAIN_WG=8 | tee -a $O/quick.txt
timeout 300 python bench/quick.py C4 DBAT_HIP_DF_CHAIN=1 DBAT_HIP_DF_CHAIN_WG=16 | tee -a $O/quick.txt
timeout 300 python bench/quick.py C3 DBAT_HIP_DF_CHAIN_WG=12 | tee -a $O/quick.txt
timeout 300 python bench/quick.py C3 DBAT_HIP_DF_CHAIN_WG=64 | tee -a $O/quick.txt
for r in 8; do timeout 300 python bench.py --config C3 --emulate-ranks $r --steps 10 --no-cpu-baseline --no-solve 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('C3 emulate 8:', d['ms_per_step'], {k:v for k,v in d.items() if k.startswith('ms_')}, d.get('kernel_ms'))"; done | tee -a $O/quick.txt
DBAT_HIP_DF_CHAIN=0 timeout 300 python bench.py --config C3 --emulate-ranks 8 --steps 10 --no-cpu-baseline --no-solve 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('C3 emulate 8 (chain off):', d['ms_per_step'], {k:v for k,v in d.items() if k.startswith('ms_')}, d.get('kernel_ms'))" | tee -a $O/quick.txt
CFGS="C3" bash bench/r05_chain2.sh | head -9
