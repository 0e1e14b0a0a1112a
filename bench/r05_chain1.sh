# round 5, first contact of the chain role: parity subset, then kernel times with and without it
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_chain1; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "step_parity or cholesky or product_switches or camcal_known" > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt
tail -5 $O/pytest.txt
for c in C1 C2 C3 C4; do
  timeout 300 python bench/quick.py $c DBAT_HIP_DF_CHAIN=0
  timeout 300 python bench/quick.py $c
  timeout 300 python bench/quick.py $c DBAT_HIP_DF_CHAIN_WG=8
  timeout 300 python bench/quick.py $c DBAT_HIP_DF_CHAIN_WG=64
  timeout 300 python bench/quick.py $c DBAT_HIP_DF_L2=1
  timeout 300 python bench/quick.py $c DBAT_HIP_DF_L2=1 DBAT_HIP_DF_CHAIN=0
done 2>&1 | tee $O/quick.txt
DBAT_HIP_PLAN_STATS=1 timeout 300 python bench/quick.py C3 2>&1 | tail -5 | tee -a $O/quick.txt
