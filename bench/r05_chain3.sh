cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_chain3; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "step_parity or cholesky or product_switches or camcal_known" > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt
tail -3 $O/pytest.txt
for c in C1 C2 C3 C4; do
  timeout 300 python bench/quick.py $c DBAT_HIP_DF_CHAIN=0
  timeout 300 python bench/quick.py $c
done 2>&1 | tee $O/quick.txt
CFGS="C3 C1" bash bench/r05_chain2.sh
