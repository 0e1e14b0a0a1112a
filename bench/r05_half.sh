# half tiles (two four-wave workgroups per CU) against the eight-wave kernel, C3 and C1
cd $GRAFT_REPO_ROOT
for c in C3 C1; do
  timeout 300 python bench/quick.py $c
  timeout 300 python bench/quick.py $c DBAT_HIP_CMAX=14 DBAT_HIP_SIG_HALF=0
  timeout 300 python bench/quick.py $c DBAT_HIP_CMAX=14
  timeout 300 python bench/quick.py $c DBAT_HIP_CMAX=12
done
DBAT_HIP_CMAX=14 timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "step_parity or all_dampings or product_switches" 2>&1 | tail -3
( timeout 900 python bench/fuzz_irregular.py 40 7301 ) > gpurun_out/r05_fuzz/fuzz_irregular.txt 2>&1; echo "fuzz_irregular rc=$?"; tail -2 gpurun_out/r05_fuzz/fuzz_irregular.txt
