# the same sweeps on the round's last kernels, other seeds
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_fuzz2; mkdir -p $O
( timeout 900 python bench/fuzz_step.py 40 8001 ) > $O/fuzz_step.txt 2>&1; echo "fuzz_step rc=$?"; tail -1 $O/fuzz_step.txt
( timeout 900 python bench/fuzz_irregular.py 40 8301 ) > $O/fuzz_irregular.txt 2>&1; echo "fuzz_irregular rc=$?"; tail -1 $O/fuzz_irregular.txt
( timeout 900 python bench/fuzz_solve.py 16 8401 ) > $O/fuzz_solve.txt 2>&1; echo "fuzz_solve rc=$?"; tail -1 $O/fuzz_solve.txt
( timeout 900 python bench/fuzz_multishard.py 30 8501 ) > $O/fuzz_multishard.txt 2>&1; echo "fuzz_multishard rc=$?"; tail -1 $O/fuzz_multishard.txt
( timeout 900 python bench/fuzz_det.py 40 8601 ) > $O/fuzz_det.txt 2>&1; echo "fuzz_det rc=$?"; tail -1 $O/fuzz_det.txt
