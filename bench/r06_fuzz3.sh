# the randomised sweeps on the round's last code, new seeds
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_fuzz3; mkdir -p $O
( timeout 1200 python bench/fuzz_step.py 60 9401 ) > $O/fuzz_step.txt 2>&1; echo "fuzz_step rc=$?"; tail -1 $O/fuzz_step.txt
( timeout 900 python bench/fuzz_irregular.py 60 9101 ) > $O/fuzz_irregular.txt 2>&1; echo "fuzz_irregular rc=$?"; tail -1 $O/fuzz_irregular.txt
( timeout 900 python bench/fuzz_det.py 40 9301 ) > $O/fuzz_det.txt 2>&1; echo "det rc=$?"; tail -2 $O/fuzz_det.txt
( timeout 900 python bench/fuzz_multishard.py 30 9201 ) > $O/fuzz_multishard.txt 2>&1; echo "multishard rc=$?"; tail -2 $O/fuzz_multishard.txt
( timeout 900 python bench/fuzz_solve.py 16 9501 ) > $O/fuzz_solve.txt 2>&1; echo "fuzz_solve rc=$?"; tail -1 $O/fuzz_solve.txt
