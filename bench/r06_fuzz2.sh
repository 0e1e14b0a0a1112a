cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_fuzz; mkdir -p $O
( FUZZ_IRREGULAR=1 timeout 1200 python bench/fuzz_multishard.py 30 8201 ) > $O/fuzz_multishard_irregular.txt 2>&1; echo "multishard rc=$?"; tail -2 $O/fuzz_multishard_irregular.txt
( FUZZ_IRREGULAR=1 timeout 1200 python bench/fuzz_det.py 40 8301 ) > $O/fuzz_det_irregular.txt 2>&1; echo "det rc=$?"; tail -2 $O/fuzz_det_irregular.txt
( timeout 900 python bench/fuzz_step.py 30 8401 ) > $O/fuzz_step.txt 2>&1; echo "fuzz_step rc=$?"; tail -1 $O/fuzz_step.txt
( timeout 900 python bench/fuzz_solve.py 12 8501 ) > $O/fuzz_solve.txt 2>&1; echo "fuzz_solve rc=$?"; tail -1 $O/fuzz_solve.txt
