# the randomised sweeps of round 5 (new seeds), with the chain role of the factorisation squeezed to one / two workgroups
# now and then (ticket starts, hand-overs)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_fuzz; mkdir -p $O
( timeout 900 python bench/fuzz_step.py 40 7001 ) > $O/fuzz_step.txt 2>&1; echo "fuzz_step rc=$?"; tail -2 $O/fuzz_step.txt
( DBAT_HIP_DF_CHAIN_WG=1 timeout 900 python bench/fuzz_step.py 30 7101 ) > $O/fuzz_step_wg1.txt 2>&1; echo "fuzz_step WG=1 rc=$?"; tail -1 $O/fuzz_step_wg1.txt
( DBAT_HIP_DF_CHAIN=1 DBAT_HIP_DF_CHAIN_WG=2 timeout 900 python bench/fuzz_step.py 30 7201 ) > $O/fuzz_step_wg2.txt 2>&1; echo "fuzz_step WG=2 rc=$?"; tail -1 $O/fuzz_step_wg2.txt
( timeout 900 python bench/fuzz_irregular.py 40 7301 ) > $O/fuzz_irregular.txt 2>&1; echo "fuzz_irregular rc=$?"; tail -2 $O/fuzz_irregular.txt
( timeout 900 python bench/fuzz_solve.py 16 7401 ) > $O/fuzz_solve.txt 2>&1; echo "fuzz_solve rc=$?"; tail -2 $O/fuzz_solve.txt
( timeout 900 python bench/fuzz_multishard.py 30 7501 ) > $O/fuzz_multishard.txt 2>&1; echo "fuzz_multishard rc=$?"; tail -2 $O/fuzz_multishard.txt
( timeout 900 python bench/fuzz_det.py 60 7601 ) > $O/fuzz_det.txt 2>&1; echo "fuzz_det rc=$?"; tail -2 $O/fuzz_det.txt
