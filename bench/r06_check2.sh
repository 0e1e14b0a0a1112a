cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_fuzz; mkdir -p $O
( time python -m pytest tests/test_fullsize_parity.py -m gpu -x -q -k "oracles_full_sparse" ) 2>&1 | tail -5
( FUZZ_IRREGULAR=1 timeout 1200 python bench/fuzz_det.py 30 8701 ) > $O/fuzz_det_irregular2.txt 2>&1; echo "det rc=$?"; tail -1 $O/fuzz_det_irregular2.txt
