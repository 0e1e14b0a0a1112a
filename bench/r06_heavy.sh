# round 6: the matrix-core path of the heavy / giant points -- its tests, the randomised sweep, the camcal bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_heavy; mkdir -p $O
python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "every_point or giant or mixed or camcal" 2>&1 | tail -8
( timeout 1200 python bench/fuzz_irregular.py ${1:-60} ${2:-8001} ) > $O/fuzz_irregular.txt 2>&1; echo "fuzz_irregular rc=$?"; tail -3 $O/fuzz_irregular.txt; grep -c "k_heavy\|heavy" $O/fuzz_irregular.txt
grep DISAGREES $O/fuzz_irregular.txt | head
