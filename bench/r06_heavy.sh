# round 6: the matrix-core path of the heavy / giant points -- its tests, the randomised sweep, the dense-scene bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_heavy; mkdir -p $O
python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "every_point or giant or mixed or camcal or determin" 2>&1 | tail -3
( timeout 1200 python bench/fuzz_irregular.py ${1:-40} ${2:-8601} ) > $O/fuzz_irregular.txt 2>&1; echo "fuzz_irregular rc=$?"; tail -1 $O/fuzz_irregular.txt
( FUZZ_IRREGULAR=1 timeout 1200 python bench/fuzz_det.py 20 8701 ) > $O/fuzz_det_irregular.txt 2>&1; echo "det rc=$?"; tail -1 $O/fuzz_det_irregular.txt
for c in dense:48x16384 camcal; do DBAT_AMD_LIB=prof DBAT_HIP_ABLATE=32 python bench.py --config $c --no-cpu-baseline --no-solve --steps 3 --warmup 1 2>&1 | grep "heavy_z prof" | tail -1; done
bash bench/r06_dense.sh 2>&1 | grep -v passed
