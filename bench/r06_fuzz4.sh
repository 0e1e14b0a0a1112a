cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_fuzz3; mkdir -p $O
( FUZZ_KEEP_GOING=1 timeout 1500 python bench/fuzz_multishard.py 40 9201 ) > $O/fuzz_multishard.txt 2>&1; echo "multishard rc=$?"; grep -c "ranks" $O/fuzz_multishard.txt; grep "DISAGREES" $O/fuzz_multishard.txt; tail -1 $O/fuzz_multishard.txt
