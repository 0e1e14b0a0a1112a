# the whole GPU suite, smoke, the default bench line, the four configs' step breakdown
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_final2; mkdir -p $O
bash bench/r06_full.sh r06_final2
for c in C1 C2 C4; do python bench/quick.py $c 2>&1 | tail -4; done > $O/quick.txt 2>&1; cat $O/quick.txt
