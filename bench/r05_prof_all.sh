# round 5 profiles: C3 (default bench), C1, C2, C4 -- kernel trace + PMC passes (separate runs), summaries under gpurun_out/
cd $GRAFT_REPO_ROOT
for c in ${CFGS:-C3 C1 C2 C4}; do
  STEPS=5 bash bench/prof_config.sh $c prof_r05_$c r05 sq 2>&1 | tail -3
  mkdir -p gpurun_out/r05_summaries
  python bench/summarise_config.py gpurun_out/prof_r05_$c r05 $c gpurun_out/r05_summaries/r05_$(echo $c | tr A-Z a-z) "Round 5, $c" 2>&1 | tail -2
done
ls gpurun_out/r05_summaries
