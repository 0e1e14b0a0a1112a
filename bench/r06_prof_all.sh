# round 6 profiles: the synthetic configurations, the reference's demo projects and the dense-visibility scene -- kernel trace +
# PMC passes (separate runs); the summaries are written by bench/summarise_config.py on the box into gpurun_out/r06_summaries
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_summaries
for c in ${CFGS:-C3 C1 C2 C4 camcal roma roma-selfcal dense:48x16384}; do
  n=$(echo $c | tr A-Z a-z | tr ':' '_')
  STEPS=5 bash bench/prof_config.sh $c prof_r06_$n r06 sq 2>&1 | tail -2
  python bench/summarise_config.py gpurun_out/prof_r06_$n r06 $c gpurun_out/r06_summaries/r06_$n "Round 6, $c" > /dev/null 2>&1
  cp profiles/traffic.json gpurun_out/r06_summaries/traffic.json
done
ls gpurun_out/r06_summaries
