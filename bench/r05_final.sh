# end of round 5: the whole GPU suite, the default bench line, smoke
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_final
python -m pytest tests -q -m gpu 2>&1 | tail -4 > gpurun_out/r05_final/pytest_gpu.txt; cat gpurun_out/r05_final/pytest_gpu.txt
python bench.py > gpurun_out/r05_final/bench_default.json 2> gpurun_out/r05_final/bench_default.err; echo "bench rc=$?"; tail -c 600 gpurun_out/r05_final/bench_default.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
