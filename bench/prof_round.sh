# Round profile on the GPU box: kernel trace, HBM traffic (PMC, separate passes), matrix-core
# activity, bench lines of all configs.  Output under gpurun_out/<dir>; summarise with
# bench/summarise_profiles.py into profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-prof_r02}
P=${2:-r02}
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o $P -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-solve > $O/bench_stdout.json 2> $O/kt.err; echo kt rc=$?
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o ${P}_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-solve > /dev/null 2> $O/fetch.err; echo fetch rc=$?
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o ${P}_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-solve > /dev/null 2> $O/write.err; echo write rc=$?
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -o ${P}_mfma -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-solve > /dev/null 2> $O/mfma.err; echo mfma rc=$?
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo bench rc=$?
python bench.py --config C2 --no-cpu-baseline > $O/bench_c2.json 2>/dev/null; echo c2 rc=$?
python bench.py --config C1 --no-cpu-baseline > $O/bench_c1.json 2>/dev/null; echo c1 rc=$?
python bench.py --config C4 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_c4.json 2>/dev/null; echo c4 rc=$?
DBAT_BENCH_FORCE_COMM=1 python bench.py --no-cpu-baseline > $O/bench_comm1.json 2>/dev/null; echo comm rc=$?
find $O -name "*kernel_trace.csv" -size +20M -delete
find $O -name "*counter_collection.csv" -size +20M -delete
ls -la $O $O/*/* | head -40
