# Profile of ONE configuration on the GPU box: kernel trace, HBM traffic (FETCH_SIZE / WRITE_SIZE in
# separate --pmc passes), matrix-core activity, and the vector / LDS / wait counters of the SQ.
#   bash bench/prof_config.sh C3 prof_r03_c3 r03 [sq]
# Output under gpurun_out/<dir>; bench/summarise_config.py turns it into profiles/<prefix>_<cfg>_*.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
C=${1:-C3}
O=$R/gpurun_out/${2:-prof_r03_$C}
P=${3:-r03}
SQ=${4:-sq}
ST=${STEPS:-5}
mkdir -p $O
cd $R
B="python3 bench.py --config $C --no-cpu-baseline --no-solve"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o $P -- $B --steps $ST --warmup 2 > $O/bench_stdout.json 2> $O/kt.err; echo kt rc=$?
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o ${P}_fetch -- $B --steps 2 --warmup 1 > /dev/null 2> $O/fetch.err; echo fetch rc=$?
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o ${P}_write -- $B --steps 2 --warmup 1 > /dev/null 2> $O/write.err; echo write rc=$?
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -o ${P}_mfma -- $B --steps 2 --warmup 1 > /dev/null 2> $O/mfma.err; echo mfma rc=$?
if [ "$SQ" = "sq" ]; then
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d $O/sq1 -o ${P}_sq1 -- $B --steps 2 --warmup 1 > /dev/null 2> $O/sq1.err; echo sq1 rc=$?
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA --output-format csv -d $O/sq2 -o ${P}_sq2 -- $B --steps 2 --warmup 1 > /dev/null 2> $O/sq2.err; echo sq2 rc=$?
  rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_SMEM --output-format csv -d $O/sq3 -o ${P}_sq3 -- $B --steps 2 --warmup 1 > /dev/null 2> $O/sq3.err; echo sq3 rc=$?
fi
python bench.py --config $C --no-cpu-baseline > $O/bench_line.json 2> $O/bench_line.err; echo bench rc=$?
find $O -name "*kernel_trace.csv" -size +20M -delete
find $O -name "*counter_collection.csv" -size +20M -delete
find $O -name "*agent_info.csv" -delete
du -sh $O
