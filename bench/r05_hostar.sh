cd $GRAFT_REPO_ROOT
for n in 2 4; do
DBAT_BENCH_HOST_ALLREDUCE=1 timeout 600 python bench.py --gpus $n --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/hostar_r05_$n.json 2> gpurun_out/hostar_r05_$n.err; echo "n=$n rc=$?"; tail -1 gpurun_out/hostar_r05_$n.json | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d.get('value'), d.get('ms_per_step'), d.get('error'), (d.get('multi_gpu') or {}).get('per_rank_ms'), (d.get('solve') or {}).get('sigma0'))"
done
