for c in C1 C2 C3; do
  echo "$c"
  DBAT_HIP_PLAN_STATS=1 timeout 300 python bench.py --config $c --steps 10 --warmup 2 2>/tmp/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernel_ms'], d['config']['n_tiles'])"
  grep "plan\]" /tmp/err.txt | head -2
done
for bm in 2 8; do echo "C1 bmax $bm";  DBAT_HIP_TILE_BMAX=$bm timeout 300 python bench.py --config C1 --steps 10 --warmup 2 2>/tmp/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernel_ms'], d['config']['n_tiles'])"; done
for bm in 4 16; do echo "C2 bmax $bm";  DBAT_HIP_TILE_BMAX=$bm timeout 300 python bench.py --config C2 --steps 10 --warmup 2 2>/tmp/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernel_ms'], d['config']['n_tiles'])"; done
