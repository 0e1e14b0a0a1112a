for c in 12 16 18 21; do
  echo "CMAX=$c"
  DBAT_HIP_CMAX=$c timeout 300 python bench.py --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernel_ms'], d['config']['n_tiles'])"
done
