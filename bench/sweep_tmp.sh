for a in 32 96 160 224; do
  echo "ablate $a"
  DBAT_HIP_ABLATE=$a timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/tmp/e.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['kernel_ms']['k_build_tile2'])"
  grep "tile2 prof" /tmp/e.txt | tail -1
done
