echo "PC16 NBUF2"
timeout 300 python bench.py --config C3 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernel_ms'])"
cp dbat_amd/libdbat_hip.so /tmp/keep.so; cp dbat_amd/libdbat_hip_pc8.so dbat_amd/libdbat_hip.so
echo "PC8 NBUF4"
timeout 300 python -m pytest tests -m gpu -x -q -k "step_parity or C1_full" 2>&1 | grep -E "passed|failed" | head -2
timeout 300 python bench.py --config C3 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernel_ms'])"
cp /tmp/keep.so dbat_amd/libdbat_hip.so
