#!/bin/bash
# time k_build_sig with parts switched off (results are wrong, only the kernel time matters)
for a in 0 1 2 3 4 7 8 16 31; do
  DBAT_AMD_LIB=prof DBAT_HIP_ABLATE=$a python bench.py --no-cpu-baseline --no-solve --steps 10 --warmup 2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('ablate $a', {k: round(v, 3) for k, v in j['kernel_ms'].items()}, 'build', round(j['ms_build_schur'], 3))
"
done
