#!/bin/bash
# time k_build_sig with parts switched off (results are wrong, only the kernel time matters): bits 1 products (and operand
# reads), 2 pass-2 evaluation, 4 pass 1 with one camera, 8 chunk flush, 16 tile flush, 64 no pass 2, 128 no camera in pass 1,
# 256 no gathers in pass 2, 512 no panel writes
for a in ${@:-0 1 2 3 4 7 8 16 31 64 192 223 256 512 771}; do
  DBAT_AMD_LIB=prof DBAT_HIP_ABLATE=$a python bench.py --config ${CFG:-C3} --no-cpu-baseline --no-solve --steps 10 --warmup 2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('ablate $a', {k: round(v, 3) for k, v in j['kernel_ms'].items()}, 'build', round(j['ms_build_schur'], 3))
"
done
