#!/usr/bin/env python
"""bundle() five times on the same structure (C3 by default): where the time of every call goes (E.time inside the loop,
its hipEvent stages, the host's three parts)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from dbat_amd import bundle, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else 'C3'
s, _ = synth.make_scene(cfg)
for k in range(5):
    t0 = time.perf_counter()
    r, ok, it, s0, E = bundle(s, 'lm')
    w = time.perf_counter() - t0
    print('call %d: wall %.3f  E.time %.4f  iters %d  stages %s  host %s' % (
        k, w, E.time, it, {a: round(b * 1e3, 2) for a, b in E.timeStages.items()},
        {a: (round(b, 4) if not isinstance(b, bool) else b) for a, b in E.timeHost.items()}), flush=True)
    if len(sys.argv) > 2: time.sleep(float(sys.argv[2]))
