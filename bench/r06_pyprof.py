#!/usr/bin/env python
"""cProfile of the third bundle(s, 'lm') on the same structure (cached handle): the host side of a call, function by function."""
import sys, cProfile, pstats, io
sys.path.insert(0, '.')
from dbat_amd import bundle, bundle_cov, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else 'C3'
s, _ = synth.make_scene(cfg)
for _ in range(2): r = bundle(s, 'lm')
pr = cProfile.Profile(); pr.enable()
r = bundle(s, 'lm')
pr.disable()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats('cumulative').print_stats(28); print(st.getvalue()[:6000])
pr = cProfile.Profile(); pr.enable()
C = bundle_cov(r[0], r[4], 'CEO')
pr.disable()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats('cumulative').print_stats(14); print(st.getvalue()[:3500])
