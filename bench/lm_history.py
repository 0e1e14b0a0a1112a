#!/usr/bin/env python
"""Where do the GPU's and the oracle's Levenberg-Marquardt histories part?  Prints, per
iteration, the relative difference of the iterates (E.trace), of the residual norms and of
lambda, for the ill-conditioned self-calibrating cases of tests/test_hip_parity.py."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np
import dbat_oracle as o
from dbat_amd import bundle
from helpers import camcal_struct, synth_struct, relerr

cases = [('camcal3', lambda: camcal_struct(3)), ('tiny-selfcal', lambda: synth_struct('tiny', 'selfcal')[0]),
         ('tiny-groups4', lambda: synth_struct('tiny', 'groups4')[0]), ('tiny-plain', lambda: synth_struct('tiny', 'plain')[0])]
for name, make in cases:
    s = make()
    res, ok, it, s0, E = bundle(s, 'lm')
    ro, oko, ito, s0o, Eo = o.bundle(s, 'lm')
    T, To = np.asarray(E.trace), np.asarray(Eo.trace)
    lam, lamo = E.damping.__dict__['lambda'], Eo.damping.__dict__['lambda']
    print('%s: gpu %d iterations, oracle %d; final relerr x %.2e' % (name, it, ito, relerr(E.x, Eo.x)))
    n = min(T.shape[1], To.shape[1])
    for k in range(n):
        print('  it %2d  x %.2e  res %.2e (%.10g | %.10g)  lambda %.2e (%.3g | %.3g)'
              % (k, relerr(T[:, k], To[:, k]), abs(E.res[k] - Eo.res[k]) / Eo.res[k] if k < min(len(E.res), len(Eo.res)) else np.nan,
                 E.res[k] if k < len(E.res) else np.nan, Eo.res[k] if k < len(Eo.res) else np.nan,
                 abs(lam[k] - lamo[k]) / max(lamo[k], 1e-300) if k < min(len(lam), len(lamo)) else np.nan,
                 lam[k] if k < len(lam) else np.nan, lamo[k] if k < len(lamo) else np.nan))
