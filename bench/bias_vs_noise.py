#!/usr/bin/env python
"""Self-calibration at full size: estimate minus truth against the image noise (GPU box).

DBAT's residual is formed in the CORRECTED image space, v = pinhole(Q) - brown(u_measured; K, P)
(res_euler_brown_1.m:84-95): the lens correction is applied to the noisy measurement and its
coefficients are unknowns, so the Jacobian wrt K, P, pp is evaluated at noisy coordinates -- an
errors-in-variables estimator, whose bias does not shrink with the number of observations while the
posterior standard deviation does.  This script separates that from a solver error: with noise 0 the
adjustment must return the generator's truth to rounding; the offset must grow with the square of the
noise.

    python bench/bias_vs_noise.py C2 0 0.125 0.25 0.5 1
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from dbat_amd import _hip, bundle, bundle_cov, synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'C2'
noises = [float(a) for a in sys.argv[2:]] or [0.0, 0.25, 0.5]
print('| config | noise px | iters | sigma0 | cc - truth (mm) per IO block | K1 - truth | sigma_cc | max |EO pos - truth| (m) |')
print('|---|---|---|---|---|---|---|---|')
for nz in noises:
    s, truth = synth.make_scene(name, noise_px=nz)
    fixed = ~np.asarray(s.bundle.est.EO, bool)[:6]
    s.EO.val[:6][fixed] = truth['EO'][fixed]
    res, ok, iters, s0, E = bundle(s, 'lm', store_trace=False)
    blocks = np.unique(s.IO.struct.block[0])
    lead = [int(np.flatnonzero(s.IO.struct.block[0] == b)[0]) for b in blocks]
    sd = None
    if nz > 0:
        CIO, = bundle_cov(res, E, 'CIO'),
        sd = np.sqrt(CIO.diagonal()).reshape(res.IO.val.shape, order='F')[0, lead]
    h = _hip.Handle(s)          # the last iterate, whatever the status code (noise 0: the relative test never fires)
    IOe, EOe, OPe = h.deserialize(E.x)
    h.close()
    d_cc = IOe[0, lead] - truth['IO'][0, lead]
    d_k1 = IOe[5, lead] - truth['IO'][5, lead]
    print('| %s | %g | %d | %.6g | %s | %s | %s | %.3g |' % (
        name, nz, iters, s0, np.array2string(d_cc, precision=3, floatmode='maxprec'), np.array2string(d_k1, precision=3),
        np.array2string(sd, precision=3) if sd is not None else '-', np.abs(EOe[:3] - truth['EO'][:3]).max()), flush=True)
