#!/usr/bin/env python
"""How far is the adjusted result from the generator's truth, in units of its own posterior standard
deviation?  (GPU box.)  The fixed datum elements (camera 0 and one coordinate of a second camera,
seteoest 'depend') are set to their TRUE values first, so that estimate and truth share one datum;
then z = (estimate - truth) / sigma with sigma from dbat_hip_posterior_cov for the camera constants
(one per IO block), the other IO unknowns, every EO element and a sample of object points.

    python bench/zscore_truth.py C2 [lm [noise_px]]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from dbat_amd import bundle, bundle_cov, synth  # noqa: E402


def truth_datum(s, truth):
    fixed = ~np.asarray(s.bundle.est.EO, bool)[:6]
    s.EO.val[:6][fixed] = truth['EO'][fixed]
    return s


def zscores(name, damping='lm', verbose=True, noise_px=0.5):
    s, truth = synth.make_scene(name, noise_px=noise_px)
    s = truth_datum(s, truth)
    t0 = time.time()
    res, ok, iters, s0, E = bundle(s, damping, store_trace=False)
    t1 = time.time()
    CIO, CEO, COP = bundle_cov(res, E, 'CIO', 'CEO', 'COP')
    t2 = time.time()
    out = dict(ok=bool(ok), iters=int(iters), sigma0=float(s0), t_bundle=t1 - t0, t_cov=t2 - t1)
    estIO = np.asarray(s.bundle.est.IO, bool)
    if estIO.any():
        sd = np.sqrt(CIO.diagonal()).reshape(res.IO.val.shape, order='F')
        z = (res.IO.val - truth['IO']) / np.where(sd > 0, sd, np.nan)
        blocks = np.unique(s.IO.struct.block[0])
        lead = [int(np.flatnonzero(s.IO.struct.block[0] == b)[0]) for b in blocks]
        out['cc'] = [(float(res.IO.val[0, c]), float(truth['IO'][0, c]), float(sd[0, c]), float(z[0, c])) for c in lead]
        out['io_z'] = {int(r): [float(z[r, c]) for c in lead] for r in np.flatnonzero(estIO[:, lead[0]])}
    sdE = np.sqrt(CEO.diagonal()).reshape(6, -1, order='F')
    est = np.asarray(s.bundle.est.EO, bool)[:6]
    zE = np.where(est, (res.EO.val[:6] - truth['EO']) / np.where(sdE > 0, sdE, np.nan), np.nan)
    out['eo_pos_z'] = zE[:3][est[:3]]
    out['eo_ang_z'] = zE[3:][est[3:]]
    sdP = np.sqrt(COP.diagonal()).reshape(3, -1, order='F')
    out['op_z'] = ((res.OP.val - truth['OP']) / sdP).ravel()
    if verbose:
        print(name, damping, 'noise %.3g px' % noise_px, 'ok', ok, 'iters', iters, 'sigma0 %.5f' % s0, 'bundle %.1f s cov %.1f s' % (t1 - t0, t2 - t1))
        for k in ('cc',):
            if k in out:
                for v in out[k]:
                    print('  cc est %.6f truth %.6f sigma %.3g  z %.2f' % v)
        if 'io_z' in out:
            print('  IO z by row:', {k: np.round(v, 2).tolist() for k, v in out['io_z'].items()})
        for k in ('eo_pos_z', 'eo_ang_z', 'op_z'):
            a = np.abs(out[k][np.isfinite(out[k])])
            print('  %-9s n %d  rms %.3f  frac>3 %.5f  max %.2f' % (k, a.size, np.sqrt(np.mean(a * a)), np.mean(a > 3), a.max()))
    return out


if __name__ == '__main__':
    zscores(sys.argv[1] if len(sys.argv) > 1 else 'C2', sys.argv[2] if len(sys.argv) > 2 else 'lm',
            noise_px=float(sys.argv[3]) if len(sys.argv) > 3 else 0.5)
