"""Shared test helpers: known-answer project set-up and seeded synthetic scenes."""
import json
import os

import numpy as np

from dbat_amd import loadpm as L

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def camcal_struct(model=3):
    """demo/camcaldemo.m:56-100 set-up, starting from PhotoModeler's own EO/OP
    instead of resect/forwintersect (the converged solution is x0-independent
    inside the basin)."""
    prob = L.loadpm(os.path.join(GOLDEN, 'camcal-pmexport.txt'))
    s = L.prob2dbatstruct(prob, distModel=model)
    s.IO.val[0, :] = 7.3                                   # setcamvals 'default',7.3
    s.IO.val[1:3, :] = 0.5 * np.diag([1, -1]) @ s.IO.sensor.ssSize
    s.IO.val[3:, :] = 0
    s.bundle.est.IO[:] = True                              # setcamest 'all','not','sk'
    s.bundle.est.IO[4, :] = False
    if model < 3:
        s.bundle.est.IO[3, :] = False
    s.bundle.est.EO[:] = True
    s.prior.OP.isCtrl = s.OP.id > 1000
    pts = L.loadcpt(os.path.join(GOLDEN, 'camcal-fixed.txt'))
    return L.setcpt(s, pts)


def camcal_expected():
    with open(os.path.join(GOLDEN, 'camcal_expected.json')) as fh:
        return json.load(fh)


def check_camcal_against_report(res, s0, E, exp, sig=6):
    """Compare a converged camcal result with the reference report's printed
    values (6 significant digits; bundle_result_file.m:357-358 flips the sign
    of py, K and P for display)."""
    def close(a, b, digits=sig):
        if b == 0:
            return abs(a) < 10.0 ** (-digits)
        # one unit in the last printed digit: the reference stops at convTol=1e-6,
        # so its 7th digit depends on its own x0/iteration path.
        return abs(a - b) <= 1.01 * 10.0 ** (np.floor(np.log10(abs(b))) - digits + 1)
    assert E.numParams == exp['numParams'] and E.numObs == exp['numObs']
    assert E.redundancy == exp['redundancy']
    assert close(s0, exp['sigma0']), (s0, exp['sigma0'])
    io = res.IO.val[:, 0]
    rep = exp['IO_report']
    got = {'cc': io[0], 'px': io[1], 'py': -io[2], 'as': io[3], 'K1': -io[5],
           'K2': -io[6], 'K3': -io[7], 'P1': -io[8], 'P2': -io[9]}
    for k, v in rep.items():
        assert close(got[k], v, 4 if k == 'cc' else sig), (k, got[k], v)
    eo = np.array(exp['EO_report_deg'])
    ang = np.rad2deg(res.EO.val[3:6]).T
    dang = (ang - eo[:, :3] + 180.0) % 360.0 - 180.0     # x0-dependent 2*pi wraps
    assert np.abs(dang).max() < 1e-6
    assert np.abs(res.EO.val[:3].T - eo[:, 3:]).max() < 1e-6
