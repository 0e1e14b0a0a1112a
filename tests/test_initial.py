"""Initial values: dbat_amd.initial's resection and forward
intersection, unit-tested on exact synthetic data and pinned end-to-end by the
reference's committed camcal report (demo/camcaldemo.m:56-107 pipeline ->
'Number of iterations: 9', 'First error: 30873.9')."""
import numpy as np
import pytest

import dbat_oracle as o
from dbat_amd import initial as I
import initial_oracle as IO_
from dbat_amd import synth
from helpers import (camcal_demo_struct, camcal_expected, camcal_struct, check_camcal_against_report,
                     check_report_lines)


def test_derotmat3d_round_trip():
    rng = np.random.default_rng(5)
    for _ in range(20):
        ang = np.array([rng.uniform(-3, 3), rng.uniform(-1.5, 1.5), rng.uniform(-3, 3)])
        assert np.abs(I.derotmat3d(IO_._rot(ang).T) - ang).max() < 1e-12


def test_largesttriangle():
    # unit square plus an interior point: the hull drops the interior point and
    # all four corner triangles have area 1/2
    pts = np.array([[0, 1, 1, 0, 0.4], [0, 0, 1, 1, 0.5]], float)
    T, A = I.largesttriangle(pts)
    assert len(T) == 4 and not (T == 4).any() and np.allclose(A, 0.5)
    T, A = I.largesttriangle(pts, cHull=False)
    assert len(T) == 10 and np.all(np.diff(A) <= 0)
    # a stretched quadrilateral: the largest triangle is the one without the near corner
    pts = np.array([[0, 4, 4, 0.5], [0, 0, 3, 0.5]], float)
    T, A = I.largesttriangle(pts)
    assert list(T[0]) == [0, 1, 2] and A[0] == 6.0


def _exact_scene(seed=3):
    """Noise-free, distortion-free 12-camera scene at its true parameters."""
    s, truth = synth.make_scene('tiny', seed=seed, noise_px=0.0)
    s.IO.val[3:, :] = 0
    s.EO.val[:6] = truth['EO']
    s.OP.val[:] = truth['OP']
    s.bundle.est.EO[:] = True
    uv, depth = synth.project(s.IO.val, s.EO.val, s.OP.val, s.IP.cam, s.IP.pt,
                              s.IO.sensor.pxSize[0, 0])
    assert (depth < 0).all()
    s.IP.val = uv
    return s


def test_resect_recovers_exact_poses():
    s = _exact_scene()
    truth = s.EO.val.copy()
    cp = s.OP.id                                     # every point is a control point
    t = I.cleareo(s)
    assert np.isnan(t.EO.val).all()
    s1, rms, fail = IO_.resect(t, 'all', cp, 2, 0.5)
    assert not fail and rms.max() < 1e-8
    assert np.abs(s1.EO.val[:3] - truth[:3]).max() < 1e-7
    dang = (s1.EO.val[3:6] - truth[3:6] + np.pi) % (2 * np.pi) - np.pi
    assert np.abs(dang).max() < 1e-8


def test_resect_too_few_control_points_fails():
    s = _exact_scene()
    t = I.cleareo(s)
    s1, rms, fail = IO_.resect(t, [0], s.OP.id[:2])
    assert fail and np.isnan(s1.EO.val[:, 0]).all() and np.isposinf(rms[0])      # resect.m: rms(i) = bestRes = inf


def test_forwintersect_exact_and_skip_prior():
    s = _exact_scene()
    truth = s.OP.val.copy()
    s.prior.OP.use[:, :3] = True                     # three points with prior observations
    s.OP.val[:, :3] += 1.0                           # ... keep whatever they hold
    held = s.OP.val[:, :3].copy()
    s.OP.val[:, 3:] = np.nan
    t = IO_.forwintersect(s, 'all', True)
    assert np.array_equal(t.OP.val[:, :3], held)
    assert np.abs(t.OP.val[:, 3:] - truth[:, 3:]).max() < 1e-9
    t = IO_.forwintersect(s, s.OP.id[:5])              # explicit ids, priors included
    assert np.abs(t.OP.val[:, :5] - truth[:, :5]).max() < 1e-9
    assert np.isnan(t.OP.val[:, 5:]).all()


@pytest.mark.parametrize('model', [1, 2, 3, 4, 5])
def test_camcal_demo_pipeline_known_answer(model):
    """The whole camcaldemo: resection + forward intersection + GNA bundle.
    camcal-dbatreport.txt:39-43: 9 iterations, first error 30873.9, last 98.556;
    camcaldemo_allmodels.m:77-108 runs the same pipeline for every lens model
    (camcal-dbatreport-model*.txt:39-43: 9 iterations and 30873.9 each)."""
    exp = camcal_expected()['model%d' % model]
    s = camcal_demo_struct(model)
    res, ok, iters, s0, E = o.bundle(s, 'gna')
    assert ok and iters == exp['iterations'] == 9
    assert abs(E.res[0] / 30873.9 - 1) < 1e-5
    assert abs(E.res[-1] / exp['lastError'] - 1) < 1e-5
    check_camcal_against_report(res, s0, E, exp)
    from dbat_amd.report import bundle_result_lines
    import os
    from helpers import GOLDEN
    CIO, CEO, COP = o.bundle_cov(res, E, 'CIO', 'CEO', 'COP')
    lines = bundle_result_lines(res, E, CIO, CEO, COP)
    assert any('Number of iterations: 9' in l for l in lines)
    ref = os.path.join(GOLDEN, 'camcal-dbatreport.txt' if model == 3 else 'camcal-dbatreport-model%d.txt' % model)
    # every line of the report but the first error verbatim, for every lens model
    assert len(lines) >= 590 and check_report_lines(lines, ref_path=ref, demo_x0=True) >= len(lines) - 2


def test_resect_first_error_conditioning():
    """Why the first error is compared to 1e-5 and not to the report's six digits:
    camera 21's quartic (pm_resect_3pt.m:62-69) has a near-triple root, so one
    ulp in its coefficients moves the initial residual norm in the sixth digit;
    the reference's 30873.9 lies inside that band."""
    s = I.clearop(I.cleareo(camcal_struct(3)))
    cp = s.OP.id[s.prior.OP.isCtrl]
    roots = np.roots
    rng = np.random.default_rng(0)
    vals = []
    try:
        for k in range(6):
            eps = 0.0 if k == 0 else 2.2e-16
            np.roots = lambda c: roots(np.asarray(c) * (1 + eps * rng.standard_normal(5)))
            s1, _, fail = IO_.resect(s, 'all', cp, 1, 0, cp)
            x2 = IO_.forwintersect(s1, 'all', True)
            vals.append(o.bundle(x2, 'gna')[4].res[0])
    finally:
        np.roots = roots
    vals = np.array(vals)
    assert np.ptp(vals) > 0.05                        # the sixth digit is noise ...
    assert np.abs(vals / 30873.9 - 1).max() < 1e-5    # ... around the reference's value
