"""Resection on the device (dbat_hip_resect / dbat_amd.initial.resect) against the host
restatement of photogrammetry/resect.m + pm_resect_3pt.m (oracle/initial_oracle.py resect, itself pinned by the
reference's camcal report in tests/test_initial.py) and against that report's first error."""
import numpy as np
import pytest

import dbat_oracle as o
from dbat_amd import initial as I
import initial_oracle as IO_
from dbat_amd import synth
from helpers import camcal_struct, camcal_expected

pytestmark = pytest.mark.gpu


def _ang_diff(a, b):
    return np.abs((a - b + np.pi) % (2 * np.pi) - np.pi).max()


def test_resect_hip_recovers_exact_poses():
    from test_initial import _exact_scene
    s = _exact_scene()
    truth = s.EO.val.copy()
    t = I.cleareo(s)
    s1, rms, fail = I.resect(t, 'all', s.OP.id, 2, 0.5)
    assert not fail and rms.max() < 1e-8
    assert np.abs(s1.EO.val[:3] - truth[:3]).max() < 1e-7 and _ang_diff(s1.EO.val[3:6], truth[3:6]) < 1e-8
    s2, rms2, fail2 = I.resect(t, [0], s.OP.id[:2])           # too few control points
    # resect.m: rms(i) = bestRes = inf for a station without a pose -- on the device as on the host
    h2, hrms2, hfail2 = IO_.resect(t, [0], s.OP.id[:2])
    assert fail2 and hfail2 and np.isnan(s2.EO.val[:, 0]).all() and np.isposinf(rms2[0]) and np.isposinf(hrms2[0])


@pytest.mark.parametrize('n,v', [(1, 0.0), (3, 0.5), (8, 0.0)])
def test_resect_hip_matches_host_on_noisy_scene(n, v):
    """100 cameras, noisy image points, 40 scattered control points, all object points as check points: the
    same pose (to the conditioning of the quartic), the same rms per camera, the same failures."""
    s, truth = synth.make_scene('C1')
    s.OP.val[:] = truth['OP']
    cp = s.OP.id[::250]
    t = I.cleareo(s)
    h_s, h_rms, h_fail = IO_.resect(t, 'all', cp, n, v)
    d_s, d_rms, d_fail = I.resect(t, 'all', cp, n, v)
    assert h_fail == d_fail
    ok = np.isfinite(h_rms)
    assert np.array_equal(ok, np.isfinite(d_rms)) and ok.sum() > 50
    # The two root finders (companion-matrix eigenvalues on the host, Aberth-Ehrlich on the device) agree to
    # the conditioning of the quartic: ~1e-10 as a rule, 1e-5 where a nadir camera over a near-symmetric
    # triangle has close roots (measured: median 3e-10, max 1.3e-5 relative in the rms)
    dr = np.abs(d_rms[ok] / h_rms[ok] - 1)
    assert np.median(dr) < 1e-8 and dr.max() < 1e-3
    dp = np.abs(d_s.EO.val[:3, ok] - h_s.EO.val[:3, ok]).max(0)
    assert np.median(dp) < 1e-7 and dp.max() < 1e-2 * 40.0                      # (flying height 40 m)
    da = np.abs((d_s.EO.val[3:6, ok] - h_s.EO.val[3:6, ok] + np.pi) % (2 * np.pi) - np.pi).max(0)
    assert np.median(da) < 1e-8 and da.max() < 1e-2


def test_resect_rejects_bad_ranges_and_null_pointers():
    """dbat_hip_resect validates before it dereferences: ranges that do not start at 0 or descend, triangle
    indices outside the image's points, and null arrays with non-empty ranges are DBAT_HIP_EINVAL, not a segfault."""
    import ctypes as C
    from dbat_amd import _hip
    lib = _hip.load()
    i64p = C.POINTER(C.c_int64)
    X = np.zeros(3 * 4); xn = np.zeros(2 * 4); P = np.zeros(12); rms = np.zeros(1)
    tri = np.array([0, 1, 2], np.int32)
    def call(pt_start, tri_start, Xp=X, xp=xn, trip=tri):
        ps = np.ascontiguousarray(pt_start, np.int64); ts = np.ascontiguousarray(tri_start, np.int64)
        return lib.dbat_hip_resect(0, 1, ps.ctypes.data_as(i64p), None if Xp is None else _hip.dptr(Xp),
                                   None if xp is None else _hip.dptr(xp), ts.ctypes.data_as(i64p),
                                   None if trip is None else trip.ctypes.data_as(_hip._ip), _hip.dptr(P), _hip.dptr(rms))
    assert call([0, 4], [0, 1]) == _hip.OK
    assert call([1, 4], [0, 1]) == _hip.EINVAL and 'start at 0' in _hip.last_error()
    assert call([0, 4], [1, 1]) == _hip.EINVAL
    assert call([0, 4], [0, 1], trip=None) == _hip.EINVAL and 'null' in _hip.last_error()
    assert call([0, 4], [0, 1], Xp=None) == _hip.EINVAL
    assert call([0, 4], [0, 1], trip=np.array([0, 1, 7], np.int32)) == _hip.EINVAL and 'triangle index' in _hip.last_error()
    assert call([0, -1], [0, 0]) == _hip.EINVAL


def test_camcal_demo_pipeline_with_device_resection():
    """demo/camcaldemo.m:56-107 with resection AND forward intersection on the GPU, then the bundle:
    camcal-dbatreport.txt:39-43 -- 9 iterations, first error 30873.9 (sixth digit: the near-triple root of image
    21's quartic, tests/test_initial.py::test_resect_first_error_conditioning), last error 98.556."""
    from dbat_amd import bundle
    exp = camcal_expected()['model3']
    s = I.clearop(I.cleareo(camcal_struct(3)))
    cp = s.OP.id[s.prior.OP.isCtrl]
    s1, rms, fail = I.resect(s, 'all', cp, 1, 0, cp)
    h1, hrms, hfail = IO_.resect(s, 'all', cp, 1, 0, cp)
    assert not fail and not hfail
    # every image but the ill-conditioned one agrees closely; that one to the conditioning of its quartic
    d = np.abs(s1.EO.val[:3] - h1.EO.val[:3]).max(0)
    assert np.sort(d)[-2] < 1e-7 and d.max() < 1e-2
    x2 = I.forwintersect(s1, 'all', True)
    res, ok, iters, s0, E = bundle(x2, 'gna')
    assert ok and iters == exp['iterations'] == 9
    assert abs(E.res[0] / 30873.9 - 1) < 1e-5
    assert abs(E.res[-1] / exp['lastError'] - 1) < 1e-5
