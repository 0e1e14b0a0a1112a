"""Host-only checks of the plan of the heavy / giant points (csrc/heavy.hpp, plan.hpp build_heavy_plan): the row
groups, slots, operand places and tasks of k_heavy_syrk replayed on the host against their definition -- the sum over
the points of z_p z_p' over all rows of the reduced system the point touches (dbat_hip_debug_heavy_plan_selftest).
No GPU: the kernels that run on these structures are covered by tests/test_hip_parity.py."""
import numpy as np
import pytest

from helpers import camcal_struct, synth_struct


def _check(st, tol=1e-10):
    assert st['on'] and st['tasks'] > 0 and st['entries'] > 0
    assert st['max_diff'] <= tol * max(st['max_abs'], 1.0), st


@pytest.mark.parametrize('model', [3, 5])
def test_camcal_every_point_in_every_image(model):
    """The reference's calibration demo (demo/camcaldemo.m:56-119): 21 images, nine IO columns: three camera groups
    (8 + 8 + 5 images) and one group for the IO columns and the right-hand side -- 45 blocks of 16 x 16, none wasted."""
    from dbat_amd import _hip
    st = _hip.heavy_plan_selftest(camcal_struct(model))
    _check(st)
    assert st['row_groups'] == 4 and st['points'] >= 90


@pytest.mark.parametrize('variant', ['plain', 'selfcal', 'imagevar', 'priors', 'groups4'])
@pytest.mark.parametrize('ks', [None, '3'])
def test_heavy_points_of_synthetic_scenes(variant, ks, monkeypatch):
    """Tiles of six cameras: every eight-ray point is heavy.  imagevar: 83 IO columns = two IO groups; groups4: the
    cameras of a point belong to different IO blocks; ks = 3: several tasks per pair of groups."""
    from dbat_amd import _hip
    monkeypatch.setenv('DBAT_HIP_CMAX', '6')
    if ks:
        monkeypatch.setenv('DBAT_HIP_HEAVY_KS', ks)
    s, _ = synth_struct('small', variant)
    st = _hip.heavy_plan_selftest(s)
    _check(st)
    assert st['points'] == s.OP.val.shape[1]
    monkeypatch.setenv('DBAT_HIP_HEAVY', '0')
    assert not _hip.heavy_plan_selftest(s)['on']


def test_giant_and_heavy_points_together():
    """Three points seen in all 300 images (more than a batch holds) beside ordinary tiled points."""
    from dbat_amd import _hip, synth
    s, truth = synth.make_scene('small', cams=300, points=400, rays=6)
    nc = s.EO.val.shape[1]
    add_cam, add_pt = [], []
    for p in (3, 77, 250):
        have = set(s.IP.cam[s.IP.pt == p].tolist())
        for c in range(nc):
            if c not in have:
                add_cam.append(c); add_pt.append(p)
    cam = np.r_[s.IP.cam, np.array(add_cam)]; pt = np.r_[s.IP.pt, np.array(add_pt)]
    order = np.lexsort((pt, cam))
    s.IP.cam, s.IP.pt = cam[order], pt[order]
    s.IP.val = np.zeros((2, len(cam))); s.IP.std = np.ones((2, len(cam)))
    s.bundle.est.IO[[0, 1, 2, 5, 6]] = True
    st = _hip.heavy_plan_selftest(s)
    _check(st)
    assert st['points'] == 3 and st['row_groups'] == (nc + 7) // 8 + 1
    lay = _hip.plan_layout_stats(s)
    assert lay['heavy_tasks'] == st['tasks'] and lay['n_tiles'] > 0


def test_two_shards_plan_their_own_heavy_points(monkeypatch):
    from dbat_amd import _hip
    monkeypatch.setenv('DBAT_HIP_CMAX', '6')
    s, _ = synth_struct('small', 'selfcal')
    n = 0
    for r in range(2):
        st = _hip.heavy_plan_selftest(s, r, 2)
        _check(st)
        n += st['points']
    assert n == s.OP.val.shape[1]
