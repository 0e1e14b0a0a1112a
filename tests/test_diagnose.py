"""dbat_amd.diagnose (bundle.m:368-446): parameter names, the column matching
behind the structural-rank diagnosis, and the null-space diagnosis, on the
host and without the GPU.  The end-to-end known answers (failure-mode demos
against their committed reports) are in test_oracle.py / test_hip_parity.py."""
import numpy as np
import scipy.sparse as sp
from scipy.sparse.csgraph import maximum_bipartite_matching

import dbat_oracle as o
from dbat_amd import diagnose as D
from helpers import camcal_struct, synth_struct


def test_maxtrans_is_maximum_and_lexicographically_first():
    rng = np.random.default_rng(0)
    for _ in range(200):
        m, n = rng.integers(1, 30), rng.integers(1, 20)
        A = sp.random(m, n, density=rng.uniform(0.02, 0.3), random_state=int(rng.integers(1 << 30)), format='csc')
        A.data[:] = 1
        p = D.maxtrans(A)
        rank = lambda B: int(np.count_nonzero(maximum_bipartite_matching(sp.csr_matrix(B), perm_type='column') >= 0))
        assert np.count_nonzero(p >= 0) == rank(A)
        rows = p[p >= 0]
        assert len(set(rows)) == len(rows) and all(A[p[j], j] != 0 for j in np.flatnonzero(p >= 0))
        for j in np.flatnonzero(p < 0):            # an unmatched column adds no rank to the columns before it
            assert rank(A[:, :j + 1]) == np.count_nonzero(p[:j] >= 0)
        assert np.array_equal(p >= 0, o.dmperm_cols(A) >= 0)      # same set as the oracle's recursive version


def test_param_types_match_oracle_and_reference_naming():
    s = camcal_struct(3)
    so = o.buildserialindices(s)
    want = o.paramtypes(so)
    # index maps as Handle.index_maps() returns them, from the oracle's serial indices
    maps = []
    for nm, rows in (('IO', s.IO.val.shape[0]), ('EO', 6), ('OP', 3)):
        ser = getattr(so.bundle.serial, nm)
        full = np.full(getattr(s, nm).val.size, -1, np.int64)
        full[ser.src] = ser.dest
        maps.append(full.reshape(getattr(s, nm).val.shape, order='F')[:rows])
    got = D.param_types(s, maps, so.bundle.serial.n)
    assert list(got) == list(want)
    assert list(got[:9]) == ['cc', 'px', 'py', 'as', 'K1', 'K2', 'K3', 'P1', 'P2']
    assert got[9] == 'EX-1' and got[9 + 125] == 'ka-21'
    assert got[9 + 126] == 'OX-1/2' and got[-1] == 'OZ-96/97'     # sequence number / id (ids start at 2)
    # control points are 'C', ids that differ from the sequence number are appended
    IOt, EOt, OPt = D.buildparamtypes(s)
    assert OPt[0, -1] == 'CX-100/1004-CP4'                 # sequence number / id - label


def test_structural_pattern_and_numerical_null_space_small():
    s, _ = synth_struct('tiny', 'plain')
    s.bundle.est.EO[:] = True                      # no datum: seven-dimensional null space
    so = o.buildserialindices(s)
    x0 = o.serialize(so)
    r, J = o.brown_euler_cam4(x0, so, jac=True)
    w = o.buildweightvector(so)
    Jw = sp.diags(np.sqrt(w)) @ J
    types = o.paramtypes(so)
    nw = D.numerical_weakness(sp.csc_matrix(Jw), types)
    assert nw.deficiency == 7 and nw.rank == len(x0) - 7
    assert np.abs(Jw @ (nw.V / np.sqrt(np.asarray(Jw.multiply(Jw).sum(0)).ravel())[:, None])).max() < 1e-6
    assert all(len(sp_.params) > 0 for sp_ in nw.suspectedParams)
    big = D.numerical_weakness(sp.csc_matrix(Jw), types, dense_limit=10)
    assert np.isnan(big.rank)


def _drop_obs(s, keep):
    s.IP.val, s.IP.std = s.IP.val[:, keep], s.IP.std[:, keep]
    s.IP.cam, s.IP.pt = s.IP.cam[keep], s.IP.pt[keep]
    return s


def test_plan_structural_rank_equals_sprank():
    """The host-side matching of libdbat_hip.so (dbat_hip_plan_structural_rank_ok)
    against the structural rank of the oracle's Jacobian, on healthy scenes and
    on scenes thinned until sub-networks lose their rows -- including cases every
    per-group counting condition passes (two cameras that only see three points
    nobody else sees: 12 rows for 21 unknowns)."""
    from dbat_amd import _hip
    rng = np.random.default_rng(11)
    seen = {True: 0, False: 0}
    isolated = 0
    for trial in range(40):
        variant = ['plain', 'selfcal', 'priors', 'imagevar'][trial % 4]
        s, _ = synth_struct('tiny', variant)
        no = s.IP.val.shape[1]
        keep = np.ones(no, bool)
        if trial >= 4:
            frac = rng.uniform(0.05, 0.6)
            keep = rng.random(no) < frac
            if trial % 3 == 0:            # an isolated pair of cameras with three private points:
                # every camera keeps 6 rows for 6 unknowns and every point 4 rows for 3,
                # but together they have 12 rows for 21 unknowns
                keep[:] = True
                vis = np.zeros((s.EO.val.shape[1], s.OP.val.shape[1]), bool)
                vis[s.IP.cam, s.IP.pt] = True
                pairs = [(a, b) for a in range(vis.shape[0]) for b in range(a + 1, vis.shape[0])
                         if np.count_nonzero(vis[a] & vis[b]) >= 3]
                ca, cb = pairs[rng.integers(len(pairs))]
                pts = rng.choice(np.flatnonzero(vis[ca] & vis[cb]), 3, replace=False)
                keep &= ~np.isin(s.IP.cam, [ca, cb]) & ~np.isin(s.IP.pt, pts)
                keep |= np.isin(s.IP.cam, [ca, cb]) & np.isin(s.IP.pt, pts)
                s.bundle.est.EO[:, [ca, cb]] = True
                isolated += 1
        s = _drop_obs(s, keep)
        so = o.buildserialindices(s)
        x0 = o.serialize(so)
        r, J = o.brown_euler_cam4(x0, so, jac=True)
        J = sp.csr_matrix(J)
        J.data[:] = 1.0
        want = int(np.count_nonzero(maximum_bipartite_matching(J, perm_type='column') >= 0)) == len(x0)
        got = _hip.plan_structural_rank_ok(s)
        assert got == want, (trial, variant, got, want)
        seen[want] += 1
    assert seen[True] >= 4 and seen[False] >= 10 and isolated >= 10
