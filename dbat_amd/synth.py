"""Seeded synthetic bundle-adjustment scenes (BASELINE.json configs 2-5).

Recipe of SURVEY.md 8(d): object points uniform in a 100 x 100 x 10 m slab,
cameras on a lawn-mower grid 40 m above the ground looking down (omega, phi
within +-10 deg, kappa uniform), the camera of the reference's roma data set
(data/script/romabundledemo/cameras/EOS5DMarkII.xml: 5616 x 3744 px,
cc 24.3581 mm, pp (18.1143, 12) mm, K1 2.174e-4, K2 -1.518e-7, lens model 3),
every point observed by its `rays` nearest cameras that have it in frame,
image noise N(0, 0.5 px) with IP.std = 1 px, x0 = truth + noise, datum by
dependency (seteoest 'depend').  Input generation only -- nothing here is on
the measured path.
"""
from __future__ import annotations

import numpy as np

from .dbatstruct import make_struct, seteoest_depend

CONFIGS = {
    # name: (cams, points, rays, self-calibration, IO groups, damping)
    'tiny': (12, 300, 6, False, 1, 'gna'),
    'small': (40, 4000, 8, False, 1, 'gna'),
    'C1': (100, 10_000, 10, False, 1, 'lmp'),
    'C2': (1000, 100_000, 10, True, 1, 'lm'),
    'C3': (1000, 1_000_000, 10, False, 1, 'lm'),
    'C4': (5000, 5_000_000, 10, True, 4, 'lm'),
}

ROMA_CAM = dict(imsz=(5616, 3744), cc=24.3581, pp=(18.1143, 12.0), sensor=(36.0, 24.0),
                K=(2.174e-4, -1.518e-7, 0.0), P=(0.0, 0.0))


def _rotmat(ang):
    """M = R1(omega) R2(phi) R3(kappa), vectorised (eulerrotmat.m:81,129-147)."""
    so, co = np.sin(ang[0]), np.cos(ang[0])
    sp, cp = np.sin(ang[1]), np.cos(ang[1])
    sk, ck = np.sin(ang[2]), np.cos(ang[2])
    n = ang.shape[1]
    R1 = np.zeros((n, 3, 3)); R2 = np.zeros((n, 3, 3)); R3 = np.zeros((n, 3, 3))
    R1[:, 0, 0] = 1; R1[:, 1, 1] = co; R1[:, 1, 2] = -so; R1[:, 2, 1] = so; R1[:, 2, 2] = co
    R2[:, 0, 0] = cp; R2[:, 0, 2] = sp; R2[:, 1, 1] = 1; R2[:, 2, 0] = -sp; R2[:, 2, 2] = cp
    R3[:, 0, 0] = ck; R3[:, 0, 1] = -sk; R3[:, 1, 0] = sk; R3[:, 1, 1] = ck; R3[:, 2, 2] = 1
    return R1 @ R2 @ R3


def _brown(a, K, P):
    """l = brown_dist(a, -K, -P) for row-stacked points a (n,2) and per-point
    coefficient rows K (n,nK), P (n,nP)."""
    Kn, Pn = -K, -P
    rho = np.sum(a * a, 1)
    rs = np.zeros(len(a))
    pw = np.ones(len(a))
    for j in range(K.shape[1]):
        pw = pw * rho
        rs = rs + Kn[:, j] * pw
    out = a + a * rs[:, None]
    if P.shape[1] >= 2:
        pTu = Pn[:, 0] * a[:, 0] + Pn[:, 1] * a[:, 1]
        out = out + Pn[:, :2] * rho[:, None] + 2 * pTu[:, None] * a
    return out


def project(IO, EO, OP, cam, pt, px, nK=3, nP=2):
    """Pixel coordinates u such that the model-3 residual of (cam, pt) is zero
    (res_euler_brown_1.m:84-95 inverted by fixed-point iteration)."""
    M = _rotmat(EO[3:6])
    X = np.einsum('nji,nj->ni', M[cam], (OP[:, pt] - EO[:3, cam]).T)   # M'(Q-q0)
    f = IO[0, cam]
    lhs = -f[:, None] * X[:, :2] / X[:, 2:3]
    K = IO[5:5 + nK, cam].T
    P = IO[5 + nK:5 + nK + nP, cam].T
    lhs_c = np.clip(lhs, -100.0, 100.0)      # far out-of-frame points: keep the iteration finite
    a = lhs_c.copy()
    for _ in range(14):                      # solve brown(a) = lhs (contraction ~0.1)
        a = np.clip(a + (lhs_c - _brown(a, K, P)), -200.0, 200.0)
    a = np.where(np.abs(lhs) < 100.0, a, lhs)
    b1, b2 = IO[3, cam], IO[4, cam]
    x0 = (a[:, 0] - b2 * a[:, 1]) / (1 + b1)  # a = [1+b1 b2; 0 1] x
    x1 = a[:, 1]
    u = (x0 + IO[1, cam]) / px
    v = -(x1 + IO[2, cam]) / px
    return np.stack([u, v]), X[:, 2]


def make_scene(name='C1', seed=None, cams=None, points=None, rays=None, selfcal=None,
               groups=None, noise_px=0.5, verbose=False):
    """Build a DBAT struct for a named config.  Returns (s, truth) where truth
    holds the noise-free IO/EO/OP."""
    from scipy.spatial import cKDTree
    nc, npnt, k, sc, ng, damping = CONFIGS[name]
    nc = cams or nc; npnt = points or npnt; k = rays or k
    sc = sc if selfcal is None else selfcal
    ng = groups or ng
    if seed is None:
        seed = 20240 + list(CONFIGS).index(name)
    rng = np.random.default_rng(seed)
    # cameras: lawn-mower grid over the slab, 40 m above ground
    nx = int(np.ceil(np.sqrt(nc)))
    ny = int(np.ceil(nc / nx))
    L = 100.0 * min(1.0, np.sqrt(nc / 100.0))         # slab side; 100 m from 100 cameras up
    gx = (np.arange(nx) + 0.5) * L / nx
    gy = (np.arange(ny) + 0.5) * L / ny
    cx, cy = [], []
    for j in range(ny):
        xs = gx if j % 2 == 0 else gx[::-1]
        cx.extend(xs); cy.extend([gy[j]] * nx)
    EO = np.zeros((6, nc))
    EO[0] = np.array(cx[:nc]) + rng.normal(0, 0.3, nc)
    EO[1] = np.array(cy[:nc]) + rng.normal(0, 0.3, nc)
    EO[2] = 40.0 + rng.normal(0, 0.5, nc)
    EO[3] = rng.uniform(-1, 1, nc) * np.deg2rad(10)
    EO[4] = rng.uniform(-1, 1, nc) * np.deg2rad(10)
    EO[5] = rng.uniform(-np.pi, np.pi, nc)
    # interior orientation (prob2dbatstruct.m:202-254 sign conventions)
    cam0 = ROMA_CAM
    px = cam0['sensor'][1] / cam0['imsz'][1]
    IO = np.zeros((10, nc))
    IO[0] = cam0['cc']; IO[1] = cam0['pp'][0]; IO[2] = -cam0['pp'][1]
    IO[5:8] = -np.array(cam0['K'])[:, None]; IO[8:10] = -np.array(cam0['P'])[:, None]
    group = (np.arange(nc) * ng) // nc
    if ng > 1:
        IO[0] *= 1 + 0.01 * rng.uniform(-1, 1, ng)[group]
    IOblock = np.tile(group + 1, (10, 1))
    # visibility: the k nearest cameras (horizontal distance) that have the point in
    # frame.  Points with fewer than k such cameras are redrawn, so every point
    # has exactly k rays (n_obs = k * n_points by construction).
    tree = cKDTree(EO[:2].T)
    kq = min(nc, max(4 * k, 48)) if nc < 500 else min(nc, k + 6)
    OP = np.zeros((3, npnt))
    cam_sel = np.zeros((npnt, k), np.int64)
    uv_sel = np.zeros((npnt, k, 2))
    todo = np.arange(npnt)
    for _round in range(200):
        if len(todo) == 0:
            break
        m = len(todo)
        Q = np.stack([rng.uniform(0, L, m), rng.uniform(0, L, m), rng.uniform(0, 10, m)])
        _, cand = tree.query(Q[:2].T, k=kq)
        cand = cand.reshape(m, kq)
        cam_rep = cand.ravel()
        uv, depth = project(IO, EO, Q, cam_rep, np.repeat(np.arange(m), kq), px)
        ok = ((uv[0] > 0) & (uv[0] < cam0['imsz'][0]) & (uv[1] > 0) & (uv[1] < cam0['imsz'][1])
              & (depth < 0)).reshape(m, kq)
        good = ok.sum(1) >= k
        take = ok & (np.cumsum(ok, 1) <= k)
        gi = np.flatnonzero(good)
        idx = np.argsort(~take[gi], axis=1, kind='stable')[:, :k]   # first k visible candidates
        OP[:, todo[gi]] = Q[:, gi]
        cam_sel[todo[gi]] = np.take_along_axis(cand[gi], idx, 1)
        uvr = uv.reshape(2, m, kq)
        uv_sel[todo[gi], :, 0] = np.take_along_axis(uvr[0][gi], idx, 1)
        uv_sel[todo[gi], :, 1] = np.take_along_axis(uvr[1][gi], idx, 1)
        todo = todo[~good]
    if len(todo):
        raise RuntimeError('could not place %d points with %d rays' % (len(todo), k))
    cam_s = cam_sel.ravel()
    pt_s = np.repeat(np.arange(npnt), k)
    uv_s = uv_sel.reshape(-1, 2).T
    order = np.lexsort((pt_s, cam_s))                 # image-major, ascending OP
    cam_s, pt_s, uv_s = cam_s[order], pt_s[order], uv_s[:, order]
    if verbose:
        cnt = np.bincount(pt_s, minlength=npnt)
        print('scene %s: %d cams %d pts %d obs (rays/pt min %d max %d)'
              % (name, nc, npnt, len(cam_s), cnt.min(), cnt.max()))
    ip = uv_s + rng.normal(0, noise_px, uv_s.shape)
    truth = dict(IO=IO.copy(), EO=EO.copy(), OP=OP.copy())
    # initial values: truth + noise
    EO0 = EO.copy()
    EO0[:3] += rng.normal(0, 0.05, (3, nc))
    EO0[3:] += rng.normal(0, np.deg2rad(0.1), (3, nc))
    OP0 = OP + rng.normal(0, 0.05, OP.shape)
    estIO = np.zeros((10, nc), bool)
    if sc:
        estIO[[0, 1, 2, 5, 6, 7, 8, 9]] = True          # cc px py K1-3 P1-2
    s = make_struct(IO, EO0, OP0, ip, cam_s, pt_s, px, ip_std=1.0, distModel=3,
                    estIO=estIO, IOblock=IOblock)
    s = seteoest_depend(s, 0)
    s.damping = damping
    return s, truth
