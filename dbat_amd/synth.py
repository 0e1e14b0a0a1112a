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


def _n_workers():
    import os
    try:
        return max(1, min(16, len(os.sched_getaffinity(0))))
    except AttributeError:
        return max(1, min(16, os.cpu_count() or 1))


def _select_rays(IO, EO, M, Q, cand, px, imsz, nK, k):
    """For every point the first k candidate cameras that have it in frame:
    (good, cam_sel (m,k), u_sel (m,k), v_sel (m,k)); rows of points with fewer
    than k such cameras are undefined and `good` is False there.

    `project` for the case of one lens (K equal in all cameras, P = 0) on
    coordinate columns, in cache-sized chunks over a thread pool (NumPy
    releases the GIL inside its loops).  Element for element the arithmetic
    of `project`/`_brown`; a zero tangential term is not added."""
    from concurrent.futures import ThreadPoolExecutor
    m, kq = cand.shape
    Kn = [-float(IO[5 + j, 0]) for j in range(nK)]
    good = np.empty(m, bool)
    cam_sel = np.empty((m, k), np.int64); u_sel = np.empty((m, k)); v_sel = np.empty((m, k))
    Mf = M.reshape(len(M), 9)
    step = max(1, (1 << 15) // kq)

    def work(lo):
        hi = min(m, lo + step)
        c = cand[lo:hi].ravel()
        rep = lambda a: np.repeat(a[lo:hi], kq)
        d0 = rep(Q[0]) - EO[0, c]; d1 = rep(Q[1]) - EO[1, c]; d2 = rep(Q[2]) - EO[2, c]
        Mc = Mf[c]
        X0 = Mc[:, 0] * d0 + Mc[:, 3] * d1 + Mc[:, 6] * d2          # M'(Q-q0)
        X1 = Mc[:, 1] * d0 + Mc[:, 4] * d1 + Mc[:, 7] * d2
        X2 = Mc[:, 2] * d0 + Mc[:, 5] * d1 + Mc[:, 8] * d2
        f = IO[0, c]
        l0 = -f * X0 / X2; l1 = -f * X1 / X2
        c0 = np.clip(l0, -100.0, 100.0); c1 = np.clip(l1, -100.0, 100.0)
        a0 = c0.copy(); a1 = c1.copy()
        for _ in range(14):
            rho = a0 * a0 + a1 * a1
            rs = Kn[0] * rho
            pw = rho
            for j in range(1, nK):
                pw = pw * rho
                rs = rs + Kn[j] * pw
            a0 = np.clip(a0 + (c0 - (a0 + a0 * rs)), -200.0, 200.0)
            a1 = np.clip(a1 + (c1 - (a1 + a1 * rs)), -200.0, 200.0)
        a0 = np.where(np.abs(l0) < 100.0, a0, l0); a1 = np.where(np.abs(l1) < 100.0, a1, l1)
        x0 = (a0 - IO[4, c] * a1) / (1 + IO[3, c])
        u = (x0 + IO[1, c]) / px
        v = -(a1 + IO[2, c]) / px
        ok = ((u > 0) & (u < imsz[0]) & (v > 0) & (v < imsz[1]) & (X2 < 0)).reshape(-1, kq)
        good[lo:hi] = ok.sum(1) >= k
        if ok[:, :k].all():                                          # the usual case: the k nearest all see it
            idx = slice(0, k)
            cam_sel[lo:hi] = cand[lo:hi, :k]
            u_sel[lo:hi] = u.reshape(-1, kq)[:, :k]; v_sel[lo:hi] = v.reshape(-1, kq)[:, :k]
        else:
            take = ok & (np.cumsum(ok, 1, dtype=np.int8) <= k)
            idx = np.argsort(~take, axis=1, kind='stable')[:, :k]    # first k visible candidates
            cam_sel[lo:hi] = np.take_along_axis(cand[lo:hi], idx, 1)
            u_sel[lo:hi] = np.take_along_axis(u.reshape(-1, kq), idx, 1)
            v_sel[lo:hi] = np.take_along_axis(v.reshape(-1, kq), idx, 1)

    with ThreadPoolExecutor(_n_workers()) as ex:
        list(ex.map(work, range(0, m, step)))
    return good, cam_sel, u_sel, v_sel


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
    Mrot = _rotmat(EO[3:6])
    kq = min(nc, max(4 * k, 48)) if nc < 500 else min(nc, k + 6)
    OP = np.zeros((3, npnt))
    cam_sel = np.zeros((npnt, k), np.int64)
    u_sel = np.zeros((npnt, k)); v_sel = np.zeros((npnt, k))
    todo = np.arange(npnt)
    for _round in range(200):
        if len(todo) == 0:
            break
        m = len(todo)
        Q = np.stack([rng.uniform(0, L, m), rng.uniform(0, L, m), rng.uniform(0, 10, m)])
        _, cand = tree.query(Q[:2].T, k=kq, workers=_n_workers())
        cand = cand.reshape(m, kq)
        good, cs, us, vs = _select_rays(IO, EO, Mrot, Q, cand, px, cam0['imsz'], 3, k)
        if m == npnt and good.all():
            OP[:], cam_sel, u_sel, v_sel = Q, cs, us, vs
        else:
            gi = np.flatnonzero(good)
            OP[:, todo[gi]] = Q[:, gi]
            cam_sel[todo[gi]] = cs[gi]; u_sel[todo[gi]] = us[gi]; v_sel[todo[gi]] = vs[gi]
        todo = todo[~good]
    if len(todo):
        raise RuntimeError('could not place %d points with %d rays' % (len(todo), k))
    cam_s = cam_sel.ravel()
    pt_s = np.repeat(np.arange(npnt), k)
    # image-major, ascending OP: pt_s is ascending, so a stable sort by camera is enough
    # (16-bit keys take NumPy's radix sort)
    order = (np.argsort(cam_s.astype(np.uint16), kind='stable') if nc < 65536
             else np.lexsort((pt_s, cam_s)))
    cam_s, pt_s = cam_s[order], pt_s[order]
    uv_s = np.empty((2, len(order)), order='F')
    uv_s[0] = u_sel.ravel()[order]; uv_s[1] = v_sel.ravel()[order]
    if verbose:
        cnt = np.bincount(pt_s, minlength=npnt)
        print('scene %s: %d cams %d pts %d obs (rays/pt min %d max %d)'
              % (name, nc, npnt, len(cam_s), cnt.min(), cnt.max()))
    ip = uv_s
    ip += rng.normal(0, noise_px, uv_s.shape)
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


def make_dense_scene(cams=48, points=16384, selfcal=True, groups=1, seed=3, noise_px=0.5):
    """Every point in every image -- the visibility of the reference's camera-calibration demo
    (demo/camcaldemo.m:56-119: 21 images, every target in all of them) at a size of one's choosing: the cameras and
    points of make_scene('small', cams, points), then every point projected into every camera.  Distortion-free lens, so
    that projections far outside the image format stay defined.  selfcal: cc, px, py, K1, K2 estimated (groups > 1:
    independent IO blocks for runs of cameras).  Returns (s, truth)."""
    s, truth = make_scene('small', cams=cams, points=points, rays=6, seed=seed)
    s.IO.val[5:10] = 0.0
    truth['IO'][5:10] = 0.0
    nc = s.EO.val.shape[1]
    px = float(np.ravel(s.IO.sensor.pxSize)[0])
    cam = np.repeat(np.arange(nc, dtype=np.int32), points)
    pt = np.tile(np.arange(points, dtype=np.int32), nc)
    uv, depth = project(truth['IO'], truth['EO'], truth['OP'], cam, pt, px, nK=3, nP=2)
    if not np.all(depth < 0):
        raise RuntimeError('a point behind a camera')
    rng = np.random.default_rng(seed)
    s.IP.val = np.asfortranarray(uv + rng.normal(0, noise_px, uv.shape))
    s.IP.std = np.ones(uv.shape, order='F')
    s.IP.cam, s.IP.pt = cam, pt
    if selfcal:
        s.bundle.est.IO[[0, 1, 2, 5, 6]] = True
        if groups > 1:
            s.IO.struct.block[:] = (1 + (np.arange(nc) * groups) // nc)[None, :]
    return s, truth
