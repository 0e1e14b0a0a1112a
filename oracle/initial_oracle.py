"""CPU restatement of the reference's initial-value stage: spatial resection of the camera
stations from control points and forward intersection of the object points.  TEST INFRASTRUCTURE, like
dbat_oracle.py: only tests/ may import it; the product (dbat_amd.initial) runs both on the device
(dbat_hip_resect, dbat_hip_forwintersect) and is tested against these functions.

Paths relative to /root/reference/code/:
  photogrammetry/resect.m:42-131        -> resect
  photogrammetry/pm_resect_3pt.m:27-147 -> pm_resect_3pt
  misc/largesttriangle.m:16-41          -> largesttriangle
  photogrammetry/derotmat3d.m:17-19     -> derotmat3d
  bundle/cammodel/pm_multilenscorr1.m:45-69 + pm_lens1.m:36-72 -> lenscorr1
  photogrammetry/forwintersect.m:19-46 (+ pm_multiforwintersect.m, pm_forwintersect3.m) -> forwintersect

Pinned by the camcal demo's committed report (tests/test_initial.py: the whole camcaldemo.m:56-107 pipeline
reproduces 9 iterations and the printed first error) and by the sxb script's first error.
"""
from itertools import combinations

import numpy as np


def _copy_struct(s):
    import copy
    return copy.deepcopy(s)

def lenscorr1(s):
    """Measured image points in mm (y up), corrected for lens distortion with
    the backward Brown polynomial in the measured point, as
    pm_multilenscorr1.m:45-69 does it for every distortion model (the
    aspect/skew terms are ignored there as well, pm_multilenscorr1.m:161)."""
    nK, nP = s.IO.model.nK, s.IO.model.nP
    cam = s.IP.cam
    IO = s.IO.val[:, cam]
    px = s.IO.sensor.pxSize[:, cam]
    if getattr(s.IO.sensor, 'samePxSize', False):
        px = np.tile(s.IO.sensor.pxSize[0, cam], (2, 1))     # U=sensor.pxSize(1)
    q = np.stack([px[0] * s.IP.val[0], -px[1] * s.IP.val[1]])
    xb, yb = q[0] - IO[1], q[1] - IO[2]
    r2 = xb * xb + yb * yb
    Kr = np.zeros_like(r2)
    pw = np.ones_like(r2)
    for j in range(nK):
        pw = pw * r2
        Kr = Kr + IO[5 + j] * pw
    dx, dy = xb * Kr, yb * Kr
    if nP >= 2:
        P1, P2 = IO[5 + nK], IO[6 + nK]
        P3 = 1 + IO[7 + nK] if nP > 2 else 1.0
        dx = dx + (P1 * (r2 + 2 * xb * xb) + 2 * P2 * xb * yb) * P3
        dy = dy + (P2 * (r2 + 2 * yb * yb) + 2 * P1 * xb * yb) * P3
    return np.stack([q[0] - dx, q[1] - dy])

def _hull(pts):
    """Indices of the points on the convex hull of a 2-by-n point set
    (sorted), standing in for convhulln (largesttriangle.m:20)."""
    n = pts.shape[1]
    if n <= 3:
        return list(range(n))
    order = sorted(range(n), key=lambda i: (pts[0, i], pts[1, i]))

    def half(seq):
        h = []
        for i in seq:
            while len(h) >= 2:
                a, b = pts[:, h[-2]], pts[:, h[-1]]
                if (b[0] - a[0]) * (pts[1, i] - a[1]) - (b[1] - a[1]) * (pts[0, i] - a[0]) <= 0:
                    h.pop()
                else:
                    break
            h.append(i)
        return h
    return sorted(set(half(order)) | set(half(order[::-1])))

def largesttriangle(pts, cHull=True):
    """All point triplets (rows of T, 0-based) sorted by descending area A
    (largesttriangle.m:16-41)."""
    idx = _hull(pts) if cHull else list(range(pts.shape[1]))
    T = np.array(list(combinations(idx, 3)), int).reshape(-1, 3)
    x, y = pts[0][T], pts[1][T]
    A = 0.5 * np.abs(x[:, 0] * (y[:, 1] - y[:, 2]) + x[:, 1] * (y[:, 2] - y[:, 0])
                     + x[:, 2] * (y[:, 0] - y[:, 1]))
    o = np.argsort(-A, kind='stable')
    return T[o], A[o]

def derotmat3d(M):
    """omega, phi, kappa of a world-to-camera rotation (derotmat3d.m:17-19)."""
    return np.array([np.arctan2(-M[2, 1], M[2, 2]), np.arcsin(M[2, 0]),
                     np.arctan2(-M[1, 0], M[0, 0])])

def _line_angle(a, b):
    # subspace(a,b) of two unit vectors: the angle between the lines they span
    return np.arccos(min(1.0, abs(float(a @ b))))

def pm_resect_3pt(X, x, use, behind=False, relax=False):
    """Three-point resection (Haralick et al. 1994, Grunert's solution) with
    the remaining points selecting among the up-to-four poses
    (pm_resect_3pt.m:27-147).  X 3-by-n object points, x 2-by-n normalised
    image points, use boolean n-vector with three set.  Returns (P, PP, res)."""
    use = np.asarray(use, bool)
    if np.count_nonzero(use) != 3:
        raise ValueError('Can only use 3 points for resection')
    XT, xT = X, x
    X = X[:, use]
    d = np.vstack([x[:, use], np.ones(3)])
    d = d / np.linalg.norm(d, axis=0)
    a = np.linalg.norm(X[:, 2] - X[:, 1])
    b = np.linalg.norm(X[:, 2] - X[:, 0])
    c = np.linalg.norm(X[:, 1] - X[:, 0])
    ca = np.cos(_line_angle(d[:, 1], d[:, 2]))
    cb = np.cos(_line_angle(d[:, 0], d[:, 2]))
    cg = np.cos(_line_angle(d[:, 0], d[:, 1]))
    m = (a * a - c * c) / (b * b)
    p = (a * a + c * c) / (b * b)
    bc = (b * b - c * c) / (b * b)
    ba = (b * b - a * a) / (b * b)
    A4 = (m - 1) ** 2 - 4 * c * c / (b * b) * ca ** 2
    A3 = 4 * (m * (1 - m) * cb + 2 * c * c / (b * b) * ca ** 2 * cb - (1 - p) * ca * cg)
    A2 = 2 * (m ** 2 + 2 * m ** 2 * cb ** 2 + 2 * bc * ca ** 2 + 2 * ba * cg ** 2
              - 4 * p * ca * cb * cg - 1)
    A1 = 4 * (-m * (1 + m) * cb + 2 * a * a / (b * b) * cg ** 2 * cb - (1 - p) * ca * cg)
    A0 = (1 + m) ** 2 - 4 * a * a / (b * b) * cg ** 2
    v = np.roots([A4, A3, A2, A1, A0])
    if not relax:
        v = np.real(v[np.abs(np.imag(v) / np.abs(v)) < 1e-3])
    else:
        v = np.unique(np.real(v))
    with np.errstate(divide='ignore', invalid='ignore'):
        u = ((-1 + m) * v ** 2 - 2 * m * cb * v + 1 + m) / (2 * (cg - v * ca))
        s1 = np.sqrt(b * b / (1 + v ** 2 - 2 * v * cb))
    s3, s2 = v * s1, u * s1
    ok = (s1 >= 0) & (s2 >= 0) & (s3 >= 0)
    s123 = np.unique(np.stack([s1[ok], s2[ok], s3[ok]], 1), axis=0) if ok.any() else np.zeros((0, 3))

    def frame(p0, pb, pc):
        ob, oc = pb - p0, pc - p0
        r1 = ob / np.linalg.norm(ob)
        r2 = np.cross(ob, oc)
        r2 = r2 / np.linalg.norm(r2)
        r3 = np.cross(ob, np.cross(ob, oc))
        r3 = r3 / np.linalg.norm(r3)
        return np.stack([r1, r2, r3], 1)
    oR = frame(X[:, 0], X[:, 2], X[:, 1])
    PP, res = [], []
    for s in s123:
        cx = s[None, :] * d
        if behind:
            cx = -cx
        cRo = frame(cx[:, 0], cx[:, 2], cx[:, 1]) @ oR.T
        centre = X[:, 0] - cRo.T @ cx[:, 0]
        P = cRo @ np.hstack([np.eye(3), -centre[:, None]])
        h = P @ np.vstack([XT, np.ones(XT.shape[1])])
        res.append(np.sqrt(np.mean(np.sum((h[:2] / h[2] - xT) ** 2, 0))))
        PP.append(P)
    res = np.array(res)
    P = PP[int(np.argmin(res))] if len(res) else None
    return P, PP, res

def resect(s0, cams='all', cpId=None, n=1, v=0.0, chkId=None):
    """Spatial resection of the listed camera stations from the control points
    with ids cpId; returns (s, rms, fail) as resect.m:1 does.  Of the
    triangles of control points seen by a camera, the n largest in the image
    with at least v times the largest area are tried; the check points chkId
    (default: every object point) pick the best pose."""
    s = _copy_struct(s0)
    nc = s0.EO.val.shape[1]
    cams = range(nc) if isinstance(cams, str) and cams == 'all' else list(cams)
    cpId = np.asarray(cpId)
    chkId = s0.OP.id if chkId is None else np.asarray(chkId)
    keepId = np.union1d(cpId, chkId)
    xy = lenscorr1(s0)
    rms = np.full(len(cams), np.nan)
    fail = False
    for k, ci in enumerate(cams):
        IO = s0.IO.val[:, ci]
        rows = np.flatnonzero(s0.IP.cam == ci)
        ids = s0.OP.id[s0.IP.pt[rows]]
        is_cp = np.isin(ids, cpId)
        if np.count_nonzero(is_cp) > 3:
            mea = rows[is_cp]
            T, A = largesttriangle(xy[:, mea])
            take = (np.arange(len(A)) < n) & (A >= v * A[0])
            tryId = ids[is_cp][T[take]]
        elif np.count_nonzero(is_cp) == 3:
            tryId = ids[is_cp][None, :]
        else:
            tryId = np.zeros((0, 3), int)
        keep = np.isin(ids, keepId)
        pt2 = xy[:, rows[keep]]
        pt2N = np.stack([(pt2[0] - IO[1]) / -IO[0], (pt2[1] - IO[2]) / -IO[0]])   # K\homogeneous(pt2)
        pt3 = s0.OP.val[:, s0.IP.pt[rows[keep]]]
        visId = ids[keep]
        best, bestP = np.inf, None
        for useId in tryId:
            P, _, res = pm_resect_3pt(pt3, pt2N, np.isin(visId, useId), True)
            if len(res) and res.min() < best:
                best, bestP = res.min(), P
        rms[k] = best
        if bestP is not None:
            # euclidean(null(P)), through the SVD as the reference does it: with
            # project coordinates of 1e6 m this carries |C|^2*eps ~ 1e-4 m of rounding
            # noise that -R'*t would not, and the reference's committed first
            # errors (data/script/sxb/result/report.txt:42) include it
            nv = np.linalg.svd(bestP)[2][-1]
            s.EO.val[:3, ci] = nv[:3] / nv[3]
            s.EO.val[3:6, ci] = derotmat3d(bestP[:, :3])
        else:
            fail = True
            s.EO.val[:6, ci] = np.nan
    return s, rms, fail

def _rot(ang):
    so, co = np.sin(ang[0]), np.cos(ang[0])
    sp, cp = np.sin(ang[1]), np.cos(ang[1])
    sk, ck = np.sin(ang[2]), np.cos(ang[2])
    R1 = np.array([[1, 0, 0], [0, co, -so], [0, so, co]])
    R2 = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    R3 = np.array([[ck, -sk, 0], [sk, ck, 0], [0, 0, 1]])
    return R1 @ R2 @ R3

def forwintersect(s0, ids='all', skipPrior=False):
    """Object points by forward intersection of the lens-corrected image rays:
    the point minimising the summed squared distance to its rays, which is
    what the stacked system of pm_forwintersect3.m:55-82 solves.  Points seen
    from fewer than two stations become NaN (pm_multiforwintersect.m:41).
    With skipPrior, points that are fixed or carry prior observations keep
    their values (forwintersect.m:32-36)."""
    if not np.all(np.isfinite(s0.EO.val)):
        raise ValueError('Bad or uninitialized EO data')
    if not np.all(np.isfinite(s0.IO.val)):
        raise ValueError('Bad or uninitialized IO data')
    s = _copy_struct(s0)
    npnt = s0.OP.val.shape[1]
    do = np.ones(npnt, bool) if isinstance(ids, str) and ids == 'all' else np.isin(s0.OP.id, ids)
    if skipPrior:
        do &= np.all(s0.bundle.est.OP, 0) & ~np.any(s0.prior.OP.use, 0)
    cam, pt = s0.IP.cam, s0.IP.pt
    xy = lenscorr1(s0)
    IO = s0.IO.val[:, cam]
    # ray through the corrected point: K^-1 [x;y;1] with K=[-f 0 px;0 -f py;0 0 1]
    d_cam = np.stack([xy[0] - IO[1], xy[1] - IO[2], -IO[0]])
    M = np.stack([_rot(s0.EO.val[3:6, i]) for i in range(s0.EO.val.shape[1])])
    d = np.einsum('nij,jn->in', M[cam], d_cam)
    d = d / np.linalg.norm(d, axis=0)
    c = s0.EO.val[:3, cam]
    A = np.zeros((npnt, 3, 3))
    b = np.zeros((npnt, 3))
    Pj = np.eye(3)[None] - np.einsum('in,jn->nij', d, d)
    np.add.at(A, pt, Pj)
    np.add.at(b, pt, np.einsum('nij,jn->ni', Pj, c))
    rays = np.bincount(pt, minlength=npnt)
    ok = do & (rays >= 2)
    s.OP.val[:, do & (rays < 2)] = np.nan
    s.OP.val[:, ok] = np.linalg.solve(A[ok], b[ok][:, :, None])[:, :, 0].T
    return s
