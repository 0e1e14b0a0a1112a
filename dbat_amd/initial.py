"""Initial values for the bundle: spatial resection of the camera stations from
control points and forward intersection of the object points, on the device.

  resect        -> dbat_hip_resect (csrc/resect.hpp k_resect: one wave per image solves the quartics of the
                   candidate triangles and scores the poses); the host keeps what resect.m does with MATLAB
                   built-ins around it: lens correction, the choice of the triangles, centre / angles from the
                   winning 3 x 4 matrix                                  (photogrammetry/resect.m:42-131)
  forwintersect -> dbat_hip_forwintersect (k_forwintersect)              (photogrammetry/forwintersect.m:19-46)
  largesttriangle (misc/largesttriangle.m:16-41), derotmat3d (photogrammetry/derotmat3d.m:17-19),
  lenscorr1 (bundle/cammodel/pm_multilenscorr1.m:45-69 + pm_lens1.m:36-72), cleareo, clearop (misc/)

The CPU restatements these are tested against live in oracle/initial_oracle.py.
"""
from itertools import combinations

import numpy as np

def cleareo(s):
    """Set every EO parameter that is estimated and has no prior observation
    to NaN (misc/cleareo.m)."""
    s.EO.val[:s.bundle.est.EO.shape[0]][s.bundle.est.EO & ~s.prior.EO.use] = np.nan
    return s

def clearop(s):
    """Set every OP coordinate that is estimated and has no prior
    observation to NaN (misc/clearop.m)."""
    s.OP.val[s.bundle.est.OP & ~s.prior.OP.use] = np.nan
    return s

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

def resect(s0, cams='all', cpId=None, n=1, v=0.0, chkId=None, device=0):
    """Spatial resection of the listed camera stations from the control points with ids cpId; returns
    (s, rms, fail) as resect.m:1 does (rms = Inf for a station without a pose, resect.m: bestRes=inf).  The
    per-camera work runs on the GPU (dbat_hip_resect: one wave per camera solves the quartics of the candidate
    triangles and scores the poses against the check points chkId -- default: every object point).  Of the
    triangles of control points seen by a camera, the n largest in the image with at least v times the largest
    area are tried."""
    from . import _hip
    from .dbatstruct import copy_struct
    s = copy_struct(s0)
    nc = s0.EO.val.shape[1]
    cams = list(range(nc)) if isinstance(cams, str) and cams == 'all' else list(cams)
    cpId = np.asarray(cpId)
    chkId = s0.OP.id if chkId is None else np.asarray(chkId)
    keepId = np.union1d(cpId, chkId)
    xy = lenscorr1(s0)
    pt_start, tri_start, Xs, xs, tris = [0], [0], [], [], []
    for ci in cams:
        IO = s0.IO.val[:, ci]
        rows = np.flatnonzero(s0.IP.cam == ci)
        ids = s0.OP.id[s0.IP.pt[rows]]
        is_cp = np.isin(ids, cpId)
        if np.count_nonzero(is_cp) > 3:
            T, A = largesttriangle(xy[:, rows[is_cp]])
            take = (np.arange(len(A)) < n) & (A >= v * A[0])
            tryId = ids[is_cp][T[take]]
        elif np.count_nonzero(is_cp) == 3:
            tryId = ids[is_cp][None, :]
        else:
            tryId = np.zeros((0, 3), int)
        keep = np.isin(ids, keepId)
        pt2 = xy[:, rows[keep]]
        xs.append(np.stack([(pt2[0] - IO[1]) / -IO[0], (pt2[1] - IO[2]) / -IO[0]]))   # K\homogeneous(pt2)
        Xs.append(s0.OP.val[:, s0.IP.pt[rows[keep]]])
        visId = ids[keep]
        for useId in tryId:          # pm_resect_3pt takes the three points in the order they are seen (use mask)
            tris.append(np.flatnonzero(np.isin(visId, useId)))
        pt_start.append(pt_start[-1] + int(np.count_nonzero(keep)))
        tri_start.append(len(tris))
    for t in tris:
        if len(t) != 3:
            raise ValueError('Can only use 3 points for resection')
    P, rms = _hip.resect_poses(pt_start, np.concatenate(Xs, 1) if Xs else np.zeros((3, 0)),
                               np.concatenate(xs, 1) if xs else np.zeros((2, 0)), tri_start,
                               np.array(tris, np.int32).reshape(-1, 3), device=device)
    fail = False
    for k, ci in enumerate(cams):
        if np.all(np.isfinite(P[k])):
            nv = np.linalg.svd(P[k])[2][-1]          # euclidean(null(P)), as resect() above
            s.EO.val[:3, ci] = nv[:3] / nv[3]
            s.EO.val[3:6, ci] = derotmat3d(P[k][:, :3])
        else:
            fail = True
            s.EO.val[:6, ci] = np.nan
    return s, rms, fail

def forwintersect(s0, ids='all', skipPrior=False, device=0):
    """Object points by forward intersection of the lens-corrected image rays: the point minimising the summed
    squared distance to its rays (pm_forwintersect3.m:55-82), the per-point 3 x 3 systems built and solved on
    the GPU (dbat_hip_forwintersect: one lane per image ray in the point-major batches of the bundle core).
    Points seen from fewer than two stations become NaN (pm_multiforwintersect.m:41).  With skipPrior, points
    that are fixed or carry prior observations keep their values (forwintersect.m:32-36)."""
    from . import _hip
    from .dbatstruct import copy_struct
    if not np.all(np.isfinite(s0.EO.val)):
        raise ValueError('Bad or uninitialized EO data')
    if not np.all(np.isfinite(s0.IO.val)):
        raise ValueError('Bad or uninitialized IO data')
    s = copy_struct(s0)
    npnt = s0.OP.val.shape[1]
    do = np.ones(npnt, bool) if isinstance(ids, str) and ids == 'all' else np.isin(s0.OP.id, ids)
    if skipPrior:
        do &= np.all(s0.bundle.est.OP, 0) & ~np.any(s0.prior.OP.use, 0)
    h = _hip.Handle(s0, device=device)
    try:
        s.OP.val = h.forwintersect(h.serialize(), s0.OP.val, skip=~do)
    finally:
        h.close()
    return s


resect_hip = resect                   # (names of rounds 2-3)
forwintersect_hip = forwintersect
