"""Text-table inputs of DBAT's XML script layer -> DBAT struct, and initial
object points by forward intersection.

Host-side mirror of `file/loadimagepts.m`, `file/loadeotable.m`
(comma-separated tables with `#` comments, `data/script/*/...txt`), the
camera conventions of `classes/@DBATCamera/DBATCamera.m:58-135` (py, K, P are
stored with PhotoModeler's sign and negated internally; aspect difference
= 1 - aspect; pixel size = sensor ./ image) and `script/setdbatcamsandimages.m`.
`forwintersect` supplies initial OP values as `photogrammetry/forwintersect.m`
does (linear intersection of the rays); it is not on the measured path.
"""
from __future__ import annotations

import lzma

import numpy as np

from .dbatstruct import make_struct


def load_table(path):
    """Comma-separated numeric table, `#` comment lines, .xz transparently."""
    opener = lzma.open if str(path).endswith('.xz') else open
    rows = []
    with opener(path, 'rt') as fh:
        for line in fh:
            line = line.strip()
            if not line or line.startswith('#'):
                continue
            rows.append([float(t) for t in line.split(',') if t.strip() != ''])
    return np.array(rows, float)


def camera_io(cc, pp, K, P, aspect=1.0, skew=0.0):
    """IO column [cc; px; py; as; sk; K..; P..] in DBAT's internal convention
    from storable (PhotoModeler-sign) values (DBATCamera.m:58-135)."""
    return np.concatenate([[cc, pp[0], -pp[1], 1.0 - aspect, skew], -np.asarray(K, float),
                           -np.asarray(P, float)])


def struct_from_tables(io_col, sensor, imsz, eo_table, mark_table, mark_fmt='im,id,x,y', sxy=1.0,
                       distModel=3, nK=3, nP=2, eo_degrees=True):
    """One shared camera, EO table `id,x,y,z,omega,phi,kappa`, image points."""
    order = np.argsort(eo_table[:, 0], kind='stable')
    eo_table = eo_table[order]
    nc = eo_table.shape[0]
    im_ids = eo_table[:, 0].astype(np.int64)
    EO = eo_table[:, 1:7].T.copy()
    if eo_degrees:
        EO[3:6] = np.deg2rad(EO[3:6])
    cols = [c.strip() for c in mark_fmt.split(',')]
    im = mark_table[:, cols.index('im')].astype(np.int64)
    pid = mark_table[:, cols.index('id')].astype(np.int64)
    xy = mark_table[:, [cols.index('x'), cols.index('y')]]
    ids = np.unique(pid)
    pt = np.searchsorted(ids, pid)
    cam = np.searchsorted(im_ids, im)
    order = np.lexsort((pt, cam))                     # image-major, ascending OP
    cam, pt, xy = cam[order], pt[order], xy[order]
    px = np.array([sensor[0] / imsz[0], sensor[1] / imsz[1]])
    IO = np.tile(np.asarray(io_col, float)[:, None], (1, nc))
    OP = np.full((3, len(ids)), np.nan)
    s = make_struct(IO, EO, OP, xy.T, cam, pt, np.tile(px[:, None], (1, nc)), ip_std=float(sxy),
                    distModel=distModel, nK=nK, nP=nP)
    s.OP.id = ids
    s.EO.id = im_ids
    return s


def _rot(ang):
    so, co = np.sin(ang[0]), np.cos(ang[0])
    sp, cp = np.sin(ang[1]), np.cos(ang[1])
    sk, ck = np.sin(ang[2]), np.cos(ang[2])
    R1 = np.array([[1, 0, 0], [0, co, -so], [0, so, co]])
    R2 = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    R3 = np.array([[ck, -sk, 0], [sk, ck, 0], [0, 0, 1]])
    return R1 @ R2 @ R3


def forwintersect(s):
    """Initial object points: least-squares intersection of the image rays
    (lens-corrected with the current IO), photogrammetry/forwintersect.m:27-46."""
    nK, nP = s.IO.model.nK, s.IO.model.nP
    cam, pt = s.IP.cam, s.IP.pt
    IO = s.IO.val[:, cam]
    sz = s.IO.sensor.pxSize[0, cam]
    x = np.stack([sz * s.IP.val[0] - IO[1], -sz * s.IP.val[1] - IO[2]])
    a = np.stack([(1 + IO[3]) * x[0] + IO[4] * x[1], x[1]])
    rho = np.sum(a * a, 0)
    rs = np.zeros_like(rho)
    pw = np.ones_like(rho)
    for j in range(nK):
        pw = pw * rho
        rs = rs - IO[5 + j] * pw
    l = a * (1 + rs)
    if nP >= 2:
        p1, p2 = -IO[5 + nK], -IO[6 + nK]
        pTu = p1 * a[0] + p2 * a[1]
        l = l + np.stack([p1 * rho + 2 * pTu * a[0], p2 * rho + 2 * pTu * a[1]])
    # ray in the camera frame: lhs = -f X/Z = l  =>  X ~ [l; -f]
    d_cam = np.stack([l[0], l[1], -IO[0]])
    M = np.stack([_rot(s.EO.val[3:6, i]) for i in range(s.EO.val.shape[1])])     # (nc,3,3)
    d = np.einsum('nij,jn->in', M[cam], d_cam)
    d = d / np.linalg.norm(d, axis=0)
    c = s.EO.val[:3, cam]
    npnt = s.OP.val.shape[1]
    A = np.zeros((npnt, 3, 3))
    b = np.zeros((npnt, 3))
    P = np.eye(3)[None] - np.einsum('in,jn->nij', d, d)                          # I - d d'
    np.add.at(A, pt, P)
    np.add.at(b, pt, np.einsum('nij,jn->ni', P, c))
    s.OP.val = np.linalg.solve(A, b[:, :, None])[:, :, 0].T.copy()
    return s
