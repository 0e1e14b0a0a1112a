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


def forwintersect(s):
    """Initial object points by forward intersection of every point
    (photogrammetry/forwintersect.m:27-46); see dbat_amd.initial."""
    from .initial import forwintersect as fwd
    return fwd(s, 'all')
