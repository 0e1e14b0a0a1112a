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
    s.IO.sensor.ssSize = np.tile(np.asarray(sensor, float)[:, None], (1, nc))
    s.IO.sensor.imSize = np.tile(np.asarray(imsz, float)[:, None], (1, nc))
    return s


def struct_from_script(io_col, sensor, imsz, im_ids, mark_tables, ctrl=None, check=None,
                       distModel=3, nK=3, nP=2, im_names=None):
    """DBAT struct of an XML-script project without loaded EO: one shared
    camera, images im_ids, image-point tables [(table, 'id,im,x,y', sxy), ...]
    (script/parseimagepts.m: each file with its own standard deviation),
    control and check points as loadpm.loadcpt returns them
    (script/setdbatpts.m:1-60): OP ids are the union of all three lists,
    control points with a non-zero standard deviation become prior
    observations, check points only keep their reference position.  EO and
    free OP start as NaN (to be filled by resection / forward intersection)."""
    im_ids = np.asarray(im_ids, np.int64)
    nc = len(im_ids)
    im, pid, xy, std = [], [], [], []
    for table, fmt, sxy in mark_tables:
        cols = [c.strip() for c in fmt.split(',')]
        im.append(table[:, cols.index('im')].astype(np.int64))
        pid.append(table[:, cols.index('id')].astype(np.int64))
        xy.append(table[:, [cols.index('x'), cols.index('y')]])
        std.append(np.full(len(table), float(sxy)))
    im, pid, xy, std = np.concatenate(im), np.concatenate(pid), np.vstack(xy), np.concatenate(std)
    empty = dict(id=np.zeros(0, np.int64), name=[], pos=np.zeros((3, 0)), std=np.zeros((3, 0)))
    ctrl = ctrl or empty
    check = check or empty
    ids = np.unique(np.concatenate([np.asarray(ctrl['id'], np.int64), np.asarray(check['id'], np.int64), pid]))
    pt = np.searchsorted(ids, pid)
    cam = np.searchsorted(im_ids, im)
    order = np.lexsort((pid, im))                      # sortrows([im; id]') (setdbatpts.m:43)
    cam, pt, xy, std = cam[order], pt[order], xy[order], std[order]
    px = np.array([sensor[0] / imsz[0], sensor[1] / imsz[1]])
    IO = np.tile(np.asarray(io_col, float)[:, None], (1, nc))
    s = make_struct(IO, np.zeros((6, nc)), np.zeros((3, len(ids))), xy.T, cam, pt, np.tile(px[:, None], (1, nc)),
                    ip_std=np.tile(std[None, :], (2, 1)), distModel=distModel, nK=nK, nP=nP)
    s.EO.val[:] = np.nan
    s.OP.val[:] = np.nan
    s.OP.id, s.OP.rawId, s.EO.id = ids, ids.copy(), im_ids
    if im_names is not None:
        s.EO.name = list(im_names)
    s.OP.label = [''] * len(ids)
    s.prior.OP.isCtrl = np.isin(ids, ctrl['id'])
    s.prior.OP.isCheck = np.isin(ids, check['id'])
    for pts in (ctrl, check):                          # check points last, as setdbatpts.m:14-26
        for k, i in enumerate(np.searchsorted(ids, np.asarray(pts['id'], np.int64))):
            s.prior.OP.val[:, i] = pts['pos'][:, k]
            s.prior.OP.std[:, i] = pts['std'][:, k]
            s.OP.label[i] = pts['name'][k]
    sd = s.prior.OP.std
    s.prior.OP.use = ~np.isnan(sd) & (sd != 0) & ~s.prior.OP.isCheck[None, :]
    s.IO.sensor.ssSize = np.tile(np.asarray(sensor, float)[:, None], (1, nc))
    s.IO.sensor.imSize = np.tile(np.asarray(imsz, float)[:, None], (1, nc))
    return s


def set_script_defaults(s):
    """The script operations `set_initial_values` io/op = loaded and
    `set_bundle_estimate_params` io=false, eo=true, op=default
    (script/private/parsesetinitialopvalues.m, parsesetbundleestop.m:56-59):
    loaded prior positions become the initial OP values; every EO parameter,
    every non-control point and every control point coordinate with a
    non-zero standard deviation is estimated."""
    s.OP.val = s.prior.OP.val.copy()
    s.bundle.est.IO[:] = False
    s.bundle.est.EO[:] = True
    s.bundle.est.OP = ~s.prior.OP.isCtrl[None, :] | (s.prior.OP.std != 0)
    return s


def forwintersect(s, device=0):
    """Initial object points by forward intersection of every point
    (photogrammetry/forwintersect.m:27-46), on the device: dbat_amd.initial.forwintersect."""
    from .initial import forwintersect as fwd
    return fwd(s, 'all', device=device)
