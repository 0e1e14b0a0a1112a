"""PhotoModeler text-export loader -> DBAT struct.

Host-side mirror of the reference's `file/loadpm.m:108-330` (file format) and
`misc/prob2dbatstruct.m:198-398` (conventions: py flip :227, K/P sign flip
:233-236, angles stored kappa,phi,omega in degrees :265-267, pixel size /
aspect :243-254, IP ordering :343-365).  Needed to run the reference's own
demo inputs (known-answer fixtures) through `bundle()`.
"""
from __future__ import annotations

import numpy as np

from .dbatstruct import make_struct


def _nums(line):
    out = []
    for tok in line.split():
        try:
            out.append(float(tok))
        except ValueError:
            break
    return out


def loadpm(path):
    """Parse a PhotoModeler export (loadpm.m:108-330).  Returns a dict `prob`."""
    with open(path, 'rt') as fh:
        lines = fh.read().split('\n')
    it = iter(lines)
    title = next(it)
    tol = _nums(next(it))
    defStd = _nums(next(it))
    defCam = _nums(next(it))
    defCamStd = _nums(next(it))
    imSz = tol[2:4] if len(tol) > 2 else [np.nan, np.nan]
    images = []
    for line in it:                                   # loadpm.m:125-170
        toks = line.split()
        if not toks or not toks[0].lstrip('-').isdigit():
            break
        name = line.strip()[len(toks[0]):].strip()
        outer = _nums(next(it))[1:]
        outerStd = _nums(next(it))[1:]
        next(it)                                      # outerCov (blank => NaN)
        inner = _nums(next(it))[1:]
        innerStd = _nums(next(it))[1:]
        images.append(dict(imName=name, outer=outer, outerStd=outerStd,
                           inner=inner, innerStd=innerStd))

    def table(ncol):
        rows = []
        for line in it:
            v = _nums(line)
            if not v:
                break
            rows.append(v)
        return np.array(rows, float).reshape(-1, ncol) if rows else np.zeros((0, ncol))

    ctrlPts = table(7)                                # loadpm.m:196-216
    objPts = table(7)                                 # loadpm.m:221-241
    markPts = table(6)                                # loadpm.m:248-268
    return dict(title=title, tol=tol, defStd=defStd, defCam=np.array(defCam),
                defCamStd=np.array(defCamStd), imSz=np.array(imSz, float),
                images=images, ctrlPts=ctrlPts, objPts=objPts, markPts=markPts)


def prob2dbatstruct(prob, distModel=1):
    """prob -> DBAT struct (prob2dbatstruct.m:198-398), block-invariant IO.

    Default distModel=1 as in the reference (prob2dbatstruct.m:396-397); every
    demo overrides it to 3.
    """
    nImages = len(prob['images'])
    nK, nP = 3, 2
    inner = np.tile(np.asarray(prob['defCam'], float)[:, None], (1, nImages))
    imSz = np.tile(np.asarray(prob['imSz'], float)[:, None], (1, nImages))
    IO = np.full((5 + nK + nP, nImages), np.nan)
    IO[1:3] = np.diag([1.0, -1.0]) @ inner[1:3]       # :227 flip y
    IO[0] = inner[0]
    IO[5:5 + nK] = -inner[5:5 + nK]                   # :233
    IO[5 + nK:5 + nK + nP] = -inner[5 + nK:5 + nK + nP]   # :236
    sensorSize = inner[3:5]
    pixelSize = sensorSize / imSz
    aspect = 1 - pixelSize[0] / pixelSize[1]          # :247
    pixelSize = pixelSize[[1, 1]]
    IO[3] = aspect
    IO[4] = 0.0
    outer = np.array([im['outer'] for im in prob['images']], float).T
    EO = np.full((6, nImages), np.nan)
    EO[:3] = outer[:3]
    EO[3:6] = outer[[5, 4, 3]] / 180 * np.pi          # :265-267
    # Object points (:296-320)
    objPts, ctrlPts = prob['objPts'], prob['ctrlPts']
    ids = np.unique(np.concatenate([ctrlPts[:, 0], objPts[:, 0]])).astype(np.int64)
    order = np.argsort(objPts[:, 0], kind='stable')
    OPid = objPts[order, 0].astype(np.int64)
    if len(OPid) != len(ids):
        raise ValueError('control points without object point entries unsupported')
    OP = objPts[order, 1:4].T.copy()
    priorCP = np.full(OP.shape, np.nan)
    priorCPstd = np.full(OP.shape, np.nan)
    isCtrl = np.isin(OPid, ctrlPts[:, 0])
    if len(ctrlPts):
        _, ia, ib = np.intersect1d(OPid, ctrlPts[:, 0], return_indices=True)
        priorCP[:, ia] = ctrlPts[ib, 1:4].T
        priorCPstd[:, ia] = ctrlPts[ib, 4:7].T
    # Mark points (:322-365): image-major, ascending OP id inside an image.
    mp = prob['markPts']
    cols_cam, cols_pt, cols_xy, cols_std = [], [], [], []
    pos = {int(v): k for k, v in enumerate(OPid)}
    for i in range(nImages):
        m = mp[mp[:, 0] == i]
        m = m[np.argsort(m[:, 1], kind='stable')]
        for row in m:
            k = pos.get(int(row[1]))
            if k is None:
                continue
            cols_cam.append(i); cols_pt.append(k)
            cols_xy.append(row[2:4]); cols_std.append(row[4:6])
    ip_val = np.array(cols_xy, float).T
    ip_std = np.array(cols_std, float).T
    if np.any(ip_std == 0):                           # :367-375
        ip_std[:] = 1.0
    estOP = ~(priorCPstd == 0)                        # :387
    useOP = np.tile(isCtrl & ~np.all(priorCPstd == 0, 0), (3, 1))
    s = make_struct(IO, EO, OP, ip_val, cols_cam, cols_pt, pixelSize,
                    ip_std=ip_std, distModel=distModel, nK=nK, nP=nP,
                    estOP=estOP, priorOP=(useOP, priorCP, priorCPstd))
    s.OP.id = OPid
    s.IO.sensor.ssSize = sensorSize
    s.IO.sensor.imSize = imSz
    s.prior.OP.isCtrl = isCtrl
    s.EO.name = [im['imName'] for im in prob['images']]
    return s


def loadcpt(path):
    """Control point file `id,name,x,y,z[,sx,sy,sz]` (file/loadcpt.m)."""
    ids, names, pos, std = [], [], [], []
    with open(path, 'rt') as fh:
        for line in fh:
            line = line.strip()
            if not line or line.startswith('#'):
                continue
            t = [x.strip() for x in line.split(',')]
            ids.append(int(t[0])); names.append(t[1])
            pos.append([float(v) for v in t[2:5]])
            std.append([float(v) for v in t[5:8]] if len(t) >= 8 else [0.0, 0.0, 0.0])
    return dict(id=np.array(ids), name=names, pos=np.array(pos).T, std=np.array(std).T)


def setcpt(s, pts):
    """Install control points (misc/matchcpt.m + misc/setcpt.m:1-40).

    Fixed control points (std==0) are removed from the unknowns and carry no
    prior observation; others become prior observations.
    """
    _, i, j = np.intersect1d(s.OP.id, pts['id'], return_indices=True)
    s.prior.OP.val[:, i] = pts['pos'][:, j]
    s.OP.val[:, i] = pts['pos'][:, j]
    s.prior.OP.std[:, i] = pts['std'][:, j]
    s.prior.OP.isCtrl[i] = True
    if not hasattr(s.OP, 'label') or s.OP.label is None:
        s.OP.label = [''] * s.OP.val.shape[1]
    for a, b in zip(i, j):                                   # setcpt.m:33-37: non-blank labels
        if pts['name'][b]:
            s.OP.label[a] = pts['name'][b]
    isFixed = np.all(pts['std'][:, j] == 0, 0)
    s.prior.OP.use[:, i] = np.tile(~isFixed, (3, 1))
    s.bundle.est.OP[:, i] = np.tile(~isFixed, (3, 1))
    return s
