"""The DBAT struct: the data boundary of `bundle()`.

Host-side mirror of the struct produced by the reference's
`misc/prob2dbatstruct.m` (field documentation at prob2dbatstruct.m:12-186).
Only the fields the bundle hot path reads or writes are carried (SURVEY.md
section 8(b)):

    s.IO.val            nIOrows x nImages  [cc; px; py; as; sk; K1..; P1..]
    s.IO.model          distModel (nImages,), nK, nP
    s.IO.sensor.pxSize  2 x nImages (both rows = pixel height)
    s.IO.struct.block   nIOrows x nImages integer block ids (shared IO)
    s.EO.val            6 x nImages [X;Y;Z;omega;phi;kappa] (radians)
    s.EO.struct.block   6 x nImages
    s.OP.val            3 x nOP
    s.IP.val / .std     2 x nIP pixels, image-major, ascending OP within image
    s.IP.cam / .pt      image / OP column of every IP column (0-based here;
                        replaces the sparse IP.vis / IP.ix pair of the
                        reference, prob2dbatstruct.m:343-365)
    s.bundle.est.*      logical masks, same shapes as the value arrays
    s.prior.*.use/val/std   prior observations
    s.post.*            filled by bundle()

Indices are 0-based; array shapes and row meanings are the reference's.
"""
from __future__ import annotations

import copy
import types

import numpy as np

NS = types.SimpleNamespace


def make_struct(IO, EO, OP, ip_val, ip_cam, ip_pt, pxSize, *, ip_std=None,
                distModel=3, nK=3, nP=2, estIO=None, estEO=None, estOP=None,
                IOblock=None, EOblock=None, priorIO=None, priorEO=None,
                priorOP=None):
    """Assemble a DBAT struct from plain arrays.

    Defaults follow prob2dbatstruct.m:380-397: IO fixed, EO free, OP free, no
    prior observations, one IO block shared by all images, distinct EO blocks.
    `prior*` are (use, val, std) triples.
    """
    IO = np.array(IO, dtype=float, order='F')
    EO = np.array(EO, dtype=float, order='F')
    OP = np.array(OP, dtype=float, order='F')
    ip_val = np.array(ip_val, dtype=float, order='F')
    nc, no = EO.shape[1], ip_val.shape[1]
    if IO.shape != (5 + nK + nP, nc):
        raise ValueError('IO.val must be (5+nK+nP) x nImages')
    if EO.shape[0] != 6 or OP.shape[0] != 3 or ip_val.shape[0] != 2:
        raise ValueError('bad array shapes')
    px = np.asarray(pxSize, float)
    if px.ndim == 0:
        px = np.full((2, nc), float(px))
    elif px.ndim == 1:
        px = np.tile(px[None, :], (2, 1))
    std = np.ones((2, no)) if ip_std is None else np.asarray(ip_std, float)
    if std.ndim == 0:
        std = np.full((2, no), float(std))
    # IP.sigmas = the distinct standard deviations (one sort of 2*no values unless they are all equal)
    sigmas = np.array([std.flat[0]]) if std.size and std.min() == std.max() else np.unique(std)

    # (every array column-major, as MATLAB holds it and as the C ABI takes it: marshalling a struct is then a matter of
    # views -- dbat_amd._hip._flat -- and a second bundle() on a 10 M observation project finds its cached handle in
    # milliseconds)
    def prior(p, val):
        if p is None:
            return NS(use=np.zeros(val.shape, bool, order='F'),
                      val=np.full(val.shape, np.nan, order='F'),
                      std=np.full(val.shape, np.nan, order='F'))
        return NS(use=np.array(p[0], bool, order='F'), val=np.array(p[1], float, order='F'),
                  std=np.array(p[2], float, order='F'))

    s = NS()
    s.IO = NS(val=IO,
              model=NS(distModel=np.full(nc, int(distModel)), nK=int(nK), nP=int(nP)),
              sensor=NS(pxSize=np.array(px, float, order='F')),
              struct=NS(block=(np.ones(IO.shape, np.int32, order='F') if IOblock is None
                               else np.array(IOblock, np.int32, order='F'))))
    s.EO = NS(val=EO,
              struct=NS(block=np.array(np.tile(np.arange(1, nc + 1), (6, 1)) if EOblock is None else EOblock, np.int32, order='F')))
    s.OP = NS(val=OP, id=np.arange(1, OP.shape[1] + 1))      # prob2dbatstruct.m: OP.id
    s.IP = NS(val=ip_val, std=np.array(std, float, order='F'),
              cam=np.array(ip_cam, np.int32),              # (int32: what the C ABI takes -- no conversion per bundle() call)
              pt=np.array(ip_pt, np.int32),
              sigmas=sigmas)
    s.bundle = NS(est=NS(
        IO=np.zeros(IO.shape, bool, order='F') if estIO is None else np.array(estIO, bool, order='F'),
        EO=np.ones(EO.shape, bool, order='F') if estEO is None else np.array(estEO, bool, order='F'),
        OP=np.ones(OP.shape, bool, order='F') if estOP is None else np.array(estOP, bool, order='F')),
        serial=None, deserial=None)
    s.prior = NS(IO=prior(priorIO, IO), EO=prior(priorEO, EO), OP=prior(priorOP, OP))
    s.post = NS()
    validate(s)
    return s


def validate(s):
    """Invariants the hot path relies on.

    IP columns must be image-major with ascending OP index inside each image
    (prob2dbatstruct.m:349-365); the reference's residual-only and Jacobian
    branches of multi_res only agree on row order under that invariant
    (multi_res.m:46-53 vs :143-144).
    """
    cam, pt = s.IP.cam, s.IP.pt
    nc, npnt = s.EO.val.shape[1], s.OP.val.shape[1]
    if cam.shape != pt.shape or cam.shape[0] != s.IP.val.shape[1]:
        raise ValueError('IP.cam/IP.pt/IP.val size mismatch')
    if cam.size:
        if cam.min() < 0 or cam.max() >= nc or pt.min() < 0 or pt.max() >= npnt:
            raise ValueError('IP.cam/IP.pt out of range')
        key = cam.astype(np.int64) * np.int64(npnt) + pt
        if np.any(np.diff(key) <= 0):
            raise ValueError('IP columns must be image-major with strictly '
                             'ascending OP index within each image')
    dm = np.unique(s.IO.model.distModel)
    if dm.size != 1:
        # brown_euler_cam4.m:31-33
        raise ValueError('Mixed lens distortion models not implemented.')


def copy_struct(s):
    return copy.deepcopy(s)


def share_struct(s):
    """A new struct whose namespaces are copies and whose ARRAYS are shared with s: what bundle() works on.  bundle() never
    writes into an array of its input -- it replaces IO.val / EO.val / OP.val / prior.*.use by new arrays and adds s.post
    -- so the caller's struct stays as it was (MATLAB's value semantics, bundle.m:1) without a pass over the 10 M
    observations of a large project; the result shares its untouched arrays (IP.*, masks, blocks) with the input."""
    if isinstance(s, NS):
        return NS(**{k: share_struct(v) for k, v in vars(s).items()})
    return s


def seteoest_depend(s, camNo=0):
    """Datum by dependency: misc/seteoest.m:90-128 ('depend').

    Fixes all six EO parameters of the base camera and the largest camera
    offset coordinate among the other cameras.
    """
    base = s.EO.val[:3, camNo]
    offset = s.EO.val[:3, :] - base[:, None]
    i, j = np.nonzero(offset == offset.max())
    s.bundle.est.EO[:] = True
    s.bundle.est.EO[:, camNo] = False
    s.bundle.est.EO[i, j] = False
    return s
