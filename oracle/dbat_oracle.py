"""CPU oracle: NumPy/SciPy restatement of DBAT's damped bundle-adjustment hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (`dbat_amd/`) may
import this module.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` use it, and only as the checker / the timed
CPU baseline -- never as the thing shipped.

Parity pinning: the reference is MATLAB and cannot be run here (no
matlab/octave/mex in the image).  The oracle is pinned instead against the
reference's own committed end-to-end reports (tests/golden/camcal_*: sigma0,
parameter count, converged IO/EO values to their printed precision) and by the
reference's own derivative self-test method (analytic vs central-difference
Jacobians, thresholds 1e-8, `cameramodel/private/full_self_test.m:17-56`).
The solver boundary itself (MATLAB R2020a sparse `mldivide`) is third-party
arithmetic absent from /root/reference: "parity unpinned" at that boundary
except through those end-to-end reports.

Every function cites the reference file:line it restates (paths relative to
/root/reference/code/).  Arrays follow the reference's column conventions
(IO.val is nIOrows x nImages etc.); indices are 0-based here.
"""
from __future__ import annotations

import types
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
from scipy.sparse.csgraph import structural_rank

NS = types.SimpleNamespace

# ----------------------------------------------------------------------------
# F7: 3-D side primitives
# ----------------------------------------------------------------------------

_GEN = {
    1: np.array([[0., 0, 0], [0, 0, -1], [0, 1, 0]]),
    2: np.array([[0., 0, 1], [0, 0, 0], [-1, 0, 0]]),
    3: np.array([[0., -1, 0], [1, 0, 0], [0, 0, 0]]),
}


def _rot(axis, a):
    """bundle/cameramodel/eulerrotmat.m:129-147 (R1, R2, R3)."""
    c, s = np.cos(a), np.sin(a)
    if axis == 1:
        return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])
    if axis == 2:
        return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def eulerrotmat(ang, seq=123, fixed=False, jac=False):
    """bundle/cameramodel/eulerrotmat.m:81-124.

    M = M1*M2*M3 (moving axes) or M3*M2*M1 (fixed axes); dM.dA is 9x3 with
    column-major vec(M) rows.
    """
    i1, i2, i3 = seq // 100, (seq % 100) // 10, seq % 10
    M1, M2, M3 = _rot(i1, ang[0]), _rot(i2, ang[1]), _rot(i3, ang[2])
    P1, P2, P3 = _GEN[i1], _GEN[i2], _GEN[i3]
    M = M3 @ M2 @ M1 if fixed else M1 @ M2 @ M3
    if not jac:
        return M
    if fixed:
        dA1, dA2, dA3 = M @ P1, M3 @ M2 @ P2 @ M1, P3 @ M
    else:
        dA1, dA2, dA3 = P1 @ M, M1 @ M2 @ P2 @ M3, M @ P3
    dA = np.stack([dA1.flatten('F'), dA2.flatten('F'), dA3.flatten('F')], 1)
    return M, dA


def world2cam(P, p0, M, jac=False):
    """bundle/cameramodel/world2cam.m:46-49,76-82; lin3.m:41,63-66; xlat3.m:41.

    Q = M*(P - p0).  Jacobian blocks per point: dP = M, dP0 = -M,
    dM = kron(X', I3) (3x9, column-major vec(M)).
    """
    X = P - p0[:, None]
    Q = M @ X
    if not jac:
        return Q
    n = P.shape[1]
    dP = np.broadcast_to(M, (n, 3, 3))
    dP0 = np.broadcast_to(-M, (n, 3, 3))
    dM = np.zeros((n, 3, 9))
    for b in range(3):
        for a in range(3):
            dM[:, a, a + 3 * b] = X[b]
    return Q, dP, dP0, dM


def pinhole(P, jac=False):
    """bundle/cameramodel/pinhole.m:39,54-66."""
    Q = P[:2] / P[2]
    if not jac:
        return Q
    n = P.shape[1]
    d = np.zeros((n, 2, 3))
    i3 = 1.0 / P[2]
    d[:, 0, 0] = i3
    d[:, 1, 1] = i3
    d[:, 0, 2] = -P[0] * i3 ** 2
    d[:, 1, 2] = -P[1] * i3 ** 2
    return Q, d


def eulerpinhole2(P, p0, ang, f, jac=False):
    """bundle/cameramodel/eulerpinhole2.m:51-67,97-106.

    Q = f*pinhole(M'*(P-p0)), M = eulerrotmat(ang,123,moving).
    Returns Q (2xn) and dict(dA, dF, dP0, dP) of per-point blocks.
    """
    if not jac:
        return f * pinhole(world2cam(P, p0, eulerrotmat(ang).T))
    M, dM = eulerrotmat(ang, 123, False, jac=True)
    perm = np.array([1, 4, 7, 2, 5, 8, 3, 6, 9]) - 1  # eulerpinhole2.m:57
    MT = M.T
    dMT = dM[perm, :]
    W2C, dW_P, dW_P0, dW_M = world2cam(P, p0, MT, jac=True)
    PH, dPH = pinhole(W2C, jac=True)
    Q = f * PH
    d = {
        'dF': PH.T[:, :, None].copy(),                      # :97  vec(PH)
        'dA': f * np.einsum('nij,njk,kl->nil', dPH, dW_M, dMT),  # :100
        'dP': f * np.einsum('nij,njk->nik', dPH, dW_P),     # :103
        'dP0': f * np.einsum('nij,njk->nik', dPH, dW_P0),   # :106
    }
    return Q, d


# ----------------------------------------------------------------------------
# F8: 2-D linear chain
# ----------------------------------------------------------------------------

def scale2(U, k):
    """bundle/cameramodel/scale2.m:41."""
    return k * U


def aniscale2(U, k):
    """bundle/cameramodel/aniscale2.m:42-43."""
    return k[:, None] * U


def aniscale2b(U, k):
    """bundle/cameramodel/aniscale2b.m:41."""
    return np.array([1 + k, 1.0])[:, None] * U


def xlat2(U, c):
    """bundle/cameramodel/xlat2.m:41."""
    return U + c[:, None]


def affine2mat(b):
    """bundle/cameramodel/affine2mat.m:38."""
    return np.array([[1 + b[0], b[1]], [0.0, 1.0]])


def affine2(U, b):
    """bundle/cameramodel/affine2.m:41-42."""
    return affine2mat(b) @ U


def skew(U, k):
    """bundle/cameramodel/skew.m:41."""
    return np.array([[1.0, k], [0.0, 1.0]]) @ U


# ----------------------------------------------------------------------------
# F9: Brown lens distortion
# ----------------------------------------------------------------------------

def lens_rad2(U):
    """bundle/cameramodel/lens_rad2.m:39."""
    return np.sum(U ** 2, 0)


def power_vec(x, nn):
    """bundle/cameramodel/power_vec.m:42,62-67: rows x.^1 .. x.^nn; dv = j*x^(j-1)."""
    e = np.arange(1, nn + 1)[:, None]
    v = np.power(x[None, :], e)
    dv = e * np.power(x[None, :], e - 1)
    return v, dv


def rad_scale(u, c, jac=False):
    """bundle/cameramodel/rad_scale.m:44-50,72-75: v = sum_j c_j r2^j."""
    r2 = lens_rad2(u)
    pv, dpv = power_vec(r2, len(c))
    v = pv.T @ c
    if not jac:
        return v
    dC = pv.T                                 # (n, nC)
    dU = (dpv.T @ c)[:, None] * (2 * u.T)     # (n, 2): kron(I,c')*dpv*dr2
    return v, dC, dU


def tang_scale(u, p, jac=False):
    """bundle/cameramodel/tang_scale.m:42-45,66-87."""
    uTu = np.sum(u ** 2, 0)
    pTu = p @ u
    v = p[:, None] * uTu + 2 * pTu * u
    if not jac:
        return v
    n = u.shape[1]
    u12, u22, u1u2 = u[0] ** 2, u[1] ** 2, u[0] * u[1]
    dP = np.zeros((n, 2, 2))
    dP[:, 0, 0] = uTu + 2 * u12
    dP[:, 0, 1] = 2 * u1u2
    dP[:, 1, 0] = 2 * u1u2
    dP[:, 1, 1] = uTu + 2 * u22
    dU = np.zeros((n, 2, 2))
    dU[:, 0, 0] = 2 * (2 * p[0] * u[0] + pTu)
    dU[:, 1, 0] = 2 * (p[0] * u[1] + p[1] * u[0])
    dU[:, 0, 1] = 2 * (p[0] * u[1] + p[1] * u[0])
    dU[:, 1, 1] = 2 * (2 * p[1] * u[1] + pTu)
    return v, dP, dU


def brown_rad(u, K, jac=False):
    """bundle/cameramodel/brown_rad.m:48-52,73-94: v = u .* rad_scale(u,K)."""
    n = u.shape[1]
    if len(K) == 0:
        v = np.zeros_like(u)
        if not jac:
            return v
        return v, np.zeros((n, 2, 0)), np.zeros((n, 2, 2))
    if not jac:
        return u * rad_scale(u, K)
    rs, dC, dUr = rad_scale(u, K, jac=True)
    v = u * rs
    dK = u.T[:, :, None] * dC[:, None, :]                       # :76-79
    dU = u.T[:, :, None] * dUr[:, None, :]                      # :82 u*drs.dU
    dU[:, 0, 0] += rs
    dU[:, 1, 1] += rs
    return v, dK, dU


def brown_tang(u, P, jac=False):
    """bundle/cameramodel/brown_tang.m:58-70,91-137."""
    n = u.shape[1]
    pm = len(P)
    if pm == 0:
        v = np.zeros_like(u)
        if not jac:
            return v
        return v, np.zeros((n, 2, 0)), np.zeros((n, 2, 2))
    if not jac:
        v = tang_scale(u, P[:2])
        if pm > 2:
            v = v * (1 + rad_scale(u, P[2:]))
        return v
    ts, dPt, dUt = tang_scale(u, P[:2], jac=True)
    v = ts
    if pm <= 2:
        return v, dPt, dUt
    rs, dC, dUr = rad_scale(u, P[2:], jac=True)
    v = ts * (1 + rs)
    dP = np.concatenate([(1 + rs)[:, None, None] * dPt,
                         ts.T[:, :, None] * dC[:, None, :]], 2)   # :95-104
    dU = dUt * (1 + rs)[:, None, None] + ts.T[:, :, None] * dUr[:, None, :]
    return v, dP, dU


def brown_dist(u, K, P, jac=False):
    """bundle/cameramodel/brown_dist.m:52-57,83-89: v = u + rad + tang."""
    if not jac:
        return u + brown_rad(u, K) + brown_tang(u, P)
    br, dK, dUr = brown_rad(u, K, jac=True)
    bt, dP, dUt = brown_tang(u, P, jac=True)
    v = u + br + bt
    dU = np.eye(2)[None] + dUr + dUt
    return v, dK, dP, dU


# ----------------------------------------------------------------------------
# F6: per-camera residual functions (lens distortion models 2..5)
# ----------------------------------------------------------------------------

_FLIP = np.array([1.0, -1.0])


def res_euler_brown(model, Q, q0, ang, f, u, sz, u0, K, P, b, jac=False):
    """bundle/cameramodel/res_euler_brown_{0,1,2,3}.m (distModel 2,3,4,5).

    _0: :80-90,140-166   _1: :84-95,149-178   _2: :85-95,162-177
    _3: :87-98,165-180.  v = eulerpinhole2(Q,q0,ang,-f) - rhs(u,...).
    Returns v (2xn) and, if jac, dict of per-point Jacobian blocks
    dQ,dQ0,dA (n,2,3), dF (n,2,1), dU0 (n,2,2), dK (n,2,nK), dP (n,2,nP),
    dB (n,2,2).
    """
    n = Q.shape[1]
    I2 = np.broadcast_to(np.eye(2), (n, 2, 2))
    flipM = np.diag(_FLIP)
    if not jac:
        lhs = eulerpinhole2(Q, q0, ang, -f)
        s = aniscale2(scale2(u, sz), _FLIP)
        if model == 2:
            rhs = brown_dist(xlat2(s, -u0), -K, -P)
        elif model == 3:
            rhs = brown_dist(affine2(xlat2(s, -u0), b), -K, -P)
        elif model == 4:
            rhs = affine2(brown_dist(xlat2(s, -u0), -K, -P), b)
        elif model == 5:
            rhs = skew(brown_dist(xlat2(aniscale2b(s, b[0]), -u0), -K, -P), b[1])
        else:
            raise ValueError('bad model')
        return lhs - rhs

    lhs, dl = eulerpinhole2(Q, q0, ang, -f, jac=True)
    s = aniscale2(scale2(u, sz), _FLIP)
    A = affine2mat(b)
    if model == 2:
        x = xlat2(s, -u0)
        l, dLK, dLP, dLU = brown_dist(x, -K, -P, jac=True)
        rhs = l
        dU0 = dLU                                   # _0:161  dL.dU*dX.dC
        dK, dP = dLK, dLP
        dB = np.zeros((n, 2, 2))
    elif model == 3:
        x = xlat2(s, -u0)
        a = A @ x
        l, dLK, dLP, dLU = brown_dist(a, -K, -P, jac=True)
        rhs = l
        dU0 = dLU @ A                               # _1:167-169
        dK, dP = dLK, dLP                           # _1:170-175
        dAB = np.zeros((n, 2, 2))                   # affine2.m:74-75
        dAB[:, 0, :] = x.T
        dB = -np.einsum('nij,njk->nik', dLU, dAB)   # _1:176-178
    elif model == 4:
        x = xlat2(s, -u0)
        l, dLK, dLP, dLU = brown_dist(x, -K, -P, jac=True)
        rhs = A @ l
        dU0 = A @ dLU                               # _2: dA.dU*dL.dU*dX.dC
        dK = np.einsum('ij,njk->nik', A, dLK)
        dP = np.einsum('ij,njk->nik', A, dLP)
        dAB = np.zeros((n, 2, 2))
        dAB[:, 0, :] = l.T
        dB = -dAB                                   # _2: -dA.dB
    elif model == 5:
        as_ = aniscale2b(s, b[0])
        x = xlat2(as_, -u0)
        l, dLK, dLP, dLU = brown_dist(x, -K, -P, jac=True)
        SK = np.array([[1.0, b[1]], [0.0, 1.0]])
        rhs = SK @ l
        dU0 = SK @ dLU
        dK = np.einsum('ij,njk->nik', SK, dLK)
        dP = np.einsum('ij,njk->nik', SK, dLP)
        dASK = np.zeros((n, 2, 1))                  # aniscale2b.m dK
        dASK[:, 0, 0] = s[0]
        dSKK = np.zeros((n, 2, 1))                  # skew.m dK
        dSKK[:, 0, 0] = l[1]
        dB = -np.concatenate([np.einsum('ij,njk,nkl->nil', SK, dLU, dASK),
                              dSKK], 2)             # _3:180
    else:
        raise ValueError('bad model')
    v = lhs - rhs
    d = {
        'dQ': dl['dP'], 'dQ0': dl['dP0'], 'dA': dl['dA'],
        'dF': -dl['dF'],
        'dU0': dU0, 'dK': dK, 'dP': dP, 'dB': dB,
    }
    del I2, flipM
    return v, d


# ----------------------------------------------------------------------------
# F2: serialisation indices
# ----------------------------------------------------------------------------

def _serializeblock(block, est, use_obs):
    """misc/buildserialindices.m:162-221 (serializeblock).

    Returns leading (bool array), serial(src,dest,obs) and deserial(dest,src)
    with `src`/`dest` into the column-major flattened parameter array /
    the block-local x (0-based).
    """
    block = np.array(block, dtype=np.int64, copy=True)
    est = np.asarray(est, bool)
    block[~est] = 0
    leading = np.zeros(block.shape, bool)
    simple = True
    for i in range(block.shape[0]):
        row = block[i]
        vals, first = np.unique(row, return_index=True)
        nz = vals != 0
        if np.count_nonzero(nz) != np.count_nonzero(row):
            simple = False
        leading[i, first[nz]] = True
    lead_f = leading.flatten('F')
    src = np.flatnonzero(lead_f)
    dest = np.arange(len(src))
    obs = np.flatnonzero(np.asarray(use_obs, bool).flatten('F')[src])
    dist = np.full(block.shape, -1, np.int64)
    dist_f = dist.flatten('F')
    dist_f[src] = dest
    dist = dist_f.reshape(block.shape, order='F')
    if not simple:
        for k in range(len(dest)):
            i, j = np.nonzero(dist == k)
            i, j = i[0], j[0]
            in_block = block[i, :] == block[i, j]
            dist[i, in_block] = k
    dist_f = dist.flatten('F')
    ddest = np.flatnonzero(dist_f >= 0)
    dsrc = dist_f[ddest]
    return leading, NS(src=src, dest=dest, obs=obs), NS(dest=ddest, src=dsrc)


def buildserialindices(s):
    """misc/buildserialindices.m:69-159.  x order = [IO ; EO ; OP]."""
    IOlead, IOser, IOdes = _serializeblock(s.IO.struct.block, s.bundle.est.IO,
                                           s.prior.IO.use)
    EOlead, EOser, EOdes = _serializeblock(s.EO.struct.block, s.bundle.est.EO,
                                           s.prior.EO.use)
    nOP = s.OP.val.shape[1]
    _, OPser, OPdes = _serializeblock(np.tile(np.arange(1, nOP + 1), (3, 1)),
                                      s.bundle.est.OP, s.prior.OP.use)
    n = 0
    for ser, des in ((IOser, IOdes), (EOser, EOdes), (OPser, OPdes)):
        ser.dest = ser.dest + n
        des.src = des.src + n
        n += len(ser.dest)
    s.IO.struct.leading = IOlead
    s.EO.struct.leading = EOlead
    s.prior.IO.use = np.asarray(s.prior.IO.use, bool) & IOlead      # :138
    s.prior.EO.use = np.asarray(s.prior.EO.use, bool) & EOlead      # :139
    s.bundle.serial = NS(IO=IOser, EO=EOser, OP=OPser, n=n)
    s.bundle.deserial = NS(IO=IOdes, EO=EOdes, OP=OPdes, n=n)
    nobs = [2 * s.IP.val.shape[1], len(IOser.obs), len(EOser.obs), len(OPser.obs)]
    base = np.concatenate([[0], np.cumsum(nobs)])                   # indvec.m
    s.post = getattr(s, 'post', NS())
    s.post.res = NS(ix=NS(IP=np.arange(base[0], base[1]),
                          IO=np.arange(base[1], base[2]),
                          EO=np.arange(base[2], base[3]),
                          OP=np.arange(base[3], base[4]),
                          n=int(base[4])))
    return s


def serialize(s):
    """misc/serialize.m:14-18."""
    x = np.full(s.bundle.serial.n, np.nan)
    x[s.bundle.serial.IO.dest] = s.IO.val.flatten('F')[s.bundle.serial.IO.src]
    x[s.bundle.serial.EO.dest] = s.EO.val.flatten('F')[s.bundle.serial.EO.src]
    x[s.bundle.serial.OP.dest] = s.OP.val.flatten('F')[s.bundle.serial.OP.src]
    return x


def paramtypes(s):
    """Names of the elements of x: misc/buildparamtypes.m:52-128 scattered by
    misc/serialize.m:20-25."""
    nK, nP = s.IO.model.nK, s.IO.model.nP
    io = ['cc', 'px', 'py', 'as', 'sk'] + ['K%d' % k for k in range(1, nK + 1)] + ['P%d' % k for k in range(1, nP + 1)]
    nc = s.IO.val.shape[1]
    blk = np.asarray(s.IO.struct.block)
    IOt = []
    for j in range(nc):                                      # :58-77
        for i, b in enumerate(io):
            if nc == 1 or np.unique(blk).size == 1:
                IOt.append(b)
            elif np.all(blk == blk[:1]):
                IOt.append('%s-%d' % (b, j + 1))
            else:
                IOt.append('%s-%d(%d)' % (b, j + 1, blk[i, j]))
    ne = s.EO.val.shape[1]
    eid = np.asarray(getattr(s.EO, 'id', np.arange(1, ne + 1)))
    ids = np.any(eid != np.arange(1, ne + 1))                # :85-96
    EOt = []
    for j in range(ne):
        tail = '' if ne == 1 else ('-%d(%d)' % (j + 1, eid[j]) if ids else '-%d' % (j + 1))
        EOt += [b + tail for b in ('EX', 'EY', 'EZ', 'om', 'ph', 'ka')] + [''] * (s.EO.val.shape[0] - 6)
    npnt = s.OP.val.shape[1]
    oid = np.asarray(getattr(s.OP, 'id', np.arange(1, npnt + 1)))
    raw = np.asarray(getattr(s.OP, 'rawId', oid))
    lab = getattr(s.OP, 'label', None)
    ctrl = np.asarray(getattr(s.prior.OP, 'isCtrl', np.zeros(npnt, bool)), bool)
    chk = np.asarray(getattr(s.prior.OP, 'isCheck', np.zeros(npnt, bool)), bool)
    OPt = []
    for j in range(npnt):                                    # :100-128
        pre = 'H' if chk[j] else 'C' if ctrl[j] else 'O'
        tail = ''
        if npnt > 1:
            tail = '-%d' % (j + 1)
            if oid[j] != j + 1:
                tail += '/%d' % oid[j]
            if raw[j] != oid[j]:
                tail += '/%d' % raw[j]
            if lab is not None and lab[j]:
                tail += '-' + lab[j]
        OPt += [pre + c + tail for c in 'XYZ']
    t = np.empty(s.bundle.serial.n, object)
    for ser, names in ((s.bundle.serial.IO, IOt), (s.bundle.serial.EO, EOt), (s.bundle.serial.OP, OPt)):
        t[ser.dest] = np.array(names, object)[ser.src]
    return t


def dmperm_cols(J):
    """Row matched to each column by a maximum matching that takes the columns
    in order (what dmperm(J) returns for a tall J, bundle.m:436); -1 = none.
    Kuhn's algorithm, recursive, on the stored pattern of J."""
    import sys
    J = sp.csc_matrix(J)
    m, n = J.shape
    owner = np.full(m, -1)

    def try_col(j, seen):
        for i in J.indices[J.indptr[j]:J.indptr[j + 1]]:
            if owner[i] < 0:
                owner[i] = j
                return True
        for i in J.indices[J.indptr[j]:J.indptr[j + 1]]:
            if i not in seen:
                seen.add(i)
                if try_col(owner[i], seen):
                    owner[i] = j
                    return True
        return False
    old = sys.getrecursionlimit()
    sys.setrecursionlimit(max(old, 20000))
    try:
        for j in range(n):
            try_col(j, set())
    finally:
        sys.setrecursionlimit(old)
    p = np.full(n, -1)
    p[owner[owner >= 0]] = np.flatnonzero(owner >= 0)
    return p


def weakness(code, final, types):
    """bundle.m:372-446: post-mortem of codes -2 (numerical rank and null
    space of the scaled normal matrix; one dense eigen-decomposition in place
    of spnrank + eigs) and -4 (dmperm)."""
    W = NS(structural=None, numerical=NS(rank=final.weighted.J.shape[1], deficiency=0))
    J = final.weighted.J
    n = J.shape[1]
    if code == -2:
        cn = np.sqrt(np.asarray(J.multiply(J).sum(0)).ravel())
        Js = sp.csc_matrix(J) @ sp.diags(1.0 / np.where(cn == 0, 1.0, cn))
        JTJ = (Js.T @ Js).toarray()
        d, V = np.linalg.eigh(JTJ)
        null = np.abs(d) <= n * np.finfo(float).eps * np.abs(d).max()
        W.numerical = NS(rank=int(n - null.sum()), deficiency=int(null.sum()), suspectedParams=[])
        k = np.flatnonzero(null)
        k = k[np.argsort(np.abs(d[k]), kind='stable')]
        W.numerical.V, W.numerical.d, W.numerical.trace = V[:, k], d[k], np.trace(JTJ)
        for j in range(len(k)):                                  # :411-423
            v = V[:, k[j]]
            o = np.argsort(-np.abs(v), kind='stable')
            o = o[np.abs(v[o]) > np.mean([np.sqrt(1 / n), np.abs(v[o[0]])])]
            W.numerical.suspectedParams.append(NS(values=v[o], indices=o + 1, params=list(types[o])))
    if code == -4:
        p = dmperm_cols(J)
        W.structural = NS(dmperm=p + 1, rank=int((p >= 0).sum()), deficiency=int((p < 0).sum()),
                          suspectedParams=list(types[p < 0]))
        W.numerical = NS(rank=np.nan, deficiency=np.nan)
    return W


def deserialize(s, x):
    """misc/deserialize.m:28-30.  Returns (IO, EO, OP) value arrays."""
    def put(val, des):
        f = val.flatten('F')
        f[des.dest] = x[des.src]
        return f.reshape(val.shape, order='F')
    return (put(s.IO.val, s.bundle.deserial.IO),
            put(s.EO.val, s.bundle.deserial.EO),
            put(s.OP.val, s.bundle.deserial.OP))


# ----------------------------------------------------------------------------
# F3: weights
# ----------------------------------------------------------------------------

def buildweightvector(s):
    """misc/buildweightmatrix.m:13-43.  Returns the diagonal of W (=1/var)."""
    std_mm = s.IP.std * s.IO.sensor.pxSize[:, s.IP.cam]            # :20
    var = np.full(s.post.res.ix.n, np.nan)
    var[s.post.res.ix.IP] = (std_mm ** 2).flatten('F')
    var[s.post.res.ix.IO] = s.prior.IO.std.flatten('F')[
        np.flatnonzero(s.prior.IO.use.flatten('F'))] ** 2
    var[s.post.res.ix.EO] = s.prior.EO.std.flatten('F')[
        np.flatnonzero(s.prior.EO.use.flatten('F'))] ** 2
    var[s.post.res.ix.OP] = s.prior.OP.std.flatten('F')[
        np.flatnonzero(s.prior.OP.use.flatten('F'))] ** 2
    return 1.0 / var


# ----------------------------------------------------------------------------
# F10: prior observations
# ----------------------------------------------------------------------------

def prior_obs(x, s, jac=False):
    """bundle/lsa/prior_obs.m:26-72."""
    out_f, out_J = [], []
    for name in ('IO', 'EO', 'OP'):
        ser = getattr(s.bundle.serial, name)
        pv = getattr(s.prior, name).val.flatten('F')
        cols = ser.dest[ser.obs]
        f = x[cols] - pv[ser.src[ser.obs]]
        out_f.append(f)
        if jac:
            out_J.append(sp.csr_matrix((np.ones(len(f)), (np.arange(len(f)), cols)),
                                       shape=(len(f), len(x))))
    return (out_f, out_J) if jac else out_f


# ----------------------------------------------------------------------------
# F5: multi_res
# ----------------------------------------------------------------------------

def _unpackio(col, nK, nP):
    """bundle/cameramodel/private/unpackio.m:4-8."""
    return col[1:3], col[0], col[5:5 + nK], col[5 + nK:5 + nK + nP], col[3:5]


def _trimkp(K, first_is_pair):
    """bundle/cameramodel/multi_res.m:318-340."""
    nz = np.flatnonzero(K)
    if len(nz) == 0:
        return K[:0]
    i = nz[-1] + 1
    if first_is_pair and i == 1:
        i = 2
    return K[:i]


def _cam_ranges(s):
    """IP columns are image-major (misc/prob2dbatstruct.m:343-365)."""
    cam = s.IP.cam
    nc = s.EO.val.shape[1]
    if np.any(np.diff(cam) < 0):
        raise ValueError('IP columns must be image-major')
    start = np.searchsorted(cam, np.arange(nc), 'left')
    end = np.searchsorted(cam, np.arange(nc), 'right')
    return start, end


def multi_res(s, IO, EO, OP, model, jac=False):
    """bundle/cameramodel/multi_res.m:20-55 (residual) / :56-315 (Jacobian).

    IO/EO/OP are the deserialised value arrays.  Returns r (2*no, image rows
    interleaved x,y, image-major) and, if jac, the sparse J (2*no x n).
    """
    nK, nP = s.IO.model.nK, s.IO.model.nP
    start, end = _cam_ranges(s)
    no = s.IP.val.shape[1]
    xy = np.full((2, no), np.nan)
    if jac:
        n = s.bundle.serial.n
        def destcols(val, des):
            d = np.full(val.size, -1, np.int64)
            d[des.dest] = des.src
            return d.reshape(val.shape, order='F')
        dIO = destcols(s.IO.val, s.bundle.deserial.IO)       # :58-59
        dEO = destcols(s.EO.val, s.bundle.deserial.EO)
        dOP = destcols(s.OP.val, s.bundle.deserial.OP)
        rows, cols, vals = [], [], []
        oprows, opcols, opvals = [], [], []
    for i in range(EO.shape[1]):
        a, e = start[i], end[i]
        if e == a:
            continue
        pp, f, K, P, b = _unpackio(IO[:, i], nK, nP)
        sz = s.IO.sensor.pxSize[0, i]                        # :97,138  sz(1)
        center, ang = EO[:3, i], EO[3:6, i]
        pt = s.IP.pt[a:e]
        obj = OP[:, pt]
        imPts = s.IP.val[:, a:e]
        if not jac:
            K2, P2 = _trimkp(K, False), _trimkp(P, True)     # :36-37
            xy[:, a:e] = res_euler_brown(model, obj, center, ang, f, imPts, sz,
                                         pp, K2, P2, b)
            continue
        cIO = np.asarray(s.bundle.est.IO[:, i], bool)
        cpp, cf, cK, cP, cb = _unpackio(cIO, nK, nP)
        ppIx, fIx, Kix, Pix, bIx = _unpackio(dIO[:, i], nK, nP)
        if not cK.any():
            K = _trimkp(K, False)                            # :105-107
        if not cP.any():
            P = _trimkp(P, True)                             # :108-110
        if cK.any():
            k1, k0 = np.flatnonzero(cK), np.flatnonzero(~cK)
            if len(k0) and k0.min() < k1.max():
                raise ValueError('Illegal cK vector')        # :182-186
        if cP.any():
            p1, p0 = np.flatnonzero(cP), np.flatnonzero(~cP)
            if np.any(p0 <= 1) or (len(p0) and p0.min() < p1.max()):
                raise ValueError('Illegal cP vector')        # :205-209
        v, d = res_euler_brown(model, obj, center, ang, f, imPts, sz, pp, K, P, b,
                               jac=True)
        ni = e - a
        xy[:, a:e] = v
        r0 = 2 * a + 2 * np.arange(ni)                       # block rows (x rows)

        def add(blockJ, colidx):
            # blockJ: (ni,2,k) ; colidx: (k,) global columns (>=0)
            k = blockJ.shape[2]
            rr = (r0[:, None, None] + np.arange(2)[None, :, None]
                  + np.zeros((1, 1, k), np.int64))
            cc = np.broadcast_to(colidx[None, None, :], (ni, 2, k))
            rows.append(rr.ravel()); cols.append(cc.ravel()); vals.append(blockJ.ravel())

        if cpp.any():
            add(d['dU0'][:, :, cpp], ppIx[cpp])              # :148-164
        if cf:
            add(d['dF'], np.array([fIx]))                    # :166-177
        if cK.any():
            add(d['dK'][:, :, :len(cK)][:, :, cK], Kix[cK])  # :179-199
        if cP.any():
            add(d['dP'][:, :, :len(cP)][:, :, cP], Pix[cP])  # :201-222
        if cb.any():
            add(d['dB'][:, :, cb], bIx[cb])                  # :224-241
        cEO = np.asarray(s.bundle.est.EO[:6, i], bool)
        if cEO.any():
            blk = np.concatenate([d['dQ0'], d['dA']], 2)     # :244-273
            add(blk[:, :, cEO], dEO[:6, i][cEO])
        cOP = np.asarray(s.bundle.est.OP[:, pt], bool)       # (3, ni)
        if cOP.any():
            dQ = d['dQ']                                     # (ni,2,3) :275-294
            for c in range(3):
                m = cOP[c]
                if not m.any():
                    continue
                for rr_ in range(2):
                    oprows.append(r0[m] + rr_)
                    opcols.append(dOP[c, pt[m]])
                    opvals.append(dQ[m, rr_, c])
    r = xy.flatten('F')
    if not jac:
        return r
    ii = np.concatenate(rows + oprows) if rows or oprows else np.zeros(0, np.int64)
    jj = np.concatenate(cols + opcols) if cols or opcols else np.zeros(0, np.int64)
    vv = np.concatenate(vals + opvals) if vals or opvals else np.zeros(0)
    J = sp.csc_matrix((vv, (ii, jj)), shape=(2 * no, n))       # :313
    return r, J


# ----------------------------------------------------------------------------
# F4: brown_euler_cam4
# ----------------------------------------------------------------------------

def brown_euler_cam4(x, s, jac=False):
    """bundle/cameramodel/brown_euler_cam4.m:22-33,122-183 (models 2..5)."""
    IO, EO, OP = deserialize(s, x)
    dm = np.unique(s.IO.model.distModel)
    if len(dm) != 1:
        raise ValueError('Mixed lens distortion models not implemented.')  # :31-33
    model = int(dm[0])
    if model == 1:
        # the legacy model 1 is the model the flexible model 2 replicates (bundle.m:49-51:
        # "2 - Flexible Photogrammetry, no affine (replica of 1)"; the committed camcal
        # reports of both agree to every printed digit): evaluated as model 2
        if np.any(np.asarray(s.IO.val)[3:5] != 0) or np.any(np.asarray(s.bundle.est.IO)[3:5]):
            raise ValueError('lens distortion model 1 has no aspect/skew')
        model = 2
    if model not in (2, 3, 4, 5):
        raise ValueError('oracle restates distModel 1..5 only')
    ix = s.post.res.ix
    f = np.full(ix.n, np.nan)
    if not jac:
        f[ix.IP] = multi_res(s, IO, EO, OP, model)
        fpre = prior_obs(x, s)
        f[ix.IO], f[ix.EO], f[ix.OP] = fpre
        return f
    fobs, Jobs = multi_res(s, IO, EO, OP, model, jac=True)
    fpre, Jpre = prior_obs(x, s, jac=True)
    f[ix.IP] = fobs
    f[ix.IO], f[ix.EO], f[ix.OP] = fpre
    J = sp.vstack([Jobs] + Jpre).tocsc()                       # :173-182
    return f, J


# ----------------------------------------------------------------------------
# F11: normal-equation solve (MATLAB `\` on the sparse SPD normal matrix)
# ----------------------------------------------------------------------------

class SingularWarning(Exception):
    pass


def normal_solve(H, g):
    """q = H \\ g for the sparse symmetric normal matrix.

    Restates the MATLAB built-in used at gauss_newton_armijo.m:172,
    levenberg_marquardt.m:119, levenberg_marquardt_powell.m:277 (CHOLMOD /
    UMFPACK inside R2020a `mldivide`; third-party, absent from
    /root/reference).  Dense Cholesky for small systems, SuperLU in symmetric
    mode otherwise.  Returns (q, singular_flag) where singular_flag mimics the
    'MATLAB:singularMatrix'/'nearlySingularMatrix' warnings (rcond < eps).
    """
    n = H.shape[0]
    eps = np.finfo(float).eps
    Hc = sp.csc_matrix(H)
    # Fill-reducing order: CHOLMOD (AMD) eliminates the low-degree object-point
    # columns first; a minimum-degree sort by column count restates that.
    perm = np.argsort(np.diff(Hc.indptr), kind='stable')
    if n <= 3000:
        Hd = Hc.toarray()[np.ix_(perm, perm)]
        try:
            L = np.linalg.cholesky(Hd)
            q = np.empty(n)
            q[perm] = np.linalg.solve(L.T, np.linalg.solve(L, g[perm]))
            d = np.diag(L)
            # cholmod_rcond: (min diag(L) / max diag(L))^2
            sing = bool((d.min() / d.max()) ** 2 < eps)
            return q, sing
        except np.linalg.LinAlgError:
            q, *_ = np.linalg.lstsq(Hc.toarray(), g, rcond=None)
            return q, True
    lu = spla.splu(Hc, permc_spec='MMD_AT_PLUS_A', diag_pivot_thresh=0.0,
                   options=dict(SymmetricMode=True))
    q = lu.solve(g)
    d = np.abs(lu.U.diagonal())          # = diag(L)^2 of the Cholesky factor
    sing = bool(d.min() / d.max() < eps) or not np.all(np.isfinite(q))
    return q, sing


# ----------------------------------------------------------------------------
# F12: damping loops
# ----------------------------------------------------------------------------

def term_relative(convTol):
    """bundle/bundle.m:191."""
    return lambda Jp, r: np.linalg.norm(Jp) <= convTol * np.linalg.norm(r)


def term_absolute(convTol):
    """bundle/bundle.m:188."""
    return lambda Jp, r: np.linalg.norm(r) <= convTol


def _scaled_gn(J, r):
    """gauss_newton_armijo.m:166-174 / levenberg_marquardt_powell.m:267-279."""
    Jn2 = np.asarray(J.multiply(J).sum(0)).ravel()
    Jn = np.sqrt(Jn2)
    D = sp.diags(1.0 / Jn)
    Js = (J @ D).tocsc()
    Hs = (Js.T @ Js).tocsc()
    gs = Js.T @ r
    q, sing = normal_solve(Hs, -gs)
    return D @ q, sing, Jn, Jn2, Hs, gs, Js


def gauss_newton_armijo(resFun, x0, wdiag, maxIter, termFun, sTest=True,
                        mu=0.1, alphaMin=1e-9, trace=False, vetoFun=None):
    """bundle/lsa/gauss_newton_armijo.m:86-245 (+ linesearch :249-290; veto :268-283)."""
    x = x0.copy()
    T = [x0.copy()]
    n = 0
    code = 0
    rr, alphas = [], []
    R = np.sqrt(wdiag)                                   # :104 chol(W), W diagonal
    wres = lambda t: R * resFun(t, False)
    Jp = None
    while True:
        s_, K = resFun(x, True)                          # :112
        r = R * s_
        J = sp.diags(R) @ K
        rr.append(np.sqrt(r @ r))
        if trace:
            print('Gauss-Newton-Armijo: iteration %d, residual norm=%.6g' % (n, rr[-1]))
        if n == 0 and structural_rank(sp.csr_matrix(J)) < J.shape[1]:   # :132-142
            code = -4
            p = np.full(len(x), np.nan)
            break
        p, sing, *_ = _scaled_gn(J, r)                    # :166-174
        if sTest and sing:                                # :176-184
            code = -2
            break
        Jp = J @ p
        if termFun(Jp, r):                                # :191
            break
        n += 1
        # linesearch, :249-290
        f0 = 0.5 * (r @ r)
        fp0 = r @ Jp
        alpha = 1.0
        xNew, rNew = x, r
        found = False
        while alpha >= alphaMin:
            t = x + alpha * p
            rt = wres(t)
            f = 0.5 * (rt @ rt)
            if f < f0 + mu * alpha * fp0 and not (vetoFun is not None and vetoFun(t)):   # :266-276
                xNew, rNew = t, rt
                found = True
                break
            alpha /= 2
        if not found:
            alpha = 0.0
        x, r = xNew, rNew
        alphas.append(alpha)
        T.append(x.copy())
        if alpha == 0:
            code = -3
            rr.append(rr[-1])
            break
        if n > maxIter:                                   # :226
            code = -1
            rr.append(np.sqrt(r @ r))
            break
    final = NS(unweighted=NS(r=s_, J=K), weighted=NS(r=r, J=J), p=p)
    return x, code, n, final, np.array(T).T, np.array(rr), np.array(alphas)


def levenberg_marquardt(resFun, x0, wdiag, maxIter, termFun, lambda0, lambdaMin,
                        trace=False, vetoFun=None):
    """bundle/lsa/levenberg_marquardt.m:52-250."""
    x = x0.copy()
    T = {}
    n = 0
    code = 0
    R = np.sqrt(wdiag)
    wres = lambda t: R * resFun(t, False)
    s_, K = resFun(x, True)
    r = R * s_
    J = (sp.diags(R) @ K).tocsc()
    f = 0.5 * (r @ r)
    JTJ = (J.T @ J).tocsc()
    JTr = J.T @ r
    rr = []
    nx = J.shape[1]
    if lambda0 < 0:
        lambda0 = abs(lambda0) * JTJ.diagonal().sum() / nx     # :88-90
    if lambdaMin < 0:
        lambdaMin = abs(lambdaMin) * JTJ.diagonal().sum() / nx
    lam = lambda0
    if lam < lambdaMin:
        lam = 0.0
    lambdas = [lam]
    prevLambda = np.nan
    I = sp.identity(nx, format='csc')
    Jp = None
    p = None
    while True:
        while n <= maxIter:
            p, _ = normal_solve((JTJ + lam * I).tocsc(), -JTr)     # :119
            rr.append(np.sqrt(r @ r))
            if n == 0 and structural_rank(sp.csr_matrix(J)) < nx:
                code = -4
                p = np.full(len(x), np.nan)
                break
            lambdas.append(lam)
            if trace:
                print('Levenberg-Marquardt: iteration %d, residual norm=%.6g, lambda=%.3g'
                      % (n, rr[-1], lam))
            T[n] = x.copy()
            n += 1
            Jp = J @ p
            t = x + p
            rNew = wres(t)
            fNew = 0.5 * (rNew @ rNew)
            if fNew < f and not (vetoFun is not None and vetoFun(t)):      # :170-177
                x = t
                lam = lam / 10
                if lam < lambdaMin:
                    lam = 0.0
                s_, K = resFun(x, True)
                r = R * s_
                J = (sp.diags(R) @ K).tocsc()
                f = 0.5 * (r @ r)
                JTJ = (J.T @ J).tocsc()
                JTr = J.T @ r
                break
            else:
                if lam == 0:
                    lam = lambdaMin
                else:
                    lam = lam * 10
        if code != 0:
            break
        if prevLambda == 0 and termFun(Jp, r):                   # :217
            break
        prevLambda = lam
        if n > maxIter:
            code = -1
            break
    T[n] = x.copy()
    rr.append(np.sqrt(r @ r))
    Tm = np.full((len(x), n + 1), np.nan)
    for k, v in T.items():
        if k <= n:
            Tm[:, k] = v
    final = NS(unweighted=NS(r=s_, J=K), weighted=NS(r=r, J=J), p=p)
    return x, code, n, final, Tm, np.array(rr), np.array(lambdas)


def dogleg(r, J, delta):
    """bundle/lsa/levenberg_marquardt_powell.m:232-335."""
    pGN, _sing, Jn, Jn2, Hs, gs, _ = _scaled_gn(J, r)
    if np.linalg.norm(pGN) <= delta:
        return pGN, pGN, 0
    invD2gs = Jn2 * gs
    g = Jn * gs
    lambdaStar = (g @ g) / (invD2gs @ (Hs @ invD2gs))
    CP = -lambdaStar * g
    if np.linalg.norm(CP) > delta:
        return -g / np.linalg.norm(g) * delta, pGN, 2
    A = np.sum((CP - pGN) ** 2)
    B = np.sum(2 * CP * (pGN - CP))
    C = np.sum(CP ** 2) - delta ** 2
    k = (-B + np.sqrt(B ** 2 - 4 * A * C)) / (2 * A)
    return CP + k * (pGN - CP), pGN, 1


def levenberg_marquardt_powell(resFun, x0, wdiag, maxIter, termFun, delta0, mu,
                               eta, trace=False, vetoFun=None):
    """bundle/lsa/levenberg_marquardt_powell.m:60-230."""
    x = x0.copy()
    T = {0: x0.copy()}
    n = 0
    code = 0
    delta = delta0
    deltas, rhos, steps, rr = [], [], [], []
    R = np.sqrt(wdiag)
    wres = lambda t: R * resFun(t, False)
    s_, K = resFun(x, True)
    r = R * s_
    J = (sp.diags(R) @ K).tocsc()
    f = 0.5 * (r @ r)
    p = None
    while True:
        rr.append(np.sqrt(r @ r))
        if n == 0 and structural_rank(sp.csr_matrix(J)) < J.shape[1]:
            code = -4
            p = np.full(len(x), np.nan)
            break
        p, pGN, step = dogleg(r, J, delta)
        deltas.append(delta)
        steps.append(step)
        Jp = J @ p
        if step == 0 and termFun(J @ pGN, r):                    # :134-140
            break
        t = x + p
        rt = wres(t)
        ft = 0.5 * (rt @ rt)
        veto = vetoFun is not None and bool(vetoFun(t))          # :146-150
        predicted = -(r @ Jp) - 0.5 * (Jp @ Jp)
        actual = f - ft
        rho = actual / predicted
        rhos.append(rho)
        if trace:
            print('Levenberg-Marquardt-Powell: iteration %d, residual norm=%.6g, '
                  'delta=%.3g, step=%d, rho=%.2f' % (n, rr[-1], delta, step, rho))
        if veto or rho <= mu:                                    # :166
            delta = delta / 2
            npgn = np.linalg.norm(pGN)
            if delta > npgn:
                delta = delta / 2.0 ** np.ceil(np.log2(delta / npgn))   # :176-179
        else:
            x = t
            s_, K = resFun(x, True)
            r = R * s_
            J = (sp.diags(R) @ K).tocsc()
            f = 0.5 * (r @ r)
            if rho >= eta:
                delta = delta * 2
        T[n] = x.copy()
        n += 1
        if n > maxIter:
            code = -1
            break
    T[n] = x.copy()
    Tm = np.full((len(x), max(n, 1)), np.nan)                    # :229 trims to 1:n
    for k, v in T.items():
        if k < Tm.shape[1]:
            Tm[:, k] = v
    final = NS(unweighted=NS(r=s_, J=K), weighted=NS(r=r, J=J), p=p)
    return (x, code, n, final, Tm, np.array(rr), np.array(deltas), np.array(rhos),
            np.array(steps))


def gauss_markov(resFun, x0, wdiag, maxIter, termFun, sTest=True, trace=False):
    """bundle/lsa/gauss_markov.m:52-129 with the documented semantics
    (undamped Gauss-Newton; SURVEY Appendix B item 1: the reference passes a
    function handle where gauss_markov expects convTol, so the termination
    test is restated through termFun(Jp,r))."""
    x = x0.copy()
    T = [x0.copy()]
    n = 0
    code = 0
    rr = []
    R = np.sqrt(wdiag)
    while True:
        s_, K = resFun(x, True)
        r = R * s_
        J = (sp.diags(R) @ K).tocsc()
        rr.append(np.sqrt(r @ r))
        JTJ = (J.T @ J).tocsc()
        p, sing = normal_solve(JTJ, -(J.T @ r))                  # :79
        if sTest and sing:
            code = -2
            break
        Jp = J @ p
        if termFun(Jp, r):                                       # :94
            break
        x = x + p
        n += 1
        T.append(x.copy())
        if n > maxIter:
            code = -1
            break
    final = NS(unweighted=NS(r=s_, J=K), weighted=NS(r=r, J=J), p=p)
    return x, code, n, final, np.array(T).T, np.array(rr)


# ----------------------------------------------------------------------------
# F1: bundle driver
# ----------------------------------------------------------------------------

def bundle(s, *args, termFun=None, vetoFun=None):
    """bundle/bundle.m:1-132 (args), :156-192 (setup), :267-358 (dispatch),
    :449-491 (residual scatter, sigma0).

    termFun / vetoFun: the two function handles bundle.m:168-192 builds and hands to the lsa solvers, replaceable here
    so that the solvers' callback interface can be checked (the reference's own vetoFun, @chirality, is undefined).

    Returns (s, ok, iters, s0, E) like the reference.  `s` is a deep-ish copy
    with IO/EO/OP.val updated only when code==0 (bundle.m:356-358).
    """
    import copy
    s = copy.deepcopy(s)
    maxIter, damping, singularTest, doTrace = 20, 'gna', True, False
    absTerm, convTol, pmDof = False, 1e-6, False
    for a in args:                                               # :88-132
        if isinstance(a, bool):
            if a:
                raise ValueError('chirality veto is undefined in the reference')
        elif isinstance(a, (int, float, np.integer, np.floating)):
            if float(a) == round(float(a)):
                maxIter = int(a)
            else:
                convTol = float(a)
        elif isinstance(a, str):
            la = a.lower()
            if la in ('none', 'gm', 'gna', 'lm', 'lmp'):
                damping = la
            elif la == 'trace':
                doTrace = True
            elif la == 'singulartest':
                singularTest = True
            elif la == 'nosingulartest':
                singularTest = False
            elif la == 'pmdof':
                pmDof = True
            elif la == 'dofverb':
                pass
            elif la == 'absterm':
                absTerm = True
            else:
                raise ValueError('DBAT:bundle:badInput Unknown damping')
        else:
            raise ValueError('DBAT:bundle:badInput Unknown parameter')
    for nm in ('IO', 'EO', 'OP'):                                # :137-154
        pr = getattr(s.prior, nm)
        est = np.asarray(getattr(s.bundle.est, nm), bool)
        pr.use = np.asarray(pr.use, bool) & est
    s = buildserialindices(s)                                    # :156-159
    x0 = serialize(s)                                            # :162
    resFun = lambda x, jac: brown_euler_cam4(x, s, jac)          # :165
    wdiag = buildweightvector(s)                                 # :175
    if termFun is None:
        termFun = term_absolute(convTol) if absTerm else term_relative(convTol)
    E = NS(maxIter=maxIter, convTol=convTol, absTerm=absTerm,
           singularTest=singularTest)
    if damping in ('none', 'gm'):
        x, code, iters, final, X, res = gauss_markov(resFun, x0, wdiag, maxIter,
                                                     termFun, singularTest, doTrace)
        E.damping = NS(name='gm')
    elif damping == 'gna':
        x, code, iters, final, X, res, alpha = gauss_newton_armijo(
            resFun, x0, wdiag, maxIter, termFun, singularTest, 0.1, 1e-9, doTrace, vetoFun)
        E.damping = NS(name='gna', alpha=alpha, mu=0.1, alphaMin=1e-9)
    elif damping == 'lm':
        x, code, iters, final, X, res, lam = levenberg_marquardt(
            resFun, x0, wdiag, maxIter, termFun, -1e-10, -1e-10, doTrace, vetoFun)
        E.damping = NS(name='lm', **{'lambda': lam}, lambda0=lam[0], lambdaMin=lam[0])
    elif damping == 'lmp':
        delta0 = np.linalg.norm(x0)                              # :325
        x, code, iters, final, X, res, delta, rho, step = levenberg_marquardt_powell(
            resFun, x0, wdiag, maxIter, termFun, delta0, 0.25, 0.75, doTrace, vetoFun)
        E.damping = NS(name='lmp', delta=delta, rho=rho, delta0=delta0,
                       rhoBad=0.25, rhoGood=0.75, step=step)
    E.res, E.trace, E.code, E.usedIters, E.final = res, X, code, iters, final
    ok = code == 0
    if ok:
        s.IO.val, s.EO.val, s.OP.val = deserialize(s, x)         # :356-358
    # residual scatter :449-460
    ix = s.post.res.ix
    ur = final.unweighted.r
    s.post.res.IP = (ur[ix.IP].reshape(2, -1, order='F')
                     / s.IO.sensor.pxSize[:, s.IP.cam])
    for nm in ('IO', 'EO', 'OP'):
        pr = getattr(s.prior, nm)
        arr = np.full(getattr(s, nm).val.size, np.nan)
        arr[np.flatnonzero(pr.use.flatten('F'))] = ur[getattr(ix, nm)]
        setattr(s.post.res, nm, arr.reshape(getattr(s, nm).val.shape, order='F'))
    r = final.weighted.r
    p_extra = 0
    if pmDof:                                                    # :467-471
        seen_pt = np.zeros(s.OP.val.shape[1], bool); seen_pt[s.IP.pt] = True
        seen_cam = np.zeros(s.EO.val.shape[1], bool); seen_cam[s.IP.cam] = True
        p_extra = (np.count_nonzero(~np.asarray(s.bundle.est.OP, bool)[:, seen_pt])
                   + np.count_nonzero(~np.asarray(s.bundle.est.EO, bool)[:6, seen_cam]))
    dof = len(r) + p_extra - len(x)
    s0 = np.sqrt((r @ r) / dof)                                  # :483
    s.post.sigmas = s0 * np.asarray(s.IP.sigmas)
    # sensor format updated by the estimated aspect (bundle.m:360-366)
    aspect = np.ones((2, s.IO.val.shape[1])); aspect[0] = 1.0 + s.IO.val[3]
    if hasattr(s.IO.sensor, 'imSize'):               # structs built without image sizes have no format to update
        s.post.sensor = type(s.post)(imSize=np.array(s.IO.sensor.imSize, float), pxSize=s.IO.sensor.pxSize * aspect,
                                      ssSize=s.IO.sensor.imSize * s.IO.sensor.pxSize * aspect)
    E.numObs, E.numParams, E.redundancy, E.s0 = len(r), len(x), dof, s0
    E.paramTypes = paramtypes(s)                                 # :162,368
    E.weakness = weakness(code, final, E.paramTypes)             # :372-446
    E.sigmas = s.post.sigmas
    E.x = x
    return s, ok, iters, s0, E


# ----------------------------------------------------------------------------
# Posterior covariance: bundle/bundle_cov.m
# ----------------------------------------------------------------------------

def bundle_cov(s, E, *names):
    """bundle/bundle_cov.m:1-214.  Returns, per requested name, s0^2 times a
    block of inv(J'J) with J = E.final.weighted.J (bundle_cov.m:66,211):
      'cxx'            the whole matrix in x order (:137-143)
      'cio','ceo','cop' block-diagonal (one block per IO/EO/OP column: 'any
                       correlations with other cameras/images/points are
                       ignored', :9-16, BlockDiagonalC :219-311, VectorizedCOP),
                       sized numel(val) x numel(val), zero-padded for elements
                       that were not estimated
      'ciof','ceof','copf' the full component matrices (:145-191).
    The reference obtains the blocks from a Cholesky factor of the permuted
    normal matrix (invblock.m "sqrt": B = W'W, W = inv(L) E); any exact method gives
    the same numbers, here a dense inverse (test sizes only)."""
    import scipy.sparse as sp
    J = E.final.weighted.J
    N = (J.T @ J).toarray()
    n = N.shape[0]
    try:                                                   # bundle_cov.m:92-112 (chol fails -> NaN)
        np.linalg.cholesky(N)
        Ninv = np.linalg.inv(N)
        Ninv = 0.5 * (Ninv + Ninv.T)
    except np.linalg.LinAlgError:
        Ninv = np.full((n, n), np.nan)
    out = []
    for name in names:
        name = name.lower()
        if name == 'cxx':
            out.append(E.s0 ** 2 * Ninv)
            continue
        comp = name[1:3].upper()
        val = getattr(s, comp).val
        des = getattr(s.bundle.deserial, comp)             # dest: element of val(:), src: index into x
        C = np.zeros((val.size, val.size))
        C[np.ix_(des.dest, des.dest)] = Ninv[np.ix_(des.src, des.src)]
        if not name.endswith('f'):                         # keep the diagonal blocks only (mkblkdiag)
            m = val.shape[0]
            mask = np.kron(np.eye(val.shape[1]), np.ones((m, m)))
            C = C * mask
        out.append(sp.csc_matrix(E.s0 ** 2 * C))
    return out[0] if len(out) == 1 else tuple(out)


# ----------------------------------------------------------------------------
# Test method: central-difference Jacobian (misc/jacapprox.m:35-61)
# ----------------------------------------------------------------------------

def jacapprox(fun, x, h=1e-6):
    """misc/jacapprox.m:35-61: central differences with step h."""
    x = np.asarray(x, float)
    f0 = np.asarray(fun(x)).ravel()
    J = np.zeros((len(f0), x.size))
    xf = x.ravel()
    for i in range(x.size):
        xp = xf.copy(); xp[i] += h
        xm = xf.copy(); xm[i] -= h
        J[:, i] = (np.asarray(fun(xp.reshape(x.shape))).ravel()
                   - np.asarray(fun(xm.reshape(x.shape))).ravel()) / (2 * h)
    return J


# ----------------------------------------------------------------------------
# Struct construction helper (field names follow misc/prob2dbatstruct.m:12-186)
# ----------------------------------------------------------------------------

def make_struct(IO, EO, OP, ip_val, ip_cam, ip_pt, pxSize, *, ip_std=None,
                distModel=3, nK=3, nP=2, estIO=None, estEO=None, estOP=None,
                IOblock=None, EOblock=None, priorIO=None, priorEO=None, priorOP=None):
    """Build a minimal DBAT struct.  prior* = (use, val, std) tuples."""
    IO = np.array(IO, float); EO = np.array(EO, float); OP = np.array(OP, float)
    nc, npnt, no = EO.shape[1], OP.shape[1], np.asarray(ip_val).shape[1]
    px = np.asarray(pxSize, float)
    if px.ndim == 0:
        px = np.full((2, nc), float(px))
    elif px.ndim == 1:
        px = np.tile(px[None, :], (2, 1))
    std = np.ones((2, no)) if ip_std is None else np.array(ip_std, float)
    if std.ndim == 0:
        std = np.full((2, no), float(std))

    def prior(p, val):
        if p is None:
            return NS(use=np.zeros(val.shape, bool), val=np.full(val.shape, np.nan),
                      std=np.full(val.shape, np.nan))
        return NS(use=np.array(p[0], bool), val=np.array(p[1], float),
                  std=np.array(p[2], float))
    s = NS()
    s.IO = NS(val=IO, model=NS(distModel=np.full(nc, distModel), nK=nK, nP=nP),
              sensor=NS(pxSize=px),
              struct=NS(block=(np.ones(IO.shape, np.int64) if IOblock is None
                               else np.array(IOblock, np.int64))))
    s.EO = NS(val=EO, struct=NS(block=(np.tile(np.arange(1, nc + 1), (EO.shape[0], 1))
                                       if EOblock is None else np.array(EOblock, np.int64))))
    s.OP = NS(val=OP)
    s.IP = NS(val=np.array(ip_val, float), std=std, cam=np.asarray(ip_cam, np.int64),
              pt=np.asarray(ip_pt, np.int64), sigmas=np.unique(std))
    s.bundle = NS(est=NS(IO=(np.zeros(IO.shape, bool) if estIO is None else np.array(estIO, bool)),
                         EO=(np.ones(EO.shape, bool) if estEO is None else np.array(estEO, bool)),
                         OP=(np.ones(OP.shape, bool) if estOP is None else np.array(estOP, bool))),
                  serial=None, deserial=None)
    s.prior = NS(IO=prior(priorIO, IO), EO=prior(priorEO, EO), OP=prior(priorOP, OP))
    s.post = NS()
    return s
