"""Post-mortem of a failed bundle: which parameters make the design matrix rank
deficient (bundle.m:368-446, printed under 'Problems and suggestions' by
bundle_result_file.m:58-90).

Host-side and only run after a failure: code -4 (structural rank, found at
iteration 0) is explained by a maximum matching of the Jacobian's block
pattern, code -2 (singular normal matrix) by the null space of the scaled
normal matrix, whose blocks are fetched once from the device
(dbat_hip_jacobian_blocks).

    buildparamtypes(s)            misc/buildparamtypes.m:46-136  parameter names
    param_types(s, maps, n)       misc/serialize.m:20-25        names in x order
    structural_weakness(...)      bundle.m:434-446              dmperm
    numerical_weakness(...)       bundle.m:374-428              rank + null space
"""
from types import SimpleNamespace as NS

import numpy as np
import scipy.sparse as sp


class _Namer:
    """Names of the IO / EO / OP entries (misc/buildparamtypes.m:46-136), one at a time:
    'cc', 'px', .. 'K1', 'P1' (with '-<col>' for image-variant cameras and
    '-<col>(<block>)' for mixed ones); 'EX-<n>' .. 'ka-<n>'; 'OX-<n>[/<id>]',
    control points 'CX..', check points 'HX..'."""

    def __init__(self, s):
        nK, nP = int(s.IO.model.nK), int(s.IO.model.nP)
        self.base = ['cc', 'px', 'py', 'as', 'sk'] + ['K%d' % (k + 1) for k in range(nK)] + ['P%d' % (k + 1) for k in range(nP)]
        self.nc = s.IO.val.shape[1]
        self.blk = np.asarray(s.IO.struct.block)
        self.one_block = len(np.unique(self.blk)) == 1
        # isSimple: every parameter of the column belongs to the column's own block
        self.simple = bool(np.all(self.blk == self.blk[0:1]))
        self.ne = s.EO.val.shape[1]
        self.eid = np.asarray(getattr(s.EO, 'id', np.arange(1, self.ne + 1)))
        self.use_ids = bool(np.any(self.eid != np.arange(1, self.ne + 1)))
        self.npnt = s.OP.val.shape[1]
        self.oid = np.asarray(getattr(s.OP, 'id', np.arange(1, self.npnt + 1)))
        self.raw = np.asarray(getattr(s.OP, 'rawId', self.oid))
        self.label = getattr(s.OP, 'label', None)
        self.ctrl = np.asarray(getattr(s.prior.OP, 'isCtrl', np.zeros(self.npnt, bool)), bool)
        self.chk = np.asarray(getattr(s.prior.OP, 'isCheck', np.zeros(self.npnt, bool)), bool)
        self.shapes = ((len(self.base), self.nc), (6, self.ne), (3, self.npnt))

    def io(self, i, j):
        b = self.base[i]
        if self.nc == 1 or self.one_block:
            return b
        if self.simple:
            return '%s-%d' % (b, j + 1)
        return '%s-%d(%d)' % (b, j + 1, self.blk[i, j])

    def eo(self, i, j):
        tail = '' if self.ne == 1 else ('-%d(%d)' % (j + 1, self.eid[j]) if self.use_ids else '-%d' % (j + 1))
        return ('EX', 'EY', 'EZ', 'om', 'ph', 'ka')[i] + tail

    def op(self, i, j):
        pre = 'H' if self.chk[j] else ('C' if self.ctrl[j] else 'O')
        tail = ''
        if self.npnt > 1:
            tail = '-%d' % (j + 1)
            if self.oid[j] != j + 1:
                tail += '/%d' % self.oid[j]
            if self.raw[j] != self.oid[j]:
                tail += '/%d' % self.raw[j]
            if self.label is not None and self.label[j]:
                tail += '-' + self.label[j]
        return pre + 'XYZ'[i] + tail

    def name(self, section, i, j):
        return (self.io, self.eo, self.op)[section](i, j)


def buildparamtypes(s):
    """Names of every IO / EO / OP entry (misc/buildparamtypes.m:46-136) as three object arrays."""
    nm = _Namer(s)
    out = []
    for sec, (r, c) in enumerate(nm.shapes):
        A = np.empty((r, c), object)
        for j in range(c):
            for i in range(r):
                A[i, j] = nm.name(sec, i, j)
        out.append(A)
    return tuple(out)


class ParamTypes:
    """Name of every element of x (serialize.m:20-25; E.paramTypes of bundle.m:162,368) as a read-only sequence that
    builds a name when it is asked for -- a project with three million unknowns has three million names, and the result
    file and the post-mortem read a few dozen of them (building them all took a second per bundle() at C3, 50 times the
    solve).  maps = Handle.index_maps(): position in x of every IO / EO / OP entry, -1 when fixed; the leading entry
    of a block names it.  Indexing with an integer gives a str, with a slice / index array / boolean mask an object array;
    list(E.paramTypes) gives them all."""

    def __init__(self, s, maps, n):
        self._namer = _Namer(s)
        self._n = int(n)
        self._maps = maps
        self._sec = self._ent = None

    def _index(self):
        if self._sec is None:                            # x position -> (section, column-major entry): the first (leading) entry wins
            sec = np.full(self._n, -1, np.int8)
            ent = np.zeros(self._n, np.int64)
            for k, ix in enumerate(self._maps):
                flat = ix.flatten('F')
                e = np.flatnonzero(flat >= 0)[::-1]
                sec[flat[e]] = k
                ent[flat[e]] = e
            self._sec, self._ent = sec, ent
        return self._sec, self._ent

    def _one(self, k):
        sec, ent = self._index()
        if sec[k] < 0:
            return None
        rows = self._maps[sec[k]].shape[0]
        return self._namer.name(int(sec[k]), int(ent[k] % rows), int(ent[k] // rows))

    def __len__(self):
        return self._n

    @property
    def shape(self):
        return (self._n,)

    def __getitem__(self, key):
        if isinstance(key, (int, np.integer)):
            k = int(key)
            if k < 0:
                k += self._n
            if not 0 <= k < self._n:
                raise IndexError(key)
            return self._one(k)
        idx = np.arange(self._n)[key]
        out = np.empty(idx.shape, object)
        for q, k in enumerate(idx.ravel()):
            out.ravel()[q] = self._one(int(k))
        return out

    def __iter__(self):
        return (self._one(k) for k in range(self._n))

    def __array__(self, dtype=None, copy=None):
        return self[:]

    def __eq__(self, other):
        try:
            return len(other) == self._n and all(a == b for a, b in zip(self, other))
        except TypeError:
            return NotImplemented

    def __repr__(self):
        return 'ParamTypes(%d names%s)' % (self._n, '' if not self._n else ': %s ... %s' % (self[0], self[self._n - 1]))


def param_types(s, maps, n):
    """Name of every element of x (serialize.m:20-25), built on demand: ParamTypes."""
    return ParamTypes(s, maps, n)


def _prior_rows(s, maps):
    """x positions of the prior-observation rows, in residual order IO, EO, OP
    (buildserialindices.m:138-139,200): column-major over use & leading."""
    cols = []
    for nm, ix in zip(('IO', 'EO', 'OP'), maps):
        use = np.asarray(getattr(s.prior, nm).use, bool)[:ix.shape[0]]
        flat = ix.flatten('F')
        lead = np.zeros(flat.shape, bool)
        valid = np.flatnonzero(flat >= 0)
        _, first = np.unique(flat[valid], return_index=True)
        lead[valid[first]] = True
        cols.append(flat[np.flatnonzero(use.flatten('F') & lead)])
    return np.concatenate(cols).astype(np.int64)


def jacobian_pattern(s, maps, n):
    """Block pattern of J as a CSC boolean matrix: two rows per image point
    with the estimated IO, EO and OP entries it depends on, one row per prior
    observation (brown_euler_cam4.m:103-119, prior_obs.m:50-72)."""
    IOix, EOix, OPix = maps
    no = s.IP.val.shape[1]
    rows, cols = [], []
    r0 = 2 * np.arange(no)
    for ix in (IOix[:, s.IP.cam], EOix[:, s.IP.cam], OPix[:, s.IP.pt]):
        for c in range(ix.shape[0]):
            m = ix[c] >= 0
            for rr in range(2):
                rows.append(r0[m] + rr); cols.append(ix[c][m])
    pc = _prior_rows(s, maps)
    rows.append(2 * no + np.arange(len(pc))); cols.append(pc)
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    return sp.csc_matrix((np.ones(len(rows), bool), (rows, cols)), shape=(2 * no + len(pc), n))


def maxtrans(A):
    """Maximum matching of the columns of the sparse pattern A to its rows,
    columns taken in natural order with augmenting paths, so that a column stays
    matched once it is (the set of matched columns is then the
    lexicographically first one, whatever the search order -- which is what
    MATLAB's dmperm returns for a tall matrix).  Returns p with p[j] = row
    matched to column j, -1 if none (dmperm's 0)."""
    A = sp.csc_matrix(A)
    m, n = A.shape
    indptr, indices = A.indptr, A.indices
    row_match = np.full(m, -1, np.int64)
    col_match = np.full(n, -1, np.int64)
    cheap = indptr[:-1].copy()          # rows before cheap[j] in column j are all matched
    mark = np.full(n, -1, np.int64)
    for k in range(n):
        stack, it, take = [k], {k: indptr[k]}, {}
        mark[k] = k
        found = False
        while stack and not found:
            j = stack[-1]
            while cheap[j] < indptr[j + 1]:          # cheap assignment: a free row of column j
                i = indices[cheap[j]]; cheap[j] += 1
                if row_match[i] < 0:
                    take[j] = i
                    found = True
                    break
            if found:
                break
            deeper = False
            while it[j] < indptr[j + 1]:             # else try to re-match the owner of one of its rows
                i = indices[it[j]]; it[j] += 1
                j2 = row_match[i]
                if mark[j2] != k:
                    mark[j2] = k
                    take[j] = i
                    stack.append(j2); it[j2] = indptr[j2]
                    deeper = True
                    break
            if not deeper:
                stack.pop()
        if found:
            for j in stack:
                row_match[take[j]] = j
                col_match[j] = take[j]
    return col_match


def structural_weakness(s, maps, n, types):
    """bundle.m:434-446: dmperm of the Jacobian; the unmatched columns are the
    suspected parameters."""
    p = maxtrans(jacobian_pattern(s, maps, n))
    rank = int(np.count_nonzero(p >= 0))
    return NS(dmperm=p + 1, rank=rank, deficiency=n - rank,
              suspectedParams=list(types[p < 0]))


def weighted_jacobian(h, s, x):
    """Weighted Jacobian at x as a scipy CSC matrix, assembled on the host from
    the per-observation blocks the device computes (multi_res.m:300-313),
    prior rows and weights as buildweightmatrix.m:13-43."""
    maps = h.index_maps()
    IOix, EOix, OPix = maps
    JEO, JOP, JIO = h.jacobian_blocks(x)
    no = s.IP.val.shape[1]
    rows, cols, vals = [], [], []
    r0 = 2 * np.arange(no)
    for blk, ix in ((JEO, EOix[:, s.IP.cam]), (JOP, OPix[:, s.IP.pt]), (JIO, IOix[:, s.IP.cam])):
        for c in range(blk.shape[2]):
            m = ix[c] >= 0
            for rr in range(2):
                rows.append(r0[m] + rr); cols.append(ix[c][m]); vals.append(blk[m, rr, c])
    pc = _prior_rows(s, maps)
    rows.append(2 * no + np.arange(len(pc))); cols.append(pc); vals.append(np.ones(len(pc)))
    J = sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                      shape=(2 * no + len(pc), h.n))
    std = [(np.asarray(s.IP.std, float) * s.IO.sensor.pxSize[:, s.IP.cam]).flatten('F')]
    for nm, ix in zip(('IO', 'EO', 'OP'), maps):
        pr = getattr(s.prior, nm)
        use = np.asarray(pr.use, bool)[:ix.shape[0]]
        flat = ix.flatten('F')
        lead = np.zeros(flat.shape, bool)
        valid = np.flatnonzero(flat >= 0)
        _, first = np.unique(flat[valid], return_index=True)
        lead[valid[first]] = True
        std.append(np.asarray(pr.std, float)[:ix.shape[0]].flatten('F')[np.flatnonzero(use.flatten('F') & lead)])
    return sp.diags(1.0 / np.concatenate(std)) @ J


def numerical_weakness(J, types, dense_limit=4000):
    """bundle.m:374-428 on the weighted Jacobian J: numerical rank of the
    column-scaled J and, when deficient, the null-space vectors of the scaled
    normal matrix with the parameters that dominate each of them.  The
    reference estimates the rank with the third-party spnrank and the vectors
    with eigs; here one symmetric eigen-decomposition serves both (rank = number
    of eigenvalues above n*eps*largest).  Beyond dense_limit unknowns the rank
    is reported as not estimated (NaN), as the reference does when spnrank
    fails (bundle.m:383-386)."""
    n = J.shape[1]
    out = NS(rank=float('nan'), deficiency=float('nan'), suspectedParams=[])
    if n > dense_limit or not np.all(np.isfinite(J.data)):
        return out
    cn = np.sqrt(np.asarray(J.multiply(J).sum(0)).ravel())
    cn[cn == 0] = 1.0
    Js = J @ sp.diags(1.0 / cn)
    JTJ = (Js.T @ Js).toarray()
    d, V = np.linalg.eigh(JTJ)
    null = np.abs(d) <= n * np.finfo(float).eps * np.abs(d).max()
    out.rank = int(n - np.count_nonzero(null))
    out.deficiency = int(np.count_nonzero(null))
    if out.deficiency:
        k = np.flatnonzero(null)
        k = k[np.argsort(np.abs(d[k]), kind='stable')]
        out.V, out.d, out.trace = V[:, k], d[k], float(np.trace(JTJ))
        avg = np.sqrt(1.0 / n)
        for j in range(len(k)):
            v = out.V[:, j]
            order = np.argsort(-np.abs(v), kind='stable')
            keep = np.abs(v[order]) > 0.5 * (avg + np.abs(v[order[0]]))
            order = order[keep]
            out.suspectedParams.append(NS(values=v[order], indices=order + 1, params=list(types[order])))
    return out
