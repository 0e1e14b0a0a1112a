"""DBAT's result file (bundle/bundle_result_file.m) from a bundle result: the
lines a reader diffs against the reference's committed reports -- rank
diagnosis and problems, status, sigma0, counts, options, iteration summary,
every camera parameter with its posterior standard deviation, significance
and high correlations, every exterior orientation parameter likewise, point
coverage of the images, ray counts, image residual statistics, point
precision, intersection angles, control and check point tables.  Same
labels, nesting, units and printf formats as the reference (file:line cited
per block).  Bookkeeping lines (project names, dates, host, timings) are
not produced.

    lines = bundle_result_lines(s, E, CIO, CEO, COP)      # list of str
    bundle_result_file(s, E, CIO, CEO, COP, 'report.txt')

CIO, CEO, COP are the block-diagonal posterior covariances of bundle_cov()
(CIO may be the full 'CIOF' matrix: then correlations between cameras are
listed too, as the reference does).  Without them -- a failed bundle has
none -- the report stops after the iteration summary.
"""
from __future__ import annotations

import os

import numpy as np

_P = '   '          # one indentation level (bundle_result_file.m:34-40)
CORR_THRESHOLD = 0.95     # bundle_result_file.m:20
SIG_THRESHOLD = 0.95      # :22


def _g(x):
    """MATLAB's %g: NaN and Inf are spelled with capitals."""
    x = float(x)
    return 'NaN' if np.isnan(x) else ('-Inf' if x < 0 else 'Inf') if np.isinf(x) else '%g' % x


def _pretty(prefix, values, min_len=np.inf, max_len=-np.inf):
    """pretty_print (bundle_result_file.m:940-954): pad the names to a common width."""
    name_len = np.array([len(v[0]) for v in values])
    width = max(min(min_len, name_len.max()), max_len) + 1
    return ['%s%s%s%s' % (prefix, name, ' ' * max(int(width - len(name)), 0), text) for name, text in values]


def _dense_block(C, ix):
    ix = np.asarray(ix)
    sub = C[np.ix_(ix, ix)]
    return np.asarray(sub.todense()) if hasattr(sub, 'todense') else np.asarray(sub)


def _corr_pairs(C, thres):
    """(i, j, value) with i > j of the correlations above thres in magnitude
    (tril(corrmat(C)), private/high_*_correlations.m)."""
    import scipy.sparse as sp
    C = sp.csc_matrix(C)
    d = np.sqrt(np.maximum(C.diagonal(), 0))
    T = sp.tril(C, -1).tocoo()
    ok = (d[T.row] > 0) & (d[T.col] > 0)
    r = np.zeros(len(T.data))
    r[ok] = T.data[ok] / (d[T.row[ok]] * d[T.col[ok]])
    keep = np.abs(r) > thres
    order = np.lexsort((T.row[keep], T.col[keep]))            # find(): column-major
    return T.row[keep][order], T.col[keep][order], r[keep][order]


def _io_uniq(s):
    """Columns of IO.struct.block that start a camera, their numbers and whether
    all parameters of the column share one block (parseblockvariant.m:20-28)."""
    blk = np.asarray(s.IO.struct.block)
    _, first, inv = np.unique(blk.T, axis=0, return_index=True, return_inverse=True)
    uniq = np.zeros(blk.shape[1], bool)
    uniq[first] = True
    order = np.argsort(first)
    no = np.empty(len(first), int)
    no[order] = np.arange(1, len(first) + 1)
    return uniq, no[np.ravel(inv)], np.all(blk == blk[:1], axis=0)


def _significance(s, CIO):
    """Chi-square p-values of the distortion parameters against zero
    (private/test_distortion_params.m): K individually (pk) and cumulatively
    (pkc), P1 and P2 together (pp), aspect and skew (pb)."""
    from scipy.stats import chi2
    x = np.asarray(s.IO.val, float)
    R, nc = x.shape
    nK = int(s.IO.model.nK)
    est = np.asarray(s.bundle.est.IO, bool)
    pk, pkc = np.full((nK, nc), np.nan), np.full((nK, nc), np.nan)
    pp, pb = np.full((2, nc), np.nan), np.full((2, nc), np.nan)
    uniq = _io_uniq(s)[0]
    for j in np.flatnonzero(uniq):
        for i in range(nK):
            if est[5 + i, j]:
                pk[i, j] = chi2.cdf(x[5 + i, j] ** 2 / _dense_block(CIO, [j * R + 5 + i])[0, 0], 1)
            if est[5:6 + i, j].all():
                ix = j * R + 5 + np.arange(i + 1)
                v = x[5:6 + i, j]
                pkc[i, j] = chi2.cdf(float(v @ np.linalg.solve(_dense_block(CIO, ix), v)), i + 1)
        ii = 5 + nK + np.arange(2)
        if R >= 7 + nK and est[ii, j].all():
            v = x[ii, j]
            # test_distortion_params.m:49 stores the joint P1/P2 value as P(j,:) -- row j of a
            # 2-by-nCams array, i.e. for the first camera in the P1 row of every camera and
            # nothing in the P2 row.  The reports are written that way; so is this.
            if j < 2:
                pp[j, :] = chi2.cdf(float(v @ np.linalg.solve(_dense_block(CIO, j * R + ii), v)), 2)
        for i in range(2):
            if est[3 + i, j]:
                pb[i, j] = chi2.cdf(x[3 + i, j] ** 2 / _dense_block(CIO, [j * R + 3 + i])[0, 0], 1)
    return pk, pp, pb, pkc


def _problem_lines(E, n_processing=None, flags=()):
    """'Problems and suggestions' (bundle_result_file.m:142-175,185-213): the rank
    diagnosis of E.weakness, the number of processing problems and their lines."""
    p, p2, p3, p4, p5, p6 = (_P * k for k in range(1, 7))
    out = [p + 'Problems and suggestions:', p2 + 'Project Problems:']
    w = getattr(E, 'weakness', None)
    st = getattr(w, 'structural', None)
    if st is None:
        out.append(p3 + 'Structural rank: ok.')
    else:
        out.append(p3 + 'Structural rank: %d (deficiency: %d)' % (st.rank, st.deficiency))
        out.append(p4 + 'DMPERM suggests the following parameters have problems:')
        out += [p5 + str(name) for name in st.suspectedParams]
    nu = getattr(w, 'numerical', None)
    if nu is None or nu.deficiency == 0:
        out.append(p3 + 'Numerical rank: ok.')
    elif np.isnan(nu.rank):
        out.append(p3 + 'Numerical rank: not tested.')
    else:
        out.append(p3 + 'Numerical rank: %d (deficiency: %d)' % (nu.rank, nu.deficiency))
        out.append(p4 + 'Null-space suggest the following parameters are part of the problem:')
        for i, sp in enumerate(nu.suspectedParams):
            out.append(p5 + 'Vector %d (eigenvalue %g):' % (i + 1, nu.d[i]))
            out += [p6 + '(%s, %.3g)' % (name, v) for name, v in zip(sp.params, sp.values)]
    if n_processing is not None:
        out.append(p2 + 'Problems related to the processing: (%d)' % n_processing)
    if int(E.code) != 0:
        out.append(p3 + 'Bundle failed with code %d (see below for details).' % int(E.code))
    msgs = ('One or more of the camera parameter has a high correlation (see below).',
            'One or more of the camera station parameters has a high correlation (see below).',
            'One or more of the object point coordinates has a high correlation.',
            'One or more estimated lens and/or affine distortion coefficients failed significance test (see below).')
    out += [p3 + m for m, f in zip(msgs, flags) if f]
    return out


def _vis(s):
    """Visibility (points x images) and the column of each image point
    (IP.vis, IP.ix of prob2dbatstruct.m)."""
    npnt, nc = s.OP.val.shape[1], s.EO.val.shape[1]
    ix = np.full((npnt, nc), -1, np.int64)
    ix[s.IP.pt, s.IP.cam] = np.arange(len(s.IP.pt))
    return ix >= 0, ix


def _hull_area(pts):
    from .initial import _hull
    if pts.shape[1] < 3:
        return 0.0
    h = _hull(pts)
    c = pts[:, h].mean(1)
    ang = np.arctan2(pts[1, h] - c[1], pts[0, h] - c[0])
    q = pts[:, np.asarray(h)[np.argsort(ang)]]
    return 0.5 * abs(float(np.sum(q[0] * np.roll(q[1], -1) - np.roll(q[0], -1) * q[1])))


def _coverage(s, cams, union):
    """Convex-hull, rectangular and radial coverage of the images by their
    measured points (misc/coverage.m), per image or for all of them together."""
    def one(pts, i):
        tot = float(np.prod(s.IO.sensor.imSize[:, i]))
        if pts.shape[1] == 0:
            return 0.0, 0.0, 0.0
        px = s.IO.sensor.pxSize[:, i]
        pp = s.IO.val[1:3, i]
        to_mm = lambda q: np.stack([q[0] * px[0] - pp[0], -q[1] * px[1] - pp[1]])   # PP\(S\[q;1])
        im = s.IO.sensor.imSize[:, i]
        xx, yy = np.array([0.5, im[0] + 0.5]), np.array([0.5, im[1] + 0.5])
        corners = np.stack([xx[[0, 0, 1, 1]], yy[[0, 1, 1, 0]]])
        max_rad = np.sqrt(np.sum(to_mm(corners) ** 2, 0)).max()
        crr = np.sqrt(np.sum(to_mm(pts) ** 2, 0)).max() / max_rad
        cr = float(np.prod(pts.max(1) - pts.min(1))) / tot
        return _hull_area(pts) / tot, cr, crr
    if union:
        sel = np.isin(s.IP.cam, cams)
        return one(s.IP.val[:, sel], cams[0])
    return tuple(np.array(v) for v in zip(*[one(s.IP.val[:, s.IP.cam == i], i) for i in cams]))


def _angles(s, vis):
    """Largest angle between two rays of each object point (misc/angles.m)."""
    a = np.full(s.OP.val.shape[1], np.nan)
    for i in range(len(a)):
        cc = s.EO.val[:3, vis[i]]
        if cc.shape[1] == 1:
            a[i] = 0.0
        elif cc.shape[1] > 1:
            d = s.OP.val[:, i:i + 1] - cc
            dn = d / np.sqrt(np.sum(d * d, 0))
            a[i] = np.arccos(np.abs(np.clip(dn.T @ dn, -1, 1))).max()
    return a


def _quality_lines(s, E, COP, vop):
    p, p2, p3, p4, p5, p6 = (_P * k for k in range(1, 7))
    out = [p + 'Quality', p2 + 'Photographs']
    ne = s.EO.val.shape[1]
    out += _pretty(p3, [('Total number:', '%d' % ne), ('Numbers used:', '%d' % ne)])
    uniq, no, simple = _io_uniq(s)
    est_io = np.asarray(s.bundle.est.IO, bool)
    out.append(p2 + 'Cameras')
    out.append(p3 + 'Total number: %d (%d simple, %d mixed)' % (uniq.sum(), (uniq & simple).sum(), (uniq & ~simple).sum()))
    have_im = hasattr(s.IO.sensor, 'imSize')
    for i in np.flatnonzero(uniq):
        out.append(p3 + 'Camera%d:' % no[i])
        cams = np.flatnonzero(no == no[i])
        out += _pretty(p4, [('Calibration:', 'yes' if est_io[:, i].any() else '<not available>'),
                            ('Number of photos using camera:', '%d' % len(cams))])
        if have_im:
            c, cr, crr = _coverage(s, cams, False)
            uc, ucr, ucrr = _coverage(s, cams, True)
            rnd = lambda v: int(np.floor(v * 100 + 0.5))
            fmt = lambda a, u: '%d%%-%d%% (%d%% average, %d%% union)' % (rnd(a.min()), rnd(a.max()), rnd(a.mean()), rnd(u))
            out.append(p4 + 'Photo point coverage:')
            out += _pretty(p5, [('Rectangular:', fmt(cr, ucr)), ('Convex hull:', fmt(c, uc)), ('Radial:', fmt(crr, ucrr))])
    out.append(p2 + 'Photo Coverage')
    out.append(p3 + 'Reference points outside calibrated region:')
    for i in np.flatnonzero(uniq):
        out.append(p4 + 'Camera %d: %s' % (no[i], 'none' if est_io[:, i].any() else '<not available>'))
    # ---- Point measurements (:568-628)
    npnt = s.OP.val.shape[1]
    ctrl = np.asarray(getattr(s.prior.OP, 'isCtrl', np.zeros(npnt, bool)), bool)
    chk = np.asarray(getattr(s.prior.OP, 'isCheck', np.zeros(npnt, bool)), bool)
    vis, ipix = _vis(s)
    rays = vis.sum(1)
    out.append(p2 + 'Point Measurements')
    out.append(p3 + 'Number of control pts: %d' % ctrl.sum())
    out.append(p3 + 'Number of check pts: %d' % chk.sum())
    out.append(p3 + 'Number of object pts: %d' % (~ctrl & ~chk).sum())

    def hist(r):
        return [p4 + '%d points with %d rays.' % (np.count_nonzero(r == k), k) for k in np.unique(r)]
    if ctrl.any():
        r = rays[ctrl]
        n0, rr = np.count_nonzero(r == 0), r[r != 0]
        head = 'CP ray count: %dx0, ' % n0 if n0 else 'CP ray count: '
        out.append(p3 + head + '%d-%d (%.1f avg)' % (rr.min(), rr.max(), rr.mean()))
        out += hist(r)
    else:
        out.append(p3 + 'CP ray count: -')
    if chk.any():
        r = rays[chk]
        out.append(p3 + 'CCP ray count: %d-%d (%.1f avg)' % (r.min(), r.max(), r.mean()))
        out += hist(r)
    else:
        out.append(p3 + 'CCP ray count: -')
    if (~ctrl).any():
        r = rays[~ctrl]
        out.append(p3 + 'OP ray count: %d-%d (%.1f avg)' % (r.min(), r.max(), r.mean()))
        out += hist(r)
    else:
        out.append(p3 + 'OP ray count: -')
    # ---- Point marking residuals (:630-672, bundle_residuals.m)
    ids = np.asarray(s.OP.id)
    pt_res = np.sqrt(np.sum(np.asarray(s.post.res.IP, float) ** 2, 0))
    res = np.zeros(vis.shape)
    res[s.IP.pt, s.IP.cam] = pt_res
    out.append(p2 + 'Point Marking Residuals')
    out.append(p3 + 'Overall point RMS: %.3f pixels' % np.sqrt(np.mean(pt_res ** 2)))
    out.append(p3 + 'Mark point residuals:')
    k = int(np.argmax(res.flatten('F')))
    out.append(p4 + 'Maximum: %.3f pixels (OP %d on photo %d)' % (res.flatten('F')[k], ids[k % npnt], k // npnt + 1))
    with np.errstate(divide='ignore', invalid='ignore'):
        mean_op = np.sqrt((res ** 2).sum(1) / rays)
        n_photo = vis.sum(0)
        mean_photo = np.sqrt((res ** 2).sum(0) / n_photo)
    out.append(p3 + 'Object point residuals (RMS over all images of a point):')
    a, b = int(np.nanargmin(mean_op)), int(np.nanargmax(mean_op))
    out.append(p4 + 'Minimum: %.3f pixels (OP %d over %d images)' % (mean_op[a], ids[a], rays[a]))
    out.append(p4 + 'Maximum: %.3f pixels (OP %d over %d images)' % (mean_op[b], ids[b], rays[b]))
    out.append(p3 + 'Photo residuals (RMS over all points in an image):')
    a, b = int(np.nanargmin(mean_photo)), int(np.nanargmax(mean_photo))
    out.append(p4 + 'Minimum: %.3f pixels (photo %d over %d points)' % (mean_photo[a], a + 1, n_photo[a]))
    out.append(p4 + 'Maximum: %.3f pixels (photo %d over %d points)' % (mean_photo[b], b + 1, n_photo[b]))
    # ---- Point precision (:674-700)
    var = COP.diagonal().reshape(s.OP.val.shape, order='F').astype(float)
    sd_all = np.sqrt(np.maximum(var, 0))
    var[~np.asarray(s.bundle.est.OP, bool)] = np.nan
    tstd = np.sqrt(var.sum(0))
    out.append(p2 + 'Point Precision')
    out.append(p3 + 'Total standard deviation (RMS of X/Y/Z std):')
    g2 = lambda v: 'NaN' if np.isnan(v) else '%.2g' % v

    def arg(v, f):                                   # MATLAB's min/max skip NaN; all NaN -> NaN, index 1
        return 0 if np.all(np.isnan(v)) else int(f(v))
    a, b = arg(tstd, np.nanargmin), arg(tstd, np.nanargmax)
    out.append(p4 + 'Minimum: %s (OP %d)' % (g2(tstd[a]), ids[a]))
    out.append(p4 + 'Maximum: %s (OP %d)' % (g2(tstd[b]), ids[b]))
    sd = np.sqrt(var)
    for c, ax in enumerate('XYZ'):
        j = arg(sd[c], np.nanargmax)
        out.append(p3 + 'Maximum %s standard deviation: %s (OP %d)' % (ax, g2(sd[c, j]), ids[j]))
    # ---- Points with high correlations (:702-724)
    vi, vj, vv = vop
    out.append(p3 + 'Points with high correlations')
    out.append(p4 + 'Points with correlation above 95%%: %d' % np.count_nonzero(np.abs(vv) > 0.95))
    out.append(p4 + 'Points with correlation above 99%%: %d' % np.count_nonzero(np.abs(vv) > 0.99))
    if np.count_nonzero(np.abs(vv) > 0.95):
        out.append(p4 + 'Points with highest correlations:')
        printed = []
        for q in np.argsort(-np.abs(vv), kind='stable'):
            pt = int(vi[q] // 3) + 1
            if pt not in printed:
                printed.append(pt)
                out.append(p5 + 'Points %d: %.2f' % (pt, 100 * vv[q]))
            if len(printed) >= 5:
                break
    # ---- Point angles (:726-817)
    ang = np.rad2deg(_angles(s, vis))
    label = getattr(s.OP, 'label', None) or [''] * npnt
    out.append(p2 + 'Point Angles')

    def block(name, tag, mask, with_label, mean=np.mean):
        o = [p3 + name]
        if mask.any() and np.any(rays[mask] > 0):
            a_, id_ = ang[mask], ids[mask]
            lab = [label[i] for i in np.flatnonzero(mask)]
            mn, mx = int(np.nanargmin(a_)), int(np.nanargmax(a_))
            lm = lambda i: (', label %s' % lab[i]) if (with_label and lab[i]) else ''
            o.append(p4 + 'Minimum: %.1f degrees (%s %d%s)' % (a_[mn], tag, id_[mn], lm(mn)))
            o.append(p4 + 'Maximum: %.1f degrees (%s %d%s)' % (a_[mx], tag, id_[mx], lm(mx)))
            o.append(p4 + 'Average: %.1f degrees' % mean(a_))
        else:
            o += [p4 + 'Minimum: -', p4 + 'Maximum: -', p4 + 'Average: -']
        return o
    if ctrl.any() and np.any(rays[ctrl] == 0):
        out.append(p3 + 'CP')
        out.append(p4 + 'Ignoring %d CP with 0 rays.' % np.count_nonzero(rays[ctrl] == 0))
        out += block('CP', 'CP', ctrl, True, np.nanmean)[1:]
    else:
        out += block('CP', 'CP', ctrl, True, np.nanmean)
    out += block('CCP', 'CCP', chk, True)
    is_op = ~ctrl & ~chk
    out += block('OP', 'OP', is_op, False)
    if is_op.any():
        out.append(p4 + 'Smallest angles (ID, angle [deg], vis in cameras)')
        a_, id_ = ang[is_op], ids[is_op]
        order = np.argsort(a_, kind='stable')
        srt = a_[order]
        limit = min(srt[min(3, len(srt)) - 1] * 1.1 + 0.1, 80.0)
        n = min(max(np.count_nonzero(srt < limit), 3), len(srt))
        for j in range(n):
            # the reference indexes IP.vis with the position among the plain object points
            # (bundle_result_file.m:809: s.IP.vis(i(j),:), i from sort(aOP)), which is the point
            # itself only when no control or check point precedes it; kept, so that the listing
            # reads as the reference's does
            cams = ' '.join('%4d' % (c + 1) for c in np.flatnonzero(vis[order[j]]))
            out.append(p5 + '%6d: %5.2f (%s)' % (id_[order[j]], srt[j], cams))
    # ---- Control and check measurements (:819-931)
    for title, mask, delta, sep in (('Ctrl measurements', ctrl, 'Ctrl point delta', False),
                                    ('Check measurements', chk, 'Check point delta', True)):
        out.append(p2 + title)
        if not mask.any():
            out.append(p3 + 'none')
            continue
        cix = np.flatnonzero(mask)
        pos0, std0 = s.prior.OP.val[:, cix], s.prior.OP.std[:, cix]
        pos1, std1 = s.OP.val[:, cix], sd_all[:, cix]
        g3 = lambda v: '%8s' % ('%.3g' % v)
        out.append(p3 + 'Prior')
        out.append(p3 + '%6s, %8s, %8s, %8s, %8s, %8s, %8s, %s' % ('id', 'x', 'y', 'z', 'stdx', 'stdy', 'stdz', 'label'))
        for k, i in enumerate(cix):
            out.append(p3 + '%6d, %8.3f, %8.3f, %8.3f, %s, %s, %s, %s'
                       % (ids[i], *pos0[:, k], g3(std0[0, k]), g3(std0[1, k]), g3(std0[2, k]), label[i]))
        out.append(p3 + 'Posterior')
        out.append(p3 + '%6s, %8s, %8s, %8s, %8s, %8s, %8s, %4s, %s' % ('id', 'x', 'y', 'z', 'stdx', 'stdy', 'stdz', 'rays', 'label'))
        for k, i in enumerate(cix):
            out.append(p3 + '%6d, %8.3f, %8.3f, %8.3f, %s, %s, %s, %4d, %s'
                       % (ids[i], *pos1[:, k], g3(std1[0, k]), g3(std1[1, k]), g3(std1[2, k]), rays[i], label[i]))
        out.append(p3 + 'Diff (pos=abs diff, std=rel diff)')
        out.append(p3 + '%6s, %8s, %8s, %8s, %8s, %8s, %8s, %8s, %8s, %4s, %s'
                   % ('id', 'x', 'y', 'z', 'xy', 'xyz', 'stdx', 'stdy', 'stdz', 'rays', 'label'))
        posd = pos1 - pos0
        eps = np.finfo(float).eps
        stdd = ((std1 + eps) / (std0 + eps) - 1) * 100
        for k, i in enumerate(cix):
            out.append(p3 + '%6d, %8.3f, %8.3f, %8.3f, %8.3f, %8.3f, %7.1f%%, %7.1f%%, %7.1f%%, %4d, %s'
                       % (ids[i], *posd[:, k], np.linalg.norm(posd[:2, k]), np.linalg.norm(posd[:, k]),
                          *stdd[:, k], rays[i], label[i]))
        out.append(p3 + delta)
        dn = np.sqrt(np.sum(posd ** 2, 0))
        k = int(np.argmax(dn))
        tagged = lambda i: ('%s, ' % label[i]) if (sep or label[i]) else ''
        out.append(p4 + 'Max: %.3f ou (%spt %d)' % (dn[k], tagged(cix[k]), ids[cix[k]]))
        out.append(p4 + 'Max X,Y,Z')
        for c, ax in enumerate('XYZ'):
            j = int(np.argmax(np.abs(posd[c])))
            out.append(p5 + '%s: %.3f ou (%spt %d)' % (ax, abs(posd[c, j]), tagged(cix[j]), ids[cix[j]]))
        out.append(p4 + 'RMS: %.3f ou (from %d items)' % (np.sqrt(np.mean(dn ** 2)), len(dn)))
    out.append('End of result file')
    return out


def bundle_result_lines(s, E, CIO=None, CEO=None, COP=None):
    p, p2, p3, p4, p5, p6 = (_P * k for k in range(1, 7))
    have_cov = CIO is not None and CEO is not None and COP is not None
    est_io = np.asarray(s.bundle.est.IO, bool)
    R, nc = s.IO.val.shape
    nK, nP = int(s.IO.model.nK), int(s.IO.model.nP)
    uniq, cam_no, simple = _io_uniq(s)
    if have_cov:
        # high correlations (:101-115) and significance tests (:158)
        blk = np.asarray(s.IO.struct.block)
        lead = np.zeros((R, nc), bool)                            # IO.struct.leading: first column of a block, per row
        for r in range(R):
            lead[r, np.unique(blk[r], return_index=True)[1]] = True
        lead = lead.flatten('F')
        ii, jj, vio = _corr_pairs(CIO, CORR_THRESHOLD)
        keep = lead[ii] & lead[jj]
        ii, jj, vio = ii[keep], jj[keep], vio[keep]
        m = s.EO.val.shape[0]
        ei, ej, veo = _corr_pairs(CEO, CORR_THRESHOLD)
        same = ei // m == ej // m
        ei, ej, veo = ei[same], ej[same], veo[same]
        vop = _corr_pairs(COP, CORR_THRESHOLD)
        pk, pp, pb, pkc = _significance(s, CIO)
        with np.errstate(invalid='ignore'):
            low_sig = bool(np.any(np.vstack([pk, pp, pb]) < SIG_THRESHOLD))
        flags = (len(vio) > 0, len(veo) > 0, len(vop[2]) > 0, low_sig)
        out = _problem_lines(E, sum(flags) + (int(E.code) != 0), flags)
    else:
        out = _problem_lines(E, int(int(E.code) != 0))
    nIO = int(np.count_nonzero(s.IO.struct.leading)) if hasattr(s.IO.struct, 'leading') else int(E.numParams - np.count_nonzero(np.asarray(s.bundle.est.EO, bool)[:6]) - np.count_nonzero(s.bundle.est.OP))
    nEO = int(np.count_nonzero(np.asarray(s.bundle.est.EO, bool)[:6]))
    nOP = int(np.count_nonzero(s.bundle.est.OP))
    n_prior = [int(np.count_nonzero(getattr(s.prior, nm).use)) for nm in ('IO', 'EO', 'OP')]
    nIP = 2 * s.IP.val.shape[1]
    # ---- Information from last bundle (bundle_result_file.m:179-237)
    codes = {0: 'OK', -1: 'fail (code -1: Too many iterations)', -2: 'fail (code -2: Normal matrix is singular)',
             -3: 'fail (code -3: No step length found by the line search)',
             -4: 'fail (code -4: Normal matrix is structurally rank deficient)'}           # :203-213
    out.append(p + 'Information from last bundle')
    out += _pretty(p2, [
        ('Status:', codes.get(int(E.code), 'fail (code %d: unknown code)' % int(E.code))),
        ('Sigma0:', _g(E.s0)),
        ('Sigma0 (pixels):', _g(np.ravel(s.post.sigmas)[0])),
        ('Redundancy', '%d' % E.redundancy),
        ('Number of params:', '%d (%d IO, %d EO, %d OP)' % (E.numParams, nIO, nEO, nOP)),
        ('Number of observations:', '%d (%d IP, %d IO, %d EO, %d OP)' % (E.numObs, nIP, *n_prior))])
    # ---- Processing options (:236-256)
    cu = getattr(s.IO.model, 'camUnit', 'mm')
    proj = getattr(s, 'proj', None)
    onoff = lambda b: 'on' if b else 'off'
    out.append(p2 + 'Processing options:')
    out += _pretty(p3, [
        ('Orientation:', 'on'), ('Global optimization:', 'on'), ('Calibration:', onoff(est_io.any())),
        ('Constraints:', 'off'), ('Maximum # of iterations:', '%d' % E.maxIter),
        ('Convergence tolerance:', '%g' % E.convTol),
        ('Termination criteria:', 'absolute' if E.absTerm else 'relative'),
        ('Singular test:', onoff(E.singularTest)), ('Chirality veto:', onoff(getattr(E, 'chirality', False))),
        ('Damping:', E.damping.name), ('Camera unit (cu):', cu),
        ('Object space unit (ou):', getattr(proj, 'objUnit', 'm')),
        ('Initial value comment:', getattr(proj, 'x0desc', ''))])
    # ---- Total error (:257-266)
    out.append(p2 + 'Total error:')
    out += _pretty(p3, [('Number of stages:', '1'), ('Number of iterations:', '%d' % E.usedIters),
                        ('First error:', _g(E.res[0])), ('Last error:', _g(E.res[-1]))])
    # ---- Lens distortion models (:278-288)
    out.append(p2 + 'Lens distortion models:')
    dm = np.unique(np.ravel(s.IO.model.distModel))
    out.append(p3 + ('Backward (Photogrammetry) model %d' % dm[0] if len(dm) == 1 and dm[0] > 0 else
                     'Forward (Computer Vision) model %d' % -dm[0] if len(dm) == 1 else 'Mixed Forward/Backward'))
    if not have_cov:
        return out
    # ---- Cameras (:290-463): value, deviation, significance, correlations of every camera parameter
    out.append(p2 + 'Cameras:')
    names = ['cc', 'px', 'py', 'as', 'sk'] + ['K%d' % (k + 1) for k in range(nK)] + ['P%d' % (k + 1) for k in range(nP)]
    self_cal = est_io.any(0)
    if self_cal.all():
        all_cal, any_cal = est_io.all(1), est_io.any(1)
        cal = ('yes (%s)' % ' '.join(n for n, a in zip(names, all_cal) if a)) if np.array_equal(all_cal, any_cal) \
            else 'yes (mixed parameters)'
    else:
        cal = 'no' if not self_cal.any() else 'mixed'
    out.append(p3 + 'Calibration: ' + cal)
    head = (['Camera Constant', 'px - principal point x', 'py - principal point y', 'Format width', 'Format height']
            + ['K%d - radial distortion %d' % (k + 1, k + 1) for k in range(nK)]
            + ['P%d - decentering distortion %d' % (k + 1, k + 1) for k in range(nP)]
            + ['as - off-unit aspect parameter', 'sk - skew', 'Image width', 'Image height',
               'X resolution', 'Y resolution', 'Pixel width', 'Pixel height'])
    unit = ([cu] * 5 + ['%s^(-%d)' % (cu, 2 * k + 3) for k in range(nK)] + ['%s^(-3)' % cu] * nP
            + ['', '', 'px', 'px', 'px/%s' % cu, 'px/%s' % cu, cu, cu])
    rows = [0, 1, 2, -1, -2] + [5 + k for k in range(nK + nP)] + [3, 4, -3, -4, -5, -6, -7, -8]
    io = np.array(s.IO.val, float)
    io[[2] + list(range(5, io.shape[0]))] *= -1.0                 # display signs (:357-358)
    sd_io = np.sqrt(np.maximum(CIO.diagonal(), 0)).reshape(io.shape, order='F')
    ppad = np.full((nP, nc), np.nan)
    ppad[:min(nP, 2)] = pp[:min(nP, 2)]
    sig = np.vstack([np.full((3, nc), np.nan), pb, pk, ppad])
    cum = np.vstack([np.full((5, nc), np.nan), pkc, np.full((nP, nc), np.nan)])
    ps = s.post.sensor if hasattr(s.post, 'sensor') else s.IO.sensor       # bundle.m:360-366
    sensor = np.vstack([ps.ssSize, ps.imSize, ps.imSize / ps.ssSize, ps.pxSize])
    corr_str = 'Correlations over %g%%:' % (CORR_THRESHOLD * 100)
    # symmetric list [param, cam, other param, other cam, value] (:366-368)
    cc = [(a % R, a // R, b % R, b // R, v) for a, b, v in zip(ii, jj, vio)]
    cc += [(c, d, a, b, v) for a, b, c, d, v in cc]
    for i in np.flatnonzero(uniq):
        out.append(p3 + 'Camera%d (%s)' % (cam_no[i], 'simple' if simple[i] else 'mixed'))
        out.append(p4 + 'Lens distortion model:')
        dmi = int(np.ravel(s.IO.model.distModel)[i])
        out.append(p5 + ('Backward (Photogrammetry) model %d' % dmi if dmi > 0 else 'Forward (Computer Vision) model %d' % -dmi))
        pad = len('Significance:') if self_cal[i] else len('Value:')
        for h, u, r in zip(head, unit, rows):
            val, sigma = (io[r, i], sd_io[r, i]) if r >= 0 else (sensor[-r - 1, i], np.nan)
            vals = [('Value:', ('%g %s' % (val, u)))]
            if not np.isnan(sigma) and sigma != 0:
                vals.append(('Deviation:', '%.3g %s' % (sigma, u)))
            if r >= 0 and not np.isnan(sig[r, i]):
                vals.append(('Significance:', 'p=%.2f' % sig[r, i]))
            if r >= 0 and not np.isnan(cum[r, i]):
                vals.append(('Cumulative significance:', 'p=%.2f' % cum[r, i]))
            if r >= 0 and self_cal[i]:
                hits = [(c, d, v) for a, b, c, d, v in cc if b == i and a == r]
                if hits:
                    txt = ','.join(' %s:%.1f%%' % (names[c], v * 100) if d == i else
                                   ' %s(cam%d):%.1f%%' % (names[c], d + 1, v * 100) for c, d, v in hits)
                    vals.append((corr_str, txt + '.'))
            out.append(p4 + h + ':')
            out += _pretty(p5, vals, pad, pad)
        if hasattr(s.IO.sensor, 'ssSize'):                           # :432-462
            whd = np.append(s.IO.sensor.ssSize[:, i], np.linalg.norm(s.IO.sensor.ssSize[:, i]))
            aov = np.rad2deg(2 * np.arctan(whd / (2 * s.IO.val[0, i])))
            out.append(p3 + 'Rated angle of view (h,v,d): (%.0f, %.0f, %.0f) deg' % tuple(aov))
            im, px, v = s.IO.sensor.imSize[:, i], s.IO.sensor.pxSize[:, i], s.IO.val[:, i]
            xx, yy = np.array([0.5, im[0] + 0.5]), np.array([0.5, im[1] + 0.5])
            xr = xx[[0, 0, 1, 1]] * px[0] - v[1]
            yr = yy[[0, 1, 1, 0]] * px[1] + v[2]
            r2 = xr ** 2 + yr ** 2
            K = list(v[5:5 + nK]) + [0.0] * 3
            P = list(v[5 + nK:5 + nK + nP]) + [0.0] * 2
            rad = K[0] * r2 + K[1] * r2 ** 2 + K[2] * r2 ** 3
            xc = xr * rad + P[0] * (r2 + 2 * xr ** 2) + 2 * P[0] * xr * yr
            yc = yr * rad + P[1] * (r2 + 2 * yr ** 2) + 2 * P[1] * xr * yr
            mx = np.max(np.abs(xc) + np.abs(yc))
            out.append(p3 + 'Largest distortion: %.2g %s (%.1f px, %.1f%% of half-diagonal)'
                       % (mx, cu, mx / px[0], mx / (whd[2] / 2) * 100))
    # ---- Photograph standard deviations (:465-513)
    out.append(p2 + 'Precisions / Standard Deviations:')
    out.append(p3 + 'Photograph Standard Deviations:')
    sd_eo = np.sqrt(np.maximum(CEO.diagonal(), 0)).reshape(s.EO.val.shape, order='F')
    order = [3, 4, 5, 0, 1, 2]
    scale = [180 / np.pi] * 3 + [1.0] * 3
    enames = ['Omega', 'Phi', 'Kappa', 'Xc', 'Yc', 'Zc']
    pos_of = {r: k for k, r in enumerate(order)}
    units = ['deg'] * 3 + ['ou'] * 3
    pad = len('Deviation:')
    eo_names = getattr(s.EO, 'name', None)
    ce = [(a % m, b % m, a // m, v) for a, b, v in zip(ei, ej, veo)]
    ce += [(b, a, k, v) for a, b, k, v in ce]
    for i in range(s.EO.val.shape[1]):
        out.append(p4 + 'Photo %d: %s' % (i + 1, os.path.basename(str(eo_names[i])) if eo_names is not None else ''))
        for nme, u, r, sc in zip(enames, units, order, scale):
            vals = [('Value:', '%.6f %s' % (sc * s.EO.val[r, i], u))]
            if sd_eo[r, i] != 0:
                vals.append(('Deviation:', '%.3g %s' % (sc * sd_eo[r, i], u)))
            hits = [(b, v) for a, b, k, v in ce if k == i and a == r and b in pos_of]
            if hits:
                vals.append((corr_str, ','.join(' %s:%.1f%%' % (enames[pos_of[b]], v * 100) for b, v in hits) + '.'))
            out.append(p5 + nme + ':')
            out += _pretty(p6, vals, pad, pad)
    return out + _quality_lines(s, E, COP, vop)


def bundle_result_file(s, E, CIO=None, CEO=None, COP=None, path='report.txt'):
    lines = ['Damped Bundle Adjustment Toolbox result file']
    lines += bundle_result_lines(s, E, CIO, CEO, COP)
    with open(path, 'w') as fh:
        fh.write('\n'.join(lines) + '\n')
    return lines
