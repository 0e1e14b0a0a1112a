"""Numeric subset of DBAT's result file (bundle/bundle_result_file.m): the lines
a reader diffs against the reference's committed reports -- status, sigma0,
parameter/observation counts, iteration summary, every camera parameter and
every exterior orientation parameter with its posterior standard deviation,
and the point-precision summary.  Same labels, nesting, units and printf
formats as the reference (file:line cited per block); bookkeeping lines (dates,
host, timings), significance tests, correlation listings, coverage and ray
statistics are not produced.

    lines = bundle_result_lines(s, E, CIO, CEO, COP)      # list of str
    bundle_result_file(s, E, CIO, CEO, COP, 'report.txt')

CIO, CEO, COP are the block-diagonal posterior covariances of bundle_cov().
"""
from __future__ import annotations

import os

import numpy as np

_P = '   '          # one indentation level (bundle_result_file.m:34-40)


def _g(x):
    """MATLAB's %g: NaN and Inf are spelled with capitals."""
    x = float(x)
    return 'NaN' if np.isnan(x) else ('-Inf' if x < 0 else 'Inf') if np.isinf(x) else '%g' % x


def _pretty(prefix, values, min_len=np.inf, max_len=-np.inf):
    """pretty_print (bundle_result_file.m:940-954): pad the names to a common width."""
    name_len = np.array([len(v[0]) for v in values])
    width = max(min(min_len, name_len.max()), max_len) + 1
    return ['%s%s%s%s' % (prefix, name, ' ' * max(int(width - len(name)), 0), text) for name, text in values]


def _problem_lines(E):
    """'Problems and suggestions' (bundle_result_file.m:142-175,196-199): the rank
    diagnosis of E.weakness and the failure line.  The count of processing
    problems also covers correlation and significance tests, which are not
    produced, so that one line is left out."""
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
    if int(E.code) != 0:
        out.append(p3 + 'Bundle failed with code %d (see below for details).' % int(E.code))
    return out


def bundle_result_lines(s, E, CIO=None, CEO=None, COP=None):
    """Report lines; without covariances (a failed bundle has none) the report
    stops after the iteration summary."""
    p, p2, p3, p4, p5, p6 = (_P * k for k in range(1, 7))
    out = _problem_lines(E)
    nIO = int(np.count_nonzero(s.IO.struct.leading)) if hasattr(s.IO.struct, 'leading') else int(E.numParams - np.count_nonzero(s.bundle.est.EO) - np.count_nonzero(s.bundle.est.OP))
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
    # ---- Total error (:257-266)
    out.append(p2 + 'Total error:')
    out += _pretty(p3, [('Number of stages:', '1'), ('Number of iterations:', '%d' % E.usedIters),
                        ('First error:', _g(E.res[0])), ('Last error:', _g(E.res[-1]))])
    if CIO is None or CEO is None or COP is None:
        return out
    # ---- Cameras (:293-440): value and deviation of every camera parameter
    out.append(p2 + 'Cameras:')
    nK, nP = int(s.IO.model.nK), int(s.IO.model.nP)
    head = (['Camera Constant', 'px - principal point x', 'py - principal point y', 'Format width', 'Format height']
            + ['K%d - radial distortion %d' % (k + 1, k + 1) for k in range(nK)]
            + ['P%d - decentering distortion %d' % (k + 1, k + 1) for k in range(nP)]
            + ['as - off-unit aspect parameter', 'sk - skew', 'Image width', 'Image height',
               'X resolution', 'Y resolution', 'Pixel width', 'Pixel height'])
    cu = getattr(s.IO.model, 'camUnit', 'mm')
    unit = ([cu] * 5 + ['%s^(-%d)' % (cu, 2 * k + 3) for k in range(nK)] + ['%s^(-3)' % cu] * nP
            + ['', '', 'px', 'px', 'px/%s' % cu, 'px/%s' % cu, cu, cu])
    rows = [0, 1, 2, -1, -2] + [5 + k for k in range(nK + nP)] + [3, 4, -3, -4, -5, -6, -7, -8]
    io = np.array(s.IO.val, float)
    io[[2] + list(range(5, io.shape[0]))] *= -1.0                 # display signs (:357-358)
    sd_io = np.sqrt(np.maximum(CIO.diagonal(), 0)).reshape(io.shape, order='F')
    ps = s.post.sensor if hasattr(s.post, 'sensor') else s.IO.sensor       # bundle.m:360-366
    sensor = np.vstack([ps.ssSize, ps.imSize, ps.imSize / ps.ssSize, ps.pxSize])
    blocks = np.asarray(s.IO.struct.block)
    seen, cam_no = set(), 0
    est_any = np.asarray(s.bundle.est.IO, bool)
    for i in range(io.shape[1]):
        key = tuple(blocks[:, i])
        if key in seen:
            continue
        seen.add(key); cam_no += 1
        out.append(p3 + 'Camera%d (simple)' % cam_no)
        out.append(p4 + 'Lens distortion model:')
        out.append(p5 + 'Backward (Photogrammetry) model %d' % int(np.ravel(s.IO.model.distModel)[i]))
        pad = len('Significance:') if est_any[:, i].any() else len('Value:')
        for h, u, r in zip(head, unit, rows):
            val, sigma = (io[r, i], sd_io[r, i]) if r >= 0 else (sensor[-r - 1, i], 0.0)
            vals = [('Value:', ('%g %s' % (val, u)))]
            if sigma != 0 and not np.isnan(sigma):
                vals.append(('Deviation:', '%.3g %s' % (sigma, u)))
            out.append(p4 + h + ':')
            out += _pretty(p5, vals, pad, pad)
    # ---- Photograph standard deviations (:441-503)
    out.append(p2 + 'Precisions / Standard Deviations:')
    out.append(p3 + 'Photograph Standard Deviations:')
    m = s.EO.val.shape[0]
    sd_eo = np.sqrt(np.maximum(CEO.diagonal(), 0)).reshape(s.EO.val.shape, order='F')
    order = [3, 4, 5, 0, 1, 2]
    scale = [180 / np.pi] * 3 + [1.0] * 3
    names = ['Omega', 'Phi', 'Kappa', 'Xc', 'Yc', 'Zc']
    units = ['deg'] * 3 + ['ou'] * 3
    pad = len('Deviation:')
    eo_names = getattr(s.EO, 'name', None)
    for i in range(s.EO.val.shape[1]):
        out.append(p4 + 'Photo %d: %s' % (i + 1, os.path.basename(str(eo_names[i])) if eo_names is not None else ''))
        for nme, u, r, sc in zip(names, units, order, scale):
            vals = [('Value:', '%.6f %s' % (sc * s.EO.val[r, i], u))]
            if sd_eo[r, i] != 0:
                vals.append(('Deviation:', '%.3g %s' % (sc * sd_eo[r, i], u)))
            out.append(p5 + nme + ':')
            out += _pretty(p6, vals, pad, pad)
    # ---- Point precision (:674-700)
    var = COP.diagonal().reshape(s.OP.val.shape, order='F').astype(float)
    var[~np.asarray(s.bundle.est.OP, bool)] = np.nan
    tstd = np.sqrt(var.sum(0))
    ids = np.asarray(s.OP.id)
    out.append(p2 + 'Point Precision')
    out.append(p3 + 'Total standard deviation (RMS of X/Y/Z std):')
    out.append(p4 + 'Minimum: %.2g (OP %d)' % (np.nanmin(tstd), ids[np.nanargmin(tstd)]))
    out.append(p4 + 'Maximum: %.2g (OP %d)' % (np.nanmax(tstd), ids[np.nanargmax(tstd)]))
    sd = np.sqrt(var)
    for c, ax in enumerate('XYZ'):
        j = int(np.nanargmax(sd[c]))
        out.append(p3 + 'Maximum %s standard deviation: %.2g (OP %d)' % (ax, sd[c, j], ids[j]))
    return out


def bundle_result_file(s, E, CIO=None, CEO=None, COP=None, path='report.txt'):
    lines = ['Damped Bundle Adjustment Toolbox result file (numeric subset, dbat_amd.report)']
    lines += bundle_result_lines(s, E, CIO, CEO, COP)
    with open(path, 'w') as fh:
        fh.write('\n'.join(lines) + '\n')
    return lines
