"""Shared test helpers: known-answer project set-up and seeded synthetic scenes."""
import json
import os

import numpy as np

from dbat_amd import loadpm as L

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _golden_path(name):
    """Path of a fixture; .xz fixtures are unpacked next to pytest's temp files."""
    path = os.path.join(GOLDEN, name)
    if os.path.exists(path):
        return path
    import lzma
    import tempfile
    out = os.path.join(tempfile.gettempdir(), 'dbat_amd_golden_' + name)
    if not os.path.exists(out):
        with lzma.open(path + '.xz', 'rb') as fi, open(out + '.part%d' % os.getpid(), 'wb') as fo:
            fo.write(fi.read())
        os.replace(out + '.part%d' % os.getpid(), out)
    return out


def camcal_struct(model=3, export='camcal-pmexport.txt', ctrl=True):
    """demo/camcaldemo.m:56-100 set-up, starting from PhotoModeler's own EO/OP
    instead of resect/forwintersect (the converged solution is x0-independent
    inside the basin)."""
    prob = L.loadpm(_golden_path(export))
    s = L.prob2dbatstruct(prob, distModel=model)
    s.IO.val[0, :] = 7.3                                   # setcamvals 'default',7.3
    s.IO.val[1:3, :] = 0.5 * np.diag([1, -1]) @ s.IO.sensor.ssSize
    s.IO.val[3:, :] = 0
    s.bundle.est.IO[:] = True                              # setcamest 'all','not','sk'
    s.bundle.est.IO[4, :] = False
    if model < 3:
        s.bundle.est.IO[3, :] = False
    s.bundle.est.EO[:] = True
    if not ctrl:                                           # camcaldemo_no_datum.m:65
        s.proj = type(s.post)(objUnit='m', x0desc='')
        return s
    s.prior.OP.isCtrl = s.OP.id > 1000
    pts = L.loadcpt(os.path.join(GOLDEN, 'camcal-fixed.txt'))
    s.proj = type(s.post)(objUnit='m', x0desc='Camera calibration from EXIF value')      # camcaldemo.m:107
    return L.setcpt(s, pts)


def camcal_failure_struct(kind):
    """The failure-mode demos on their own inputs: camcaldemo_1ray.m (all but
    one observation of OP 88 removed), camcaldemo_missing_obs.m (no observation
    of OP 13 and 60) -- same pipeline as camcaldemo -- and camcaldemo_no_datum.m
    (no control points, PhotoModeler's EO/OP as initial values)."""
    if kind == 'no-datum':
        return camcal_struct(3, ctrl=False)
    return camcal_demo_struct(3, 'camcal-pmexport-%s.txt' % kind)


def camcal_failures_expected():
    with open(os.path.join(GOLDEN, 'camcal_failures_expected.json')) as fh:
        return json.load(fh)


def check_failure_against_report(ok, iters, s0, E, exp):
    assert not ok and E.code == exp['code'] and iters == exp['iterations'] == 0
    assert (E.numParams, E.numObs, E.redundancy) == (exp['numParams'], exp['numObs'], exp['redundancy'])
    if np.isnan(exp['sigma0']):
        assert np.isnan(s0) and np.isnan(E.res[0])
    else:       # values at x0: 1e-5, see check_report_lines
        assert abs(s0 / exp['sigma0'] - 1) < 1e-5 and abs(E.res[0] / exp['firstError'] - 1) < 1e-5
    w = E.weakness
    if exp['structural'] is None:
        assert w.structural is None
    else:
        assert (w.structural.rank, w.structural.deficiency) == (exp['structural']['rank'], exp['structural']['deficiency'])
        assert list(w.structural.suspectedParams) == exp['structural']['suspectedParams']
        assert np.isnan(w.numerical.rank) and exp['numerical_not_tested']
    if exp['numerical'] is not None:
        assert (w.numerical.rank, w.numerical.deficiency) == (exp['numerical']['rank'], exp['numerical']['deficiency'])
        assert len(w.numerical.suspectedParams) == exp['numerical']['deficiency']
        # every null-space vector is one: J v = 0 to rounding, eigenvalues ~ eps
        assert np.abs(w.numerical.d).max() < 1e-12


def camcal_demo_struct(model=3, export='camcal-pmexport.txt'):
    """The whole demo/camcaldemo.m:56-107 set-up: EXIF camera, EO cleared and
    computed by 3-point resection from the four control points, free OP cleared
    and computed by forward intersection.  With these initial values the
    reference's committed report also pins the iteration count and the first
    residual norm."""
    from dbat_amd import initial as I
    import initial_oracle as IO_
    s = I.clearop(I.cleareo(camcal_struct(model, export)))
    assert np.isnan(s.EO.val).all() and np.isnan(s.OP.val[:, ~s.prior.OP.isCtrl]).all()
    cpId = s.OP.id[s.prior.OP.isCtrl]
    s1, rms, fail = IO_.resect(s, 'all', cpId, 1, 0, cpId)
    assert not fail
    return IO_.forwintersect(s1, 'all', True)


def camcal_expected():
    with open(os.path.join(GOLDEN, 'camcal_expected.json')) as fh:
        return json.load(fh)


def check_camcal_against_report(res, s0, E, exp, sig=6):
    """Compare a converged camcal result with the reference report's printed
    values (6 significant digits; bundle_result_file.m:357-358 flips the sign
    of py, K and P for display)."""
    def close(a, b, digits=sig):
        if b == 0:
            return abs(a) < 10.0 ** (-digits)
        # one unit in the last printed digit: the reference stops at convTol=1e-6,
        # so its 7th digit depends on its own x0/iteration path.
        return abs(a - b) <= 1.01 * 10.0 ** (np.floor(np.log10(abs(b))) - digits + 1)
    assert E.numParams == exp['numParams'] and E.numObs == exp['numObs']
    assert E.redundancy == exp['redundancy']
    assert close(s0, exp['sigma0']), (s0, exp['sigma0'])
    io = res.IO.val[:, 0]
    rep = exp['IO_report']
    got = {'cc': io[0], 'px': io[1], 'py': -io[2], 'as': io[3], 'K1': -io[5],
           'K2': -io[6], 'K3': -io[7], 'P1': -io[8], 'P2': -io[9]}
    for k, v in rep.items():
        assert close(got[k], v, 4 if k == 'cc' else sig), (k, got[k], v)
    eo = np.array(exp['EO_report_deg'])
    ang = np.rad2deg(res.EO.val[3:6]).T
    dang = (ang - eo[:, :3] + 180.0) % 360.0 - 180.0     # x0-dependent 2*pi wraps
    assert np.abs(dang).max() < 1e-6
    assert np.abs(res.EO.val[:3].T - eo[:, 3:]).max() < 1e-6


def check_camcal_cov_against_report(s, CIO, CEO, COP, exp):
    """Posterior standard deviations against the "Deviation" values of the
    reference report (3 significant digits; angles printed in degrees) and its
    point-precision summary (2 significant digits, bundle_result_file.m:674-700)."""
    def close(a, b, digits):
        return abs(a - b) <= 0.51 * 10.0 ** (np.floor(np.log10(abs(b))) - digits + 1) * 1.02
    sd_io = np.sqrt(CIO.diagonal()).reshape(s.IO.val.shape, order='F')[:, 0]
    got = {'cc': sd_io[0], 'px': sd_io[1], 'py': sd_io[2], 'as': sd_io[3], 'K1': sd_io[5],
           'K2': sd_io[6], 'K3': sd_io[7], 'P1': sd_io[8], 'P2': sd_io[9]}
    for k, v in exp['IO_deviation'].items():
        assert close(got[k], v, 3), ('IO', k, got[k], v)
    sd_eo = np.sqrt(CEO.diagonal()).reshape(s.EO.val.shape, order='F')[:6].T   # X Y Z om ph ka
    dev = np.array(exp['EO_deviation'], float)                                # om ph ka [deg] X Y Z
    for i in range(dev.shape[0]):
        for j in range(3):
            assert close(np.rad2deg(sd_eo[i, 3 + j]), dev[i, j], 3), ('EO angle', i, j)
            assert close(sd_eo[i, j], dev[i, 3 + j], 3), ('EO pos', i, j)
    var = COP.diagonal().reshape(s.OP.val.shape, order='F')
    var = np.where(np.asarray(s.bundle.est.OP, bool), var, np.nan)
    tstd = np.sqrt(var.sum(0))
    ids = np.asarray(s.OP.id)
    ts = exp['OP_total_std']
    assert close(np.nanmin(tstd), ts['min'], 2) and close(np.nanmax(tstd), ts['max'], 2)
    assert ids[np.nanargmax(tstd)] == ts['max_id']
    for c in range(3):
        assert close(np.sqrt(np.nanmax(var[c])), exp['OP_max_std'][c], 2), ('OP', c)


def check_report_lines(lines, ref_path=os.path.join(GOLDEN, 'camcal-dbatreport.txt'), demo_x0=False,
                       ref_lines=None, x0_lines=('First error:',), x0_tol=1e-5):
    """Every line of dbat_amd.report's output must occur, in order, in the
    reference's committed result file -- verbatim, or with numbers that differ by
    one unit in the last printed digit (the reference stops at convTol 1e-6) or by
    a full turn (angles).  The iteration count and the first error depend on the
    demo's initial values (EXIF + resection): they are compared only with
    demo_x0 (camcal_demo_struct), the first error to 1e-5 -- camera 21's
    resection quartic has a near-triple root, so one ulp in its coefficients moves
    the sixth digit (tests/test_oracle.py::test_resect_first_error_conditioning).
    Returns the number of verbatim matches."""
    import re
    ref = [l.rstrip() for l in (ref_lines if ref_lines is not None else open(ref_path).read().splitlines())]
    num = re.compile(r'[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?')
    skel = lambda l: num.sub('#', l)

    def close(a, b):
        if a == b:
            return True
        fa, fb = float(a), float(b)
        digits = len(re.sub(r'[eE].*$', '', b).replace('-', '').replace('+', '').replace('.', '').lstrip('0')) or 1
        tol = 1.01 * 10.0 ** (np.floor(np.log10(abs(fb))) - digits + 1) if fb != 0 else 1e-12
        return abs(fa - fb) <= tol or abs(abs(fa - fb) - 360.0) <= tol

    pos, verbatim = 0, 0
    for l in lines:
        l = l.rstrip()
        if ('Number of iterations:' in l or 'First error:' in l) and not demo_x0:
            continue
        key = [k for k in x0_lines if l.strip().startswith(k)]
        if key and 'nan' not in l.lower():
            # values at the resected x0 (a failed bundle's sigma0 and last error are such values too)
            want = [float(num.findall(r)[-1]) for r in ref if r.strip().startswith(key[0])][0]
            assert abs(float(num.findall(l)[-1]) / want - 1) < x0_tol, (l, want)
            continue
        hit = None
        for q in range(pos, len(ref)):
            if ref[q] == l:
                hit = q; verbatim += 1
                break
            if skel(ref[q]) == skel(l):
                na, nb = num.findall(l), num.findall(ref[q])
                if len(na) == len(nb) and all(close(a, b) for a, b in zip(na, nb)):
                    hit = q
                    break
        assert hit is not None, 'line not found in the reference report: %r' % l
        pos = hit + 1
    return verbatim


# ---------------------------------------------------------------------------
# seeded synthetic variants used by the parity tests
# ---------------------------------------------------------------------------

def synth_struct(name='tiny', variant='plain', seed=None):
    """Synthetic scene (dbat_amd.synth) with optional complications:
       'plain'    fixed IO, datum by dependency
       'selfcal'  one shared IO block, cc px py K1-3 P1-2 estimated
       'imagevar' image-variant principal point (one px,py block per image),
                  cc,K shared  (demo/romabundledemo_imagevariant.m:42-48)
       'priors'   EO position priors on every 3rd camera, OP priors on control
                  points, a few fixed control points, non-uniform IP.std
       'groups4'  four camera groups with independent self-calibrated IO blocks
                  (BASELINE.json configs[4]; script/setdbatcamsandimages.m:28,
                  buildserialindices.m:162-221)
    """
    from dbat_amd import synth
    if variant == 'groups4':
        return synth.make_scene(name, seed=seed, selfcal=True, groups=4)
    s, truth = synth.make_scene(name, seed=seed)
    nc, npnt = s.EO.val.shape[1], s.OP.val.shape[1]
    rng = np.random.default_rng(12345)
    if variant == 'selfcal':
        s.bundle.est.IO[[0, 1, 2, 5, 6, 7, 8, 9]] = True
    elif variant == 'imagevar':
        s.bundle.est.IO[[0, 1, 2, 5, 6]] = True
        s.IO.struct.block[1:3] = np.arange(1, nc + 1)[None, :]
    elif variant == 'priors':
        s.bundle.est.EO[:] = True                       # datum from priors instead
        cams = np.arange(0, nc, 3)
        s.prior.EO.use[0:3, cams] = True
        s.prior.EO.val[0:3, cams] = truth['EO'][0:3, cams] + rng.normal(0, 0.02, (3, len(cams)))
        s.prior.EO.std[0:3, cams] = 0.02
        cps = np.arange(0, npnt, 37)
        s.prior.OP.use[:, cps] = True
        s.prior.OP.val[:, cps] = truth['OP'][:, cps] + rng.normal(0, 0.01, (3, len(cps)))
        s.prior.OP.std[:, cps] = np.array([[0.01], [0.01], [0.02]])
        fixed = np.arange(5, npnt, 53)
        s.OP.val[:, fixed] = truth['OP'][:, fixed]
        s.bundle.est.OP[:, fixed] = False
        s.IP.std = s.IP.std * (1 + (np.arange(s.IP.std.shape[1]) % 3)[None, :] * 0.5)
        s.IP.sigmas = np.unique(s.IP.std)
    elif variant != 'plain':
        raise ValueError(variant)
    return s, truth


def relerr(a, b):
    """misc/relerr.m: Frobenius relative error."""
    a, b = np.asarray(a, float), np.asarray(b, float)
    d = np.linalg.norm(b.ravel())
    return np.linalg.norm((a - b).ravel()) / (d if d > 0 else 1.0)


def script_forwintersect(s):
    """The script operation forward_intersection: on the device where there is one (the product path,
    dbat_amd.loadtables.forwintersect -> dbat_hip_forwintersect), by the oracle's restatement on the CPU box."""
    import torch
    if torch.cuda.is_available():
        from dbat_amd import loadtables as T
        return T.forwintersect(s)
    import initial_oracle
    return initial_oracle.forwintersect(s, 'all')


def roma_struct():
    """data/script/romabundledemo/romabundledemo.xml: loaded camera and EO,
    OP by forward intersection, estimate cc, pp, K1, K2 (no aspect, skew, P,
    K3), datum by dependency on camera 1."""
    from dbat_amd import loadtables as T
    from dbat_amd.dbatstruct import seteoest_depend
    exp = roma_expected()
    ci = exp['camera_in']
    eo = T.load_table(os.path.join(GOLDEN, 'roma-initial_eo.txt'))
    mk = T.load_table(os.path.join(GOLDEN, 'roma-markpts.txt.xz'))
    h = ci['sensor_height']
    sensor = (ci['image'][0] * h / ci['image'][1], h)          # sensor 'auto,24', aspect 1
    io = T.camera_io(ci['cc'], ci['pp'], ci['K'], ci['P'])
    s = T.struct_from_tables(io, sensor, ci['image'], eo, mk, 'im,id,x,y', 1.0, distModel=ci['model'])
    s = script_forwintersect(s)
    s.bundle.est.IO[[0, 1, 2, 5, 6]] = True
    s.EO.name = exp['image_paths']
    s.proj = type(s.post)(objUnit='m', x0desc='')
    return seteoest_depend(s, 0)


PRAGUE = {'c1': ('cam', 'fixed', False), 'c2': ('cam', 'weighted', True), 's1': ('sxb', 'f-op0', False),
          's2': ('sxb', 'w-op0', True), 's3': ('sxb', 'w-op1', True), 's4': ('sxb', 'wsmart', True)}


def prague_struct(label):
    """demo/prague2016_pm.m:82-190 for experiment label c1, c2, s1..s4: the
    PhotoModeler export with its loaded, fixed camera; control points from
    the reference file, shifted by the mean offset to PhotoModeler's frame
    (:150-166), fixed or weighted; EO by resection, OP by forward intersection.
    The demo runs the legacy lens model 1 (prob2dbatstruct's default), the model
    that model 2 replicates (bundle.m:49-51); oracle and library evaluate it as
    model 2.  Returns (s, path of the committed report)."""
    from dbat_amd import initial as I
    import initial_oracle as IO_
    site, stub, weighted = PRAGUE[label]
    prob = L.loadpm(_golden_path('prague-%s-%s-pmexport.txt' % (site, stub)))
    s = L.prob2dbatstruct(prob, distModel=1)
    s.bundle.est.IO[:] = False                             # setcamvals 'loaded'; setcamest 'not','all'
    pts = L.loadcpt(os.path.join(GOLDEN, 'prague-%s-ctrlpts-%s.txt' % (site, 'weighted' if weighted else 'fixed')))
    pm = prob['ctrlPts']
    assert np.all(np.isin(pm[:, 0], pts['id']))
    _, ia, ib = np.intersect1d(pm[:, 0].astype(int), pts['id'], return_indices=True)
    offset = pm[ia, 1:4].T - pts['pos'][:, ib]
    pts['pos'] = pts['pos'] + offset.mean(1)[:, None]
    s = L.setcpt(s, pts)
    s.proj = type(s.post)(objUnit='m', x0desc='')
    s = I.clearop(I.cleareo(s))
    cpId = s.OP.id[s.prior.OP.isCtrl]
    s, rms, fail = IO_.resect(s, 'all', cpId, 1, 0, cpId)
    assert not fail
    return IO_.forwintersect(s, 'all', True), os.path.join(GOLDEN, 'prague-%s-%s-dbatreport.txt' % (site, stub))


def sxb_prior_eo_struct(use_prior_eo):
    """demo/sxb_prior_eo.m:33-83: the Strasbourg smart-point project in its
    1e6-m frame, weighted control points, loaded fixed camera, and optionally
    prior observations (0.05 m) of four of the five camera positions from
    ref/fake-camera-positions.txt (misc/setprioreo.m).  Returns (s, report)."""
    from dbat_amd import initial as I
    import initial_oracle as IO_
    prob = L.loadpm(_golden_path('prague-sxb-wsmart-with-orient-pmexport.txt'))
    s = L.prob2dbatstruct(prob, distModel=1)               # the demo's model; evaluated as model 2 (bundle.m:49-51)
    s.bundle.est.IO[:] = False
    s = L.setcpt(s, L.loadcpt(os.path.join(GOLDEN, 'prague-sxb-ctrlpts-weighted.txt')))
    s.proj = type(s.post)(objUnit='m', x0desc='')
    if use_prior_eo:
        names = [os.path.basename(n) for n in s.EO.name]
        for line in open(os.path.join(GOLDEN, 'prague-sxb-fake-camera-positions.txt')):
            if line.startswith('#') or not line.strip():
                continue
            tok = [v.strip() for v in line.split(',')]
            i = names.index(tok[0])
            pos, sd = np.array([float(v) for v in tok[1:4]]), float(tok[4])
            s.prior.EO.val[:3, i] = pos
            s.EO.val[:3, i] = pos
            s.prior.EO.std[:3, i] = sd
            s.prior.EO.use[:3, i] = sd != 0
            s.bundle.est.EO[:3, i] = sd != 0
    s = I.clearop(I.cleareo(s))
    cpId = s.OP.id[s.prior.OP.isCtrl]
    s, rms, fail = IO_.resect(s, 'all', cpId, 1, 0, cpId)
    assert not fail
    return IO_.forwintersect(s, 'all', True), os.path.join(
        GOLDEN, 'prague-sxb-%sprior-eo-dbatreport.txt' % ('' if use_prior_eo else 'no-'))


def sxb_expected():
    with open(os.path.join(GOLDEN, 'sxb_expected.json')) as fh:
        return json.load(fh)


def sxb_struct():
    """data/script/sxb/sxb.xml: five aerial images, fixed camera, 14 control
    points as prior observations (0.02/0.02/0.04 m) and two check points, marked
    points at 0.5 px and smart points at 1 px, project coordinates of 1e6 m;
    operations set_initial_values (loaded), set_bundle_estimate_params (io
    false, eo true, op default), spatial_resection, forward_intersection
    (script/parseops.m:36-43)."""
    from dbat_amd import initial as I
    import initial_oracle as IO_
    from dbat_amd import loadtables as T
    exp = sxb_expected()
    cam = exp['camera']
    pts = L.loadcpt(os.path.join(GOLDEN, 'sxb-control.txt'))

    def pick(keep):
        m = np.isin(pts['id'], exp['check_ids']) == keep
        return dict(id=pts['id'][m], name=[n for n, k in zip(pts['name'], m) if k], pos=pts['pos'][:, m],
                    std=pts['std'][:, m])
    io = T.camera_io(cam['cc'], cam['pp'], cam['K'], cam['P'])
    marks = [(T.load_table(os.path.join(GOLDEN, 'sxb-%s.txt' % nm)), 'id,im,x,y', exp['sxy'][nm])
             for nm in ('markpts', 'smartpts')]
    s = T.struct_from_script(io, cam['sensor'], cam['image'], exp['images'], marks, pick(False), pick(True),
                             distModel=cam['model'], im_names=exp['image_paths'])
    s = T.set_script_defaults(s)
    s.proj = type(s.post)(objUnit='m', x0desc='')
    cpId = s.OP.id[s.prior.OP.isCtrl]
    s, rms, fail = IO_.resect(s, 'all', cpId, 1, 0, cpId)
    assert not fail
    return IO_.forwintersect(s, 'all', True)


def check_sxb_against_report(res, s0, E, iters, exp):
    """data/script/sxb/result/report.txt:19-43 and its photo blocks.  The
    converged values are compared to the printed precision; the first error to
    1e-4: the resected camera centres carry |C|^2*eps ~ 1e-4 m of SVD rounding
    noise at these coordinates (dbat_amd/initial.py, resect)."""
    rep = exp['report']
    six = lambda a, b: abs(a - b) <= 1.01 * 10.0 ** (np.floor(np.log10(abs(b))) - 5)
    assert (E.numParams, E.numObs, E.redundancy) == (rep['numParams'], rep['numObs'], rep['redundancy'])
    assert (rep['nIO'], rep['nEO'], rep['nOP']) == (0, 30, 1143)
    assert iters == rep['iterations'] == 4
    assert six(s0, rep['sigma0']) and six(res.post.sigmas[0], rep['sigma0_px'])
    assert abs(E.res[0] / rep['firstError'] - 1) < 1e-4 and six(E.res[-1], rep['lastError'])
    eo = np.array(rep['EO_report_deg'])
    assert np.abs(np.rad2deg(res.EO.val[3:6]).T - eo[:, :3]).max() < 1.5e-6
    assert np.abs(res.EO.val[:3].T - eo[:, 3:]).max() < 2e-6          # metres, at 1e6 m


def roma_demo_struct(variant):
    """demo/romabundledemo{,_selfcal,_imagevariant}.m:57-84: PhotoModeler's
    camera (printed in roma-dbatreport.txt:58-92: 36.036 x 24 mm format as an
    off-unit aspect of 0.000998889 on square 24/3744 mm pixels), fixed or
    self-calibrated without skew, or with one principal point per image; OP by
    forward intersection, datum by dependency on camera 1.  The image points
    are those of the script fixture (same 181 122 observations)."""
    from dbat_amd import loadtables as T
    from dbat_amd.dbatstruct import seteoest_depend
    cam = roma_variants_expected()['fixed']['IO_report']
    eo = T.load_table(os.path.join(GOLDEN, 'roma-initial_eo.txt'))
    mk = T.load_table(os.path.join(GOLDEN, 'roma-markpts.txt.xz'))
    io = T.camera_io(cam['cc'], (cam['px'], cam['py']), [cam['K1'], cam['K2'], cam['K3']], [cam['P1'], cam['P2']],
                     aspect=1.0 - cam['as'])
    s = T.struct_from_tables(io, (36.0, 24.0), (5616, 3744), eo, mk, 'im,id,x,y', 1.0, distModel=3)
    s = script_forwintersect(s)
    if variant != 'fixed':
        s.bundle.est.IO[:] = True                              # setcamest 'all','not','sk'
        s.bundle.est.IO[4] = False
    if variant == 'imagevariant':                              # romabundledemo_imagevariant.m:63
        s.IO.struct.block[1:3] = np.arange(1, s.EO.val.shape[1] + 1)[None, :]
    return seteoest_depend(s, 0)


def roma_variants_expected():
    with open(os.path.join(GOLDEN, 'roma_variants_expected.json')) as fh:
        return json.load(fh)


def check_roma_variant(res, s0, E, exp, CIO=None):
    """sigma0, counts, and -- when estimated -- the camera values (and their
    posterior deviations) of roma-dbatreport{,-selfcal,-imagevariant}.txt, six
    (three) significant digits; report signs as bundle_result_file.m:357-358."""
    close = lambda a, b, d: abs(a - b) <= 1.01 * 10.0 ** (np.floor(np.log10(abs(b))) - d + 1) if b else abs(a) < 1e-12
    assert (E.numParams, E.numObs, E.redundancy) == (exp['numParams'], exp['numObs'], exp['redundancy'])
    assert close(s0, exp['sigma0'], 6), (s0, exp['sigma0'])
    io = res.IO.val[:, 0]
    got = {'cc': io[0], 'px': io[1], 'py': -io[2], 'as': io[3], 'K1': -io[5], 'K2': -io[6], 'K3': -io[7],
           'P1': -io[8], 'P2': -io[9]}
    for k, v in exp['IO_report'].items():
        assert close(got[k], v, 6), (k, got[k], v)
    if CIO is not None:
        sd = np.sqrt(CIO.diagonal()).reshape(res.IO.val.shape, order='F')[:, 0]
        gsd = {'cc': sd[0], 'px': sd[1], 'py': sd[2], 'as': sd[3], 'K1': sd[5], 'K2': sd[6], 'K3': sd[7],
               'P1': sd[8], 'P2': sd[9]}
        assert len(exp['IO_deviation']) >= 8
        for k, v in exp['IO_deviation'].items():
            assert abs(gsd[k] - v) <= 0.52 * 10.0 ** (np.floor(np.log10(abs(v))) - 2), ('dev', k, gsd[k], v)


def roma_expected():
    with open(os.path.join(GOLDEN, 'roma_expected.json')) as fh:
        return json.load(fh)


def check_roma_against_result(res, s0, E, iters, exp):
    rep, cam = exp['report'], exp['camera_out']
    assert E.numParams == rep['numParams'] and E.numObs == rep['numObs']
    assert E.redundancy == rep['redundancy'] and iters == rep['iterations']
    assert abs(s0 - rep['sigma0']) < 1.01e-6
    assert abs(E.res[0] - rep['firstError']) < 0.0051 * 1.01       # printed with 6 significant digits
    assert abs(E.res[-1] - rep['lastError']) < 0.00051 * 1.01
    io = res.IO.val[:, 0]
    # result/EOS5DMarkII.xml is written with %.18g; the reference stops at convTol=1e-6
    assert abs(io[0] - cam['cc']) < 1e-7 * cam['cc']
    assert abs(io[1] - cam['pp'][0]) < 1e-7 * cam['pp'][0]
    assert abs(-io[2] - cam['pp'][1]) < 1e-7 * cam['pp'][1]
    assert abs(-io[5] - cam['K'][0]) < 1e-6 * abs(cam['K'][0])
    assert abs(-io[6] - cam['K'][1]) < 1e-6 * abs(cam['K'][1])
    eo = np.array(rep['EO_report_deg'])
    dang = (np.rad2deg(res.EO.val[3:6]).T - eo[:, :3] + 180.0) % 360.0 - 180.0
    assert np.abs(dang).max() < 1.01e-6
    assert np.abs(res.EO.val[:3].T - eo[:, 3:]).max() < 1.01e-6


# ---------------------------------------------------------------- C ABI driver (tests/abi_c_driver.c)
def build_abi_c_driver():
    """Compile tests/abi_c_driver.c as plain C11 against include/dbat_hip.h and link it with
    libdbat_hip.so.  Returns the path of the executable."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, 'tests', 'abi_c_driver.out')
    src = os.path.join(root, 'tests', 'abi_c_driver.c')
    lib = os.path.join(root, 'dbat_amd')
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(
            os.path.join(root, 'include', 'dbat_hip.h'))):
        subprocess.run(['gcc', '-std=c11', '-Wall', '-Wextra', '-pedantic', '-Werror', '-O1', '-I' + os.path.join(root, 'include'),
                        src, '-o', out, '-L' + lib, '-ldbat_hip', '-Wl,-rpath,' + lib, '-Wl,-rpath,/opt/rocm/lib'],
                       check=True, capture_output=True)
    return out


def dump_problem(s, path):
    """The arrays of dbat_hip_problem in declaration order (abi_c_driver.c::load)."""
    from dbat_amd import _hip
    p, keep = _hip.problem_from_struct(s)
    hdr = np.array([p.n_images, p.n_points, p.n_obs, p.dist_model, p.nK, p.nP, 5 + p.nK + p.nP, 0], np.int64)
    order = ['ip_cam', 'ip_pt', 'ip_val', 'ip_std', 'IO_val', 'px_size', 'EO_val', 'OP_val', 'est_IO', 'est_EO',
             'est_OP', 'IO_block', 'EO_block', 'prior_IO_use', 'prior_IO_val', 'prior_IO_std', 'prior_EO_use',
             'prior_EO_val', 'prior_EO_std', 'prior_OP_use', 'prior_OP_val', 'prior_OP_std']
    with open(path, 'wb') as f:
        f.write(hdr.tobytes())
        for k in order:
            f.write(keep[k].tobytes())


def oracle_lm_reordered(s, seed=1, max_iter=20, conv_tol=1e-6):
    """The oracle's Levenberg-Marquardt loop (levenberg_marquardt.m:52-250) on the SAME problem with
    the rows of r and J visited in a different order: every sum over the rows (r'r, J'J, J'r) then
    rounds differently, nothing else changes.  Returns (x, code, iters, rr, lambdas).

    Used to decide where an iteration count is a property of the problem and where it is rounding
    noise of the reference algorithm itself (tests/test_hip_parity.py::check_history)."""
    import copy
    import dbat_oracle as o
    s = copy.deepcopy(s)
    for nm in ('IO', 'EO', 'OP'):
        pr = getattr(s.prior, nm)
        pr.use = np.asarray(pr.use, bool) & np.asarray(getattr(s.bundle.est, nm), bool)
    s = o.buildserialindices(s)
    x0 = o.serialize(s)
    w = o.buildweightvector(s)
    perm = np.random.default_rng(seed).permutation(len(w))

    def res_fun(x, jac):
        if not jac:
            return o.brown_euler_cam4(x, s, False)[perm]
        r, J = o.brown_euler_cam4(x, s, True)
        return r[perm], J.tocsr()[perm].tocsc()

    x, code, n, final, T, rr, lam = o.levenberg_marquardt(res_fun, x0, w[perm], max_iter, o.term_relative(conv_tol),
                                                          -1e-10, -1e-10)
    return x, code, n, rr, lam


def lm_decision_margins(s, max_iter=20, conv_tol=1e-6):
    """The oracle's LM loop with its decisions instrumented: (iters, smallest relative margin |fNew - f| / f of
    any accept/reject decision `fNew < f` (levenberg_marquardt.m:166), distance of the termination test's
    ratio ||Jp|| / (tol ||r||) from 1 (as a factor >= 1) over all its evaluations (:217))."""
    import copy
    import dbat_oracle as o
    s = copy.deepcopy(s)
    for nm in ('IO', 'EO', 'OP'):
        pr = getattr(s.prior, nm)
        pr.use = np.asarray(pr.use, bool) & np.asarray(getattr(s.bundle.est, nm), bool)
    s = o.buildserialindices(s)
    x0 = o.serialize(s)
    w = o.buildweightvector(s)
    state = {'f': None, 'margin': np.inf, 'term': np.inf}

    def res_fun(x, jac):
        out = o.brown_euler_cam4(x, s, jac)
        r = out[0] if jac else out
        f = 0.5 * float(np.sum(w * r * r))
        if jac:
            state['f'] = f
        elif state['f']:
            state['margin'] = min(state['margin'], abs(f - state['f']) / state['f'])
        return out

    def term_fun(Jp, r):
        ratio = np.linalg.norm(Jp) / (conv_tol * np.linalg.norm(r))
        state['term'] = min(state['term'], max(ratio, 1 / ratio) if ratio > 0 else np.inf)
        return ratio <= 1

    x, code, n, final, T, rr, lam = o.levenberg_marquardt(res_fun, x0, w, max_iter, term_fun, -1e-10, -1e-10)
    return n, state['margin'], state['term']


def lm_count_is_stable(s, ito, margin=1e-9):
    """Is the oracle's LM iteration count `ito` a property of the problem?  Yes if every accept/reject decision
    and the termination test were taken with a margin far above rounding: an implementation that agrees with
    the oracle to ~1e-11 per iterate (GPU: different elimination order, closed-form model) then takes the same
    decisions.  Sampling re-ordered summations instead is not enough: camcal model 3 gives 6 iterations under
    five row permutations and 16 under the sixth (its last accepted step lowers ||r|| by 2e-14 relative)."""
    n, m, t = lm_decision_margins(s)
    return n == ito and m > margin and t > 1.0 + 1e-3
