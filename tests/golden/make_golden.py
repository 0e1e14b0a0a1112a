"""Generate the known-answer fixtures under tests/golden/ from the reference's
committed demo inputs and reports (run in the authoring container only; the
GPU box has no /root/reference).

Inputs copied verbatim (data files, not source):
  data/dbat/pmexports/camcal-pmexport.txt, data/dbat/ref/camcal-fixed.txt
  data/script/romabundledemo/measurements/markpts.txt (xz-compressed),
  data/script/romabundledemo/prior/initial_eo.txt
  data/dbat/pmexports/camcal-pmexport-{1ray,missing-obs}.txt (xz-compressed):
  the inputs of the failure-mode demos camcaldemo_1ray / _missing_obs
Expected values parsed from the reference's committed reports:
  data/dbat/dbatexports/camcal-dbatreport-{1ray,missing-obs,no-datum}.txt
  (status code, structural / numerical rank, DMPERM's suspected parameters,
  counts, sigma0, first error) -> camcal_failures_expected.json
  data/dbat/dbatexports/camcal-dbatreport{,-model2..5}.txt
  (camcal-dbatreport.txt is also kept whole for the line-by-line comparison of
  dbat_amd.report's output, tests/test_oracle.py::test_report_lines)
"""
import json
import os
import re
import shutil

REF = '/root/reference/data/dbat'
HERE = os.path.dirname(os.path.abspath(__file__))


def parse_report(path):
    txt = open(path).read()
    out = {}
    g = lambda pat: re.search(pat, txt)
    out['sigma0'] = float(g(r'Sigma0:\s+([-\d.eE+]+)').group(1))
    out['redundancy'] = int(g(r'Redundancy\s+(\d+)').group(1))
    m = g(r'Number of params:\s+(\d+) \((\d+) IO, (\d+) EO, (\d+) OP\)')
    out['numParams'] = int(m.group(1))
    out['nIO'], out['nEO'], out['nOP'] = int(m.group(2)), int(m.group(3)), int(m.group(4))
    out['numObs'] = int(g(r'Number of observations:\s+(\d+)').group(1))
    out['iterations'] = int(g(r'Number of iterations:\s+(\d+)').group(1))
    out['lastError'] = float(g(r'Last error:\s+([-\d.eE+]+)').group(1))
    io = {}
    for key, pat in [('cc', r'Camera Constant:\s+Value:\s+([-\d.eE+]+)'),
                     ('px', r'px - principal point x:\s+Value:\s+([-\d.eE+]+)'),
                     ('py', r'py - principal point y:\s+Value:\s+([-\d.eE+]+)'),
                     ('K1', r'K1 - radial distortion 1:\s+Value:\s+([-\d.eE+]+)'),
                     ('K2', r'K2 - radial distortion 2:\s+Value:\s+([-\d.eE+]+)'),
                     ('K3', r'K3 - radial distortion 3:\s+Value:\s+([-\d.eE+]+)'),
                     ('P1', r'P1 - decentering distortion 1:\s+Value:\s+([-\d.eE+]+)'),
                     ('P2', r'P2 - decentering distortion 2:\s+Value:\s+([-\d.eE+]+)'),
                     ('as', r'as - off-unit aspect parameter:\s+Value:\s+([-\d.eE+]+)')]:
        m = g(pat)
        if m:
            io[key] = float(m.group(1))
    out['IO_report'] = io   # report sign convention: py, K, P flipped vs IO.val
    # posterior standard deviations ("Deviation" lines; bundle_result_file.m:128-130 from bundle_cov)
    iod = {}
    for key, pat in [('cc', r'Camera Constant:\s+Value:[^\n]*\n\s+Deviation:\s+([-\d.eE+]+)'),
                     ('px', r'px - principal point x:\s+Value:[^\n]*\n\s+Deviation:\s+([-\d.eE+]+)'),
                     ('py', r'py - principal point y:\s+Value:[^\n]*\n\s+Deviation:\s+([-\d.eE+]+)'),
                     ('K1', r'K1 - radial distortion 1:\s+Value:[^\n]*\n\s+Deviation:\s+([-\d.eE+]+)'),
                     ('K2', r'K2 - radial distortion 2:\s+Value:[^\n]*\n\s+Deviation:\s+([-\d.eE+]+)'),
                     ('K3', r'K3 - radial distortion 3:\s+Value:[^\n]*\n\s+Deviation:\s+([-\d.eE+]+)'),
                     ('P1', r'P1 - decentering distortion 1:\s+Value:[^\n]*\n\s+Deviation:\s+([-\d.eE+]+)'),
                     ('P2', r'P2 - decentering distortion 2:\s+Value:[^\n]*\n\s+Deviation:\s+([-\d.eE+]+)'),
                     ('as', r'as - off-unit aspect parameter:\s+Value:[^\n]*\n\s+Deviation:\s+([-\d.eE+]+)')]:
        m = g(pat)
        if m:
            iod[key] = float(m.group(1))
    out['IO_deviation'] = iod
    m = g(r'Total standard deviation \(RMS of X/Y/Z std\):\s+Minimum: ([-\d.eE+]+) \(OP (\d+)\)\s+Maximum: ([-\d.eE+]+) \(OP (\d+)\)')
    if m:
        out['OP_total_std'] = {'min': float(m.group(1)), 'min_id': int(m.group(2)),
                               'max': float(m.group(3)), 'max_id': int(m.group(4))}
        out['OP_max_std'] = [float(g(r'Maximum %s standard deviation: ([-\d.eE+]+)' % c).group(1)) for c in 'XYZ']
    photos, pdev = [], []
    blocks = re.split(r'Photo (\d+): \S+', txt)
    for i in range(1, len(blocks), 2):
        b = blocks[i + 1]
        vals, devs = [], []
        for key in ('Omega', 'Phi', 'Kappa', 'Xc', 'Yc', 'Zc'):
            m = re.search(key + r':\s+Value:\s+([-\d.]+)', b)
            vals.append(float(m.group(1)))
            m = re.search(key + r':\s+Value:[^\n]*\n\s+Deviation:\s+([-\d.eE+]+)', b)
            devs.append(float(m.group(1)) if m else None)
        assert int(blocks[i]) == len(photos) + 1
        photos.append(vals); pdev.append(devs)
    out['EO_report_deg'] = photos   # omega, phi, kappa [deg], Xc, Yc, Zc
    out['EO_deviation'] = pdev      # same order; angles in degrees
    return out


def main():
    shutil.copy(os.path.join(REF, 'pmexports/camcal-pmexport.txt'),
                os.path.join(HERE, 'camcal-pmexport.txt'))
    shutil.copy(os.path.join(REF, 'ref/camcal-fixed.txt'),
                os.path.join(HERE, 'camcal-fixed.txt'))
    # the committed result file of camcaldemo itself (an output of the reference: data, not source)
    shutil.copy(os.path.join(REF, 'dbatexports/camcal-dbatreport.txt'), os.path.join(HERE, 'camcal-dbatreport.txt'))
    exp = {'model3': parse_report(os.path.join(REF, 'dbatexports/camcal-dbatreport.txt'))}
    for m in (1, 2, 4, 5):
        exp['model%d' % m] = parse_report(
            os.path.join(REF, 'dbatexports/camcal-dbatreport-model%d.txt' % m))
        # kept whole as well, for the line-by-line comparison of the other lens models
        dst = os.path.join(HERE, 'camcal-dbatreport-model%d.txt' % m)
        shutil.copy(os.path.join(REF, 'dbatexports/camcal-dbatreport-model%d.txt' % m), dst)
        os.chmod(dst, 0o644)
    with open(os.path.join(HERE, 'camcal_expected.json'), 'w') as fh:
        json.dump(exp, fh, indent=1)
    print({k: (v['sigma0'], len(v['EO_report_deg'])) for k, v in exp.items()})
    roma()
    failures()
    sxb()
    roma_variants()
    prague()


def parse_failure(path):
    txt = open(path).read()
    g = lambda pat: re.search(pat, txt)
    out = {}
    m = g(r'Structural rank: (\d+) \(deficiency: (\d+)\)')
    out['structural'] = None if m is None else {'rank': int(m.group(1)), 'deficiency': int(m.group(2))}
    if m:
        blk = txt[txt.index('DMPERM suggests'):txt.index('Numerical rank')]
        out['structural']['suspectedParams'] = [l.strip() for l in blk.splitlines()[1:] if l.strip()]
    m = g(r'Numerical rank: (\d+) \(deficiency: (\d+)\)')
    out['numerical'] = None if m is None else {'rank': int(m.group(1)), 'deficiency': int(m.group(2))}
    out['numerical_not_tested'] = 'Numerical rank: not tested.' in txt
    out['code'] = int(g(r'Status:\s+fail \(code (-\d+):').group(1))
    out['status'] = g(r'Status:\s+(fail[^\n]*)').group(1).strip()
    num = lambda pat: float(g(pat).group(1))
    out['sigma0'] = num(r'Sigma0:\s+(\S+)')
    out['redundancy'] = int(g(r'Redundancy\s+(\d+)').group(1))
    out['numParams'] = int(g(r'Number of params:\s+(\d+)').group(1))
    out['numObs'] = int(g(r'Number of observations:\s+(\d+)').group(1))
    out['iterations'] = int(g(r'Number of iterations:\s+(\d+)').group(1))
    out['firstError'] = num(r'First error:\s+(\S+)')
    out['lastError'] = num(r'Last error:\s+(\S+)')
    # the head of the report up to the camera section, for the line-by-line check
    out['head'] = txt[:txt.index('      Cameras:')].splitlines()
    return out


def failures():
    import lzma
    exp = {}
    for kind in ('1ray', 'missing-obs', 'no-datum'):
        if kind != 'no-datum':
            with open(os.path.join(REF, 'pmexports/camcal-pmexport-%s.txt' % kind), 'rb') as fi, \
                    lzma.open(os.path.join(HERE, 'camcal-pmexport-%s.txt.xz' % kind), 'wb', preset=9) as fo:
                fo.write(fi.read())
        exp[kind] = parse_failure(os.path.join(REF, 'dbatexports/camcal-dbatreport-%s.txt' % kind))
    with open(os.path.join(HERE, 'camcal_failures_expected.json'), 'w') as fh:
        json.dump(exp, fh, indent=1)
    print({k: (v['code'], v['structural'] and v['structural']['rank'], v['numerical'], v['sigma0']) for k, v in exp.items()})


def prague():
    """demo/prague2016_pm.m: PhotoModeler projects of the camcal sheet ('cam': c1
    fixed / c2 weighted control points) and of the Strasbourg block ('sxb': s1
    fixed, s2 weighted control points only, s3 plus one object point, s4 plus
    365 smart points), each with its loaded (fixed) camera.  Inputs: the
    PhotoModeler exports (xz) and the control point files; known answers: the
    committed DBAT reports, kept whole.  (The -with-orient variants have
    identical exports and reports up to the bookkeeping lines.)"""
    import lzma
    P = '/root/reference/data/prague2016'
    for site, stubs in (('cam', ('fixed', 'weighted')), ('sxb', ('f-op0', 'w-op0', 'w-op1', 'wsmart'))):
        for stub in stubs:
            with open(os.path.join(P, site, 'pmexports', stub + '-no-orient-pmexport.txt'), 'rb') as fi, \
                    lzma.open(os.path.join(HERE, 'prague-%s-%s-pmexport.txt.xz' % (site, stub)), 'wb', preset=9) as fo:
                fo.write(fi.read())
            dst = os.path.join(HERE, 'prague-%s-%s-dbatreport.txt' % (site, stub))
            shutil.copy(os.path.join(P, site, 'dbatexports', stub + '-no-orient-dbatreport.txt'), dst)
            os.chmod(dst, 0o644)
        for nm in ('ctrlpts-fixed.txt', 'ctrlpts-weighted.txt'):
            dst = os.path.join(HERE, 'prague-%s-%s' % (site, nm))
            shutil.copy(os.path.join(P, site, 'ref', nm), dst)
            os.chmod(dst, 0o644)
    # demo/sxb_prior_eo.m: the wsmart project in the original 1e6-m frame, with and without
    # prior observations of four camera positions
    with open(os.path.join(P, 'sxb/pmexports/wsmart-with-orient-pmexport.txt'), 'rb') as fi, \
            lzma.open(os.path.join(HERE, 'prague-sxb-wsmart-with-orient-pmexport.txt.xz'), 'wb', preset=9) as fo:
        fo.write(fi.read())
    for src, dst in (('sxb/ref/fake-camera-positions.txt', 'prague-sxb-fake-camera-positions.txt'),
                     ('sxb/dbatexports/sxb-prior-eo-dbatreport.txt', 'prague-sxb-prior-eo-dbatreport.txt'),
                     ('sxb/dbatexports/sxb-no-prior-eo-dbatreport.txt', 'prague-sxb-no-prior-eo-dbatreport.txt')):
        shutil.copy(os.path.join(P, src), os.path.join(HERE, dst))
        os.chmod(os.path.join(HERE, dst), 0o644)
    print('prague fixtures copied')


def roma_variants():
    """Known answers of demo/romabundledemo{,_selfcal,_imagevariant}.m from their
    committed reports (data/dbat/dbatexports/roma-dbatreport*.txt).  The demos
    load roma-pmexport.txt, which the reference does not ship; the script
    fixtures above hold the same 90 561 image points, and the PhotoModeler
    camera is printed in the fixed-camera report.  Only datum-independent
    values are kept (sigma0, counts, camera values and deviations): the demos
    fix the datum on camera 1 of the export, whose EO is not available."""
    exp = {}
    for variant, fn in (('fixed', 'roma-dbatreport.txt'), ('selfcal', 'roma-dbatreport-selfcal.txt'),
                        ('imagevariant', 'roma-dbatreport-imagevariant.txt')):
        rep = parse_report(os.path.join(REF, 'dbatexports', fn))
        exp[variant] = {k: rep[k] for k in ('sigma0', 'redundancy', 'numParams', 'nIO', 'nEO', 'nOP', 'numObs',
                                             'IO_report', 'IO_deviation')}
    with open(os.path.join(HERE, 'roma_variants_expected.json'), 'w') as fh:
        json.dump(exp, fh, indent=1)
    print('roma variants', {k: (v['sigma0'], v['numParams']) for k, v in exp.items()})


def sxb():
    """Script inputs of data/script/sxb (5 aerial images, 16 control/check
    points with prior standard deviations, 381 OP, project coordinates of
    1e6 m, two image-point files with different standard deviations) and its
    committed result file."""
    R = '/root/reference/data/script/sxb'
    for src, dst in (('measurements/markpts.txt', 'sxb-markpts.txt'), ('measurements/smartpts.txt', 'sxb-smartpts.txt'),
                     ('reference/sxb-control.txt', 'sxb-control.txt'), ('result/report.txt', 'sxb-report.txt')):
        shutil.copy(os.path.join(R, src), os.path.join(HERE, dst))
        os.chmod(os.path.join(HERE, dst), 0o644)
    rep = parse_report(os.path.join(R, 'result/report.txt'))
    txt = open(os.path.join(R, 'result/report.txt')).read()
    rep['firstError'] = float(re.search(r'First error:\s+([-\d.eE+]+)', txt).group(1))
    rep['sigma0_px'] = float(re.search(r'Sigma0 \(pixels\):\s+([-\d.eE+]+)', txt).group(1))
    xml = open(os.path.join(R, 'sxb.xml')).read()
    tag = lambda t_: re.search(r'<%s>([^<]*)</%s>' % (t_, t_), xml).group(1)
    nums = lambda v: [float(x) for x in v.split(',')]
    exp = {'camera': {'sensor': nums(tag('sensor')), 'image': nums(tag('image')), 'cc': float(tag('cc')),
                      'pp': nums(tag('pp')), 'K': nums(tag('K')), 'P': nums(tag('P')), 'model': int(tag('model'))},
           'check_ids': [351, 410], 'sxy': {'markpts': 0.5, 'smartpts': 1.0},
           'images': [int(l.split(',')[0]) for l in open(os.path.join(R, 'images/images.txt')) if l[0] != '#'],
           'image_paths': [l.split(',')[1].strip() for l in open(os.path.join(R, 'images/images.txt')) if l[0] != '#'],
           'report': rep}
    with open(os.path.join(HERE, 'sxb_expected.json'), 'w') as fh:
        json.dump(exp, fh, indent=1)
    print('sxb', rep['sigma0'], rep['numParams'], rep['numObs'], rep['iterations'], rep['firstError'], rep['lastError'])


def roma():
    """Script inputs of data/script/romabundledemo (60 images, 26 321 OP,
    90 561 image points) and the known answers of its committed result."""
    import lzma
    R = '/root/reference/data/script/romabundledemo'
    with open(os.path.join(R, 'measurements/markpts.txt'), 'rb') as fi, \
            lzma.open(os.path.join(HERE, 'roma-markpts.txt.xz'), 'wb', preset=9) as fo:
        fo.write(fi.read())
    shutil.copy(os.path.join(R, 'prior/initial_eo.txt'), os.path.join(HERE, 'roma-initial_eo.txt'))
    cam_in = open(os.path.join(R, 'cameras/EOS5DMarkII.xml')).read()
    cam_out = open(os.path.join(R, 'result/EOS5DMarkII.xml')).read()
    tag = lambda txt, t: re.search(r'<%s>([^<]*)</%s>' % (t, t), txt).group(1)
    nums = lambda v: [float(x) for x in v.split(',')]
    rep = parse_report(os.path.join(R, 'result/report.txt'))
    rep['firstError'] = float(re.search(r'First error:\s+([-\d.eE+]+)',
                                        open(os.path.join(R, 'result/report.txt')).read()).group(1))
    exp = {
        'camera_in': {'image': nums(tag(cam_in, 'image')), 'sensor_height': 24.0,
                      'cc': float(tag(cam_in, 'cc')), 'pp': nums(tag(cam_in, 'pp')),
                      'K': nums(tag(cam_in, 'K')), 'P': nums(tag(cam_in, 'P')),
                      'model': int(tag(cam_in, 'model'))},
        'camera_out': {'cc': float(tag(cam_out, 'cc')), 'pp': nums(tag(cam_out, 'pp')),
                       'K': nums(tag(cam_out, 'K')), 'sensor': nums(tag(cam_out, 'sensor'))},
        'report': rep,
        'image_paths': [l.split(',')[1].strip() for l in open(os.path.join(R, 'images/images.txt')) if l.strip() and l[0] != '#'],
    }
    # the committed result file whole, for the line-by-line comparison
    shutil.copy(os.path.join(R, 'result/report.txt'), os.path.join(HERE, 'roma-report.txt'))
    os.chmod(os.path.join(HERE, 'roma-report.txt'), 0o644)
    with open(os.path.join(HERE, 'roma_expected.json'), 'w') as fh:
        json.dump(exp, fh, indent=1)
    print('roma', rep['sigma0'], rep['numParams'], rep['iterations'], rep['firstError'], rep['lastError'])


if __name__ == '__main__':
    main()
