"""The reference's own workloads as bench scenes: the only timings DBAT publishes are the
'Execution times -> Bundle' lines of its committed demo reports.  Built from the fixtures under tests/golden with the
product's loaders and the device's initial-value stage only (no oracle import here):

  roma          demo/romabundledemo.m:56-88         60 images / 26 321 OP / 90 561 image points, fixed IO
                data/dbat/dbatexports/roma-dbatreport.txt:23-24,39-45      GNA, 5 iterations, 5.21 s
  roma-selfcal  demo/romabundledemo_selfcal.m       the same, 9 IO unknowns (no skew)
                data/dbat/dbatexports/roma-dbatreport-selfcal.txt:46       GNA, 5 iterations, 11.81 s
  camcal        demo/camcaldemo.m:56-119            21 images / 100 OP (96 free) / 2 074 image points, 9 IO unknowns;
                EO by 3-point resection, OP by forward intersection (on the device: dbat_amd.initial)
                data/dbat/dbatexports/camcal-dbatreport.txt:23-24,39-45    GNA, 9 iterations, 1.17 s

  sxb           data/script/sxb/sxb.xml             5 aerial images / 2 434 observations, fixed camera, 14 control points as
                weighted PRIOR OBSERVATIONS (42 rows of J below the image rows, lsa/prior_obs.m:45-72), project
                coordinates of 1e6 m; data/script/sxb/result/report.txt:20,41-45   GNA, 4 iterations, 0.07 s

(host 'slartibartfast', MATLAB R2020a, CPU model and core count not recorded: SURVEY 6.)
"""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')

PUBLISHED = {
    'roma': {'damping': 'gna', 'iterations': 5, 'bundle_s': 5.21, 'sigma0': 0.623075,
             'source': 'data/dbat/dbatexports/roma-dbatreport.txt:23-24,39-45 (demo/romabundledemo.m:88)'},
    'roma-selfcal': {'damping': 'gna', 'iterations': 5, 'bundle_s': 11.81, 'sigma0': 0.566548,
                     'source': 'data/dbat/dbatexports/roma-dbatreport-selfcal.txt:46 (demo/romabundledemo_selfcal.m)'},
    'camcal': {'damping': 'gna', 'iterations': 9, 'bundle_s': 1.17, 'sigma0': 1.6148,
               'source': 'data/dbat/dbatexports/camcal-dbatreport.txt:23-24,39-45 (demo/camcaldemo.m:119)'},
    'sxb': {'damping': 'gna', 'iterations': 4, 'bundle_s': 0.07, 'sigma0': 1.1786,
            'source': 'data/script/sxb/result/report.txt:20,41-45 (script/parseops.m:36-59 on data/script/sxb/sxb.xml)'},
}


def names():
    return tuple(PUBLISHED)


def _roma(selfcal):
    from dbat_amd import loadtables as T
    from dbat_amd.dbatstruct import seteoest_depend
    with open(os.path.join(GOLDEN, 'roma_variants_expected.json')) as fh:
        cam = json.load(fh)['fixed']['IO_report']          # PhotoModeler's camera as roma-dbatreport.txt:58-92 prints it
    eo = T.load_table(os.path.join(GOLDEN, 'roma-initial_eo.txt'))
    mk = T.load_table(os.path.join(GOLDEN, 'roma-markpts.txt.xz'))
    io = T.camera_io(cam['cc'], (cam['px'], cam['py']), [cam['K1'], cam['K2'], cam['K3']], [cam['P1'], cam['P2']],
                     aspect=1.0 - cam['as'])
    s = T.struct_from_tables(io, (36.0, 24.0), (5616, 3744), eo, mk, 'im,id,x,y', 1.0, distModel=3)
    s = T.forwintersect(s)                                  # romabundledemo.m:70 (dbat_hip_forwintersect)
    if selfcal:
        s.bundle.est.IO[:] = True                           # setcamest 'all','not','sk'
        s.bundle.est.IO[4] = False
    return seteoest_depend(s, 0)                            # datum by dependency on camera 1


def _camcal():
    from dbat_amd import initial as I
    from dbat_amd import loadpm as L
    prob = L.loadpm(os.path.join(GOLDEN, 'camcal-pmexport.txt'))
    s = L.prob2dbatstruct(prob, distModel=3)
    s.IO.val[0, :] = 7.3                                    # setcamvals 'default',7.3  (camcaldemo.m:56-100)
    s.IO.val[1:3, :] = 0.5 * np.diag([1, -1]) @ s.IO.sensor.ssSize
    s.IO.val[3:, :] = 0
    s.bundle.est.IO[:] = True                               # setcamest 'all','not','sk'
    s.bundle.est.IO[4, :] = False
    s.bundle.est.EO[:] = True
    s.prior.OP.isCtrl = s.OP.id > 1000
    s = L.setcpt(s, L.loadcpt(os.path.join(GOLDEN, 'camcal-fixed.txt')))
    s = I.clearop(I.cleareo(s))
    cp = s.OP.id[s.prior.OP.isCtrl]
    s, rms, fail = I.resect(s, 'all', cp, 1, 0, cp)         # camcaldemo.m:103 (dbat_hip_resect)
    if fail:
        raise RuntimeError('camcal: resection failed')
    return I.forwintersect(s, 'all', True)                  # camcaldemo.m:107


def _sxb():
    """data/script/sxb/sxb.xml with the product's loaders: initial values by dbat_hip_resect / dbat_hip_forwintersect."""
    from dbat_amd import initial as I
    from dbat_amd import loadpm as L
    from dbat_amd import loadtables as T
    with open(os.path.join(GOLDEN, 'sxb_expected.json')) as fh:
        exp = json.load(fh)
    cam = exp['camera']
    pts = L.loadcpt(os.path.join(GOLDEN, 'sxb-control.txt'))

    def pick(keep):
        m = np.isin(pts['id'], exp['check_ids']) == keep
        return dict(id=pts['id'][m], name=[n for n, k in zip(pts['name'], m) if k], pos=pts['pos'][:, m], std=pts['std'][:, m])
    io = T.camera_io(cam['cc'], cam['pp'], cam['K'], cam['P'])
    marks = [(T.load_table(os.path.join(GOLDEN, 'sxb-%s.txt' % nm)), 'id,im,x,y', exp['sxy'][nm]) for nm in ('markpts', 'smartpts')]
    s = T.struct_from_script(io, cam['sensor'], cam['image'], exp['images'], marks, pick(False), pick(True),
                             distModel=cam['model'], im_names=exp['image_paths'])
    s = T.set_script_defaults(s)
    cp = s.OP.id[s.prior.OP.isCtrl]
    s, rms, fail = I.resect(s, 'all', cp, 1, 0, cp)         # spatial_resection (script/parseops.m:36-43)
    if fail:
        raise RuntimeError('sxb: resection failed')
    return I.forwintersect(s, 'all', True)


def make(name):
    """(DBAT struct, published record) of one of the reference's demo projects."""
    if name == 'roma':
        return _roma(False), PUBLISHED[name]
    if name == 'roma-selfcal':
        return _roma(True), PUBLISHED[name]
    if name == 'camcal':
        return _camcal(), PUBLISHED[name]
    if name == 'sxb':
        return _sxb(), PUBLISHED[name]
    raise KeyError(name)
