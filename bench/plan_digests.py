#!/usr/bin/env python
"""Digests of the host plan over a battery of scenes (host only, no GPU): every layout path of csrc/plan.hpp --
fixed IO / self-calibration / image-variant IO / priors / shared EO, heavy and giant points, small batches, several
ranks (domain sharding and the replicated fall-back).  Two builds of the library -- or the same build with different
DBAT_HIP_PLAN_THREADS -- must print identical files:

    DBAT_AMD_LIB=/tmp/libdbat_old.so python bench/plan_digests.py /tmp/old.json [--big]
    DBAT_HIP_PLAN_THREADS=3 python bench/plan_digests.py /tmp/new3.json [--big]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
from dbat_amd import _hip, synth  # noqa: E402
import helpers as H  # noqa: E402


def scenes(big):
    for v in ('plain', 'selfcal', 'imagevar', 'priors', 'groups4'):
        yield 'tiny-' + v, H.synth_struct('tiny', v)[0], {}
        yield 'small-' + v, H.synth_struct('small', v)[0], {}
    yield 'camcal3', H.camcal_struct(3), {}
    yield 'camcal-demo', H.camcal_demo_struct(3), {}
    yield 'roma', H.roma_demo_struct('fixed'), {}
    yield 'roma-selfcal', H.roma_demo_struct('selfcal'), {}
    yield 'roma-imagevariant', H.roma_demo_struct('imagevariant'), {}
    yield 'sxb', H.sxb_struct(), {}
    s = synth.make_scene('small', cams=140, points=500, rays=6)[0]
    yield 'small-heavy-cmax4', s, {'DBAT_HIP_CMAX': '4'}
    yield 'small-bt128', s, {'DBAT_HIP_BT': '128'}
    yield 'small-untiled', s, {'DBAT_HIP_CMAX': '0'}
    yield 'small-sig2', H.synth_struct('small', 'selfcal')[0], {'DBAT_HIP_SIG': '2'}
    yield 'small-sig0', H.synth_struct('small', 'plain')[0], {'DBAT_HIP_SIG': '0'}
    s = synth.make_scene('small', rays=12)[0]
    yield 'small-rays12', s, {}
    yield 'C1', synth.make_scene('C1')[0], {}
    yield 'C1-selfcal', synth.make_scene('C1', selfcal=True)[0], {}
    if big:
        yield 'C2', synth.make_scene('C2')[0], {}
        yield 'C3', synth.make_scene('C3')[0], {}


def main():
    out_path = sys.argv[1]
    big = '--big' in sys.argv
    res = {}
    for name, s, env in scenes(big):
        for k, v in env.items():
            os.environ[k] = v
        try:
            shards = [(0, 1)]
            if name in ('small-plain', 'small-selfcal', 'small-groups4', 'C1', 'roma', 'small-priors', 'C3'):
                shards += [(0, 2), (1, 2), (2, 3), (0, 4), (3, 4), (5, 8)]
            for r, n in shards:
                res['%s@%d/%d' % (name, r, n)] = _hip.plan_digest(s, r, n)
            if name in ('small-plain', 'C1'):
                os.environ['DBAT_HIP_MG_REPLICATED'] = '1'
                res['%s@1/2-replicated' % name] = _hip.plan_digest(s, 1, 2)
                del os.environ['DBAT_HIP_MG_REPLICATED']
        finally:
            for k in env:
                del os.environ[k]
        print(name, 'ok', flush=True)
    json.dump(res, open(out_path, 'w'), indent=0, sort_keys=True)


if __name__ == '__main__':
    main()
