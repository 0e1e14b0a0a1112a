import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests')); sys.path.insert(0, os.path.join(R, 'oracle'))
import numpy as np
import test_hip_parity as T
from dbat_amd import _hip
for name in ['camcal3', 'tiny-selfcal', 'small-plain']:
    s = dict(T.cases())[name]()
    so, x0, w = T.oracle_setup(s)
    ref = None
    for chain, dbg in (('0', '0'), ('1', '0'), ('1', '1'), ('1', '2'), ('1', '6'), ('1', '8'), ('1', '9'), ('1', '15')):
        os.environ['DBAT_HIP_DF_CHAIN'] = chain
        os.environ['DBAT_HIP_DF_DBG'] = dbg
        h = _hip.Handle(s)
        bad = 0
        for rep in range(20):
            p, st = h.linearize_solve(x0, 0.0, True)
            if ref is None: ref = p
            e = np.linalg.norm(p - ref) / np.linalg.norm(ref)
            bad += not (e < 1e-9)
        print(name, 'chain', chain, 'dbg', dbg, 'bad', bad, 'of 20', flush=True)
        h.close()
