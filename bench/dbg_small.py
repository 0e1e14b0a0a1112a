import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np
import dbat_oracle as o
from helpers import synth_struct, relerr
from dbat_amd import bundle
s, truth = synth_struct('small', 'plain')
for damping in ('gna', 'lm', 'lmp'):
    res, ok, iters, s0, E = bundle(s, damping, store_trace=True) if False else bundle(s, damping)
    ro, oko, ito, s0o, Eo = o.bundle(s, damping)
    print(damping, ok, oko, iters, ito, relerr(E.x, Eo.x), s0, s0o)
    for nm in ('res', 'lambda_'):
        if hasattr(E, nm): print('  hip', nm, np.array(getattr(E, nm))[:10])
        if hasattr(Eo, nm): print('  ora', nm, np.array(getattr(Eo, nm))[:10])
