"""Time of the posterior covariance (CEO, COP) at C3 on one GPU."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dbat_amd import synth, _hip
s, _ = synth.make_scene(sys.argv[1] if len(sys.argv) > 1 else 'C3')
h = _hip.Handle(s)
x = h.serialize()
for i in range(3):
    t0 = time.perf_counter()
    CEO, CIO, COP = h.posterior_cov(x, 1.0)
    print('posterior_cov: %.3f s' % (time.perf_counter() - t0), 'mean point std', np.sqrt(np.trace(COP, axis1=1, axis2=2)).mean(),
          'min eig>0', bool(np.all(np.linalg.eigvalsh(COP[::1000]) > 0)))
