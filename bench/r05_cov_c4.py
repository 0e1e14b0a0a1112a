"""Posterior covariance at C3 / C4: time and device memory of the selected inversion against the dense inverse."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import torch
from dbat_amd import _hip, synth
for name in sys.argv[1:]:
    s, _ = synth.make_scene(name)
    h = _hip.Handle(s)
    x0 = h.serialize()
    opt = _hip.default_options('gna'); opt.store_trace = 0
    x, res, rr, damp, aux, T = h.solve(x0, opt)
    out = {}
    for mode in ('selected', 'dense'):
        if mode == 'dense': os.environ['DBAT_HIP_COV_DENSE'] = '1'
        else: os.environ.pop('DBAT_HIP_COV_DENSE', None)
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        t0 = time.perf_counter(); a = h.posterior_cov(x, float(res.sigma0)); t1 = time.perf_counter() - t0
        free1 = torch.cuda.mem_get_info()[0]
        t0 = time.perf_counter(); a = h.posterior_cov(x, float(res.sigma0)); t2 = time.perf_counter() - t0
        out[mode] = (t1, t2, (free0 - free1) / 1e6, a)
        print('%s %-9s first call %.1f ms, second %.1f ms, device memory kept by the handle after the first call +%.0f MB' % (name, mode, t1 * 1e3, t2 * 1e3, (free0 - free1) / 1e6), flush=True)
    A, B = out['selected'][3], out['dense'][3]
    print('%s   blocks agree to %.1e (CEO), %.1e (COP)' % (name, np.abs(A[0] - B[0]).max() / np.abs(B[0]).max(), np.abs(A[2] - B[2]).max() / np.abs(B[2]).max()))
    info = h.info()
    print('%s   reduced system NS = %d: dense S %.0f MB; compact factor tiles: %s' % (name, info['NS'], 8e-6 * info['NS'] ** 2, {k: v for k, v in h.chol_stats().items()}))
    h.close()
