"""The C++/OpenMP CPU baseline (bench/cpu_ref.cpp) against the oracle: same
unknown ordering, same residual, and the same damped step
p = (J'J + lambda I) \\ (-J'r) as levenberg_marquardt.m:81-82,119 -- so that the
`cpu_baseline` number of bench.py is the time of a correct computation."""
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

import dbat_oracle as o
from helpers import synth_struct, relerr

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'bench'))


@pytest.fixture(scope='module')
def cpu_ref():
    import subprocess
    bench = os.path.join(os.path.dirname(__file__), '..', 'bench')
    subprocess.run(['make', '-C', bench], check=True, capture_output=True)
    import cpu_ref as m
    m.load()
    return m


@pytest.mark.parametrize('name,variant', [('tiny', 'plain'), ('tiny', 'selfcal'), ('tiny', 'imagevar'),
                                          ('tiny', 'groups4'), ('small', 'plain'), ('small', 'groups4'),
                                          ('tiny', 'priors'), ('small', 'priors')])   # prior observations: rows of J (prior_obs.m:45-72)
@pytest.mark.parametrize('threads', [1, 3])
def test_cpu_ref_step_matches_oracle(cpu_ref, name, variant, threads):
    s, _ = synth_struct(name, variant)
    if variant == 'priors':                       # (bundle.m:137-154: priors of parameters that are not estimated are dropped)
        import copy
        s = copy.deepcopy(s)
        for nm in ('IO', 'EO', 'OP'):
            pr = getattr(s.prior, nm)
            pr.use = np.asarray(pr.use, bool) & np.asarray(getattr(s.bundle.est, nm), bool)
    so = o.buildserialindices(s)
    x0 = o.serialize(so)
    w = o.buildweightvector(so)
    R = np.sqrt(w)
    r_, K = o.brown_euler_cam4(x0, so, jac=True)
    r = R * r_
    J = (sp.diags(R) @ K).tocsc()
    JTJ = (J.T @ J).tocsc()
    c = cpu_ref.CpuRef(s, threads=threads)
    try:
        assert c.n == len(x0) and c.threads == threads
        assert np.array_equal(c.serialize(), x0)
        assert c.nnz['J'] == J.nnz
        lam = 1e-4 * JTJ.diagonal().sum() / J.shape[1]
        p_o, _ = o.normal_solve((JTJ + lam * sp.identity(J.shape[1])).tocsc(), -(J.T @ r))
        p, st = c.lm_step(x0, lam)
        assert st['code'] == 0
        assert abs(st['f'] - 0.5 * r @ r) <= 1e-12 * st['f']
        assert abs(st['trace'] - JTJ.diagonal().sum()) <= 1e-12 * st['trace']
        assert relerr(p, p_o) < 1e-8
        rt = R * o.brown_euler_cam4(x0 + p, so)
        assert abs(st['f_trial'] - 0.5 * rt @ rt) <= 1e-10 * st['f_trial']
        # lambda given as a fraction of trace/n (levenberg_marquardt.m:88-90)
        p2, st2 = c.lm_step(x0, -1e-4)
        assert abs(st2['lam'] - lam) <= 1e-12 * lam and relerr(p2, p) < 1e-10
    finally:
        c.close()
