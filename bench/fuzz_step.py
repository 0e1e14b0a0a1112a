"""Randomised sweep, outside the test-suite: scenes of random shape (cameras, points, rays per point
3 ... 13, fixed IO / self-calibration with 1, 2 or 4 IO blocks, long and short signature groups) --
the device's Gauss-Newton and damped steps against the oracle's sparse solve, signature kernels forced
on and off.  Prints one line per scene; exits non-zero on the first disagreement.
    python bench/fuzz_step.py [n_scenes] [first_seed]"""
import os, sys
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, 'oracle')); sys.path.insert(0, os.path.join(R_, 'tests'))
import numpy as np, scipy.sparse as sp
import dbat_oracle as o
from dbat_amd import synth, _hip
from test_hip_parity import oracle_setup

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 24
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
relerr = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
worst = 0.0
for sd in range(seed0, seed0 + n_scenes):
    rng = np.random.default_rng(sd)
    cams = int(rng.integers(24, 140))
    rays = int(rng.integers(3, 14))               # (two rays per point: barely determined networks, singular now and then)
    points = int(rng.integers(300, 6000))
    selfcal = bool(rng.integers(0, 2))
    groups = int(rng.choice([1, 1, 2, 4])) if selfcal else 1
    if selfcal and rays > 11: rays = 11            # six rows per camera + IO rows + right-hand side <= 80
    s, _ = synth.make_scene('small', seed=1000 + sd, cams=cams, points=points, rays=rays, selfcal=selfcal, groups=groups)
    so, x0, w = oracle_setup(s)
    Rw = np.sqrt(w)
    r_o, K = o.brown_euler_cam4(x0, so, jac=True)
    J = (sp.diags(Rw) @ K).tocsc()
    p_o, *_ = o._scaled_gn(J, Rw * r_o)
    JTJ = (J.T @ J).tocsc()
    lam = 1e-4 * JTJ.diagonal().sum() / J.shape[1]
    q_o, _ = o.normal_solve((JTJ + lam * sp.identity(J.shape[1])).tocsc(), -(J.T @ (Rw * r_o)))
    line = 'seed %3d: %3d cams %5d pts %2d rays selfcal=%d groups=%d |' % (sd, cams, points, rays, selfcal, groups)
    for sig in ('0', '2'):
        os.environ['DBAT_HIP_SIG'] = sig
        h = _hip.Handle(s)
        try:
            p_h, st = h.linearize_solve(x0, 0.0, True)
            q_h, _ = h.linearize_solve(x0, lam, False)
            e1, e2 = relerr(p_h, p_o), relerr(q_h, q_o)
            line += ' %s: GN %.1e LM %.1e' % (h.build_kernel_name(), e1, e2)
            worst = max(worst, e1, e2)
            if not (e1 < 1e-6 and e2 < 1e-6) or st['singular']:
                print(line, ' <-- DISAGREES', flush=True); sys.exit(1)
        finally:
            h.close()
    print(line, flush=True)
print('worst relative difference of a step over %d scenes: %.2e' % (n_scenes, worst))
