"""Randomised sweep of the deterministic mode (outside the test-suite): scenes of random shape as bench/fuzz_step.py draws
them, signature kernels forced on / off / automatic, thinned visibility now and then -- five linearise + solve steps with
dbat_hip_set_deterministic repeat BIT FOR BIT, and the step stays within 1e-6 of the default mode's (1e-8 for the small
systems drawn here is the rule; the bound printed is the worst seen).
    python bench/fuzz_det.py [n_scenes] [first_seed]"""
import os, sys
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_)
import numpy as np
from dbat_amd import synth, _hip

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 24
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
worst = 0.0
for sd in range(seed0, seed0 + n_scenes):
    rng = np.random.default_rng(sd)
    cams = int(rng.integers(24, 140)); rays = int(rng.integers(3, 12)); points = int(rng.integers(300, 6000))
    selfcal = bool(rng.integers(0, 2)); groups = int(rng.choice([1, 1, 2, 4])) if selfcal else 1
    s, _ = synth.make_scene('small', seed=3000 + sd, cams=cams, points=points, rays=rays, selfcal=selfcal, groups=groups)
    sig = str(rng.choice(['0', '2', '']))
    if sig: os.environ['DBAT_HIP_SIG'] = sig
    else: os.environ.pop('DBAT_HIP_SIG', None)
    if os.environ.get('FUZZ_IRREGULAR'):     # bench/fuzz_irregular.py's family: thinned visibility, heavy / giant points (csrc/heavy.hpp), forced kernel paths
        sys.path.insert(0, os.path.join(R_, 'oracle')); sys.path.insert(0, os.path.join(R_, 'tests')); sys.path.insert(0, os.path.join(R_, 'bench'))
        from fuzz_irregular import irregular_scene, KNOBS
        s, env, desc = irregular_scene(sd)
        for k in KNOBS: os.environ.pop(k, None)
        env.pop('DBAT_HIP_BT', None)         # (128-observation batches: no tiles, the deterministic column lists -- covered by the suite)
        os.environ.update(env)
        cams, points, rays, sig = s.EO.val.shape[1], s.OP.val.shape[1], 0, env.get('DBAT_HIP_SIG', '') + ' ' + ' '.join('%s=%s' % (k[9:], v) for k, v in env.items())
    lam = float(rng.choice([0.0, 1e-3]))
    h = _hip.Handle(s)
    try:
        x0 = h.serialize()
        p_def, st = h.linearize_solve(x0, lam, True)
        h.set_deterministic(True)
        ref = None; bad = 0
        for i in range(5):
            p, st2 = h.linearize_solve(x0, lam, True)
            if ref is None: ref = p.copy()
            bad += not np.array_equal(p, ref, equal_nan=True)     # (a singular scene: NaN steps, in every repeat)
        e = float(np.linalg.norm(ref - p_def) / np.linalg.norm(p_def))
        worst = max(worst, e)
        print('seed %d: %d cams %d pts %d rays selfcal %d groups %d SIG=%s lambda %g: %d of 5 repeats differ, vs default %.1e%s'
              % (sd, cams, points, rays, selfcal, groups, sig or 'auto', lam, bad, e, ' singular' if st['singular'] else ''), flush=True)
        if bad or not (e < 1e-6 or st['singular']):
            print('DISAGREEMENT'); sys.exit(1)
    finally:
        h.close()
print('deterministic sweep: %d scenes, worst distance from the default mode %.1e' % (n_scenes, worst))
