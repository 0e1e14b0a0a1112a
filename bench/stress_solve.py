"""Many complete solves in a row (all damping loops, fixed IO and self-calibration): every one must
end with code 0 after the same number of iterations, in about the same time -- a lost flag or a
stale mailbox value in the launch-fused loops shows up as a different count, an abort or a stall."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dbat_amd import synth, _hip
for name, kw, reps in (('C1', {}, 60), ('C1', dict(selfcal=True), 40), ('C2', {}, 15), ('C3', {}, 6)):
    s, _ = synth.make_scene(name, **kw)
    for damping in ('gna', 'lm', 'lmp'):
        h = _hip.Handle(s)
        opt = _hip.default_options(damping)
        x0 = h.serialize()
        its, times, xs = [], [], []
        for i in range(reps if damping == 'lm' else max(reps // 4, 2)):
            t0 = time.perf_counter()
            x, res, rr, damp, aux, T = h.solve(x0, opt)
            times.append(time.perf_counter() - t0)
            assert res.code == 0, (name, damping, i, res.code)
            its.append(res.iters); xs.append(x)
        spread = max(np.linalg.norm(x - xs[0]) / np.linalg.norm(xs[0]) for x in xs)
        print('%s %s %-3s: %d solves, iterations %s, time median %.2f ms max %.2f ms, x spread %.1e'
              % (name, kw, damping, len(its), sorted(set(its)), 1e3 * np.median(times), 1e3 * max(times), spread), flush=True)
        assert len(set(its)) == 1 or damping != 'gna'
        h.close()
