"""Randomised sweep of the domain-sharded path, outside the test-suite: scenes of random shape (cameras,
points, rays per point, fixed IO / self-calibration with 1, 2 or 4 IO blocks, camera grids and random
clouds), 2 ... 8 ranks as threads on ONE GPU (sums through the host), signature kernels forced on and
off -- every rank's Gauss-Newton and damped step, step scalars and a complete LM solve against the
one-rank result.  Prints one line per scene; exits non-zero on the first disagreement.
    python bench/fuzz_multishard.py [n_scenes] [first_seed]"""
import os, sys, threading
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, 'oracle')); sys.path.insert(0, os.path.join(R_, 'tests')); sys.path.insert(0, os.path.join(R_, 'bench'))
import numpy as np
from dbat_amd import synth, _hip
from test_hip_parity import _ThreadComm

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 24
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
relerr = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def run_ranks(world, fn):
    shared = {'buf': [None] * world, 'bar': threading.Barrier(world)}
    out, err = [None] * world, []

    def run(rank):
        try:
            out[rank] = fn(_ThreadComm(rank, world, shared))
        except Exception:   # noqa: BLE001
            import traceback
            err.append(traceback.format_exc())
            shared['bar'].abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(900) for t in th]
    if err:
        raise RuntimeError(err[0])
    return out


def main():
    worst = 0.0
    nbad = 0
    for sd in range(seed0, seed0 + n_scenes):
        rng = np.random.default_rng(7000 + sd)
        cams = int(rng.integers(30, 400))
        rays = int(rng.integers(3, 12))
        points = int(rng.integers(500, 12000))
        selfcal = bool(rng.integers(0, 2))
        groups = int(rng.choice([1, 1, 2, 4])) if selfcal else 1
        world = int(rng.choice([2, 2, 3, 4, 5, 8]))
        sig = str(rng.choice(['0', '2', '']))
        if sig: os.environ['DBAT_HIP_SIG'] = sig
        else: os.environ.pop('DBAT_HIP_SIG', None)
        base = 'C1' if rng.integers(0, 2) else 'small'
        if os.environ.get('FUZZ_IRREGULAR'):          # bench/fuzz_irregular.py's family: thinned visibility, control points in many images, forced kernel paths
            from fuzz_irregular import irregular_scene, KNOBS
            s, env, desc = irregular_scene(sd)
            for k in KNOBS: os.environ.pop(k, None)
            os.environ.update(env)
            base, cams, points, rays = 'irr', s.EO.val.shape[1], s.OP.val.shape[1], 0
            sig = env.get('DBAT_HIP_SIG', '')
        else:
            s, _ = synth.make_scene(base, seed=2000 + sd, cams=cams, points=points, rays=rays, selfcal=selfcal, groups=groups)
        cam_owner, subtree = _hip.plan_domain_map(s, world)
        h = _hip.Handle(s)
        try:
            x0 = h.serialize()
            p1, st1 = h.linearize_solve(x0, 0.0, True)
            lam = 1e-4 * st1['trace'] / h.n
            q1, st2 = h.linearize_solve(x0, lam, False)
            opt = _hip.default_options('lm'); opt.store_trace = 0
            xs1, res1, *_ = h.solve(x0, opt)
        finally:
            h.close()

        def work(comm):
            hh = _hip.Handle(s, shard_rank=comm.rank, shard_count=comm.world_size)
            try:
                hh.set_allreduce(comm.allreduce_ptr)
                p, st = hh.linearize_solve(x0, 0.0, True)
                q, stq = hh.linearize_solve(x0, lam, False)
                xs, res, *_ = hh.solve(x0, opt)
                return p, st, q, stq, xs, res.code, res.iters, res.sigma0
            finally:
                hh.close()

        out = run_ranks(world, work)
        line = 'seed %3d: %s %3d cams %5d pts %2d rays selfcal=%d groups=%d sig=%-1s | %d ranks, %3d top cams, sharded=%d |' % (
            sd, base, cams, points, rays, selfcal, groups, sig, world, int(np.count_nonzero(np.asarray(cam_owner) < 0)), subtree)
        bad = False
        e = [0.0, 0.0, 0.0]
        for p, st, q, stq, xs, code, iters, s0 in out:
            ep, eq = relerr(p, p1), relerr(q, q1)
            e[0] = max(e[0], ep); e[1] = max(e[1], eq); e[2] = max(e[2], relerr(xs, xs1))
            for k in ('f', 'JpJp', 'rJp', 'pp', 'trace'):
                if res1.code == -4 and k != 'f': continue       # structurally singular (sprank(J) < n): the solve refuses, the steps mean nothing
                # the scalars of a step are quadratic in it: on a weak network (an undamped step that the two summation
                # orders move by 1e-7) they follow the step, not the 1e-8 of a well-conditioned one
                tp = 1e-8 if k in ('f', 'trace') else max(1e-8, 4 * ep)
                tq = 1e-8 if k in ('f', 'trace') else max(1e-8, 4 * eq)
                if not abs(st[k] - st1[k]) <= tp * abs(st1[k]) or not abs(stq[k] - st2[k]) <= tq * abs(st2[k]): bad = True
            if code != res1.code or (st['singular'] != st1['singular']): bad = True
            if code == 0 and abs(s0 - res1.sigma0) > 1e-7 * res1.sigma0: bad = True
        line += ' GN %.1e LM %.1e solve %.1e (code %d, %d/%d its)' % (e[0], e[1], e[2], out[0][5], out[0][6], res1.iters)
        worst = max(worst, e[0], e[1])
        if bad or not (e[0] < 1e-6 and e[1] < 1e-6) or (res1.code == 0 and e[2] > 1e-6):
            print(line, ' <-- DISAGREES', flush=True); nbad += 1
            if not os.environ.get('FUZZ_KEEP_GOING'): sys.exit(1)
            continue
        print(line, flush=True)
    print('worst relative difference of a step over %d scenes: %.2e; %d disagreements' % (n_scenes, worst, nbad))
    sys.exit(1 if nbad else 0)


if __name__ == '__main__':
    main()
