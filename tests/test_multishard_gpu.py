"""Several ranks on ONE GPU (threads + an all-reduce through the host): the domain-sharded path --
csrc/nd.hpp, the two-phase factorisation of csrc/chol_df.hpp (own domain + shares of the top
separators, all-reduce of the top tiles, top separators + backward substitution) -- against the
one-rank result, for 2, 3 and 4 ranks, all dampings, fixed and self-calibrated IO (dense IO rows =
part of the top), prior observations, ranks whose domain is empty, the signature kernels forced on,
and the replicated fall-back.  (tests/test_parallel_cpu.py: the same scheme in NumPy over gloo.)"""
import threading

import numpy as np
import pytest

from helpers import relerr, synth_struct
from test_hip_parity import _ThreadComm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hip():
    from dbat_amd import _hip
    import torch
    assert torch.cuda.is_available(), 'gpu tests need a GPU'
    _hip.load()
    return _hip


class _CountingComm(_ThreadComm):
    """... and counts what travels: (count, bytes) per collective."""

    def __init__(self, rank, world, shared):
        super().__init__(rank, world, shared)
        self.sizes = []

    def allreduce_ptr(self, ptr, count, stream):
        self.sizes.append(int(count))
        return super().allreduce_ptr(ptr, count, stream)


def _run_ranks(s, world, fn):
    shared = {'buf': [None] * world, 'bar': threading.Barrier(world)}
    out, err = [None] * world, []
    comms = [_CountingComm(r, world, shared) for r in range(world)]

    def run(rank):
        try:
            out[rank] = fn(comms[rank])
        except Exception as e:   # noqa: BLE001
            import traceback
            err.append(traceback.format_exc())
            shared['bar'].abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(900) for t in th]
    assert not err, err[0]
    return out, comms


def _scene(kind):
    from dbat_amd import synth
    if kind == 'c1':                       # 100 cameras: two domains of 30 at world 2; at world 4 two of the four are empty
        return synth.make_scene('C1')[0]
    if kind == 'c1-selfcal':
        return synth.make_scene('C1', selfcal=True)[0]
    if kind == 'grid-selfcal4':            # 15 x 15 cameras, six rays: four non-empty domains; four IO blocks
        return synth.make_scene('C1', cams=225, points=6000, rays=6, selfcal=True, groups=4)[0]
    if kind == 'small-priors':             # 40 cameras, EO / OP prior observations, fixed points, non-uniform weights
        return synth_struct('small', 'priors')[0]
    raise ValueError(kind)


@pytest.mark.parametrize('world', [2, 3, 4])
@pytest.mark.parametrize('kind,damping', [('c1', 'gna'), ('c1', 'lm'), ('c1', 'lmp'), ('c1', 'gm'),
                                          ('c1-selfcal', 'gna'), ('c1-selfcal', 'lm'), ('grid-selfcal4', 'gna'),
                                          ('grid-selfcal4', 'lmp'), ('small-priors', 'gna'), ('small-priors', 'lm')])
def test_domain_shards_match_single(hip, kind, damping, world):
    from dbat_amd import bundle
    s = _scene(kind)
    cam_owner, subtree = hip.plan_domain_map(s, world)
    assert subtree
    ref = bundle(s, damping)
    out, comms = _run_ranks(s, world, lambda comm: bundle(s, damping, comm=comm))
    for rank in range(world):
        res, ok, iters, s0, E = out[rank]
        assert ok == ref[1] and E.code == ref[4].code
        assert relerr(E.x, ref[4].x) < 1e-8, (rank, relerr(E.x, ref[4].x))
        assert abs(s0 - ref[3]) < 1e-9 * ref[3]
        if damping != 'lm':
            assert iters == ref[2]
            assert relerr(E.res, ref[4].res) < 1e-9
        assert relerr(res.post.res.IP, ref[0].post.res.IP) < 1e-6
    # what travelled inside the factorisation: whole 64 x 64 tiles of the top separators, never the whole system
    # (the other large collectives are the gathers of results: z vectors of NZ entries, residual rows)
    nc, npnt, no = s.EO.val.shape[1], s.OP.val.shape[1], s.IP.val.shape[1]
    NS = 6 * nc + (len(np.unique(s.IO.struct.block[0])) * 8 if np.any(s.bundle.est.IO) else 0)
    results = {NS + 3 * npnt, 2 * no, 3 * npnt, 9 * npnt}
    tiles = [c for c in comms[0].sizes if c % 4096 == 0 and c not in results]
    ntop = int(np.count_nonzero(cam_owner < 0)) * 6 + (NS - 6 * nc)
    assert tiles and len(set(tiles)) == 1, sorted(set(comms[0].sizes))
    if np.count_nonzero(cam_owner >= 0) > 0 and ntop < 0.6 * NS:
        assert tiles[0] < 0.5 * NS * NS, (tiles[0], NS, ntop)
    assert 2 * NS + 8 in comms[0].sizes                         # [J_c'r | diag | sums] per linearisation


@pytest.mark.parametrize('world', [2, 4])
def test_domain_shards_step_scalars_and_covariance(hip, world):
    """linearize_solve on every rank: the complete step and the same scalars (f, ||J p||^2, g'p, ||p||^2,
    trace(J'J)) as one rank; gradient and column norms complete on every rank; the posterior covariance blocks
    (computed from the whole system summed over the ranks) as one rank's."""
    s = _scene('grid-selfcal4')
    h = hip.Handle(s)
    try:
        x0 = h.serialize()
        p1, st1 = h.linearize_solve(x0, 0.0, True)
        g1, n1 = h.gradient(), h.colnorms()
        q1, st2 = h.linearize_solve(x0, 1e-3 * st1['trace'] / h.n, False)
        C1 = h.posterior_cov(x0, 1.0)
    finally:
        h.close()

    def work(comm):
        hh = hip.Handle(s, shard_rank=comm.rank, shard_count=comm.world_size)
        try:
            hh.set_allreduce(comm.allreduce_ptr)
            p, st = hh.linearize_solve(x0, 0.0, True)
            g, n = hh.gradient(), hh.colnorms()
            q, stq = hh.linearize_solve(x0, 1e-3 * st1['trace'] / hh.n, False)
            C = hh.posterior_cov(x0, 1.0)
            return p, st, g, n, q, stq, C
        finally:
            hh.close()

    out, comms = _run_ranks(s, world, work)
    for p, st, g, n, q, stq, C in out:
        assert relerr(p, p1) < 1e-9 and relerr(q, q1) < 1e-9
        for k in ('f', 'JpJp', 'rJp', 'pp', 'trace'):
            assert abs(st[k] - st1[k]) <= 1e-9 * abs(st1[k]), k
            assert abs(stq[k] - st2[k]) <= 1e-9 * abs(st2[k]), k
        assert relerr(g, g1) < 1e-11 and relerr(n, n1) < 1e-11
        for a, b in zip(C, C1):
            assert relerr(a, b) < 1e-8


@pytest.mark.parametrize('seed', [5, 8, 12, 17, 26, 29])
def test_domain_shards_random_scenes(hip, seed, monkeypatch):
    """Scenes of random shape (cameras, points, rays, self-calibration with 1 ... 4 IO blocks, 2 ... 8 ranks, signature
    kernels on / off) from bench/fuzz_multishard.py's family -- these seeds are the ones a broken point-shard range
    once showed up on: every rank's Gauss-Newton and damped step and the step scalars as one rank's."""
    from dbat_amd import synth
    rng = np.random.default_rng(7000 + seed)
    cams, rays, points = int(rng.integers(30, 400)), int(rng.integers(3, 12)), int(rng.integers(500, 12000))
    selfcal = bool(rng.integers(0, 2))
    groups = int(rng.choice([1, 1, 2, 4])) if selfcal else 1
    world = int(rng.choice([2, 2, 3, 4, 5, 8]))
    sig = str(rng.choice(['0', '2', '']))
    if sig:
        monkeypatch.setenv('DBAT_HIP_SIG', sig)
    base = 'C1' if rng.integers(0, 2) else 'small'
    s = synth.make_scene(base, seed=2000 + seed, cams=cams, points=points, rays=rays, selfcal=selfcal, groups=groups)[0]
    cam_owner, subtree = hip.plan_domain_map(s, world)
    owner = hip.plan_point_owner(s, world)
    co = cam_owner[s.IP.cam]
    assert subtree and np.all((co < 0) | (co == owner[s.IP.pt]))
    h = hip.Handle(s)
    try:
        x0 = h.serialize()
        p1, st1 = h.linearize_solve(x0, 0.0, True)
        lam = 1e-4 * st1['trace'] / h.n
        q1, st2 = h.linearize_solve(x0, lam, False)
    finally:
        h.close()

    def work(comm):
        hh = hip.Handle(s, shard_rank=comm.rank, shard_count=comm.world_size)
        try:
            hh.set_allreduce(comm.allreduce_ptr)
            p, st = hh.linearize_solve(x0, 0.0, True)
            q, stq = hh.linearize_solve(x0, lam, False)
            return p, st, q, stq
        finally:
            hh.close()

    out, _comms = _run_ranks(s, world, work)
    for p, st, q, stq in out:
        assert relerr(p, p1) < 1e-7 and relerr(q, q1) < 1e-9, (world, relerr(p, p1), relerr(q, q1))
        for k in ('f', 'JpJp', 'rJp', 'pp', 'trace'):
            assert abs(st[k] - st1[k]) <= 1e-8 * abs(st1[k]), k
            assert abs(stq[k] - st2[k]) <= 1e-8 * abs(st2[k]), k


@pytest.mark.parametrize('name,world', [('roma', 2), ('roma', 3), ('camcal', 2)])
def test_domain_shards_real_projects(hip, name, world):
    """Real data under domain sharding: the roma script project (60 images with irregular visibility, 26 321
    points, five IO parameters estimated: tile kernels instead of signature groups, dense IO rows in the top) and
    the camcal project (21 images that all see each other: no domain can be cut, everything is 'top' -- the
    degenerate case) give every rank the one-rank result."""
    from dbat_amd import bundle
    from helpers import roma_struct, camcal_struct
    s = roma_struct() if name == 'roma' else camcal_struct(3)
    cam_owner, subtree = hip.plan_domain_map(s, world)
    assert subtree
    if name == 'camcal':
        assert np.all(cam_owner < 0)
    ref = bundle(s, 'gna', store_trace=False)
    out, comms = _run_ranks(s, world, lambda comm: bundle(s, 'gna', comm=comm, store_trace=False))
    for res, ok, iters, s0, E in out:
        assert ok == ref[1] and iters == ref[2]
        assert relerr(E.x, ref[4].x) < 1e-8 and abs(s0 - ref[3]) < 1e-9 * ref[3]


def test_domain_shards_failure_is_seen_by_every_rank(hip):
    """A rank-deficient domain (no datum: seven-dimensional null space) fails inside ONE rank's local
    factorisation or in the top separators; every rank must return the same code (-2), none may hang."""
    from dbat_amd import bundle
    s = _scene('c1')
    s.bundle.est.EO[:] = True
    ref = bundle(s, 'gna')
    assert ref[4].code == -2
    out, _ = _run_ranks(s, 2, lambda comm: bundle(s, 'gna', comm=comm))
    assert all(o[4].code == -2 and not o[1] for o in out)


def test_shard_without_communicator_is_a_usage_error(hip):
    """A handle that is one of several domain shards but has neither a communicator nor an all-reduce callback cannot
    factor (its schedule publishes the top separators only after the sum over the ranks): DBAT_HIP_EINVAL with a
    message, not a spurious 'singular' (ADVICE r03)."""
    from dbat_amd import synth
    s, _ = synth.make_scene('C1')
    h = hip.Handle(s, shard_rank=0, shard_count=2)
    try:
        if h.info()['domain_sharding']:
            with pytest.raises(hip.DbatHipError) as e:
                h.linearize_solve(h.serialize(), 0.0, True)
            assert e.value.code == hip.EINVAL and 'communicator' in str(e.value)
    finally:
        h.close()


def test_caller_term_function_is_refused_on_a_shard(hip):
    """dbat_hip_options.term_fun needs J*p and r as whole vectors on the host; a shard holds the rows of its own
    observations only: DBAT_HIP_EUNSUPPORTED before anything is launched (include/dbat_hip.h), not a half vector."""
    from dbat_amd import synth
    s, _ = synth.make_scene('C1')
    h = hip.Handle(s, shard_rank=0, shard_count=2)
    try:
        opt = hip.default_options('gna')
        calls = []
        with pytest.raises(hip.DbatHipError) as e:
            h.solve(h.serialize(), opt, term_fun=lambda Jp, r: calls.append(1) or False)
        assert e.value.code == hip.EUNSUPPORTED and 'term_fun' in str(e.value) and not calls
    finally:
        h.close()


@pytest.mark.parametrize('env', ['DBAT_HIP_MG_REPLICATED=1', 'DBAT_HIP_SIG=2', 'DBAT_HIP_SIG=0'])
def test_domain_shards_variants(hip, env, monkeypatch):
    """The replicated fall-back (contiguous point ranges, envelope of the whole system summed, every rank factors
    everything) and the domain scheme under the signature kernels forced on / off."""
    from dbat_amd import bundle
    name, val = env.split('=')
    s = _scene('c1')
    ref = bundle(s, 'gna')
    monkeypatch.setenv(name, val)
    assert hip.plan_domain_map(s, 2)[1] == (name != 'DBAT_HIP_MG_REPLICATED')
    out, comms = _run_ranks(s, 2, lambda comm: bundle(s, 'gna', comm=comm))
    for res, ok, iters, s0, E in out:
        assert ok and iters == ref[2] and relerr(E.x, ref[4].x) < 1e-8
