"""N>1 path on CPU: world_size-2 gloo run of tests/_gloo_worker.py."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sharded_reduced_system_gloo_world2():
    env = dict(os.environ)
    env['MASTER_ADDR'] = '127.0.0.1'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
           '--master-addr', '127.0.0.1', '--master-port', '29571',
           os.path.join(ROOT, 'tests', '_gloo_worker.py')]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert 'GLOO_OK world=2' in out.stdout


def _run_domain_worker(world, port):
    env = dict(os.environ)
    env['MASTER_ADDR'] = '127.0.0.1'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=%d' % world,
           '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.join(ROOT, 'tests', '_gloo_domain_worker.py')]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert 'GLOO_DOMAIN_OK world=%d' % world in out.stdout


def test_domain_sharded_solve_gloo_world2():
    """Domain sharding (csrc/nd.hpp) on the plan's ownership maps, world size 2: tests/_gloo_domain_worker.py."""
    _run_domain_worker(2, 29573)


def test_domain_sharded_solve_gloo_world4():
    _run_domain_worker(4, 29575)


def test_domain_map_invariant_and_balance():
    """No object point sees interior cameras of two domains, whatever the number of ranks; the observations are
    balanced over the ranks (the cut between two ranks is placed by bisection on the interiors' weight, points
    that only see top-separator cameras fill up the lightest ranks)."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import numpy as np
    from dbat_amd import _hip, synth
    s, _ = synth.make_scene('C1', cams=400, points=20000, rays=8)
    for w in (2, 3, 4, 8):
        cam_owner, subtree = _hip.plan_domain_map(s, w)
        owner = _hip.plan_point_owner(s, w)
        assert subtree
        co = cam_owner[s.IP.cam]
        assert np.all((co < 0) | (co == owner[s.IP.pt]))
        cnt = np.bincount(owner[s.IP.pt], minlength=w)
        assert cnt.max() <= 1.1 * cnt.mean(), (w, cnt)
        assert 0 < np.count_nonzero(cam_owner < 0) < 0.75 * len(cam_owner)
    cam_owner, subtree = _hip.plan_domain_map(s, 1)
    assert not subtree and np.all(cam_owner == -1)
    # scenes of random shape (bench/fuzz_multishard.py runs the same family through the GPU): every point's owner is
    # the domain of its interior cameras, for even and odd numbers of ranks
    rng = np.random.default_rng(11)
    for _ in range(6):
        cams, rays, points = int(rng.integers(30, 300)), int(rng.integers(3, 11)), int(rng.integers(500, 6000))
        selfcal = bool(rng.integers(0, 2))
        s, _t = synth.make_scene('C1' if rng.integers(0, 2) else 'small', seed=int(rng.integers(1, 10 ** 6)), cams=cams,
                                 points=points, rays=rays, selfcal=selfcal, groups=2 if selfcal else 1)
        for w in (2, 3, 5):
            cam_owner, subtree = _hip.plan_domain_map(s, w)
            owner = _hip.plan_point_owner(s, w)
            co = cam_owner[s.IP.cam]
            assert np.all((co < 0) | (co == owner[s.IP.pt])), (cams, rays, points, w)


def test_point_owner_partitions_points():
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import numpy as np
    from dbat_amd import _hip
    from helpers import synth_struct
    s, _ = synth_struct('small')
    for w in (1, 2, 4, 8):
        owner = _hip.plan_point_owner(s, w)
        assert owner.min() == 0 and owner.max() == w - 1
        cnt = np.bincount(owner[s.IP.pt], minlength=w)        # observations per rank
        assert cnt.max() - cnt.min() <= 0.1 * cnt.mean() + 16
