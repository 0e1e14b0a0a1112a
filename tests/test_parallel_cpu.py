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
