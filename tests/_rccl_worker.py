"""Multi-rank GPU worker (launched by test_two_ranks_rccl_match_single): each rank
runs bundle() on its shard of the object points with the RCCL communicator
inside libdbat_hip.so and compares with the one-GPU run on the same device.
DBAT_TEST_HOST_ALLREDUCE=1 (test_processes_share_one_gpu_through_the_host): the ranks share GPU 0 and sum
through host memory over gloo -- the same processes, launcher and domain sharding on a one-GPU box."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from dbat_amd import bundle, bundle_cov  # noqa: E402
from dbat_amd.parallel import Comm  # noqa: E402
from helpers import synth_struct, relerr  # noqa: E402


def main():
    via_host = os.environ.get('DBAT_TEST_HOST_ALLREDUCE') == '1'
    local = 0 if via_host else int(os.environ['LOCAL_RANK'])
    torch.cuda.set_device(local)
    dist.init_process_group('gloo')
    comm = Comm(via_host=via_host)
    assert comm.device == local
    s, _ = synth_struct('small', 'priors')
    for damping in ('gna', 'lm', 'lmp'):
        ref = bundle(s, damping, device=local)
        res, ok, iters, s0, E = bundle(s, damping, comm=comm)
        assert ok == ref[1] and E.code == ref[4].code
        assert relerr(E.x, ref[4].x) < 1e-8, relerr(E.x, ref[4].x)
        assert abs(s0 - ref[3]) < 1e-9 * ref[3]
        if damping != 'lm':
            assert iters == ref[2]
        assert relerr(res.post.res.IP, ref[0].post.res.IP) < 1e-6
    print('RANK_OK %d' % comm.rank, flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
