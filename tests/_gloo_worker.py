"""world_size-2 worker (gloo, CPU): checks the N>1 path's host logic.

Each rank takes the object points the product's plan assigns to it
(dbat_hip_plan_point_owner), forms ITS share of the reduced camera system
with the oracle's Jacobian -- sum over own observations of Jc'Jc minus the
Schur correction of own points -- and the shares are summed with the product's
collective wrapper (dbat_amd.parallel.Comm, torch.distributed all_reduce).
The solved step must equal the oracle's unsharded full-matrix solve.
"""
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)

import torch.distributed as dist  # noqa: E402

import dbat_oracle as o  # noqa: E402
from dbat_amd import _hip  # noqa: E402
from dbat_amd.parallel import Comm, shard_ranges  # noqa: E402
from helpers import synth_struct  # noqa: E402


def main():
    dist.init_process_group('gloo')
    comm = Comm()
    rank, world = comm.rank, comm.world_size
    s, _ = synth_struct('tiny', 'priors')
    for nm in ('IO', 'EO', 'OP'):                    # bundle.m:137-154
        pr = getattr(s.prior, nm)
        pr.use = np.asarray(pr.use, bool) & np.asarray(getattr(s.bundle.est, nm), bool)
    so = o.buildserialindices(__import__('copy').deepcopy(s))
    x = o.serialize(so)
    w = o.buildweightvector(so)
    r_, K = o.brown_euler_cam4(x, so, jac=True)
    R = np.sqrt(w)
    r = R * r_
    J = (sp.diags(R) @ K).tocsr()
    n = J.shape[1]
    # reference: unsharded full solve (levenberg_marquardt.m:119 with lambda=0)
    p_ref, _ = o.normal_solve((J.T @ J).tocsc(), -(J.T @ r))
    # ownership from the product's plan
    owner = _hip.plan_point_owner(s, world)
    ranges = shard_ranges(s, world)
    assert sum(b - a for a, b in ranges) == s.OP.val.shape[1]
    assert set(np.unique(owner)) == set(range(world))
    nOP = len(so.bundle.serial.OP.dest)
    nC = n - nOP                                     # IO+EO columns come first in x
    # x columns of the OP section -> point index
    op_cols = so.bundle.serial.OP.dest
    op_pt = so.bundle.serial.OP.src // 3
    mine_cols = op_cols[owner[op_pt] == rank]
    # rows owned by this rank: image rows of own points, OP prior rows of own
    # points, IO/EO prior rows on rank 0
    no2 = 2 * s.IP.val.shape[1]
    row_owner = np.full(J.shape[0], -1)
    row_owner[:no2] = np.repeat(owner[s.IP.pt], 2)
    ix = so.post.res.ix
    row_owner[ix.IO] = 0
    row_owner[ix.EO] = 0
    op_obs_cols = so.bundle.serial.OP.dest[so.bundle.serial.OP.obs]
    row_owner[ix.OP] = owner[(so.bundle.serial.OP.src[so.bundle.serial.OP.obs]) // 3]
    rows = np.flatnonzero(row_owner == rank)
    Jr, rr = J[rows], r[rows]
    Jc, Jp = Jr[:, :nC].tocsc(), Jr[:, mine_cols].tocsc()
    U = (Jc.T @ Jc).toarray()
    W = (Jc.T @ Jp).toarray()
    V = (Jp.T @ Jp).toarray()                         # block diagonal (3x3 per point)
    gc, gp = Jc.T @ rr, Jp.T @ rr
    Vi = np.linalg.inv(V)
    S_part = U - W @ Vi @ W.T
    g_part = gc - W @ Vi @ gp
    buf = np.concatenate([S_part.ravel(), g_part])
    buf = comm.allreduce_numpy(buf)                   # the one exchange step per linearisation
    S, g = buf[:nC * nC].reshape(nC, nC), buf[nC * nC:]
    dc = np.linalg.solve(S, -g)
    dp = -Vi @ (gp + W.T @ dc)
    p = np.zeros(n)
    if rank == 0:
        p[:nC] = dc
    p[mine_cols] = dp
    p = comm.allreduce_numpy(p)                       # final gather of the sharded step
    err = np.linalg.norm(p - p_ref) / np.linalg.norm(p_ref)
    assert err < 1e-9, err
    assert comm.n_collectives == 2
    dist.barrier()
    if rank == 0:
        print('GLOO_OK world=%d err=%.2e' % (world, err))
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
