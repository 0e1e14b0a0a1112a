"""world_size-N worker (gloo, CPU): the DOMAIN-SHARDED solve of the reduced camera system, in NumPy,
on the ownership maps of the product's plan (dbat_hip_plan_domain_map / dbat_hip_plan_point_owner).

Every rank forms ITS share of the reduced system from the oracle's Jacobian rows of its own object
points, eliminates its own domain's cameras locally, and only the shares of the Schur complement on
the top separators are summed (one collective).  The top system is solved by every rank, each rank
back-substitutes its own cameras and points, and the masked sum of the pieces must equal the
oracle's unsharded full-matrix solve.  Checks on the way: no point sees interior cameras of two
domains; a rank's share is zero outside (own domain + top) x (own domain + top); the collective
carries the top block only.
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
from dbat_amd import _hip, synth  # noqa: E402
from dbat_amd.parallel import Comm  # noqa: E402


def main():
    dist.init_process_group('gloo')
    comm = Comm()
    rank, world = comm.rank, comm.world_size
    # 15 x 15 cameras, six rays per point: narrow separators, four non-empty domains at world 4
    s, _ = synth.make_scene('C1', cams=225, points=2500, rays=6, selfcal=True)
    so = o.buildserialindices(__import__('copy').deepcopy(s))
    x = o.serialize(so)
    w = o.buildweightvector(so)
    r_, K = o.brown_euler_cam4(x, so, jac=True)
    R = np.sqrt(w)
    r = R * r_
    J = (sp.diags(R) @ K).tocsr()
    n = J.shape[1]
    lam = 1e-6 * (J.multiply(J)).sum() / n
    p_ref, _ = o.normal_solve((J.T @ J + lam * sp.identity(n)).tocsc(), -(J.T @ r))
    cam_owner, subtree = _hip.plan_domain_map(s, world)
    owner = _hip.plan_point_owner(s, world)
    assert subtree and set(np.unique(owner)) <= set(range(world))
    nd = [int(np.count_nonzero(cam_owner == q)) for q in range(world)]
    assert min(nd) > 0, nd                                      # every rank has a domain in this scene
    # the invariant of the scheme: the interior cameras a point sees belong to its owner's domain
    co = cam_owner[s.IP.cam]
    assert np.all((co < 0) | (co == owner[s.IP.pt]))
    # x columns: IO first, then EO, then OP (buildserialindices.m:57)
    ser = so.bundle.serial
    nIO, nEO, nOP = len(ser.IO.dest), len(ser.EO.dest), len(ser.OP.dest)
    nC = nIO + nEO
    eo_cam = ser.EO.src // so.EO.val.shape[0]                   # camera of every EO column
    col_owner = np.full(nC, -1)                                 # IO columns: top
    col_owner[nIO:] = cam_owner[eo_cam]
    op_pt = ser.OP.src // 3
    mine_pts = ser.OP.dest[owner[op_pt] == rank]
    rows = np.flatnonzero(np.repeat(owner[s.IP.pt], 2) == rank)   # no prior rows in this scene
    assert J.shape[0] == 2 * s.IP.val.shape[1]
    Jr, rr = J[rows], r[rows]
    Jc, Jp = Jr[:, :nC].tocsc(), Jr[:, mine_pts].tocsc()
    dom = np.flatnonzero(col_owner == rank)
    top = np.flatnonzero(col_owner < 0)
    other = np.flatnonzero((col_owner >= 0) & (col_owner != rank))
    assert Jc[:, other].nnz == 0                                # own observations never touch another domain
    V = (Jp.T @ Jp + lam * sp.identity(len(mine_pts))).toarray()
    Vi = np.linalg.inv(V)
    W = (Jc.T @ Jp).toarray()
    S_r = (Jc.T @ Jc).toarray() - W @ Vi @ W.T                  # this rank's share; damping where it owns the column
    gp = Jp.T @ rr
    g_r = Jc.T @ rr - W @ Vi @ gp
    own_cols = np.r_[dom, top] if rank == 0 else dom            # the terms that enter once: rank 0 owns the top
    S_r[own_cols, own_cols] += lam
    assert not S_r[other].any() and not S_r[:, other].any()
    # eliminate the own domain locally; the share of the Schur complement on the top
    Sdd, Std = S_r[np.ix_(dom, dom)], S_r[np.ix_(top, dom)]
    L = np.linalg.cholesky(Sdd)
    Y = np.linalg.solve(L, Std.T)                               # L^-1 S_dt
    yd = np.linalg.solve(L, -g_r[dom])
    T_share = S_r[np.ix_(top, top)] - Y.T @ Y
    t_share = -g_r[top] - Y.T @ yd
    nt = len(top)
    buf = comm.allreduce_numpy(np.concatenate([T_share.ravel(), t_share]))      # the ONE exchange of the factorisation
    assert comm.bytes_reduced == 8 * (nt * nt + nt) and nt < nC / 2
    T, t = buf[:nt * nt].reshape(nt, nt), buf[nt * nt:]
    q_top = np.linalg.solve(T, t)                               # every rank
    q_dom = np.linalg.solve(L.T, yd - Y @ q_top)
    dc = np.zeros(nC)
    dc[top], dc[dom] = q_top, q_dom                             # other domains' steps: never needed here
    dp = -Vi @ (gp + W.T @ dc)
    p = np.zeros(n)
    p[dom] = q_dom
    if rank == 0:
        p[top] = q_top
    p[mine_pts] = dp
    p = comm.allreduce_numpy(p)                                 # masked gather of the result
    err = np.linalg.norm(p - p_ref) / np.linalg.norm(p_ref)
    assert err < 1e-8, err
    dist.barrier()
    if rank == 0:
        print('GLOO_DOMAIN_OK world=%d err=%.2e top=%d of %d columns, domains %s' % (world, err, nt, nC, nd))
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
