#!/usr/bin/env python
"""Benchmark of the bundle hot path on MI355X.

One step = one Levenberg-Marquardt iteration's device work on a synthetic
scene: residual + Jacobian blocks + J'J build + Schur complement, Cholesky
solve of the reduced camera system, back-substitution, and the residual-only
evaluation of the trial point (dbat_hip_bench_step).  Everything is resident
in HBM before the timed region.

    python bench.py --gpus N --steps K --warmup W [--config C3]

For N>1 it is launched by torch.distributed.run (one rank per GPU, RCCL); the
object points of the same scene are sharded over the ranks (strong scaling)
and the reduced system is summed with one all-reduce per iteration.  Rank 0
prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_PEAK_TFLOPS = 78.6    # MI355X FP64 vector = matrix peak (vendor)


def cpu_baseline(seconds=20.0):
    """Time the CPU oracle (a port of the reference algorithm as written:
    explicit sparse J, J'J, sparse direct solve of the FULL normal matrix) on a
    bounded sample of the workload: LM iterations on a 100-camera / 10k-point /
    100k-observation scene cut from the same generator."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import dbat_oracle as o
    import scipy.sparse as sp
    from dbat_amd import synth
    s, _ = synth.make_scene('C1')
    s = o.buildserialindices(s)
    x = o.serialize(s)
    w = o.buildweightvector(s)
    R = np.sqrt(w)
    n_it, t0 = 0, time.perf_counter()
    while True:
        r_, K = o.brown_euler_cam4(x, s, jac=True)           # residual + Jacobian
        r = R * r_
        J = (sp.diags(R) @ K).tocsc()
        JTJ = (J.T @ J).tocsc()                              # levenberg_marquardt.m:81
        lam = 1e-10 * JTJ.diagonal().sum() / J.shape[1]
        p, _ = o.normal_solve((JTJ + lam * sp.identity(J.shape[1])).tocsc(), -(J.T @ r))
        rt = R * o.brown_euler_cam4(x + p, s)                # trial point
        if rt @ rt < r @ r:
            x = x + p
        n_it += 1
        if time.perf_counter() - t0 > seconds or n_it >= 50:
            break
    dt = time.perf_counter() - t0
    no = s.IP.val.shape[1]
    return {
        'value': n_it / dt, 'unit': 'it/s', 'cores': 1, 'kind': 'port',
        'sample': 'NumPy/SciPy oracle, %d LM iterations on a 100 cam / 10k pt / %d obs scene '
                  '(1/100 of the C3 observations); full sparse normal matrix, SuperLU' % (n_it, no),
        'obs_per_s': n_it * no / dt,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', default='C3')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the dbat_hip core has no CPU path)')
    torch.cuda.set_device(local)
    comm = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        from dbat_amd.parallel import Comm
        comm = Comm()
    from dbat_amd import _hip, synth

    t_gen = time.perf_counter()
    s, _ = synth.make_scene(args.config)
    t_gen = time.perf_counter() - t_gen
    nc, npnt, no = s.EO.val.shape[1], s.OP.val.shape[1], s.IP.val.shape[1]
    t_plan = time.perf_counter()
    h = _hip.Handle(s, device=local, shard_rank=rank, shard_count=world)
    t_plan = time.perf_counter() - t_plan
    if comm is not None:
        h.set_allreduce(comm.allreduce_ptr)
    info = h.info()
    x0 = h.serialize()
    h.set_x(x0)
    # LM damping as bundle.m:301 / levenberg_marquardt.m:88-95: 1e-10*trace(J'J)/n
    _, st = h.linearize_solve(x0, 0.0, False)
    lam = 1e-10 * st['trace'] / h.n
    h.set_x(x0)

    def barrier():
        if comm is not None:
            comm.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        h.bench_step(lam, False)
    barrier()
    t0 = time.perf_counter()
    ms = np.zeros(8)
    for _ in range(args.steps):
        ms += h.bench_step(lam, False)
    barrier()
    dt = time.perf_counter() - t0
    if comm is not None:
        t = torch.tensor([dt], dtype=torch.float64, device='cuda')
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    ms /= max(args.steps, 1)

    if rank == 0:
        NS = info['NS']
        # algorithmic bytes / flops per launch; obs and
        # points are this rank's shard
        no_s, np_s = info['n_obs_shard'], info['n_pts_shard']
        bytes_build = 40 * no_s + 24 * np_s + 48 * nc + 8 * NS * NS
        flops_chol = NS ** 3 / 3.0
        k_ms = {'k_build': ms[4], 'potrf+potrs': ms[5], 'k_backsub': ms[6], 'k_residual': ms[7]}
        dom = max(k_ms, key=k_ms.get)
        if dom == 'potrf+potrs':
            ach = flops_chol / (ms[5] * 1e-3) / 1e12
            roof = {'kernel': 'rocsolver_dpotrf+dpotrs (reduced camera system, order %d)' % NS,
                    'bound': 'mfma', 'achieved': ach, 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': ach / FP64_PEAK_TFLOPS, 'traffic': None}
        else:
            byt = {'k_build': bytes_build, 'k_backsub': 24 * no_s + 16 * no_s + 48 * np_s + 48 * nc,
                   'k_residual': 24 * no_s + 24 * np_s + 48 * nc}[dom]
            ach = byt / (k_ms[dom] * 1e-3) / 1e9
            roof = {'kernel': dom, 'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS,
                    'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS, 'traffic': None}
        out = {
            'metric': 'LM iterations/sec', 'value': args.steps / dt, 'unit': 'it/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'strong',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': '%s: %d cams / %d pts / %d obs, %s, LM step (J\'J build + Schur '
                                   'solve + back-substitution + trial residual)'
                                   % (args.config, nc, npnt, no,
                                      'self-calibrating' if info['ncolmax'] > 6 else 'fixed IO'),
                       'reduced_system_order': NS, 'n_params': h.n, 'parallelism': 'points/%d' % world,
                       'n_tiles': info['n_tiles'], 'n_batches': info['n_batches'], 'batch': info['BT']},
            'ms_build_schur': ms[0], 'ms_factor_solve': ms[1], 'ms_backsub': ms[2],
            'ms_trial_residual': ms[3], 'kernel_ms': k_ms,
            'roofline': roof,
            'host_s': {'scene_generation': t_gen, 'plan_and_upload': t_plan},
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out))
    h.close()
    if comm is not None:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
