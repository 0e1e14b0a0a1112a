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
    if world > 1 or os.environ.get('DBAT_BENCH_FORCE_COMM') == '1':
        # one rank per GPU over RCCL (a forced one-rank group exercises the same code path)
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29555')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
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
        # Algorithmic work per launch; obs/points are this
        # rank's shard.  k_build fuses K1,K3,K4 (HBM side: 40*no + 24*np + 48*nc +
        # 8*NS^2 bytes) with the Schur contraction K5 (sum_p 108*k_p + 216*k_p^2
        # flops); the larger of the two lower-bound times names its roof.
        tile3 = info['ncolmax'] <= 6 and os.environ.get('DBAT_HIP_TILE3', '1') != '0' and 'DBAT_HIP_TILE_BMAX' not in os.environ
        kname_t = (('k_build_tile3' if tile3 else 'k_build_tile2') if info['ncolmax'] <= 14 else 'k_build_tile') if info['n_tiles'] > 0 else 'k_build'
        no_s, np_s = info['n_obs_shard'], info['n_pts_shard']
        kp = np.bincount(s.IP.pt, minlength=npnt).astype(np.float64)
        flops_schur = float(np.sum(108.0 * kp + 216.0 * kp * kp)) * (no_s / max(no, 1))
        bytes_build = 40 * no_s + 24 * np_s + 48 * nc + 8 * NS * NS
        # ms[4]: the tile kernel alone (k_cam_normal and the heavy-point kernels are outside its events)
        k_ms = {kname_t: ms[4], 'potrf+potrs': ms[5], 'k_backsub': ms[6], 'k_residual': ms[7]}
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'r01_traffic.json')
        if world == 1 and args.config == 'C3' and os.path.exists(tpath):
            traffic = json.load(open(tpath)).get('traffic_bytes_per_launch')   # PMC, see profiles/
        t_build = ms[4] * 1e-3
        mfma_binds = flops_schur / (FP64_PEAK_TFLOPS * 1e12) > bytes_build / (HBM_PEAK_GBS * 1e9)
        kname = kname_t
        if mfma_binds:
            ach = flops_schur / t_build / 1e12
            roof = {'kernel': kname, 'bound': 'mfma', 'achieved': ach, 'peak': FP64_PEAK_TFLOPS,
                    'unit': 'TFLOP/s', 'frac': ach / FP64_PEAK_TFLOPS, 'traffic': traffic}
        else:
            ach = bytes_build / t_build / 1e9
            roof = {'kernel': kname, 'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS,
                    'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS, 'traffic': traffic}
        roof['algorithmic_flops'] = flops_schur
        roof['algorithmic_bytes'] = bytes_build
        roof['hbm_GBs_on_algorithmic_bytes'] = bytes_build / t_build / 1e9
        # factorisation: flops actually scheduled = 64^3-tile products of the update phase and of the
        # triangular solves (2*64^3 each) + the diagonal blocks (64^3/3); the kernel is bound by the
        # latency of its dependent chain, the roofline entry only says how far from the matrix peak that is
        cs = h.chol_stats()
        t64 = 2.0 * 64 ** 3
        flops_chol = cs['tile_products'] * t64 + (cs['tile_tasks'] - cs['tile_rows']) * t64 + cs['tile_rows'] * 64 ** 3 / 3.0
        ach_c = flops_chol / (ms[5] * 1e-3) / 1e12
        roof_chol = {'kernel': 'k_chol_df (persistent dataflow Cholesky + both substitutions; %s), order %d, '
                               '%d tile tasks, %d tile products'
                               % ('nested-dissection order' if cs['nested_dissection'] else 'natural order', NS,
                                  cs['tile_tasks'], cs['tile_products']),
                     'bound': 'mfma', 'achieved': ach_c, 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': ach_c / FP64_PEAK_TFLOPS, 'traffic': None, 'algorithmic_flops': flops_chol,
                     'dense_equivalent_flops': NS ** 3 / 3.0}
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
            'roofline': roof, 'roofline_factorisation': roof_chol,
            'host_s': {'scene_generation': t_gen, 'plan_and_upload': t_plan},
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
        else:
            out['cpu_baseline'] = None
    # RCCL writes a version banner through C stdio, which is block-buffered when stdout is a
    # pipe and would otherwise land after the result: every rank flushes it, then rank 0 prints
    # the JSON line as the last line of the job
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    sys.stdout.flush()
    if comm is not None:
        comm.barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    h.close()
    if comm is not None:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
