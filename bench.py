#!/usr/bin/env python
"""Benchmark of the bundle hot path on MI355X.

One step = one Levenberg-Marquardt iteration's device work on a synthetic
scene: residual + Jacobian blocks + J'J build + Schur complement, Cholesky
solve of the reduced camera system, back-substitution, and the residual-only
evaluation of the trial point (dbat_hip_bench_step).  Everything is resident
in HBM before the timed region.

    python bench.py --gpus N --steps K --warmup W [--config C3]

--config roma | roma-selfcal | camcal | sxb: the reference's own demo projects (bench/real_scenes.py, fixtures under
tests/golden) -- the scenes DBAT's only published timings are quoted on; the line then also carries the
shipped dbat_hip_solve with the demo's damping next to the published MATLAB figure.

N > 1: one rank per GPU.  Under torch.distributed.run (WORLD_SIZE set) this
process is one of the ranks; started plainly, it launches
`python -m torch.distributed.run --nproc-per-node N` itself before anything
touches a GPU, relays the ranks' output and exits with their code.  The object
points of the same scene are sharded over the ranks (strong scaling); the
reduced camera system is summed with one RCCL all-reduce per iteration inside
libdbat_hip.so (dbat_hip_comm_init).  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_PEAK_TFLOPS = 78.6    # MI355X FP64 vector = matrix peak (vendor)


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(s_main, name_main, budget_s=25.0, others=('C1', 'C2')):
    """The reference algorithm as written (explicit sparse J, J'*J, sparse
    Cholesky of the FULL normal matrix, levenberg_marquardt.m:81-82,119) in
    C++/OpenMP on all host cores -- bench/cpu_ref.cpp, checked against the
    oracle in tests/test_cpu_ref.py.  Timed on the bench's own scene (a
    bounded number of LM iterations) and on the smaller BASELINE configs."""
    sys.path.insert(0, os.path.join(ROOT, 'bench'))
    import cpu_ref
    from dbat_amd import synth

    def run(s, max_it, budget):
        c = cpu_ref.CpuRef(s)                      # all cores (omp_get_max_threads)
        try:
            x = c.serialize()
            p, st = c.lm_step(x, -1e-10)           # untimed: first touch of every buffer
            n_it, t_all, ms, t_it = 0, 0.0, {}, []
            while n_it < max_it and (n_it == 0 or t_all < budget):
                t0 = time.perf_counter()
                p, st = c.lm_step(x, -1e-10)
                t_it.append(time.perf_counter() - t0)
                t_all += t_it[-1]
                for k, v in st['ms'].items():
                    ms[k] = ms.get(k, 0.0) + v
                if st['code'] == 0 and st['f_trial'] < st['f']:
                    x = x + p
                n_it += 1
            t_it.sort()
            # the MEDIAN iteration (a 128-thread OpenMP run on a shared box has outliers both ways: C1 measured 2.0 / 10.2 /
            # 106.9 it/s slowest / median / fastest in round 5, and its mean 7.4 was noise -- VERDICT r05)
            return {'it_per_s': 1.0 / t_it[len(t_it) // 2], 'it_per_s_mean': n_it / t_all, 'iterations': n_it, 'threads': c.threads, 'n_params': c.n,
                    'it_per_s_slowest_median_fastest': [1.0 / t_it[-1], 1.0 / t_it[len(t_it) // 2], 1.0 / t_it[0]],
                    'setup_s': c.setup_ms * 1e-3, 'nnz': c.nnz,
                    'ms_per_phase': {k: v / n_it for k, v in ms.items()}}
        finally:
            c.close()

    out = {}
    t_start = time.perf_counter()
    for name in others:
        if name == name_main:
            continue
        out[name] = run(synth.make_scene(name)[0], 20, 4.0)
    left = max(5.0, budget_s - (time.perf_counter() - t_start))
    main = run(s_main, 10, left)
    out[name_main] = main
    no = s_main.IP.val.shape[1]
    return {
        'value': main['it_per_s'], 'unit': 'it/s', 'cores': main['threads'], 'kind': 'port',
        'cpu': cpu_model(), 'nproc': os.cpu_count(),
        'sample': 'median of %d LM iterations of bench/cpu_ref.cpp (C++17/OpenMP: explicit CSC Jacobian, J\'J by sparse '
                  'product, supernodal Cholesky of the full %d x %d normal matrix, trial residual) on the '
                  'bench scene %s itself (%d obs), %d threads'
                  % (main['iterations'], main['n_params'], main['n_params'], name_main, no, main['threads']),
        'value_slowest_median_fastest': main['it_per_s_slowest_median_fastest'],
        'obs_per_s': main['it_per_s'] * no, 'configs': out,
        'reference_published': 'DBAT MATLAB R2020a (other machine, core count unknown): roma 1.04 s/it at 181k rows, '
                               'St-Pierre 5.9 s/it at 394k rows (SURVEY 6)',
    }


def error_line(args, msg, **kw):
    """One JSON line in the shape of a result, for a run that could not be measured."""
    out = {'metric': 'LM iterations/sec', 'value': None, 'unit': 'it/s', 'n_gpus': args.gpus, 'steps': args.steps,
           'warmup': args.warmup, 'higher_is_better': True, 'error': msg}
    out.update(kw)
    print(json.dumps(out), flush=True)


def launch_ranks(args):
    """Start one rank per GPU (torch.distributed.run) as a child process; the
    parent never initialises a GPU -- not even to count the devices: torch.cuda.device_count() goes through the HIP
    runtime on ROCm builds without amdsmi, so the count is taken by a short-lived child of its own.  A run
    that cannot work says so in one JSON line and a non-zero exit code: fewer visible devices than ranks (checked here,
    before any rank is started), or ranks that failed / timed out (their own message is above the line)."""
    host_ar = os.environ.get('DBAT_BENCH_HOST_ALLREDUCE') == '1'
    try:
        r = subprocess.run([sys.executable, '-c', 'import torch; print(torch.cuda.device_count())'],
                           capture_output=True, text=True, timeout=600)
        ndev = int(r.stdout.strip().splitlines()[-1])
    except Exception as e:                               # noqa: BLE001 (a broken torch install is an answer too)
        error_line(args, 'counting the HIP devices (torch.cuda.device_count() in a child process) failed: %s' % e)
        return 2
    if ndev < args.gpus and not host_ar:
        error_line(args, '--gpus %d but %d HIP device(s) visible (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES?); '
                         'DBAT_BENCH_HOST_ALLREDUCE=1 runs the ranks on the devices that exist, sums through the host '
                         '(a functional check, not a scaling number)' % (args.gpus, ndev), visible_devices=ndev)
        return 2
    # --standalone: the launcher picks its own rendezvous port (no bind-and-close race on a shared box)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--standalone', '--local-addr', '127.0.0.1', '--nnodes=1',
           '--nproc-per-node', str(args.gpus), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    # this pool's host driver supports dmabuf IPC only: without it RCCL's ncclCommInitRank across
    # processes fails in hipIpcGetMemHandle (the image exports the same value)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    rc = subprocess.run(cmd, env=env).returncode
    if rc != 0:
        error_line(args, 'the ranks exited with code %d (their messages are above this line)' % rc, visible_devices=ndev,
                   launcher='torch.distributed.run --standalone --nproc-per-node %d' % args.gpus)
    return rc


class Watchdog:
    """A phase that can hang on a dead peer (ncclCommInitRank, the first collective) runs under a timer: on expiry the
    rank says where it was and leaves with os._exit -- torch.distributed.run then ends the other ranks and the parent
    reports the failure.  Never an exec: a process that has touched the GPU must not be replaced (this pool)."""

    def __init__(self, seconds, rank, what):
        import threading
        self.t = threading.Timer(seconds, self._fire)
        self.t.daemon = True
        self.rank, self.what, self.seconds = rank, what, seconds

    def _fire(self):
        sys.stderr.write('bench.py rank %d: no progress within %.0f s in: %s\n' % (self.rank, self.seconds, self.what))
        sys.stderr.flush()
        os._exit(3)

    def __enter__(self):
        self.t.start()
        return self

    def __exit__(self, *a):
        self.t.cancel()
        return False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', default='C3')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-solve', action='store_true', help='skip the timed dbat_hip_solve run (LM; LM-Powell for C1, as BASELINE.json quotes it)')
    ap.add_argument('--deterministic', action='store_true',
                    help='exact (order-independent) sums into the reduced system (dbat_hip_set_deterministic: bit-identical runs); reports what that costs')
    ap.add_argument('--no-one-gpu-ref', action='store_true',
                    help='N > 1: skip rank 0\'s one-GPU run of the same scene after the timed region (multi_gpu.one_gpu)')
    ap.add_argument('--emulate-ranks', type=int, default=0, metavar='R',
                    help='one GPU plays rank 0 of R (its share of the points, its domain of the reduced system, the '
                         'collectives replaced by no-ops): per-rank phase times for the scaling estimate of DESIGN.md 6; '
                         'the numbers of the step are meaningless then and "value" is null')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args))

    import torch
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and world > 1:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    host_ar = os.environ.get('DBAT_BENCH_HOST_ALLREDUCE') == '1'
    # (launched by torch.distributed.run directly -- the driver does -- there is no parent to look first)
    ndev = torch.cuda.device_count()                     # (a rank: it is about to use its GPU anyway)
    if ndev < (world if not host_ar else 1):
        if rank == 0:
            error_line(args, '%d rank(s) but %d HIP device(s) visible: the dbat_hip core has no CPU path, and every rank '
                             'needs its own GPU (DBAT_BENCH_HOST_ALLREDUCE=1: ranks share the devices that exist)' % (world, ndev),
                       visible_devices=ndev)
        else:
            time.sleep(3.0)          # the launcher ends every rank at the first exit: let rank 0's line out first
        raise SystemExit(2)
    # DBAT_BENCH_HOST_ALLREDUCE=1: the sums over the ranks go through host memory and the control-plane group
    # (gloo) instead of RCCL, and the ranks share the GPUs that exist -- the whole multi-process path (launcher,
    # rendezvous, one plan per rank, domain sharding, barriers, max over the ranks) on a box with fewer GPUs than
    # ranks.  A functional check: its numbers are not scaling numbers, and the JSON says so.
    if host_ar:
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    comm = None
    if world > 1 or os.environ.get('DBAT_BENCH_FORCE_COMM') == '1':
        # control plane only (unique id, barrier, max over ranks): gloo.  The data-path
        # collectives are RCCL inside the library (a forced one-rank group takes the same path).
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29555')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        dist.init_process_group('gloo')
        from dbat_amd.parallel import Comm
        comm = Comm()
    from dbat_amd import _hip, synth

    t_gen = time.perf_counter()
    published = None
    real = args.config in ('roma', 'roma-selfcal', 'camcal', 'sxb')
    dense = args.config.startswith('dense')      # 'dense' or 'dense:<cams>x<points>': every point in every image (heavy points only)
    if dense:
        nc_d, np_d = (int(v) for v in args.config.split(':')[1].split('x')) if ':' in args.config else (48, 16384)
        s, _ = synth.make_dense_scene(nc_d, np_d)
    elif real:
        sys.path.insert(0, os.path.join(ROOT, 'bench'))
        import real_scenes
        s, published = real_scenes.make(args.config)       # (initial values on the device: needs the GPU)
    else:
        s, _ = synth.make_scene(args.config)
    t_gen = time.perf_counter() - t_gen
    nc, npnt, no = s.EO.val.shape[1], s.OP.val.shape[1], s.IP.val.shape[1]
    # What a process pays ONCE, whatever the problem: loading the library (and rocSOLVER / rocBLAS / RCCL behind it), the
    # HIP context, the code object -- timed on a 12-camera scene, so that plan_and_upload below is what dbat_hip_create costs
    # per problem (host plan, uploads, schedule of the factorisation).
    t_first = time.perf_counter()
    _h0 = _hip.Handle(synth.make_scene('tiny')[0], device=local)
    _h0.close()
    t_first = time.perf_counter() - t_first
    t_plan = time.perf_counter()
    emu = args.emulate_ranks if world == 1 and args.emulate_ranks > 1 else 0
    h = _hip.Handle(s, device=local, shard_rank=rank, shard_count=emu or world)
    t_plan = time.perf_counter() - t_plan
    init_timeout = float(os.environ.get('DBAT_BENCH_INIT_TIMEOUT', '180'))
    if comm is not None and host_ar:
        comm.attach_host(h)
    elif comm is not None:
        with Watchdog(init_timeout, rank, 'ncclCommInitRank (dbat_hip_comm_init) of %d ranks' % world):
            comm.attach(h)                   # ncclCommInitRank inside the library
    if emu:
        h.set_allreduce(lambda ptr, count, stream: 0)       # the sums over the ranks: not performed
    if args.deterministic:
        h.set_deterministic(True)
    info = h.info()
    x0 = h.serialize()
    h.set_x(x0)
    # LM damping as bundle.m:301 / levenberg_marquardt.m:88-95: 1e-10*trace(J'J)/n
    # (several ranks: the first linearisation + solve is the first time the collectives of the data path run)
    with Watchdog(init_timeout if world > 1 else 1e9, rank, 'the first linearisation + solve (first all-reduce of the data path)'):
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
    ms = np.zeros(16)
    t_each = []
    for _ in range(args.steps):
        t1 = time.perf_counter()
        ms += h.bench_step(lam, False)                       # (returns after the step's last kernel: it reads back f)
        t_each.append(time.perf_counter() - t1)
    barrier()
    dt = time.perf_counter() - t0
    if comm is not None:
        t = torch.tensor([dt], dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    ms /= max(args.steps, 1)
    # every rank's phase times (ms per step), so that the first real multi-GPU run explains itself: rank 0 prints
    # them with the bytes each all-reduce carried and the rate that implies
    per_rank = None
    if comm is not None and world > 1:
        # ... with the sizes of every rank's shard behind them (its observations, points, matrix-core instructions of its
        # tile kernel), so that rank 0 can print north_star's table -- HBM GB/s and MFMA % of peak per rank -- itself
        tv = torch.tensor(list(ms) + [float(info['n_obs_shard']), float(info['n_pts_shard']),
                                      float(info['tile_kernel_mfma'] + info['heavy_mfma'])], dtype=torch.float64)
        gl = [torch.zeros_like(tv) for _ in range(world)]
        torch.distributed.all_gather(gl, tv)
        per_rank = [[float(v) for v in g] for g in gl]

    # N > 1: the same scene on ONE GPU of the same box, by rank 0 while the others wait -- the N = 1 figure that this run's
    # value is to be compared with (strong scaling), measured in the same process and minute
    one_gpu = None
    if world > 1 and not host_ar and not args.no_one_gpu_ref:
        if rank == 0:
            h1 = _hip.Handle(s, device=local)
            try:
                h1.set_x(x0)
                _, st1 = h1.linearize_solve(x0, 0.0, False)
                h1.set_x(x0)
                for _ in range(args.warmup):
                    h1.bench_step(lam, False)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    h1.bench_step(lam, False)
                torch.cuda.synchronize()
                dt1 = time.perf_counter() - t1
                one_gpu = {'value': args.steps / dt1, 'ms_per_step': dt1 / args.steps * 1e3,
                           'note': 'rank 0 alone, same scene, same steps, after the timed region'}
            finally:
                h1.close()
        comm.barrier()

    # the shipped loop: a real dbat_hip_solve('lm') from x0 (host control flow, scalar
    # read-backs and all), outside the timed region above
    solve = None
    # BASELINE.json: C1 is quoted with LM-Powell damping (levenberg_marquardt_powell.m:107-214), the others with LM
    solve_damping = 'lmp' if args.config == 'C1' else 'lm'
    if not args.no_solve and not emu:
        opt = _hip.default_options(solve_damping)
        opt.store_trace = 0
        barrier()
        t1 = time.perf_counter()
        xs, res, rr, damp, aux, T = h.solve(x0, opt)
        barrier()
        t_solve = time.perf_counter() - t1
        solve = {'damping': solve_damping, 'code': int(res.code), 'iterations': int(res.iters), 'linearizations': int(res.n_linearizations),
                 'residual_evals': int(res.n_residual_evals), 'solves': int(res.n_solves),
                 'time_s': float(res.time_s), 'wall_s': t_solve, 'sigma0': float(res.sigma0),
                 'it_per_s': res.iters / res.time_s if res.time_s > 0 else None,
                 'trace_only_linearizations': int(res.n_trace_only),
                 'stage_s': dict(zip(('linearise', 'factor_solve', 'backsub', 'residual', 'other'), [float(v) for v in res.stage_s])),
                 'ms_per_linearization': res.time_s / max(res.n_linearizations, 1) * 1e3}

    # the reference's demo run: the shipped loop with the demo's damping from the demo's start values
    solve_ref = None
    if real and not emu and not args.no_solve:
        opt = _hip.default_options(published['damping'])
        opt.store_trace = 0
        best = None
        for _ in range(3):                                   # (a 10 ms solve: the first run pays the allocations of its loop)
            barrier()
            xs, res, rr, damp, aux, T = h.solve(x0, opt)
            if best is None or res.time_s < best.time_s:
                best = res
        res = best
        solve_ref = {'damping': published['damping'], 'code': int(res.code), 'iterations': int(res.iters), 'time_s': float(res.time_s),
                     'sigma0': float(res.sigma0), 'it_per_s': res.iters / res.time_s if res.time_s > 0 else None,
                     'linearizations': int(res.n_linearizations), 'residual_evals': int(res.n_residual_evals),
                     'stage_s': dict(zip(('linearise', 'factor_solve', 'backsub', 'residual', 'other'), [float(v) for v in res.stage_s])),
                     'published': published,
                     'published_it_per_s': published['iterations'] / published['bundle_s'],
                     'speedup_vs_published_matlab': (published['bundle_s'] / published['iterations']) / (res.time_s / max(res.iters, 1)),
                     'note': 'published: DBAT MATLAB R2020a on another machine, CPU and core count not recorded (SURVEY 6): context, not vs_baseline'}

    if rank == 0:
        NS = info['NS']
        # Algorithmic work per launch; obs/points are this
        # rank's shard.  k_build fuses K1,K3,K4 (HBM side: 40*no + 24*np + 48*nc +
        # 8*NS^2 bytes) with the Schur contraction K5 (sum_p 108*k_p + 216*k_p^2
        # flops); the larger of the two lower-bound times names its roof.
        kname = h.build_kernel_name()
        no_s, np_s = info['n_obs_shard'], info['n_pts_shard']
        kp = np.bincount(s.IP.pt, minlength=npnt).astype(np.float64)
        flops_schur = float(np.sum(108.0 * kp + 216.0 * kp * kp)) * (no_s / max(no, 1))
        # S is written inside its sparsity pattern only: the 64 x 64 tiles of the factor's pattern bound it from above (a dense
        # 8 NS^2, SURVEY 8(d)'s count, is 289 MB at C2 that no kernel moves -- it made the C2 line read 'hbm')
        s_pattern_bytes = 8 * h.chol_stats()['tile_tasks'] * 64 * 64
        bytes_build = 40 * no_s + 24 * np_s + 48 * nc + min(8 * NS * NS, s_pattern_bytes)
        # ms[4]: the tile kernel alone (k_cam_normal and the heavy-point kernels are outside its events); ms[12], ms[13]:
        # the kernels of the heavy / giant points (csrc/heavy.hpp): camera side + k_heavy_z, then k_heavy_syrk
        heavy = info['heavy_tasks'] > 0
        k_ms = {kname: ms[4], 'k_chol_df': ms[5], 'k_backsub': ms[6], 'k_residual_cm': ms[7]}
        if heavy:
            k_ms['k_heavy_z (+ camera side of its observations)'] = ms[12]
            k_ms['k_heavy_syrk'] = ms[13]
        # the roofline entry names the Schur kernel that really dominates the build: the tile kernel, or -- scenes
        # whose points are seen by more cameras than a tile holds (the reference's camcal demo) -- k_heavy_syrk
        heavy_dominates = heavy and (info['n_tiles'] == 0 or ms[12] + ms[13] > ms[4])
        if heavy_dominates:
            kname = 'k_heavy_syrk'
            flops_schur = float(info['heavy_algorithmic_flops'])
            bytes_build = 40 * info['heavy_obs'] + 24 * info['heavy_points'] + 48 * nc + min(8 * NS * NS, s_pattern_bytes)
        # HBM traffic of the dominant kernel: PMC counters cannot be read inside this run
        # (rocprofv3 --pmc is its own pass); the figure of the committed profile of the same
        # command is attached with its source, or null when there is none for this kernel
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
        if world == 1 and os.path.exists(tpath):
            tj = json.load(open(tpath)).get(args.config, {})
            if tj.get('kernel') == kname:
                traffic, traffic_src = tj.get('traffic_bytes_per_launch'), tj.get('source')
        t_build = (ms[13] if heavy_dominates else ms[4]) * 1e-3
        mfma_binds = flops_schur / (FP64_PEAK_TFLOPS * 1e12) > bytes_build / (HBM_PEAK_GBS * 1e9)
        if mfma_binds:
            ach = flops_schur / t_build / 1e12
            roof = {'kernel': kname, 'bound': 'mfma', 'achieved': ach, 'peak': FP64_PEAK_TFLOPS,
                    'unit': 'TFLOP/s', 'frac': ach / FP64_PEAK_TFLOPS, 'traffic': traffic}
        else:
            ach = bytes_build / t_build / 1e9
            roof = {'kernel': kname, 'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS,
                    'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS, 'traffic': traffic}
        roof['traffic_source'] = traffic_src
        # what the matrix pipe really executed: the kernel uses the symmetry of Z Z' (lower triangle by 16 x 16 blocks)
        ex_flops = 2048.0 * (info['heavy_mfma'] if heavy_dominates else info['tile_kernel_mfma'])
        roof['executed_flops'] = ex_flops
        roof['executed_TFLOPs'] = ex_flops / t_build / 1e12 if t_build > 0 else None
        roof['executed_frac'] = ex_flops / t_build / 1e12 / FP64_PEAK_TFLOPS if t_build > 0 else None
        roof['note'] = ('achieved / frac: ALGORITHMIC flops sum_p(108 k + 216 k^2) (the full product Y W\', SURVEY 8(d)) per launch; '
                        'executed_*: the v_mfma_f64_16x16x4_f64 instructions the kernel issues (x 2048) -- comparable with SQ_VALU_MFMA_BUSY')
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
        # HBM-bound kernels of the step on their algorithmic bytes
        gb = lambda b, t_ms: b / (t_ms * 1e-3) / 1e9 if t_ms > 0 else None
        hbm = {'k_backsub': {'algorithmic_bytes': 40 * no_s + 48 * np_s + 48 * nc},
               'k_residual_cm': {'algorithmic_bytes': 20 * no_s + 24 * np_s + 48 * nc}}
        for k, v in hbm.items():
            v['GBs'] = gb(v['algorithmic_bytes'], k_ms[k])
            v['frac_of_8TBs'] = v['GBs'] / HBM_PEAK_GBS if v['GBs'] else None
        # whole step against SURVEY 8(d)'s per-iteration figures (dense S written and read once, dense Cholesky):
        # the lower bound max(bytes / 8 TB/s, flops / 78.6 TF) over the measured step
        bytes_iter = 40.0 * no + 48.0 * npnt + 96.0 * nc + 16.0 * NS * NS
        flops_iter = float(np.sum(108.0 * kp + 216.0 * kp * kp)) + NS ** 3 / 3.0
        t_lb = max(bytes_iter / (HBM_PEAK_GBS * 1e9), flops_iter / (FP64_PEAK_TFLOPS * 1e12))
        te = np.sort(np.asarray(t_each)) * 1e3
        # ... and against the work that is really executed (sparse factorisation, symmetric Schur products)
        flops_exec = 2048.0 * (info['heavy_mfma'] + info['tile_kernel_mfma']) + flops_chol
        bytes_exec = 40.0 * no + 48.0 * npnt + 96.0 * nc + 16.0 * cs['tile_tasks'] * 64 * 64
        t_lb_exec = max(bytes_exec / (HBM_PEAK_GBS * 1e9), flops_exec / (FP64_PEAK_TFLOPS * 1e12))
        roof_step = {'executed_flops_iter': flops_exec, 'executed_bytes_iter': bytes_exec,
                     'executed_lower_bound_ms': t_lb_exec * 1e3, 'executed_frac': t_lb_exec / (dt / args.steps),
                     'bound': 'mfma' if flops_exec / (FP64_PEAK_TFLOPS * 1e12) > bytes_exec / (HBM_PEAK_GBS * 1e9) else 'hbm',
                     'survey_dense_counts': {'bytes_iter': bytes_iter, 'flops_iter': flops_iter, 'lower_bound_ms': t_lb * 1e3},
                     'note': 'executed_*: the work the step really does (sparse factorisation: %.3g flops, symmetric Schur products) against '
                             'max(bytes / 8 TB/s, flops / 78.6 TF); survey_dense_counts: SURVEY 8(d)\'s per-iteration figures with a dense '
                             'NS^2 S and a dense NS^3/3 Cholesky, which nobody executes -- counts only, no fraction' % flops_chol}
        multi = None
        if world > 1 or emu:
            multi = {'ranks': emu or world, 'emulated_on_one_gpu': bool(emu), 'domain_sharding': bool(info['domain_sharding']),
                     'allreduce_bytes': 8 * (info['reduced_doubles_per_factorisation'] + info['vector_doubles_per_linearisation'] + 8 + 4 * (emu or world) + 1),
                     'allreduce_bytes_reduced_system': 8 * info['reduced_doubles_per_factorisation'],
                     'allreduce_bytes_vectors': 8 * info['vector_doubles_per_linearisation'],
                     'collectives_per_step': 4, 'top_separator_cams': info['n_top_cams'],
                     'ms_factor_domain': ms[8], 'ms_allreduce': ms[9], 'ms_replicated': ms[10],
                     'tasks_domain': info['tasks_domain'], 'tasks_top': info['tasks_top'],
                     'obs_this_rank': no_s, 'pts_this_rank': np_s}
            # achieved rate of the collective inside the factorisation (the top-separator tiles): algorithm bandwidth
            # bytes / time, and bus bandwidth x 2 (R - 1) / R for a ring-equivalent comparison with the 7 x 153 GB/s xGMI links
            R_ = emu or world
            if multi['ms_allreduce'] > 0 and not emu:
                alg = multi['allreduce_bytes_reduced_system'] / (multi['ms_allreduce'] * 1e-3) / 1e9
                multi['allreduce_algbw_GBs'] = alg
                multi['allreduce_busbw_GBs'] = alg * 2.0 * (R_ - 1) / R_
            keys = ('build', 'factor_solve', 'backsub', 'trial_residual', 'tile_kernel', 'k_chol_df', 'k_backsub', 'k_residual_cm',
                    'factor_domain', 'allreduce_top_tiles', 'factor_top_replicated', 'unused', 'heavy_z', 'heavy_syrk', 'unused2', 'unused3')
            if per_rank is not None:
                multi['per_rank_ms'] = [dict(zip(keys, r[:16])) for r in per_rank]
                multi['slowest_rank_ms'] = {k: max(r[i] for r in per_rank) for i, k in enumerate(keys)}
            # north_star's table, one row per rank: achieved HBM GB/s of the streaming kernels on their algorithmic bytes
            # and MFMA % of the FP64 matrix peak of the Schur kernel (algorithmic flops of the rank's points; executed:
            # the matrix-core instructions it issues x 2048).  An emulated run has rank 0's row only.
            rows_in = per_rank if per_rank is not None else [list(ms) + [float(no_s), float(np_s),
                                                                          float(info['tile_kernel_mfma'] + info['heavy_mfma'])]]
            flops_all = float(np.sum(108.0 * kp + 216.0 * kp * kp))
            table = []
            for rk, r in enumerate(rows_in):
                o_r, p_r, mf_r = r[16], r[17], r[18]
                t_tile = (r[4] + r[13]) * 1e-3
                row = {'rank': rk, 'obs': int(o_r), 'points': int(p_r), 'ms_step_device': r[0] + r[1] + r[2] + r[3]}
                if t_tile > 0:
                    fl = flops_all * o_r / max(no, 1)
                    row['schur_TFLOPs_algorithmic'] = fl / t_tile / 1e12
                    row['schur_mfma_pct_of_fp64_peak_algorithmic'] = 100.0 * fl / t_tile / 1e12 / FP64_PEAK_TFLOPS
                    row['schur_mfma_pct_of_fp64_peak_executed'] = 100.0 * 2048.0 * mf_r / t_tile / 1e12 / FP64_PEAK_TFLOPS
                if r[7] > 0:
                    row['residual_HBM_GBs'] = (20 * o_r + 24 * p_r + 48 * nc) / (r[7] * 1e-3) / 1e9
                    row['residual_pct_of_8TBs'] = 100.0 * row['residual_HBM_GBs'] / HBM_PEAK_GBS
                if r[6] > 0:
                    row['backsub_HBM_GBs'] = (40 * o_r + 48 * p_r + 48 * nc) / (r[6] * 1e-3) / 1e9
                    row['backsub_pct_of_8TBs'] = 100.0 * row['backsub_HBM_GBs'] / HBM_PEAK_GBS
                if r[1] > 0:
                    row['factor_solve_ms'] = r[1]
                table.append(row)
            multi['per_rank_roofline'] = table
            if one_gpu is not None:
                multi['one_gpu'] = one_gpu
                multi['speedup_vs_one_gpu'] = (args.steps / dt) / one_gpu['value']
            multi['measured'] = ('per-rank phase times, all-reduce time and bytes (hipEvents on the handle\'s stream), one_gpu'
                                 if not emu else 'rank 0\'s kernel times only: the collectives are no-ops in an emulated run')
        out = {
            'metric': 'LM iterations/sec', 'value': None if emu else args.steps / dt, 'unit': 'it/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3, 'ms_per_step_min_median_max': [float(te[0]), float(te[len(te) // 2]), float(te[-1])],
            'higher_is_better': True, 'scaling': 'strong', 'deterministic_sums': bool(args.deterministic),
            'vs_baseline': None, 'dtype': 'f64', 'data': ('the reference\'s demo project (fixtures under tests/golden)' if real else 'synthetic'),
            'config': {'workload': '%s: %d cams / %d pts / %d obs, %s, LM step (J\'J build + Schur '
                                   'solve + back-substitution + trial residual)'
                                   % (args.config, nc, npnt, no,
                                      'self-calibrating' if info['ncolmax'] > 6 else 'fixed IO'),
                       'reduced_system_order': NS, 'n_params': h.n, 'parallelism': 'points/%d' % world,
                       'collective': ('%s: %s' % ('all-reduce through host memory (gloo; ranks share GPUs: functional check, not a scaling number)' if host_ar else 'RCCL all-reduce in libdbat_hip.so', 'top-separator tiles + vectors (domain sharding)' if info['domain_sharding'] else 'envelope of the reduced system')) if comm is not None else None,
                       'n_tiles': info['n_tiles'], 'n_batches': info['n_batches'], 'batch': info['BT']},
            'ms_build_schur': ms[0], 'ms_factor_solve': ms[1], 'ms_backsub': ms[2],
            'ms_trial_residual': ms[3], 'kernel_ms': k_ms,
            'roofline': roof, 'roofline_factorisation': roof_chol, 'roofline_step': roof_step, 'hbm_kernels': hbm, 'multi_gpu': multi,
            'solve': solve, 'solve_it_s': solve['it_per_s'] if solve else None, 'solve_reference_demo': solve_ref,
            'host_s': {'scene_generation': t_gen, 'library_load_and_first_use': t_first, 'plan_and_upload': t_plan},
        }
        if world == 1 and not args.no_cpu_baseline and not emu:
            try:
                out['cpu_baseline'] = cpu_baseline(s, args.config, others=() if (real or dense) else ('C1', 'C2'))
            except ValueError as e:                          # (a scene the C++ port does not model)
                out['cpu_baseline'] = {'value': None, 'kind': 'port', 'note': str(e)}
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
