"""ctypes binding of the C ABI in include/dbat_hip.h (libdbat_hip.so).

The library is the product: there is no Python or CPU fallback.  If the shared
object is missing or cannot be loaded, importing callers get a loud
`DbatHipUnavailable` with the build command.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libdbat_hip.so')
# measurement only (bench/ tools): DBAT_AMD_LIB=prof loads libdbat_hip_prof.so, the same sources built with
# -DDBAT_HIP_PROFILING (make -C dbat_amd/csrc prof) -- the only build that reads DBAT_HIP_ABLATE, _DF_TRACE, ...
if os.environ.get('DBAT_AMD_LIB') == 'prof':
    LIB_PATH = os.path.join(_HERE, 'libdbat_hip_prof.so')
elif os.environ.get('DBAT_AMD_LIB', '').endswith('.so'):          # (a development build to compare with)
    LIB_PATH = os.environ['DBAT_AMD_LIB']
if LIB_PATH != os.path.join(_HERE, 'libdbat_hip.so'):
    # never silently: a stale development build must not stand in for the product library in a test or bench run
    sys.stderr.write('[dbat_amd] DBAT_AMD_LIB: loading %s instead of the product library\n' % LIB_PATH)

ABI_VERSION = 4
DAMP = {'none': 0, 'gm': 0, 'gna': 1, 'lm': 2, 'lmp': 3}

OK, EINVAL, EUNSUPPORTED, EDEVICE, ENOMEM = 0, -101, -102, -103, -104


class DbatHipUnavailable(RuntimeError):
    pass


class DbatHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__('dbat_hip error %d: %s' % (code, msg))
        self.code = code


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_bp = C.POINTER(C.c_uint8)


class Problem(C.Structure):
    _fields_ = [
        ('abi_version', C.c_int32), ('n_images', C.c_int32), ('n_points', C.c_int32),
        ('n_obs', C.c_int64), ('dist_model', C.c_int32), ('nK', C.c_int32), ('nP', C.c_int32),
        ('ip_cam', _ip), ('ip_pt', _ip), ('ip_val', _dp), ('ip_std', _dp),
        ('IO_val', _dp), ('px_size', _dp), ('EO_val', _dp), ('OP_val', _dp),
        ('est_IO', _bp), ('est_EO', _bp), ('est_OP', _bp),
        ('IO_block', _ip), ('EO_block', _ip),
        ('prior_IO_use', _bp), ('prior_IO_val', _dp), ('prior_IO_std', _dp),
        ('prior_EO_use', _bp), ('prior_EO_val', _dp), ('prior_EO_std', _dp),
        ('prior_OP_use', _bp), ('prior_OP_val', _dp), ('prior_OP_std', _dp),
        ('device', C.c_int32), ('shard_rank', C.c_int32), ('shard_count', C.c_int32),
    ]


class Options(C.Structure):
    _fields_ = [
        ('damping', C.c_int32), ('max_iter', C.c_int32), ('conv_tol', C.c_double),
        ('abs_term', C.c_int32), ('singular_test', C.c_int32), ('store_trace', C.c_int32),
        ('mu', C.c_double), ('alpha_min', C.c_double), ('lambda0', C.c_double),
        ('lambda_min', C.c_double), ('rho_bad', C.c_double), ('rho_good', C.c_double),
        ('delta0', C.c_double),
        ('term_fun', C.c_void_p), ('term_user', C.c_void_p), ('veto_fun', C.c_void_p), ('veto_user', C.c_void_p),
        ('trace_fun', C.c_void_p), ('trace_user', C.c_void_p),
    ]


TERM_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, _dp, _dp, C.c_int64)   # dbat_hip_term_fn
VETO_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, _dp, C.c_int64)        # dbat_hip_veto_fn
TRACE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_int32, C.c_double)   # dbat_hip_trace_fn


def trace_text(damping, n, res, damp, step, rho):
    """The line the reference's lsa solver prints with 'trace' (gauss_newton_armijo.m:119-128, gauss_markov.m:74-76,
    levenberg_marquardt.m:138-147, levenberg_marquardt_powell.m:160-164), from the arguments of a dbat_hip_trace_fn."""
    from fractions import Fraction
    if damping == DAMP['gna']:
        if damp != damp:
            return 'Gauss-Newton-Armijo: iteration %d, residual norm=%.2g' % (n, res)
        return 'Gauss-Newton-Armijo: iteration %d, residual norm=%.2g, last alpha=%s' % (n, res, Fraction(damp).limit_denominator(1 << 40))   # strtrim(rats(alpha)): alpha = 2^-k
    if damping == DAMP['lm']:
        if damp != damp:
            return 'Levenberg-Marquardt: iteration %d, residual norm=%.2g' % (n, res)
        return 'Levenberg-Marquardt: iteration %d, residual norm=%.2g, lambda=%.2g' % (n, res, damp)
    if damping == DAMP['lmp']:
        return 'Levenberg-Marquardt-Powell: iteration %d, residual norm=%.2g, delta=%.2g, step=%s, rho=%.1f' % (
            n, res, damp, ('GN', 'IP', 'CP')[step] if 0 <= step < 3 else '?', rho)
    return 'Gauss-Markov: iteration %d, residual norm=%.2g' % (n, res)


class Result(C.Structure):
    _fields_ = [
        ('code', C.c_int32), ('iters', C.c_int32), ('n_res', C.c_int32), ('n_damp', C.c_int32),
        ('n_trace', C.c_int32), ('sigma0', C.c_double), ('time_s', C.c_double),
        ('n_residual_evals', C.c_int32), ('n_linearizations', C.c_int32), ('n_solves', C.c_int32),
        ('n_trace_only', C.c_int32), ('stage_s', C.c_double * 5),
    ]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)

# every symbol include/dbat_hip.h declares: name -> (restype, argtypes)
_H = C.c_void_p
SYMBOLS = {
    'dbat_hip_last_error': (C.c_char_p, []),
    'dbat_hip_abi_version': (C.c_int, []),
    'dbat_hip_default_options': (C.c_int, [C.c_int32, C.POINTER(Options)]),
    'dbat_hip_plan': (C.c_int, [C.POINTER(Problem)] + [C.POINTER(C.c_int64)] * 7),
    'dbat_hip_plan_structural_rank_ok': (C.c_int, [C.POINTER(Problem), C.POINTER(C.c_int32)]),
    'dbat_hip_plan_point_owner': (C.c_int, [C.POINTER(Problem), _ip]),
    'dbat_hip_plan_serialize': (C.c_int, [C.POINTER(Problem), _dp]),
    'dbat_hip_plan_layout_stats': (C.c_int, [C.POINTER(Problem), C.POINTER(C.c_int64)]),
    'dbat_hip_plan_domain_map': (C.c_int, [C.POINTER(Problem), _ip, C.POINTER(C.c_int32)]),
    'dbat_hip_create': (C.c_int, [C.POINTER(Problem), C.POINTER(_H)]),
    'dbat_hip_destroy': (None, [_H]),
    'dbat_hip_num_params': (C.c_int64, [_H]),
    'dbat_hip_num_residuals': (C.c_int64, [_H]),
    'dbat_hip_serialize': (C.c_int, [_H, _dp]),
    'dbat_hip_deserialize': (C.c_int, [_H, _dp, _dp, _dp, _dp]),
    'dbat_hip_structural_rank_ok': (C.c_int, [_H, C.POINTER(C.c_int32)]),
    'dbat_hip_residual': (C.c_int, [_H, _dp, _dp, _dp]),
    'dbat_hip_jacobian_blocks': (C.c_int, [_H, _dp, _dp, _dp, _dp]),
    'dbat_hip_jacobian_sample': (C.c_int, [_H, _dp, C.c_int64, C.POINTER(C.c_int64), _dp, _dp, _dp, _dp]),
    'dbat_hip_linearize_solve': (C.c_int, [_H, _dp, C.c_double, C.c_int32, _dp, _dp]),
    'dbat_hip_jacobian_csc': (C.c_int, [_H, _dp, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), _dp]),
    'dbat_hip_gradient': (C.c_int, [_H, _dp]),
    'dbat_hip_colnorms': (C.c_int, [_H, _dp]),
    'dbat_hip_jtimes_sqnorm': (C.c_int, [_H, _dp, _dp]),
    'dbat_hip_jtimes': (C.c_int, [_H, _dp, _dp]),
    'dbat_hip_solve': (C.c_int, [_H, C.POINTER(Options), _dp, C.POINTER(Result), _dp, _dp, _dp, _dp]),
    'dbat_hip_final_residuals': (C.c_int, [_H, _dp, _dp]),
    'dbat_hip_comm_unique_id': (C.c_int, [_bp]),
    'dbat_hip_comm_init': (C.c_int, [_H, _bp]),
    'dbat_hip_comm_allreduce_host': (C.c_int, [_H, _dp, C.c_int64, C.c_int32]),
    'dbat_hip_set_deterministic': (C.c_int, [_H, C.c_int32]),
    'dbat_hip_set_allreduce': (C.c_int, [_H, ALLREDUCE_FN, C.c_void_p]),
    'dbat_hip_owned_mask': (C.c_int, [_H, _bp]),
    'dbat_hip_forwintersect': (C.c_int, [_H, _dp, _bp, _dp]),
    'dbat_hip_resect': (C.c_int, [C.c_int32, C.c_int32, C.POINTER(C.c_int64), _dp, _dp, C.POINTER(C.c_int64), _ip, _dp, _dp]),
    'dbat_hip_bench_step': (C.c_int, [_H, C.c_double, C.c_int32, _dp]),
    'dbat_hip_structure_key': (C.c_int, [C.POINTER(Problem), C.POINTER(C.c_uint64)]),
    'dbat_hip_handle_key': (C.c_int, [_H, C.POINTER(C.c_uint64)]),
    'dbat_hip_set_values': (C.c_int, [_H, C.POINTER(Problem)]),
    'dbat_hip_set_x': (C.c_int, [_H, _dp]),
    'dbat_hip_info': (C.c_int, [_H, C.POINTER(C.c_int64)]),
    'dbat_hip_build_kernel_name': (C.c_int, [_H, C.c_char_p, C.c_int32]),
    'dbat_hip_chol_stats': (C.c_int, [_H, C.POINTER(C.c_int64)]),
    'dbat_hip_posterior_cov': (C.c_int, [_H, _dp, C.c_double, _dp, _dp, _dp, _dp]),
}
DEBUG_SYMBOLS = {
    'dbat_hip_debug_plan_digest': (C.c_int, [C.POINTER(Problem), C.POINTER(C.c_uint64), C.c_int32, C.c_char_p, C.c_int32]),
    'dbat_hip_debug_heavy_plan_selftest': (C.c_int, [C.POINTER(Problem), _dp]),
    'dbat_hip_debug_model_eval_host': (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _dp, _dp, C.c_double,
                                                 _dp, _dp, _dp, _dp, _dp, _dp]),
}

_lib = None


def load():
    """Load libdbat_hip.so and bind every declared symbol (fails loudly)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DbatHipUnavailable(
            '%s not built: run `python -c "import __graft_entry__ as g; g.build()"` or '
            '`make -C dbat_amd/csrc` (needs hipcc).  There is no CPU fallback.' % LIB_PATH)
    try:
        # PyTorch-ROCm ships its own HIP runtime; when this library (linked against /opt/rocm) is
        # loaded first and torch afterwards, the second runtime finds no device.  Load order
        # torch -> libdbat_hip works, so make it the order always.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise DbatHipUnavailable('cannot load %s: %s' % (LIB_PATH, e)) from e
    for name, (res, args) in {**SYMBOLS, **DEBUG_SYMBOLS}.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise DbatHipUnavailable('%s does not export %s' % (LIB_PATH, name)) from e
        fn.restype = res
        fn.argtypes = args
    if lib.dbat_hip_abi_version() != ABI_VERSION:
        raise DbatHipUnavailable('ABI version mismatch')
    _lib = lib
    return lib


UNIQUE_ID_BYTES = 128


def comm_unique_id():
    """A fresh RCCL unique id (rank 0 creates it and hands it to the other ranks)."""
    buf = (C.c_uint8 * UNIQUE_ID_BYTES)()
    check(load().dbat_hip_comm_unique_id(buf))
    return bytes(buf)


def last_error():
    return load().dbat_hip_last_error().decode('utf-8', 'replace')


def check(rc):
    if rc != 0:
        raise DbatHipError(rc, last_error())


def dptr(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def _flat(a, dtype):
    """a as a flat column-major array of dtype -- WITHOUT a copy when it already is one (a Fortran-ordered or
    one-dimensional array of that type: what dbat_amd.dbatstruct builds), so that marshalling a 10 M observation struct
    for a cached handle costs microseconds, not a pass over 500 MB."""
    a = np.asarray(a)
    if a.dtype != dtype:
        a = a.astype(dtype)
    if a.ndim <= 1:
        return np.ascontiguousarray(a)
    if a.flags.f_contiguous:
        return a.reshape(-1, order='F')
    return a.flatten('F')


def _f64(a):
    return _flat(a, np.float64)


def _u8(a):
    a = np.asarray(a)
    if a.dtype != np.bool_:
        a = a.astype(bool)
    return _flat(a, np.bool_).view(np.uint8)


def _i32(a):
    return _flat(a, np.int32)


def problem_from_struct(s, device=0, shard_rank=0, shard_count=1):
    """Flatten a DBAT struct (dbat_amd.dbatstruct) into a `Problem`.

    Returns (Problem, keepalive) -- keepalive holds the numpy buffers the
    pointers refer to.
    """
    nc, npnt, no = s.EO.val.shape[1], s.OP.val.shape[1], s.IP.val.shape[1]
    dm = np.unique(s.IO.model.distModel)
    if dm.size != 1:
        raise ValueError('Mixed lens distortion models not implemented.')   # brown_euler_cam4.m:31-33
    model = int(dm[0])
    if model == 1:
        # legacy model 1 = the model that model 2 replicates (bundle.m:49-51); the library
        # implements 2..5, so a model-1 project runs as model 2
        if np.any(np.asarray(s.IO.val)[3:5] != 0) or np.any(np.asarray(s.bundle.est.IO)[3:5]):
            raise ValueError('lens distortion model 1 has no aspect/skew')
        model = 2
    if npnt >= 2 ** 31 or nc >= 2 ** 31:
        raise ValueError('too many points/images for int32 indices')
    keep = dict(
        ip_cam=_i32(s.IP.cam), ip_pt=_i32(s.IP.pt), ip_val=_f64(s.IP.val), ip_std=_f64(s.IP.std),
        IO_val=_f64(s.IO.val), px_size=_f64(s.IO.sensor.pxSize), EO_val=_f64(s.EO.val[:6]),
        OP_val=_f64(s.OP.val),
        est_IO=_u8(s.bundle.est.IO), est_EO=_u8(s.bundle.est.EO[:6]), est_OP=_u8(s.bundle.est.OP),
        IO_block=_i32(s.IO.struct.block), EO_block=_i32(s.EO.struct.block[:6]),
    )
    for nm in ('IO', 'EO', 'OP'):
        pr = getattr(s.prior, nm)
        rows = slice(0, 6) if nm == 'EO' else slice(None)
        use = np.asarray(pr.use)[rows]
        keep['prior_%s_use' % nm] = _u8(use)
        if use.any():
            keep['prior_%s_val' % nm] = _f64(np.nan_to_num(np.asarray(pr.val, float)[rows]))
            keep['prior_%s_std' % nm] = _f64(np.nan_to_num(np.asarray(pr.std, float)[rows], nan=1.0))
        else:                   # (the library reads value and standard deviation only where `use` is set: no pass over the arrays)
            keep['prior_%s_val' % nm] = _f64(np.asarray(pr.val, float)[rows])
            keep['prior_%s_std' % nm] = _f64(np.asarray(pr.std, float)[rows])
    p = Problem()
    p.abi_version = ABI_VERSION
    p.n_images, p.n_points, p.n_obs = nc, npnt, no
    p.dist_model, p.nK, p.nP = model, int(s.IO.model.nK), int(s.IO.model.nP)
    for k, v in keep.items():
        ptr_t = dict(Problem._fields_)[k]
        setattr(p, k, v.ctypes.data_as(ptr_t))
    p.device, p.shard_rank, p.shard_count = int(device), int(shard_rank), int(shard_count)
    return p, keep


class Handle:
    """RAII wrapper of dbat_hip_handle."""

    def __init__(self, s, device=0, shard_rank=0, shard_count=1, _problem=None):
        self.lib = load()
        self.prob, self._keep = _problem if _problem is not None else problem_from_struct(s, device, shard_rank, shard_count)
        h = _H()
        check(self.lib.dbat_hip_create(C.byref(self.prob), C.byref(h)))
        self.h = h
        self.n = int(self.lib.dbat_hip_num_params(h))
        self.m = int(self.lib.dbat_hip_num_residuals(h))
        self._cb = None

    def close(self):
        if getattr(self, 'h', None):
            self.lib.dbat_hip_destroy(self.h)
            self.h = None

    __del__ = close

    def key(self):
        """The structure key of the problem this handle was created from (dbat_hip_handle_key)."""
        k = (C.c_uint64 * 2)()
        check(self.lib.dbat_hip_handle_key(self.h, k))
        return (int(k[0]), int(k[1]))

    def set_values(self, s=None, _problem=None):
        """New IO / EO / OP values and prior observations of a struct with the same structure (dbat_hip_set_values);
        DbatHipError if the structure differs."""
        prob, keep = _problem if _problem is not None else problem_from_struct(
            s, self.prob.device, self.prob.shard_rank, self.prob.shard_count)
        check(self.lib.dbat_hip_set_values(self.h, C.byref(prob)))
        self.prob, self._keep = prob, keep

    def serialize(self):
        x = np.empty(self.n)
        check(self.lib.dbat_hip_serialize(self.h, dptr(x)))
        return x

    def deserialize(self, x):
        nc, npnt = self.prob.n_images, self.prob.n_points
        R = 5 + self.prob.nK + self.prob.nP
        IO, EO, OP = np.empty(R * nc), np.empty(6 * nc), np.empty(3 * npnt)
        x = np.ascontiguousarray(x, float)
        check(self.lib.dbat_hip_deserialize(self.h, dptr(x), dptr(IO), dptr(EO), dptr(OP)))
        return (IO.reshape(R, nc, order='F'), EO.reshape(6, nc, order='F'),
                OP.reshape(3, npnt, order='F'))

    def structural_rank_ok(self):
        ok = C.c_int32(0)
        check(self.lib.dbat_hip_structural_rank_ok(self.h, C.byref(ok)))
        return bool(ok.value)

    def residual(self, x, want_r=True):
        x = np.ascontiguousarray(x, float)
        r = np.zeros(self.m) if want_r else None
        f = C.c_double(0)
        check(self.lib.dbat_hip_residual(self.h, dptr(x), dptr(r), C.byref(f)))
        return (r, f.value) if want_r else f.value

    def jacobian_blocks(self, x):
        no, R = self.prob.n_obs, 5 + self.prob.nK + self.prob.nP
        x = np.ascontiguousarray(x, float)
        JEO, JOP, JIO = np.zeros(12 * no), np.zeros(6 * no), np.zeros(2 * R * no)
        check(self.lib.dbat_hip_jacobian_blocks(self.h, dptr(x), dptr(JEO), dptr(JOP), dptr(JIO)))
        return (JEO.reshape(no, 6, 2).transpose(0, 2, 1), JOP.reshape(no, 3, 2).transpose(0, 2, 1),
                JIO.reshape(no, R, 2).transpose(0, 2, 1))

    def jacobian_sample(self, x, ip_cols):
        """Residual and Jacobian blocks of the image observations ip_cols (IP columns, 0-based): (r n x 2, JEO n x 2 x 6,
        JOP n x 2 x 3, JIO n x 2 x nIOrows)."""
        idx = np.ascontiguousarray(ip_cols, np.int64)
        n, R = idx.size, 5 + self.prob.nK + self.prob.nP
        x = np.ascontiguousarray(x, float)
        r, JEO, JOP, JIO = np.zeros(2 * n), np.zeros(12 * n), np.zeros(6 * n), np.zeros(2 * R * n)
        check(self.lib.dbat_hip_jacobian_sample(self.h, dptr(x), n, idx.ctypes.data_as(C.POINTER(C.c_int64)), dptr(r), dptr(JEO),
                                                dptr(JOP), dptr(JIO)))
        return (r.reshape(n, 2), JEO.reshape(n, 6, 2).transpose(0, 2, 1), JOP.reshape(n, 3, 2).transpose(0, 2, 1),
                JIO.reshape(n, R, 2).transpose(0, 2, 1))

    def jacobian_csc(self, x, weighted=True):
        """J at x as scipy.sparse.csc_matrix (m x n): E.final.weighted.J / unweighted.J of bundle.m:341-350."""
        import scipy.sparse as sp
        x = np.ascontiguousarray(x, float)
        nnz = C.c_int64(0)
        check(self.lib.dbat_hip_jacobian_csc(self.h, dptr(x), int(bool(weighted)), C.byref(nnz), None, None, None))
        colptr = np.zeros(self.n + 1, np.int64)
        rowidx = np.zeros(max(nnz.value, 1), np.int64)
        val = np.zeros(max(nnz.value, 1))
        i64 = C.POINTER(C.c_int64)
        check(self.lib.dbat_hip_jacobian_csc(self.h, dptr(x), int(bool(weighted)), C.byref(nnz), colptr.ctypes.data_as(i64),
                                             rowidx.ctypes.data_as(i64), dptr(val)))
        return sp.csc_matrix((val[:nnz.value], rowidx[:nnz.value], colptr), shape=(self.m, self.n))

    def linearize_solve(self, x, lam=0.0, scale=True):
        x = np.ascontiguousarray(x, float)
        p, st = np.empty(self.n), np.empty(8)
        check(self.lib.dbat_hip_linearize_solve(self.h, dptr(x), float(lam), int(bool(scale)),
                                                dptr(p), dptr(st)))
        return p, dict(f=st[0], JpJp=st[1], rJp=st[2], pp=st[3], trace=st[4], singular=bool(st[5]), rcond=st[6], chol_info=int(st[7]))

    def gradient(self):
        g = np.empty(self.n)
        check(self.lib.dbat_hip_gradient(self.h, dptr(g)))
        return g

    def colnorms(self):
        g = np.empty(self.n)
        check(self.lib.dbat_hip_colnorms(self.h, dptr(g)))
        return g

    def jtimes_sqnorm(self, v):
        v = np.ascontiguousarray(v, float)
        out = C.c_double(0)
        check(self.lib.dbat_hip_jtimes_sqnorm(self.h, dptr(v), C.byref(out)))
        return out.value

    def jtimes(self, v):
        """J v at the last linearisation point: m weighted rows in the reference's row order."""
        v = np.ascontiguousarray(v, float)
        out = np.zeros(self.m)
        check(self.lib.dbat_hip_jtimes(self.h, dptr(v), dptr(out)))
        return out

    def solve(self, x0, opt, term_fun=None, veto_fun=None, trace_fun=None):
        """term_fun(Jp, r) -> bool and veto_fun(x) -> bool: the caller's own tests (bundle.m:168-192), called from the
        damping loop with numpy views of the library's vectors; an exception inside one ends the run and is re-raised.
        trace_fun(damping, n, res_norm, damp, step_type, rho): called where the reference's solver prints its 'trace' line
        (trace_text formats it)."""
        x = np.ascontiguousarray(x0, float).copy()
        raised = []

        def guarded(fn, *views):
            if raised:
                return 1
            try:
                return int(bool(fn(*views)))
            except BaseException as e:          # (never unwind through the C frames)
                raised.append(e)
                return 1
        keep = []
        if term_fun is not None:
            keep.append(TERM_FN(lambda _u, Jp, r, m: guarded(term_fun, np.ctypeslib.as_array(Jp, (m,)).copy(),
                                                             np.ctypeslib.as_array(r, (m,)).copy())))
            opt.term_fun = C.cast(keep[-1], C.c_void_p)
        if veto_fun is not None:
            keep.append(VETO_FN(lambda _u, xx, n: guarded(veto_fun, np.ctypeslib.as_array(xx, (n,)).copy())))
            opt.veto_fun = C.cast(keep[-1], C.c_void_p)
        if trace_fun is not None:
            def traced(_u, damping, n, res, damp, step, rho):
                if not raised:
                    try:
                        trace_fun(damping, n, res, damp, step, rho)
                    except BaseException as e:
                        raised.append(e)
            keep.append(TRACE_FN(traced))
            opt.trace_fun = C.cast(keep[-1], C.c_void_p)
        try:
            return self._solve(x, opt, raised)
        finally:
            if trace_fun is not None:
                opt.trace_fun = None
            if term_fun is not None:
                opt.term_fun = None
            if veto_fun is not None:
                opt.veto_fun = None

    def _solve(self, x, opt, raised):
        res = Result()
        mi = opt.max_iter
        rr = np.full(mi + 3, np.nan)
        damp = np.full(2 * mi + 4, np.nan)
        aux = np.full(2 * mi + 4, np.nan)
        # (not filled: the library writes n_trace columns and only those are handed on -- filling the 22 columns of a
        # 3 M-unknown project with NaN was 50 ms of every solve)
        trace = np.empty(self.n * (mi + 2)) if opt.store_trace else None
        check(self.lib.dbat_hip_solve(self.h, C.byref(opt), dptr(x), C.byref(res), dptr(rr),
                                      dptr(damp), dptr(aux), dptr(trace)))
        if raised:
            raise raised[0]
        T = (trace[:self.n * res.n_trace].reshape(self.n, res.n_trace, order='F')
             if trace is not None else None)
        return x, res, rr[:res.n_res], damp[:res.n_damp], aux, T

    def final_residuals(self):
        ru, rw = np.empty(self.m), np.empty(self.m)
        check(self.lib.dbat_hip_final_residuals(self.h, dptr(ru), dptr(rw)))
        return ru, rw

    def forwintersect(self, x, OP, skip=None):
        """OP (3 x n_points) with the points of skip == False replaced by the forward
        intersection of their rays at the IO / EO of x."""
        x = np.ascontiguousarray(x, float)
        out = np.ascontiguousarray(np.asarray(OP, float).flatten('F'))
        sk = None if skip is None else np.ascontiguousarray(np.asarray(skip, bool).astype(np.uint8))
        check(self.lib.dbat_hip_forwintersect(self.h, dptr(x), None if sk is None else sk.ctypes.data_as(_bp), dptr(out)))
        return out.reshape(3, -1, order='F')

    def set_deterministic(self, on=True):
        """Exact (order-independent) sums into the reduced system: bit-identical runs.  DbatHipError(EUNSUPPORTED) for
        several ranks, shared EO blocks or more than nine IO columns per camera."""
        check(self.lib.dbat_hip_set_deterministic(self.h, int(bool(on))))

    def build_kernel_name(self):
        buf = C.create_string_buffer(64)
        check(self.lib.dbat_hip_build_kernel_name(self.h, buf, 64))
        return buf.value.decode()

    def comm_init(self, unique_id):
        """Join the RCCL communicator of `unique_id` (128 bytes from comm_unique_id()
        on rank 0) as rank shard_rank of shard_count.  Collective."""
        buf = (C.c_uint8 * UNIQUE_ID_BYTES).from_buffer_copy(bytes(unique_id))
        check(self.lib.dbat_hip_comm_init(self.h, buf))

    def comm_allreduce_host(self, a, op='sum'):
        a = np.ascontiguousarray(a, np.float64).copy()
        check(self.lib.dbat_hip_comm_allreduce_host(self.h, dptr(a), a.size, {'sum': 0, 'max': 1, 'min': 2}[op]))
        return a

    def set_allreduce(self, pyfunc):
        """pyfunc(ptr:int, count:int, stream:int) -> int (0 ok)."""
        def tramp(user, buf, count, stream):
            try:
                return int(pyfunc(int(buf or 0), int(count), int(stream or 0)) or 0)
            except Exception:   # noqa: BLE001 - must not unwind through C
                import traceback
                traceback.print_exc()
                return 1
        self._cb = ALLREDUCE_FN(tramp)
        check(self.lib.dbat_hip_set_allreduce(self.h, self._cb, None))

    def owned_mask(self):
        m = np.zeros(self.n, np.uint8)
        check(self.lib.dbat_hip_owned_mask(self.h, m.ctypes.data_as(_bp)))
        return m.astype(bool)

    def index_maps(self):
        """x index of every IO/EO/OP array entry (-1 = not an unknown); the
        deserial maps of misc/buildserialindices.m:204-221."""
        if getattr(self, '_index_maps', None) is None:     # a matter of the structure: once per handle (read-only arrays)
            big = 1e15
            IO, EO, OP = self.deserialize(big + np.arange(self.n, dtype=float))
            f = lambda a: np.where(a >= big / 2, np.rint(a - big), -1).astype(np.int64)
            maps = f(IO), f(EO), f(OP)
            for a in maps:
                a.setflags(write=False)
            self._index_maps = maps
        return self._index_maps

    def posterior_cov(self, x, sigma0, want_sinv=False):
        """sigma0^2 * blocks of inv(J'J) at x (bundle_cov.m): CEO (nc,6,6), CIOu
        (nIOu,nIOu; IO unknowns in z order, see x2z), COP (np,3,3) and, if asked
        for, inv(S) (NS,NS; symmetrised, not scaled by sigma0^2)."""
        x = np.ascontiguousarray(x, float)
        NS = self.info()['NS']
        nc, npnt = int(self.prob.n_images), int(self.prob.n_points)
        nIOu = NS - 6 * nc
        CEO = np.zeros(36 * nc)
        CIO = np.zeros(max(nIOu * nIOu, 1))
        COP = np.zeros(9 * npnt)
        Sinv = np.zeros(NS * NS) if want_sinv else None
        check(self.lib.dbat_hip_posterior_cov(self.h, dptr(x), float(sigma0), dptr(CEO), dptr(CIO), dptr(COP),
                                              dptr(Sinv) if want_sinv else None))
        out = [CEO.reshape(nc, 6, 6).transpose(0, 2, 1), CIO[:nIOu * nIOu].reshape(nIOu, nIOu).T,
               COP.reshape(npnt, 3, 3).transpose(0, 2, 1)]
        if want_sinv:
            L = np.tril(Sinv.reshape(NS, NS).T)
            out.append(L + np.tril(L, -1).T)
        return tuple(out)

    def set_x(self, x):
        x = np.ascontiguousarray(x, float)
        check(self.lib.dbat_hip_set_x(self.h, dptr(x)))

    def bench_step(self, lam=0.0, scale=False):
        ms = np.zeros(16)
        check(self.lib.dbat_hip_bench_step(self.h, float(lam), int(bool(scale)), dptr(ms)))
        return ms

    def chol_stats(self):
        a = (C.c_int64 * 6)()
        check(self.lib.dbat_hip_chol_stats(self.h, a))
        keys = ('order_padded', 'tile_tasks', 'tile_products', 'tile_rows', 'nested_dissection', 'dataflow')
        return dict(zip(keys, [int(v) for v in a]))

    def info(self):
        a = (C.c_int64 * 24)()
        check(self.lib.dbat_hip_info(self.h, a))
        keys = ('NS', 'n_batches', 'max_k', 'n_obs_shard', 'n_pts_shard', 'BT', 'ncolmax', 'n_tiles',
                'domain_sharding', 'reduced_doubles_per_factorisation', 'vector_doubles_per_linearisation',
                'n_top_cams', 'factor_tile_rows', 'tasks_domain', 'tasks_top', 'tile_kernel_mfma',
                'heavy_tasks', 'heavy_mfma', 'heavy_row_groups', 'heavy_scratch_bytes', 'heavy_points', 'heavy_obs',
                'heavy_ksteps_per_task', 'heavy_algorithmic_flops')
        return dict(zip(keys, [int(v) for v in a]))


def structure_key(s, device=0, shard_rank=0, shard_count=1, _problem=None):
    """Host-only: the 128-bit key of everything a plan depends on (dbat_hip_structure_key) -- equal keys: one handle
    serves both structs through Handle.set_values."""
    prob, keep = _problem if _problem is not None else problem_from_struct(s, device, shard_rank, shard_count)
    k = (C.c_uint64 * 2)()
    check(load().dbat_hip_structure_key(C.byref(prob), k))
    return (int(k[0]), int(k[1]))


# ---- handle cache: bundle() -> bundle() -> bundle_cov() on one structure build ONE plan (the reference keeps its index
# structures in s.bundle.serial / deserial, bundle.m:156-159).  One handle (it owns device memory in proportion to the
# problem), one-rank handles only; a struct whose structure key differs -- a changed mask, block, observation -- gets a
# new handle and the cached one is destroyed.
_cached = None
cache_stats = {'hits': 0, 'misses': 0}


def acquire(s, device=0):
    """A one-rank handle for s: the cached one with s's values (dbat_hip_set_values) when the structure key matches, a
    new one otherwise.  Give it back with release()."""
    global _cached
    problem = problem_from_struct(s, device)
    h, _cached = _cached, None
    if h is not None and h.h:
        if h.key() == structure_key(None, _problem=problem):
            try:
                h.set_values(_problem=problem)
                cache_stats['hits'] += 1
                cache_stats['last'] = 'hit'
                return h
            except DbatHipError:
                pass
        h.close()
    cache_stats['misses'] += 1
    cache_stats['last'] = 'miss'
    return Handle(None, _problem=problem)


def release(h, keep=True):
    """Return a handle of acquire(): kept for the next acquire() of the same structure (keep=False, or an older cached
    handle: closed)."""
    global _cached
    if not keep or h is None or not h.h:
        if h is not None:
            h.close()
        return
    if _cached is not None and _cached is not h:
        _cached.close()
    h.set_deterministic(False)
    _cached = h


def clear_cache():
    """Destroy the cached handle (frees its device memory)."""
    global _cached
    if _cached is not None:
        _cached.close()
    _cached = None


def default_options(damping='gna'):
    o = Options()
    check(load().dbat_hip_default_options(DAMP[damping.lower()], C.byref(o)))
    return o


def plan(s, shard_rank=0, shard_count=1):
    """Host-only index plan (no GPU needed): sizes of x and r, shard range."""
    lib = load()
    p, keep = problem_from_struct(s, 0, shard_rank, shard_count)
    v = [C.c_int64(0) for _ in range(7)]
    check(lib.dbat_hip_plan(C.byref(p), *[C.byref(a) for a in v]))
    keys = ('n', 'm', 'nIO', 'nEO', 'nOP', 'pt_lo', 'pt_hi')
    return dict(zip(keys, [a.value for a in v]))


def plan_structural_rank_ok(s):
    """Host-only: does J have full structural rank (sprank(J) == n)?  No GPU needed."""
    lib = load()
    p, keep = problem_from_struct(s)
    ok = C.c_int32(0)
    check(lib.dbat_hip_plan_structural_rank_ok(C.byref(p), C.byref(ok)))
    return bool(ok.value)


def plan_point_owner(s, shard_count):
    """Host-only: rank owning every object point (no GPU needed)."""
    lib = load()
    p, keep = problem_from_struct(s, 0, 0, shard_count)
    owner = np.full(s.OP.val.shape[1], -1, np.int32)
    check(lib.dbat_hip_plan_point_owner(C.byref(p), owner.ctypes.data_as(_ip)))
    return owner


def resect_poses(pt_start, X, xn, tri_start, tri, device=0):
    """dbat_hip_resect: best 3 x 4 camera matrix (n, 3, 4) and rms (n) per camera from its candidate triangles."""
    lib = load()
    n = len(pt_start) - 1
    ps = np.ascontiguousarray(pt_start, np.int64)
    ts = np.ascontiguousarray(tri_start, np.int64)
    Xf = np.ascontiguousarray(np.asarray(X, float).flatten('F'))
    xf = np.ascontiguousarray(np.asarray(xn, float).flatten('F'))
    tf = np.ascontiguousarray(np.asarray(tri, np.int32).reshape(-1))
    P, rms = np.empty(12 * n), np.empty(n)
    i64p = C.POINTER(C.c_int64)
    check(lib.dbat_hip_resect(int(device), n, ps.ctypes.data_as(i64p), dptr(Xf), dptr(xf), ts.ctypes.data_as(i64p),
                              tf.ctypes.data_as(_ip), dptr(P), dptr(rms)))
    return P.reshape(n, 4, 3).transpose(0, 2, 1), rms


def plan_layout_stats(s, shard_rank=0, shard_count=1):
    """Host-only: which layout (tiles, signature groups, chunk lengths) the plan gives the problem."""
    lib = load()
    p, keep = problem_from_struct(s, 0, shard_rank, shard_count)
    a = (C.c_int64 * 16)()
    check(lib.dbat_hip_plan_layout_stats(C.byref(p), a))
    v = [int(x) for x in a]
    return dict(n_tiles=v[0], n_batches=v[1], n_batches_tiled=v[2], n_groups=v[3], n_group_points=v[4], n_chunks=v[5],
                chunks_by_length={'1-8': v[6], '9-16': v[7], '17-32': v[8], '33-64': v[9]}, chunks_multi_round=v[10],
                k_max=v[11], rows_max=v[12], build_sig=bool(v[13]), backsub_sig=bool(v[14]), heavy_tasks=v[15])


def plan_digest(s, shard_rank=0, shard_count=1):
    """Host-only (debug): {field: hash} of the whole host plan of this shard -- equal digests mean identical plans."""
    lib = load()
    p, keep = problem_from_struct(s, 0, shard_rank, shard_count)
    out = (C.c_uint64 * 96)()
    names = C.create_string_buffer(4096)
    n = lib.dbat_hip_debug_plan_digest(C.byref(p), out, 96, names, 4096)
    if n < 0:
        check(n)
    return dict(zip(names.value.decode().split(','), [int(v) for v in out[:n]]))


def heavy_plan_selftest(s, shard_rank=0, shard_count=1):
    """Host-only (debug): the row groups / slots / pair tasks of the heavy and giant points (csrc/heavy.hpp) replayed on
    the host against the plain sum of z_p z_p' -- see dbat_hip_debug_heavy_plan_selftest."""
    lib = load()
    p, keep = problem_from_struct(s, 0, shard_rank, shard_count)
    out = np.zeros(8)
    check(lib.dbat_hip_debug_heavy_plan_selftest(C.byref(p), dptr(out)))
    return dict(on=bool(out[0]), points=int(out[1]), row_groups=int(out[2]), tasks=int(out[3]), ksteps=int(out[4]),
                max_diff=float(out[5]), max_abs=float(out[6]), entries=int(out[7]))


def plan_domain_map(s, shard_count):
    """Host-only: (cam_owner, subtree) -- the rank whose domain every image belongs to (-1: top separator) under
    domain sharding with shard_count ranks, and whether the problem is sharded that way at all."""
    lib = load()
    p, keep = problem_from_struct(s, 0, 0, shard_count)
    owner = np.full(s.EO.val.shape[1], -1, np.int32)
    sub = C.c_int32(0)
    check(lib.dbat_hip_plan_domain_map(C.byref(p), owner.ctypes.data_as(_ip), C.byref(sub)))
    return owner, bool(sub.value)


def plan_serialize(s):
    """Host-only x0 = serialize(s) (no GPU needed)."""
    lib = load()
    p, keep = problem_from_struct(s)
    x0 = np.empty(plan(s)['n'])
    check(lib.dbat_hip_plan_serialize(C.byref(p), dptr(x0)))
    return x0


def debug_model_eval_host(model, nK, nP, EO6, IO, px, Q, uv):
    """Host evaluation of csrc/model.hpp for CPU unit tests of the closed form."""
    lib = load()
    R = 5 + nK + nP
    r, A, B, Cc = np.zeros(2), np.zeros(12), np.zeros(6), np.zeros(2 * R)
    a = [np.ascontiguousarray(v, float) for v in (EO6, IO, Q, uv)]
    check(lib.dbat_hip_debug_model_eval_host(model, nK, nP, dptr(a[0]), dptr(a[1]), float(px),
                                             dptr(a[2]), dptr(a[3]), dptr(r), dptr(A), dptr(B), dptr(Cc)))
    return r, A.reshape(6, 2).T, B.reshape(3, 2).T, Cc.reshape(R, 2).T
