"""ctypes binding of bench/libdbat_cpuref.so -- the C++/OpenMP CPU baseline
(bench/cpu_ref.cpp: explicit sparse J, J'J, supernodal sparse Cholesky of the
full normal matrix; levenberg_marquardt.m:81-82,119).  Used by bench.py's
`cpu_baseline` leg and tests/test_cpu_ref.py only."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


class _Problem(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ('nc', 'np', 'no', 'nIOrows', 'model', 'nK', 'nP', 'reserved')] + \
               [(k, C.c_void_p) for k in ('cam', 'pt', 'uv', 'std', 'px', 'IO', 'EO', 'OP',
                                          'estIO', 'estEO', 'estOP', 'IOblock', 'useIO', 'useEO', 'useOP',
                                          'priorIO', 'priorEO', 'priorOP', 'stdIO', 'stdEO', 'stdOP')]


def load():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, 'libdbat_cpuref.so')
        if not os.path.exists(path):
            raise RuntimeError('%s is missing: run `make -C bench` (or __graft_entry__.build())' % path)
        lib = C.CDLL(path)
        lib.cpuref_create.restype = C.c_void_p
        lib.cpuref_create.argtypes = [C.POINTER(_Problem), C.c_int32, C.POINTER(C.c_double)]
        lib.cpuref_destroy.argtypes = [C.c_void_p]
        lib.cpuref_num_params.restype = C.c_int64
        lib.cpuref_num_params.argtypes = [C.c_void_p]
        lib.cpuref_num_threads.restype = C.c_int32
        lib.cpuref_num_threads.argtypes = [C.c_void_p]
        lib.cpuref_nnz.restype = C.c_int64
        lib.cpuref_nnz.argtypes = [C.c_void_p, C.c_int32]
        lib.cpuref_serialize.argtypes = [C.c_void_p, C.c_void_p]
        lib.cpuref_lm_step.restype = C.c_int32
        lib.cpuref_lm_step.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
        lib.cpuref_step_norms.restype = C.c_int32
        lib.cpuref_step_norms.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = lib
    return _lib


class CpuRef:
    """One problem (a DBAT struct; prior observations of estimated parameters are rows of J, lsa/prior_obs.m:45-72) on the CPU."""

    PHASES = ('residual_jacobian', 'JtJ', 'Jtr', 'chol_leaves', 'chol_root_update', 'chol_root', 'solve',
              'trial_residual')

    def __init__(self, s, threads=0):
        lib = load()
        f64 = lambda a: np.ascontiguousarray(np.asarray(a, np.float64).flatten('F'))
        u8 = lambda a: np.ascontiguousarray(np.asarray(a, bool).flatten('F').astype(np.uint8))
        i32 = lambda a: np.ascontiguousarray(np.asarray(a).flatten('F').astype(np.int32))
        self._keep = dict(cam=i32(s.IP.cam), pt=i32(s.IP.pt), uv=f64(s.IP.val), std=f64(s.IP.std),
                          px=f64(s.IO.sensor.pxSize[0]), IO=f64(s.IO.val), EO=f64(s.EO.val[:6]), OP=f64(s.OP.val),
                          estIO=u8(s.bundle.est.IO), estEO=u8(np.asarray(s.bundle.est.EO)[:6]),
                          estOP=u8(s.bundle.est.OP), IOblock=i32(s.IO.struct.block))
        nz = lambda a: np.nan_to_num(np.asarray(a, np.float64))
        for nm, rows in (('IO', slice(None)), ('EO', slice(0, 6)), ('OP', slice(None))):
            pr = getattr(s.prior, nm)
            use = np.asarray(pr.use, bool)[rows] & np.asarray(getattr(s.bundle.est, nm), bool)[rows]
            self._keep['use' + nm] = u8(use)
            self._keep['prior' + nm] = f64(nz(pr.val)[rows])
            self._keep['std' + nm] = f64(np.where(use, nz(pr.std)[rows], 1.0))
        pb = _Problem()
        pb.nc, pb.np, pb.no = s.EO.val.shape[1], s.OP.val.shape[1], s.IP.val.shape[1]
        pb.nIOrows, pb.nK, pb.nP = s.IO.val.shape[0], int(s.IO.model.nK), int(s.IO.model.nP)
        pb.model = int(np.unique(s.IO.model.distModel)[0])
        for k, a in self._keep.items():
            setattr(pb, k, a.ctypes.data)
        ms = C.c_double(0)
        self._h = lib.cpuref_create(C.byref(pb), int(threads), C.byref(ms))
        if not self._h:
            raise RuntimeError('cpuref_create failed')
        self.setup_ms = ms.value
        self.n = int(lib.cpuref_num_params(self._h))
        self.threads = int(lib.cpuref_num_threads(self._h))
        self.nnz = {k: int(lib.cpuref_nnz(self._h, i)) for i, k in enumerate(('J', 'JtJ_lower', 'L'))}

    def serialize(self):
        x = np.empty(self.n)
        load().cpuref_serialize(self._h, x.ctypes.data)
        return x

    def lm_step(self, x, lam):
        """(p, stats): p = (J'J + lam I) \\ (-J'r) at x; lam < 0 means |lam| trace(J'J)/n."""
        x = np.ascontiguousarray(x, np.float64)
        p = np.empty(self.n)
        st = np.zeros(16)
        rc = load().cpuref_lm_step(self._h, x.ctypes.data, float(lam), p.ctypes.data, st.ctypes.data)
        out = dict(code=int(rc), f=st[0], f_trial=st[1], trace=st[2], lam=st[3],
                   ms=dict(zip(self.PHASES, st[4:12].tolist())))
        return p, out

    def step_norms(self, v):
        """(|J v|^2, r'J v, v'v) with J and r of the last lm_step."""
        v = np.ascontiguousarray(v, np.float64)
        out = np.zeros(3)
        if load().cpuref_step_norms(self._h, v.ctypes.data, out.ctypes.data) != 0:
            raise RuntimeError('cpuref_step_norms failed')
        return float(out[0]), float(out[1]), float(out[2])

    def close(self):
        if self._h:
            load().cpuref_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001
            pass
