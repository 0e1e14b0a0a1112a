"""dbat_amd -- MI355X-native damped bundle adjustment behind DBAT's bundle() API.

Host-side mirror of the reference's interface for the bundle hot path:
    dbatstruct   the DBAT struct (prob2dbatstruct.m field layout)
    driver       bundle(s, ...) -> (s, ok, iters, sigma0, E); bundle_cov(s, E, 'CIO','CEO','COP')
    loadpm       PhotoModeler export loader (known-answer fixtures)
    initial      resect / forwintersect: initial EO and OP (photogrammetry/resect.m, forwintersect.m)
    report       the result file (bundle_result_file.m)
    diagnose     post-mortem of rank-deficient problems (bundle.m:368-446)
    parallel     torch.distributed / RCCL plumbing for sharded object points
    _hip         ctypes binding of include/dbat_hip.h (libdbat_hip.so)
"""
from .dbatstruct import make_struct, seteoest_depend, validate  # noqa: F401
from .driver import bundle, bundle_cov, BadInput  # noqa: F401
