"""`bundle()` -- host-side mirror of DBAT's bundle driver over the HIP core.

Mirrors `bundle/bundle.m:1-132` (argument conventions), `:156-192` (set-up),
`:267-358` (dispatch on damping, result struct E, deserialise only if ok) and
`:449-491` (residual scatter, sigma0); the post-mortem of a rank-deficient
design matrix (`:368-446`) is in dbat_amd.diagnose.  All arithmetic of the path --
residuals, Jacobian blocks, normal equations, Schur solve, damping loops --
runs in libdbat_hip.so through the C ABI of include/dbat_hip.h; nothing here
computes them on the CPU.
"""
from __future__ import annotations

import copy
import time
import types

import numpy as np

from . import _hip, diagnose
from .dbatstruct import share_struct

NS = types.SimpleNamespace


class BadInput(ValueError):
    """error('DBAT:bundle:badInput', ...)  (bundle.m:124,130)."""


def _parse_args(args):
    """bundle.m:78-132: integer => maxIter, non-integer scalar => tol, string =>
    damping / flags, logical => chirality veto."""
    o = dict(maxIter=20, damping='gna', veto=False, singularTest=True, doTrace=False,
             dofVerb=False, pmDof=False, absTerm=False, convTol=1e-6)
    for a in args:
        if isinstance(a, (bool, np.bool_)):
            o['veto'] = bool(a)
        elif isinstance(a, (int, float, np.integer, np.floating)):
            if float(a) == round(float(a)):
                o['maxIter'] = int(a)
            else:
                o['convTol'] = float(a)
        elif isinstance(a, str):
            la = a.lower()
            if la in ('none', 'gm', 'gna', 'lm', 'lmp'):
                o['damping'] = la
            elif la == 'trace':
                o['doTrace'] = True
            elif la == 'singulartest':
                o['singularTest'] = True
            elif la == 'nosingulartest':
                o['singularTest'] = False
            elif la == 'pmdof':
                o['pmDof'] = True
            elif la == 'dofverb':
                o['dofVerb'] = True
            elif la == 'absterm':
                o['absTerm'] = True
            else:
                raise BadInput('DBAT:bundle:badInput: Unknown damping')
        else:
            raise BadInput('DBAT:bundle:badInput: Unknown parameter')
    return o


def bundle(s, *args, device=None, comm=None, store_trace=True, jacobian=False, deterministic=False,
           term_fun=None, veto_fun=None, reuse_handle=True):
    """[s,ok,iters,s0,E] = bundle(s[,maxIter][,damping][,'trace'][,tol]
    [,'absterm'][,'singulartest'|'nosingulartest'][,veto][,'pmdof'][,'dofverb'])

    `comm` (dbat_amd.parallel.Comm) shards the object points over the ranks of
    a torch.distributed group, one GPU per rank; every rank returns the full
    result.  `store_trace=False` drops E.trace (n x iterations) for very large
    problems.  `deterministic=True` sums the reduced system in a fixed order (dbat_hip_set_deterministic:
    two runs give the same bits; slower; signature-group path only).  `jacobian=True` also returns E.final.weighted.J and E.final.unweighted.J
    (scipy CSC, bundle.m:341-350) -- on request only, the solver never forms J.
    `term_fun(Jp, r) -> bool` replaces the termination test bundle() builds (bundle.m:186-192) and `veto_fun(x) -> bool`
    is the veto the lsa solvers call at every trial point (bundle.m:168-172 only ever passes the undefined `chirality`):
    the two function handles of the reference's solver interface, for callers that used the solvers directly.
    `reuse_handle=False` builds (and destroys) a handle of its own instead of using the cached one (_hip.acquire).
    The returned struct shares the arrays bundle() does not change (IP.*, masks, blocks) with its input.
    """
    o = _parse_args(args)
    if o['veto']:
        # bundle.m:169 references an undefined function `chirality`
        # (SURVEY Appendix B item 2): the reference errors at this point too.
        raise BadInput("chirality veto is not defined in the reference (bundle.m:169)")
    s = share_struct(s)          # (namespaces copied, arrays shared: nothing below writes into an array of the input)
    # bundle.m:137-154: a fixed parameter cannot be used as a prior observation
    for nm in ('IO', 'EO', 'OP'):
        pr = getattr(s.prior, nm)
        est = np.asarray(getattr(s.bundle.est, nm), bool)
        pr.use = np.asarray(pr.use, bool)
        if np.any(pr.use & ~est):
            print("Warning: Some %s parameters are set to both 'fixed' and 'observed'" % nm)
            print('Setting %s parameters to fixed' % nm)
            pr.use = pr.use & est
    rank, world = (comm.rank, comm.world_size) if comm is not None else (0, 1)
    if jacobian and world > 1:
        raise BadInput('bundle(..., jacobian=True): the explicit Jacobian (E.final.*.J) is built by one-rank handles only')
    if device is None:
        # a rank of a multi-GPU run works on the device its process selected
        device = getattr(comm, 'device', None) if comm is not None else 0
        device = 0 if device is None else device
    # Plan reuse (bundle.m:156-159 keeps s.bundle.serial / deserial between calls): one-rank runs take the cached handle
    # when the structure of s -- everything but the parameter and prior values -- is the one it was built for
    # (_hip.acquire: dbat_hip_structure_key + dbat_hip_set_values); a changed mask, block or observation builds a new plan.
    cached = world == 1 and reuse_handle
    t_host = [time.perf_counter()]
    h = _hip.acquire(s, device) if cached else _hip.Handle(s, device=device, shard_rank=rank, shard_count=world)
    t_host.append(time.perf_counter())
    done = False
    try:
        if deterministic:                                            # fixed-order sums: bit-identical runs (parity mode)
            h.set_deterministic(True)
        if comm is not None and world > 1:
            if hasattr(comm, 'attach'):
                comm.attach(h)                                       # RCCL communicator inside the library
            else:
                h.set_allreduce(comm.allreduce_ptr)                  # test hook
        x0 = h.serialize()                                           # bundle.m:162
        opt = _hip.default_options(o['damping'])
        opt.max_iter = o['maxIter']
        opt.conv_tol = o['convTol']
        opt.abs_term = int(o['absTerm'])
        opt.singular_test = int(o['singularTest'])
        opt.store_trace = int(bool(store_trace))
        # 'trace': the solver's own line per iteration, printed while the loop runs (gauss_newton_armijo.m:119-128 ...)
        live = (lambda *a: print(_hip.trace_text(*a), flush=True)) if o['doTrace'] and rank == 0 else None
        x, res, rr, damp, aux, T = h.solve(x0, opt, term_fun=term_fun, veto_fun=veto_fun, trace_fun=live)   # complete on every rank of a sharded run
        t_host.append(time.perf_counter())
        E = NS(maxIter=o['maxIter'], convTol=o['convTol'], absTerm=o['absTerm'],
               singularTest=o['singularTest'], chirality=False)
        name = 'gm' if o['damping'] in ('none', 'gm') else o['damping']
        if name == 'gm':
            E.damping = NS(name='gm')
        elif name == 'gna':
            E.damping = NS(name='gna', alpha=damp, mu=opt.mu, alphaMin=opt.alpha_min)
        elif name == 'lm':
            E.damping = NS(name='lm', **{'lambda': damp},
                           lambda0=damp[0] if len(damp) else np.nan,
                           lambdaMin=damp[0] if len(damp) else np.nan)
        else:
            mi = o['maxIter']
            rho = aux[:mi + 2]
            rho = rho[~np.isnan(rho)]
            step = aux[mi + 2:]
            step = step[~np.isnan(step)].astype(int)
            E.damping = NS(name='lmp', delta=damp, rho=rho, delta0=float(np.linalg.norm(x0)),
                           rhoBad=opt.rho_bad, rhoGood=opt.rho_good, step=step)
        E.res, E.trace, E.time = rr, T, res.time_s
        # where it went (hipEvent stage timers of the library), the E.time of bundle.m:287-294 by stage
        E.timeStages = dict(zip(('linearise', 'factor_solve', 'backsub', 'residual', 'other'), [float(v) for v in res.stage_s]))
        E.code, E.usedIters = int(res.code), int(res.iters)
        E.counters = NS(residual_evals=res.n_residual_evals, linearizations=res.n_linearizations,
                        solves=res.n_solves)
        ok = E.code == 0
        if ok:                                                       # bundle.m:356-358
            IO, EO, OP = h.deserialize(x)
            s.IO.val, s.OP.val = IO, OP
            s.EO.val = np.vstack([EO, s.EO.val[6:]]) if s.EO.val.shape[0] > 6 else EO
        # residuals at the last linearisation point (bundle.m:449-460)
        # (code -4 stops after the first linearisation: the residual is that of x0,
        # gauss_newton_armijo.m:112-142, and sigma0 below is computed from it)
        ru, rw = h.final_residuals()
        no = s.IP.val.shape[1]
        s.post = getattr(s, 'post', NS())
        s.post.res = NS()
        px = np.asarray(s.IO.sensor.pxSize)
        if px.shape[1] == 1 or not (px != px[:, :1]).any():          # one pixel size for every camera: no 2 x nObs gather
            s.post.res.IP = ru[:2 * no].reshape(2, no, order='F') / px[:, :1]
        else:
            s.post.res.IP = ru[:2 * no].reshape(2, no, order='F') / px[:, s.IP.cam]
        IOix, EOix, OPix = h.index_maps()
        ofs = 2 * no
        for nm, ixmap in (('IO', IOix), ('EO', EOix), ('OP', OPix)):
            val = getattr(s, nm).val
            rows = slice(0, 6) if nm == 'EO' else slice(None)
            use = np.asarray(getattr(s.prior, nm).use, bool)[rows]
            arr = np.full(val[rows].shape, np.nan)
            if not use.any():                                        # no prior observation of this kind: nothing to place
                setattr(s.post.res, nm, arr)                         # (the leading-element map below sorts 3 x points entries)
                continue
            # prior rows = column-major order of use & leading (buildserialindices.m:138-139,200)
            flatmap = ixmap.flatten('F')
            lead = np.zeros(flatmap.shape, bool)
            valid = np.flatnonzero(flatmap >= 0)
            _, first = np.unique(flatmap[valid], return_index=True)
            lead[valid[first]] = True
            pos = np.flatnonzero(use.flatten('F') & lead)
            flat = arr.flatten('F')
            flat[pos] = ru[ofs:ofs + len(pos)]
            ofs += len(pos)
            setattr(s.post.res, nm, flat.reshape(arr.shape, order='F'))
        E.final = NS(unweighted=NS(r=ru), weighted=NS(r=rw))
        if jacobian and E.code != -4:
            E.final.weighted.J = h.jacobian_csc(x, True)
            E.final.unweighted.J = h.jacobian_csc(x, False)
        p_extra = 0
        if o['pmDof']:                                               # bundle.m:467-471
            seen_pt = np.zeros(s.OP.val.shape[1], bool); seen_pt[s.IP.pt] = True
            seen_cam = np.zeros(s.EO.val.shape[1], bool); seen_cam[s.IP.cam] = True
            p_extra = int(np.count_nonzero(~np.asarray(s.bundle.est.OP, bool)[:, seen_pt])
                          + np.count_nonzero(~np.asarray(s.bundle.est.EO, bool)[:6, seen_cam]))
        dof = h.m + p_extra - h.n
        # bundle.m:476-483: sqrt(r'r/dof) of the weighted residual (finite for m == n with 'pmdof')
        s0 = float(np.sqrt(rw @ rw / dof)) if dof > 0 else np.nan
        if o['dofVerb']:
            print('bundle: dof=%d+%d-%d=%d.' % (h.m, p_extra, h.n, dof))
        s.post.sigmas = s0 * np.asarray(s.IP.sigmas)
        # sensor format updated by the estimated aspect (bundle.m:360-366)
        aspect = np.ones((2, s.IO.val.shape[1])); aspect[0] = 1.0 + s.IO.val[3]
        if hasattr(s.IO.sensor, 'imSize'):               # structs built without image sizes have no format to update
            s.post.sensor = type(s.post)(imSize=np.array(s.IO.sensor.imSize, float), pxSize=s.IO.sensor.pxSize * aspect,
                                          ssSize=s.IO.sensor.imSize * s.IO.sensor.pxSize * aspect)
        E.numObs, E.numParams, E.redundancy, E.s0 = h.m, h.n, dof, s0
        E.sigmas = s.post.sigmas
        E.x = x
        # post-mortem of a rank-deficient design matrix (bundle.m:368-446)
        maps = (IOix, EOix, OPix)
        E.paramTypes = diagnose.param_types(s, maps, h.n)
        E.weakness = NS(structural=None, numerical=NS(rank=h.n, deficiency=0))
        if E.code == -2:
            E.weakness.numerical = NS(rank=float('nan'), deficiency=float('nan'), suspectedParams=[])
            if world == 1:
                E.weakness.numerical = diagnose.numerical_weakness(
                    diagnose.weighted_jacobian(h, s, x), E.paramTypes)
        elif E.code == -4:
            E.weakness.structural = diagnose.structural_weakness(s, maps, h.n, E.paramTypes)
            E.weakness.numerical = NS(rank=float('nan'), deficiency=float('nan'))
        done = True
        # where the wall time of this call went on the host: the handle (a plan and its uploads, or -- a cached handle of
        # the same structure -- the key and the new values), the solve (E.time of it inside the damping loop), the result
        E.timeHost = dict(handle=t_host[1] - t_host[0], solve=t_host[2] - t_host[1], result=time.perf_counter() - t_host[2],
                          handle_reused=bool(cached and _hip.cache_stats.get('last') == 'hit'))
        return s, ok, E.usedIters, s0, E
    finally:
        if cached:
            _hip.release(h, keep=done)                               # (after an exception: destroyed, not kept)
        else:
            h.close()


CXX_MAX_N = 6000       # unknowns up to which bundle_cov offers the dense matrices 'CXX' / 'COPF' (288 MB)


def bundle_cov(s, E, *names, device=0):
    """C = bundle_cov(s, E, 'CIO' | 'CEO' | 'COP' | 'CIOF' | 'CEOF', ...)
    (bundle/bundle_cov.m:1-31): sigma0^2 times blocks of inv(J'J) at the
    bundle result (s, E) as scipy sparse matrices of size numel(val) x
    numel(val), zero-padded for elements that were not estimated; 'CIO',
    'CEO', 'COP' keep the per-column diagonal blocks only (:9-16), 'CIOF' and
    'CEOF' are the full component matrices.  The blocks come from the device
    (dbat_hip_posterior_cov: Schur pieces and a selected inversion of the compact
    factor); this function only scatters them.
    'CXX' (n x n, not zero-padded) and 'COPF' (3np x 3np) are dense by nature
    ("may require a lot of memory", bundle_cov.m:24): offered up to CXX_MAX_N
    unknowns -- the sizes the reference's own callers use them at -- from the
    weighted Jacobian of the device (dbat_hip_jacobian_csc) by one dense
    factorisation on the host, as bundle_cov.m:63-117 does it in MATLAB."""
    import scipy.sparse as sp
    names = [n.lower() for n in names]
    for n in names:
        if n not in ('cio', 'ceo', 'cop', 'ciof', 'ceof', 'cxx', 'copf'):
            raise BadInput("Bad covariance string '%s'" % n)          # bundle_cov.m:53-55
    if not names:
        return None
    dense = [n for n in names if n in ('cxx', 'copf')]
    h = _hip.acquire(s, device)          # the handle bundle() left behind, with the values of its result: one plan for both
    done = False
    try:
        if dense and h.n > CXX_MAX_N:
            raise BadInput("bundle_cov: '%s' is an n x n dense matrix and n = %d (offered up to %d unknowns; "
                           "'CIO', 'CEO', 'COP' give the blocks at any size)" % (dense[0].upper(), h.n, CXX_MAX_N))
        CXX = None
        if dense:
            J = h.jacobian_csc(np.asarray(E.x, float), True)
            N = (J.T @ J).toarray()
            import scipy.linalg as sla
            c, low = sla.cho_factor(N, lower=True)           # (fails loudly if J'J is not positive definite)
            CXX = float(E.s0) ** 2 * sla.cho_solve((c, low), np.eye(N.shape[0]))
        blocks = [n for n in names if n not in dense]
        res = None
        if blocks:
            full = any(n.endswith('f') for n in blocks)
            res = h.posterior_cov(np.asarray(E.x, float), float(E.s0), want_sinv=full)
            CEOb, CIOu, COPb = res[:3]
        ixIO, ixEO, ixOP = h.index_maps()                   # x index of every array entry, -1 = no unknown
        nc = s.EO.val.shape[1]
        done = True
    finally:
        _hip.release(h, keep=done)
    out = []
    for n in names:
        if n == 'cxx':
            out.append(CXX)
            continue
        if n == 'copf':                                      # zero-padded: rows / columns of coordinates that are not estimated
            ix = ixOP.flatten('F')
            est = np.flatnonzero(ix >= 0)
            D = np.zeros((ix.size, ix.size))
            D[np.ix_(est, est)] = CXX[np.ix_(ix[est], ix[est])]
            out.append(sp.csc_matrix(D))
            continue
        comp = n[1:3].upper()
        val = getattr(s, comp).val
        m, ncol = val.shape
        if n == 'cop':
            r = np.arange(3)
            rows = (3 * np.arange(ncol)[:, None, None] + r[None, :, None]) + 0 * r[None, None, :]
            cols = (3 * np.arange(ncol)[:, None, None] + r[None, None, :]) + 0 * r[None, :, None]
            C = sp.csc_matrix((COPb.ravel(), (rows.ravel(), cols.ravel())), shape=(val.size, val.size))
        elif n == 'ceo':
            r = np.arange(6)
            rows = (m * np.arange(ncol)[:, None, None] + r[None, :, None]) + 0 * r[None, None, :]
            cols = (m * np.arange(ncol)[:, None, None] + r[None, None, :]) + 0 * r[None, :, None]
            C = sp.csc_matrix((CEOb.ravel(), (rows.ravel(), cols.ravel())), shape=(val.size, val.size))
        else:
            # IO (shared blocks: several array entries map to one unknown) and the full matrices:
            # entry (a, b) of the result = covariance of the unknowns behind array entries a and b
            ix = (ixIO if comp == 'IO' else ixEO).flatten('F')
            if comp == 'IO':
                M, off = CIOu * 1.0, 0
                src = ix                                     # x index == IO unknown index (IO comes first in x)
            else:
                Sinv = res[3]
                M, off = float(E.s0) ** 2 * Sinv[:6 * nc, :6 * nc], 0
                # EO entry e of the array <-> z index: row r of image c -> 6c + r (rows 0..5)
                src = np.where(ix >= 0, (np.arange(val.size) % m) + 6 * (np.arange(val.size) // m), -1)
                src = np.where((np.arange(val.size) % m) < 6, src, -1)
            if n == 'ciof' and len(res) > 3:
                M = float(E.s0) ** 2 * res[3][6 * nc:, 6 * nc:]
            est = np.flatnonzero(src >= 0)
            D = np.zeros((val.size, val.size))
            if len(est):
                D[np.ix_(est, est)] = M[np.ix_(src[est] - off, src[est] - off)]
            if not n.endswith('f'):
                D *= np.kron(np.eye(ncol), np.ones((m, m)))
            C = sp.csc_matrix(D)
        out.append(C)
    return out[0] if len(out) == 1 else tuple(out)
