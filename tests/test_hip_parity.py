"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on
the same inputs, against the reference's committed known answers, and through
size-independent properties at the benchmark's sizes.

Tolerances (north_star: parameter updates and converged parameters within
1e-6 relative of the reference's full-matrix solve):
  residuals, Jacobian blocks, gradient, column norms : 1e-11 relative
  update p of one linearise+solve                    : 1e-8 relative (bar 1e-6)
  converged x                                        : 1e-8 relative (bar 1e-6)
"""
import copy
import os

import numpy as np
import pytest
import scipy.sparse as sp

import dbat_oracle as o
from helpers import (camcal_struct, camcal_expected, check_camcal_against_report, synth_struct,
                     relerr, roma_struct, roma_expected, check_roma_against_result, lm_count_is_stable)

pytestmark = pytest.mark.gpu

TOL_BLOCK = 1e-11
TOL_STEP = 1e-8
TOL_X = 1e-7          # converged parameters (stated bar 1e-6; LM stops on a 1e-6 relative step, so the
                      # last digits depend on the summation order of the atomics)


@pytest.fixture(scope='module')
def hip():
    from dbat_amd import _hip
    import torch
    assert torch.cuda.is_available(), 'gpu tests need a GPU'
    _hip.load()
    return _hip


def cases():
    out = [('camcal3', lambda: camcal_struct(3)), ('camcal2', lambda: camcal_struct(2)),
           ('camcal4', lambda: camcal_struct(4)), ('camcal5', lambda: camcal_struct(5))]
    for v in ('plain', 'selfcal', 'imagevar', 'priors', 'groups4'):
        out.append(('tiny-' + v, (lambda v=v: synth_struct('tiny', v)[0])))
    out.append(('small-plain', lambda: synth_struct('small', 'plain')[0]))
    out.append(('small-priors', lambda: synth_struct('small', 'priors')[0]))
    out.append(('small-groups4', lambda: synth_struct('small', 'groups4')[0]))
    return out


def noise_tail_start(res):
    """First index from which the residual norm no longer changes by more than
    its rounding error."""
    res = np.asarray(res)
    for k in range(1, len(res)):
        if abs(res[k] - res[k - 1]) <= 1e-11 * res[k]:
            return k
    return len(res)


def check_history(E, Eo, iters, ito, damping, s=None):
    """Iteration histories must match; for 'lm' the COUNT only where it is a property of the problem.

    levenberg_marquardt.m terminates only after an ACCEPTED undamped step
    (:177,:217); once converged, "fNew<f" compares objective values that
    differ by less than their rounding error, so the number of trailing
    trials is arithmetic noise in the reference itself: the oracle's own
    count changes when the rows of r and J are merely summed in another order
    (helpers.oracle_lm_reordered: tiny-plain 7 / 4 / 21 iterations, camcal
    model 3 6 six times and 16 the seventh).  helpers.lm_decision_margins
    measures it: the smallest relative margin |fNew - f| / f of any accept /
    reject decision is 1e-13 ... 1e-16 in EVERY case these tests use
    (tests/test_abi_cpu.py::test_lm_count_stability_helper), self-calibrating
    or not.  The count is asserted where the margins are large
    (helpers.lm_count_is_stable: at convTol = 1e-6 nowhere -- the COUNT is asserted by
    test_lm_iteration_count_where_it_is_a_property_of_the_problem, with convTol = 1e-3);
    the iterates, residual norms and lambdas are compared up to the noise tail
    everywhere, at the tolerance used everywhere else."""
    if damping != 'lm':
        assert iters == ito
        assert len(E.res) == len(Eo.res) and relerr(E.res, Eo.res) < 1e-8
        return
    k = min(noise_tail_start(Eo.res), noise_tail_start(E.res))
    # bench/lm_history.py on MI355X: iterates, residual norms and lambdas of the GPU and of the oracle
    # agree to <= 1.3e-11 at every LM iteration of camcal (all lens models) and of the self-calibrating
    # synthetic scenes (the largest difference sits at the first long step, where the rounding of two
    # different elimination orders is amplified by cond(J'J + lambda I)); 1e-8 leaves three digits
    tol = 1e-8
    assert relerr(E.res[:k], Eo.res[:k]) < tol
    lam, lamo = E.damping.__dict__['lambda'], Eo.damping.__dict__['lambda']
    assert relerr(lam[:k], lamo[:k]) < tol
    if s is not None and lm_count_is_stable(s, ito):
        assert iters == ito, 'LM iteration count %d, oracle %d (stable under re-ordered summation)' % (iters, ito)


def oracle_setup(s):
    s = copy.deepcopy(s)
    for nm in ('IO', 'EO', 'OP'):
        pr = getattr(s.prior, nm)
        pr.use = np.asarray(pr.use, bool) & np.asarray(getattr(s.bundle.est, nm), bool)
    s = o.buildserialindices(s)
    return s, o.serialize(s), o.buildweightvector(s)


def assemble_J(h, s, JEO, JOP, JIO):
    """Sparse J from the per-observation blocks + index maps (multi_res.m:300-313)."""
    IOix, EOix, OPix = h.index_maps()
    no = s.IP.val.shape[1]
    rows, cols, vals = [], [], []
    r0 = 2 * np.arange(no)
    for blk, ix in ((JEO, EOix[:, s.IP.cam]), (JOP, OPix[:, s.IP.pt]), (JIO, IOix[:, s.IP.cam])):
        k = blk.shape[2]
        for c in range(k):
            m = ix[c] >= 0
            for rr in range(2):
                rows.append(r0[m] + rr); cols.append(ix[c][m]); vals.append(blk[m, rr, c])
    J = sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                      shape=(2 * no, h.n))
    return J


@pytest.mark.parametrize('name,make', cases(), ids=[c[0] for c in cases()])
def test_residual_jacobian_parity(hip, name, make):
    s = make()
    so, x0, w = oracle_setup(s)
    h = hip.Handle(s)
    try:
        assert h.n == len(x0) and h.m == so.post.res.ix.n
        assert np.array_equal(h.serialize(), x0)
        rng = np.random.default_rng(5)
        x = x0 + 1e-3 * rng.standard_normal(len(x0)) * np.maximum(1e-3, np.abs(x0)) * 1e-2
        r_o, J_o = o.brown_euler_cam4(x, so, jac=True)
        r_h, f_h = h.residual(x)
        assert relerr(r_h, r_o) < TOL_BLOCK
        assert abs(f_h - 0.5 * np.sum(w * r_o ** 2)) <= 1e-11 * abs(f_h)
        JEO, JOP, JIO = h.jacobian_blocks(x)
        J_h = assemble_J(h, s, JEO, JOP, JIO)
        no2 = 2 * s.IP.val.shape[1]
        Jd = (J_h - J_o[:no2]).tocsc()
        assert abs(Jd).max() <= TOL_BLOCK * abs(J_o).max()
    finally:
        h.close()


@pytest.mark.parametrize('name,make', cases(), ids=[c[0] for c in cases()])
def test_step_parity(hip, name, make):
    """One linearisation + solve: Schur-complement solve on the GPU vs the
    oracle's full sparse normal-equation solve (F11)."""
    s = make()
    so, x0, w = oracle_setup(s)
    R = np.sqrt(w)
    r_o, K = o.brown_euler_cam4(x0, so, jac=True)
    r = R * r_o
    J = (sp.diags(R) @ K).tocsc()
    h = hip.Handle(s)
    try:
        # scaled Gauss-Newton step (gauss_newton_armijo.m:166-174)
        p_o, sing, Jn, Jn2, Hs, gs, Js = o._scaled_gn(J, r)
        p_h, st = h.linearize_solve(x0, 0.0, True)
        assert not st['singular']
        assert relerr(p_h, p_o) < TOL_STEP
        assert abs(st['f'] - 0.5 * r @ r) <= 1e-11 * st['f']
        Jp = J @ p_o
        assert abs(st['JpJp'] - Jp @ Jp) <= 1e-7 * (Jp @ Jp)
        assert abs(st['rJp'] - r @ Jp) <= 1e-7 * abs(r @ Jp)
        assert abs(st['pp'] - p_o @ p_o) <= 1e-7 * (p_o @ p_o)
        assert relerr(h.gradient(), J.T @ r) < 1e-10
        assert relerr(h.colnorms(), Jn) < 1e-10
        v = np.random.default_rng(2).standard_normal(len(x0))
        Jv = J @ v
        assert abs(h.jtimes_sqnorm(v) - Jv @ Jv) <= 1e-10 * (Jv @ Jv)
        assert relerr(h.jtimes(v), Jv) < 1e-10           # J v itself, in the reference's row order (termFun's first argument)
        # damped, unscaled step (levenberg_marquardt.m:119)
        JTJ = (J.T @ J).tocsc()
        lam = 1e-4 * JTJ.diagonal().sum() / J.shape[1]
        q_o, _ = o.normal_solve((JTJ + lam * sp.identity(J.shape[1])).tocsc(), -(J.T @ r))
        q_h, st2 = h.linearize_solve(x0, lam, False)
        assert relerr(q_h, q_o) < TOL_STEP
        assert abs(st2['trace'] - JTJ.diagonal().sum()) <= 1e-10 * st2['trace']
    finally:
        h.close()


@pytest.mark.parametrize('damping', ['gna', 'lm', 'lmp', 'gm'])
@pytest.mark.parametrize('model', [2, 3, 4, 5])
def test_camcal_known_answer_hip(hip, model, damping):
    """Reference's committed camcal reports (data/dbat/dbatexports/
    camcal-dbatreport{,-model*}.txt) reproduced through bundle() on the GPU."""
    from dbat_amd import bundle
    exp = camcal_expected()['model%d' % model]
    res, ok, iters, s0, E = bundle(camcal_struct(model), damping)
    assert ok and E.code == 0
    check_camcal_against_report(res, s0, E, exp)
    ro, oko, ito, s0o, Eo = o.bundle(camcal_struct(model), damping)
    check_history(E, Eo, iters, ito, damping, s=camcal_struct(model))
    assert relerr(E.x, Eo.x) < TOL_X
    assert abs(s0 - s0o) < 1e-9 * s0o
    if damping != 'lm':
        assert relerr(E.trace, Eo.trace) < 1e-7
    else:       # (LM's last steps are rounding: the columns agree to what its stopping rule leaves; the trace comes down from
                # the device in one piece at the end of the loop -- same columns, same gaps, first = x0, last = x)
        assert np.array_equal(E.trace[:, -1], E.x) and relerr(E.trace[:, 0], Eo.trace[:, 0]) < 1e-12
        if iters == ito:        # (near the end LM accepts or rejects by the last bits of f: the COUNT may differ, check_history allows it)
            assert E.trace.shape == Eo.trace.shape and np.array_equal(np.isnan(E.trace), np.isnan(Eo.trace))
            fin = ~np.isnan(Eo.trace)
            assert relerr(E.trace[fin], Eo.trace[fin]) < 1e-5
    assert relerr(res.post.res.IP, ro.post.res.IP) < 1e-6


@pytest.mark.parametrize('damping', ['gna', 'lm', 'lmp', 'gm'])
@pytest.mark.parametrize('variant', ['plain', 'selfcal', 'imagevar', 'priors', 'groups4'])
def test_synthetic_bundle_parity(hip, variant, damping):
    from dbat_amd import bundle
    s, truth = synth_struct('tiny', variant)
    res, ok, iters, s0, E = bundle(s, damping)
    ro, oko, ito, s0o, Eo = o.bundle(s, damping)
    assert ok == oko and E.code == Eo.code
    assert relerr(E.x, Eo.x) < TOL_X
    assert abs(s0 - s0o) < 1e-9 * s0o
    # levenberg_marquardt.m only terminates after an ACCEPTED undamped step
    # (:177,:217); once converged, "fNew<f" compares objective values that
    # differ by less than their rounding error, so the number of trailing
    # rejected trials is arithmetic noise in the reference algorithm itself.
    # Compare the iteration history up to that point.
    check_history(E, Eo, iters, ito, damping, s=s)
    if damping == 'gna':
        assert np.array_equal(E.damping.alpha, Eo.damping.alpha)
    if damping == 'lmp':
        assert np.array_equal(E.damping.step, Eo.damping.step)
        assert relerr(E.damping.delta, Eo.damping.delta) < 1e-9
        # rho = actual/predicted cancels catastrophically close to convergence
        assert np.abs(E.damping.rho - Eo.damping.rho).max() < 1e-3
    for nm in ('IP', 'EO', 'OP'):
        a, b = getattr(res.post.res, nm), getattr(ro.post.res, nm)
        assert np.array_equal(np.isnan(a), np.isnan(b))
        m = ~np.isnan(b)
        if m.any():
            assert relerr(a[m], b[m]) < 1e-6
    assert E.numParams == Eo.numParams and E.numObs == Eo.numObs and E.redundancy == Eo.redundancy


@pytest.mark.parametrize('name', ['tiny-plain', 'tiny-selfcal', 'tiny-priors', 'camcal3', 'camcal5', 'small-plain', 'small-priors'])
def test_lm_iteration_count_where_it_is_a_property_of_the_problem(hip, name):
    """levenberg_marquardt.m:177-217 as a COUNT.  At the default convTol = 1e-6 the last accept / reject decisions of LM
    are taken inside the rounding error of f (check_history), so no test above can assert the number of iterations.  With
    convTol = 1e-3 the loop stops at an accepted undamped step well before that: every decision `fNew < f` of the
    oracle's run has a relative margin of 1e-10 ... 1e-7 and the termination ratio is a factor 3 ... 70 from 1
    (helpers.lm_decision_margins) -- three to six orders above the 1e-12 by which the device's objective values differ
    from the oracle's.  There the count, the whole residual history and every lambda must be the oracle's."""
    from dbat_amd import bundle
    from helpers import lm_decision_margins
    s = dict(cases())[name]()
    n_o, margin, term = lm_decision_margins(s, conv_tol=1e-3)
    assert margin > 1e-10 and term > 1.5, 'not a case for this test any more: margin %.1e, termination factor %.2f' % (margin, term)
    res, ok, iters, s0, E = bundle(s, 'lm', 1e-3)
    ro, oko, ito, s0o, Eo = o.bundle(s, 'lm', 1e-3)
    assert ito == n_o
    assert ok == oko and E.code == Eo.code == 0
    assert iters == ito, 'LM iterations: device %d, oracle %d' % (iters, ito)
    assert len(E.res) == len(Eo.res) and relerr(E.res, Eo.res) < 1e-8
    lam, lamo = E.damping.__dict__['lambda'], Eo.damping.__dict__['lambda']
    assert len(lam) == len(lamo) and relerr(lam, lamo) < 1e-8
    assert relerr(E.x, Eo.x) < TOL_X and abs(s0 - s0o) < 1e-9 * s0o
    # the objective values themselves: the margin argument above rests on this agreement
    assert np.abs(np.asarray(E.res) - np.asarray(Eo.res)).max() <= 1e-11 * np.asarray(Eo.res).max()


@pytest.mark.parametrize('damping', ['gm', 'gna', 'lm', 'lmp'])
@pytest.mark.parametrize('name', ['tiny-priors', 'camcal3', 'small-plain'])
def test_caller_supplied_term_and_veto_functions(hip, name, damping):
    """The solvers' two function handles (bundle.m:168-192; gauss_newton_armijo.m:187-191,265-281,
    levenberg_marquardt.m:170-177,217, levenberg_marquardt_powell.m:134-150,166) through dbat_hip_options.term_fun /
    veto_fun.  termFun here needs the VECTORS (max norms), vetoFun rejects the first trial point it is shown: every call
    must arrive with the oracle's arguments -- Jp and r weighted, in the reference's row order, x the trial point -- the
    same number of times, and the run must take the oracle's course (step halved / lambda raised / delta halved once)."""
    from dbat_amd import bundle
    s = dict(cases())[name]()
    # LM: the SECOND point shown (lambda is 0 there: 0 -> lambdaMin -> 0, levenberg_marquardt.m:177-205).  Rejecting the
    # first one leaves (10*lambda0)/10 < lambdaMin = lambda0 to the last bit of trace(J'J): not a property of the problem
    nth = 2 if damping == 'lm' else 1

    def recorder():
        log = {'term': [], 'veto': []}

        def term(Jp, r):
            log['term'].append((np.array(Jp), np.array(r)))
            return np.abs(Jp).max() <= 1e-3 * np.abs(r).max()

        def veto(x):
            log['veto'].append(np.array(x))
            return len(log['veto']) == nth
        return log, term, veto
    lg, term, veto = recorder()
    lo, termo, vetoo = recorder()
    res, ok, iters, s0, E = bundle(s, damping, term_fun=term, veto_fun=veto)
    ro, oko, ito, s0o, Eo = o.bundle(s, damping, termFun=termo, vetoFun=vetoo)
    assert ok == oko and E.code == Eo.code == 0
    assert iters == ito and len(lg['term']) == len(lo['term']) and len(lg['veto']) == len(lo['veto'])
    assert len(lg['term']) >= 1 and (damping == 'gm' or len(lg['veto']) >= 2)
    for (Jp, r), (Jpo, ro_) in zip(lg['term'], lo['term']):
        assert Jp.shape == Jpo.shape == r.shape == ro_.shape == (E.numObs,)
        assert np.abs(r - ro_).max() <= 1e-9 * np.abs(ro_).max()
        assert np.abs(Jp - Jpo).max() <= 1e-6 * np.abs(Jpo).max() + 1e-9 * np.abs(ro_).max()
    for x, xo in zip(lg['veto'], lo['veto']):
        assert x.shape == xo.shape == (E.numParams,) and relerr(x, xo) < 1e-8
    assert len(E.res) == len(Eo.res) and relerr(E.res, Eo.res) < 1e-8 and relerr(E.x, Eo.x) < 1e-6
    if damping == 'gna':
        assert E.damping.alpha[0] == Eo.damping.alpha[0] == 0.5          # the vetoed full step
        assert np.array_equal(E.damping.alpha, Eo.damping.alpha)
    elif damping == 'lm':
        lam, lamo = E.damping.__dict__['lambda'], Eo.damping.__dict__['lambda']
        assert len(lam) == len(lamo) and relerr(lam, lamo) < 1e-8 and lam[2] == 0 and lam[3] > 0   # raised after the veto
    elif damping == 'lmp':
        assert np.array_equal(E.damping.step[:len(Eo.damping.step)], Eo.damping.step)
        assert relerr(E.damping.delta[:iters + 1], Eo.damping.delta[:ito + 1]) < 1e-8


def test_callbacks_fail_loudly(hip):
    """An exception inside a callback ends the run and comes back to the caller unchanged; a veto that never accepts
    is the reference's code -3 (gauss_newton_armijo.m:218-224)."""
    from dbat_amd import bundle
    s = dict(cases())['tiny-plain']()

    class Boom(Exception):
        pass

    def bad(Jp, r):
        raise Boom('in termFun')
    with pytest.raises(Boom):
        bundle(s, 'gna', term_fun=bad)
    res, ok, iters, s0, E = bundle(s, 'gna', veto_fun=lambda x: True)
    ro, oko, ito, s0o, Eo = o.bundle(s, 'gna', vetoFun=lambda x: True)
    assert not ok and not oko and E.code == Eo.code == -3 and iters == ito


def test_roma_script_known_answer_hip(hip):
    """The reference's roma script result (79 321 unknowns, real data, 5 IO
    estimated) reproduced by bundle() on the GPU: first/last error, iteration
    count, sigma0, camera and EO values of data/script/romabundledemo/result."""
    from dbat_amd import bundle
    s = roma_struct()
    res, ok, iters, s0, E = bundle(s, 'gna')
    assert ok and E.code == 0
    check_roma_against_result(res, s0, E, iters, roma_expected())
    for damping in ('lm', 'lmp'):
        r2, ok2, it2, s02, E2 = bundle(s, damping, store_trace=False)
        assert ok2 and abs(s02 - s0) < 1e-8 * s0
        assert relerr(E2.x, E.x) < 1e-7


def test_small_scene_all_dampings(hip):
    from dbat_amd import bundle
    s, truth = synth_struct('small', 'plain')
    for damping in ('gna', 'lm', 'lmp'):
        res, ok, iters, s0, E = bundle(s, damping)
        ro, oko, ito, s0o, Eo = o.bundle(s, damping)
        assert ok and oko
        check_history(E, Eo, iters, ito, damping, s=s)
        assert relerr(E.x, Eo.x) < TOL_X
        assert 0.4 < s0 < 0.6           # noise 0.5 px, IP.std 1 px


def test_failure_codes(hip):
    """Structural rank deficiency (code -4, camcaldemo_1ray) and a missing
    datum (code -2, camcaldemo_no_datum): same codes as the oracle."""
    from dbat_amd import bundle
    s, _ = synth_struct('tiny', 'plain')
    # a point measured in one image only: 2 rows, 3 unknowns
    p = s.IP.pt[0]
    keep = ~((s.IP.pt == p) & (np.arange(len(s.IP.pt)) != 0))
    s.IP.val, s.IP.std = s.IP.val[:, keep], s.IP.std[:, keep]
    s.IP.cam, s.IP.pt = s.IP.cam[keep], s.IP.pt[keep]
    res, ok, iters, s0, E = bundle(s, 'gna')
    ro, oko, ito, s0o, Eo = o.bundle(s, 'gna')
    assert E.code == Eo.code == -4 and not ok
    s, _ = synth_struct('tiny', 'plain')
    s.bundle.est.EO[:] = True            # no datum: 7-dimensional null space
    res, ok, iters, s0, E = bundle(s, 'gna')
    ro, oko, ito, s0o, Eo = o.bundle(s, 'gna')
    assert Eo.code == -2
    assert E.code == -2 and not ok
    # a sub-network that no per-camera / per-point count gives away: two cameras that only
    # see three points nobody else sees (12 rows, 21 unknowns) -- sprank finds it, and so
    # does the matching in the plan
    s, _ = synth_struct('tiny', 'plain')
    vis = np.zeros((s.EO.val.shape[1], s.OP.val.shape[1]), bool)
    vis[s.IP.cam, s.IP.pt] = True
    ca, cb = next((a, b) for a in range(1, vis.shape[0]) for b in range(a + 1, vis.shape[0])
                  if np.count_nonzero(vis[a] & vis[b]) >= 3)
    pts = np.flatnonzero(vis[ca] & vis[cb])[:3]
    keep = (~np.isin(s.IP.cam, [ca, cb]) & ~np.isin(s.IP.pt, pts)) | (np.isin(s.IP.cam, [ca, cb]) & np.isin(s.IP.pt, pts))
    s.IP.val, s.IP.std = s.IP.val[:, keep], s.IP.std[:, keep]
    s.IP.cam, s.IP.pt = s.IP.cam[keep], s.IP.pt[keep]
    for damping in ('gna', 'lm', 'lmp'):
        res, ok, iters, s0, E = bundle(s, damping)
        ro, oko, ito, s0o, Eo = o.bundle(s, damping)
        assert E.code == Eo.code == -4 and not ok and iters == 0
        assert E.weakness.structural.deficiency == Eo.weakness.structural.deficiency == 9
        assert list(E.weakness.structural.suspectedParams) == list(Eo.weakness.structural.suspectedParams)
    # too few iterations: code -1, s not updated (bundle.m:356-358)
    s, _ = synth_struct('tiny', 'plain')
    res, ok, iters, s0, E = bundle(s, 'gna', 1)
    ro, oko, ito, s0o, Eo = o.bundle(s, 'gna', 1)
    assert E.code == Eo.code == -1 and iters == ito
    assert np.array_equal(res.EO.val, s.EO.val)


def _shared_eo_struct(kind):
    """Images that share exterior orientation elements through EO.struct.block
    (buildserialindices.m:162-221 treats it exactly like IO.struct.block): 'station' -- images 1
    and 2 and images 5, 6, 7 taken from the same projection centres (rows X, Y, Z shared, the
    angles their own); 'rig' -- images 3 and 4 share all six elements.  The observations of the
    images that follow a leader are re-projected from the shared truth, so the network stays
    consistent and the adjustment converges."""
    from dbat_amd import synth
    s, truth = synth_struct('tiny', 'plain' if kind != 'selfcal' else 'selfcal')
    blk = s.EO.struct.block
    share = [(slice(0, 6), 4, 3)] if kind == 'rig' else [(slice(0, 3), 2, 1), (slice(0, 3), 6, 5), (slice(0, 3), 7, 5)]
    EOt = truth['EO'].copy()
    for rows, c, lead in share:
        blk[rows, c] = blk[rows, lead]
        EOt[rows, c] = EOt[rows, lead]
        s.EO.val[rows, c] = s.EO.val[rows, lead]
    px = float(np.ravel(s.IO.sensor.pxSize)[0])
    rng = np.random.default_rng(11)
    for _, c, _ in share:
        m = s.IP.cam == c
        uv, depth = synth.project(truth['IO'], EOt, truth['OP'], s.IP.cam[m], s.IP.pt[m], px)
        s.IP.val[:, m] = uv + rng.normal(0, 0.5, uv.shape)
    return s


@pytest.mark.parametrize('kind', ['station', 'rig', 'selfcal'])
def test_shared_eo_blocks(hip, kind):
    """Shared EO elements: unknown ordering, residual / Jacobian, the step of the reduced
    system, the bundle result and the posterior covariance as the oracle's (one unknown in
    the slot of the block's leading entry, fanned out on deserialise)."""
    from dbat_amd import bundle, bundle_cov
    s = _shared_eo_struct(kind)
    so, x0, w = oracle_setup(s)
    R = np.sqrt(w)
    h = hip.Handle(s)
    try:
        assert h.n == len(x0) and np.array_equal(h.serialize(), x0)
        rng = np.random.default_rng(3)
        x = x0 + 1e-5 * rng.standard_normal(len(x0)) * np.maximum(1e-3, np.abs(x0))
        r_o, K = o.brown_euler_cam4(x, so, jac=True)
        r_h, f_h = h.residual(x)
        assert relerr(r_h, r_o) < TOL_BLOCK
        J = (sp.diags(R) @ K).tocsc()
        p_o, *_ = o._scaled_gn(J, R * r_o)
        p_h, st = h.linearize_solve(x, 0.0, True)
        assert relerr(p_h, p_o) < TOL_STEP
        assert relerr(h.gradient(), J.T @ (R * r_o)) < 1e-10
        assert relerr(h.colnorms(), np.sqrt(np.asarray(J.multiply(J).sum(0)).ravel())) < 1e-10
        Jp = J @ p_o
        assert abs(st['JpJp'] - Jp @ Jp) <= 1e-7 * (Jp @ Jp)
    finally:
        h.close()
    for damping in ('gna', 'lm', 'lmp'):
        res, ok, iters, s0, E = bundle(s, damping)
        ro, oko, ito, s0o, Eo = o.bundle(s, damping)
        assert ok == oko and E.code == Eo.code
        assert relerr(E.x, Eo.x) < TOL_X and abs(s0 - s0o) < 1e-9 * s0o
        assert relerr(res.EO.val[:6], ro.EO.val[:6]) < TOL_X          # fanned out to every image of a block
        if damping == 'gna':
            assert iters == ito
            CEO, COP = bundle_cov(res, E, 'CEO', 'COP')
            CEOo, COPo = o.bundle_cov(ro, Eo, 'CEO', 'COP')
            for A, B in ((CEO, CEOo), (COP, COPo)):
                assert abs(A - B).max() <= 1e-6 * abs(B).max()


def test_unsupported_and_bad_input(hip):
    from dbat_amd import bundle
    from dbat_amd.driver import BadInput
    s, _ = synth_struct('tiny', 'plain')
    with pytest.raises(BadInput):
        bundle(s, 'newton')


def test_C1_full_size_properties(hip):
    """BASELINE config 2 (100 cams / 10k pts / 100k obs, LM-Powell): the HIP
    path against the oracle at full size, plus size-independent properties."""
    from dbat_amd import bundle
    s, truth = synth_struct('C1', 'plain')
    res, ok, iters, s0, E = bundle(s, 'lmp')
    assert ok and 0.45 < s0 < 0.55
    assert np.all(np.diff(E.res) <= 1e-9 * E.res[0])             # monotone residual norm
    # converged point is stationary: ||J'r|| tiny relative to ||J|| ||r||
    from dbat_amd import _hip
    h = _hip.Handle(res)
    try:
        p, st = h.linearize_solve(h.serialize(), 0.0, True)
        assert np.sqrt(st['JpJp']) <= 1e-5 * np.sqrt(2 * st['f'])
    finally:
        h.close()
    # recovers the truth to the noise level
    assert np.abs(res.OP.val - truth['OP']).std() < 0.1
    ro, oko, ito, s0o, Eo = o.bundle(s, 'lmp')
    assert iters == ito and relerr(E.x, Eo.x) < 1e-7


class _ThreadComm:
    """Two shards on ONE GPU: each rank is a thread with its own handle; the
    all-reduce callback exchanges buffers through the host.  Exercises the
    core's N>1 path (sharded plan, reduce hooks, min/max exchange, final
    gather) where only one device is available."""

    def __init__(self, rank, world, shared):
        self.rank, self.world_size, self.sh = rank, world, shared

    def _reduce(self, arr):
        sh = self.sh
        sh['buf'][self.rank] = arr
        sh['bar'].wait()
        total = sum(sh['buf'][r] for r in range(self.world_size))
        sh['bar'].wait()
        return total

    def allreduce_ptr(self, ptr, count, stream):
        import torch
        from dbat_amd.parallel import _DevMem
        t = torch.as_tensor(_DevMem(ptr, count), device='cuda')
        torch.cuda.synchronize()
        total = self._reduce(t.cpu().numpy().copy())
        t.copy_(torch.from_numpy(total))
        torch.cuda.synchronize()
        return 0

    def allreduce_numpy(self, a):
        return self._reduce(np.array(a, dtype=float, copy=True))


@pytest.mark.parametrize('damping', ['gna', 'lm', 'lmp', 'gna-sig'])
def test_two_shards_one_gpu_match_single(hip, damping, monkeypatch):
    import threading
    from dbat_amd import bundle
    if damping == 'gna-sig':             # the signature kernels on every shard (the default at C3 / C4)
        monkeypatch.setenv('DBAT_HIP_SIG', '2')
        damping = 'gna'
    s, truth = synth_struct('small', 'priors')
    ref = bundle(s, damping)
    shared = {'buf': [None, None], 'bar': threading.Barrier(2)}
    out = [None, None]
    err = []

    def run(rank):
        try:
            out[rank] = bundle(s, damping, comm=_ThreadComm(rank, 2, shared))
        except Exception as e:   # noqa: BLE001
            err.append(e)
            shared['bar'].abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    [t.start() for t in th]
    [t.join(300) for t in th]
    assert not err, err
    for rank in range(2):
        res, ok, iters, s0, E = out[rank]
        assert ok == ref[1] and E.code == ref[4].code
        assert relerr(E.x, ref[4].x) < 1e-8
        assert abs(s0 - ref[3]) < 1e-9 * ref[3]
        if damping != 'lm':
            assert iters == ref[2]
        assert relerr(res.post.res.IP, ref[0].post.res.IP) < 1e-6


@pytest.mark.parametrize('damping', ['gna', 'lm'])
def test_library_rccl_communicator_one_rank(hip, damping):
    """The in-library RCCL path (dbat_hip_comm_unique_id / dbat_hip_comm_init,
    ncclAllReduce on the handle's stream) with a one-rank communicator: every
    collective of the N>1 code path runs, the result is that of the plain handle."""
    from dbat_amd import _hip
    s, truth = synth_struct('small', 'priors')
    opt = _hip.default_options(damping)
    out = []
    for with_comm in (False, True):
        h = hip.Handle(s)
        try:
            if with_comm:
                h.comm_init(_hip.comm_unique_id())
                assert np.array_equal(h.comm_allreduce_host(np.arange(4.0), 'max'), np.arange(4.0))
            x, res, rr, damp, aux, T = h.solve(h.serialize(), opt)
            ru, rw = h.final_residuals()
            out.append((x, res.code, res.iters, rr, ru, rw))
        finally:
            h.close()
    (x0_, c0, i0, r0, u0, w0), (x1, c1, i1, r1, u1, w1) = out
    assert c0 == c1 == 0
    if damping != 'lm':       # LM: the count of trailing trial steps is rounding noise (check_history)
        assert i0 == i1 and relerr(r1, r0) < 1e-10
    assert relerr(x1, x0_) < 1e-9
    tol_r = 1e-9 if damping != 'lm' else 1e-6          # residuals of two end points that differ at the 1e-9 level
    assert relerr(u1, u0) < tol_r and relerr(w1, w0) < tol_r


def test_two_ranks_rccl_match_single(hip):
    """Two processes, two GPUs, RCCL inside the library (torch.distributed.run
    launches tests/_rccl_worker.py): every rank's bundle() result equals the
    one-GPU result.  Skipped on a one-GPU box."""
    import os
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', '29517', os.path.join(root, 'tests', '_rccl_worker.py')]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count('RANK_OK') == 2


@pytest.mark.parametrize('path', ['tile3', 'sig', 'column-list'])
def test_ill_conditioned_point_blocks(hip, path, monkeypatch):
    """Start values far from the solution (bench/fuzz_solve.py, seed 137: exterior orientations off by metres and
    tenths of a radian): object points seen under narrow angles (cond(V) up to 7e8) beside an image whose derivatives
    are a thousand times the others'.  cond(J'J) is 8e9 -- well inside double precision, the oracle's Cholesky of the
    full normal matrix goes through -- but the Schur complement only survives if the point blocks are eliminated
    through their FACTOR (kernels.hpp point_block_factor: V^-1 = R R', Z = W R, h = R (R' g)): with the adjugate
    inverse of rounds 1-3 the reduced system lost a pivot (code -2 at iteration 0 where the reference converges in 15
    iterations) and products with the explicit inverse cost five digits of the step.  Every build path."""
    import scipy.sparse as sp
    rng = np.random.default_rng(9000 + 137)
    rng.choice(['plain', 'selfcal', 'imagevar', 'priors', 'groups4']); rng.choice(['tiny', 'tiny', 'small'])
    s, truth = synth_struct('tiny', 'plain', seed=3000 + 137)
    push = float(rng.choice([0.0, 1.0, 3.0, 10.0, 30.0])) * 12
    assert push == 120.0
    nc, npnt = s.EO.val.shape[1], s.OP.val.shape[1]
    s.EO.val[0:3] += push * rng.normal(0, 0.05, (3, nc))
    s.EO.val[3:6] += push * rng.normal(0, 0.002, (3, nc))
    s.OP.val += push * rng.normal(0, 0.05, (3, npnt))
    so, x0, w = oracle_setup(s)
    Rw = np.sqrt(w)
    r_o, K = o.brown_euler_cam4(x0, so, jac=True)
    J = (sp.diags(Rw) @ K).tocsc()
    p_o, sing, Jn, *_ = o._scaled_gn(J, Rw * r_o)
    assert not sing
    p_qr = np.linalg.lstsq((J @ sp.diags(1.0 / Jn)).toarray(), -(Rw * r_o), rcond=None)[0] / Jn      # no normal equations at all
    assert relerr(p_o, p_qr) < 5e-6
    monkeypatch.setenv(*{'tile3': ('DBAT_HIP_SIG', '0'), 'sig': ('DBAT_HIP_SIG', '2'), 'column-list': ('DBAT_HIP_CMAX', '0')}[path])
    h = hip.Handle(s)
    try:
        assert h.build_kernel_name() == {'tile3': 'k_build_tile3', 'sig': 'k_build_sig', 'column-list': 'k_build'}[path]
        p, st = h.linearize_solve(x0, 0.0, True)
        assert st['chol_info'] == 0 and not st['singular'] and 1e-10 < st['rcond'] < 1e-7
        assert relerr(p, p_qr) < 1e-6, relerr(p, p_qr)           # measured 2 ... 4e-8 on all three paths
        lam = 1e-10 * st['trace'] / h.n
        q, st2 = h.linearize_solve(x0, lam, False)
        q_o, _ = o.normal_solve(((J.T @ J) + lam * sp.identity(J.shape[1])).tocsc(), -(J.T @ (Rw * r_o)))
        assert st2['chol_info'] == 0 and relerr(q, q_o) < 1e-5
    finally:
        h.close()
    if path == 'sig':
        from dbat_amd import bundle
        res, ok, iters, s0, E = bundle(s, 'gna')
        ro, oko, ito, s0o, Eo = o.bundle(s, 'gna')
        assert ok and oko and E.code == 0 and iters == ito == 15
        assert relerr(E.x, Eo.x) < 1e-6


@pytest.mark.parametrize('world', [2, 3])
def test_processes_share_one_gpu_through_the_host(hip, world):
    """The multi-process path on a one-GPU box: `world` processes (torch.distributed.run, tests/_rccl_worker.py) share
    GPU 0, every rank plans and runs its own domain of the dissection, and the sums over the ranks go through
    host memory over gloo (parallel.Comm.attach_host) instead of RCCL.  Every rank's bundle() equals the
    one-process result for GNA, LM and LMP."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--standalone', '--local-addr', '127.0.0.1', '--nnodes=1',
           '--nproc-per-node', str(world), os.path.join(root, 'tests', '_rccl_worker.py')]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', DBAT_TEST_HOST_ALLREDUCE='1')
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count('RANK_OK') == world


def test_bench_two_processes_on_one_gpu(hip):
    """`python bench.py --gpus 2` end to end on a one-GPU box (DBAT_BENCH_HOST_ALLREDUCE=1: both ranks on GPU 0, sums
    through the host): the launcher, one plan per rank, the barriers and the max over the ranks, the JSON line
    of rank 0 -- and the LM solve inside it ends where the one-process run ends."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, os.path.join(root, 'bench.py'), '--config', 'C1', '--steps', '3', '--warmup', '1', '--no-cpu-baseline']
    one = subprocess.run(base, capture_output=True, text=True, timeout=900, cwd=root)
    assert one.returncode == 0, one.stderr[-3000:]
    two = subprocess.run(base + ['--gpus', '2'], capture_output=True, text=True, timeout=900, cwd=root,
                         env=dict(os.environ, DBAT_BENCH_HOST_ALLREDUCE='1'))
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-3000:]
    j1, j2 = json.loads(one.stdout.strip().splitlines()[-1]), json.loads(two.stdout.strip().splitlines()[-1])
    assert j2['n_gpus'] == 2 and j2['multi_gpu']['ranks'] == 2 and j2['multi_gpu']['domain_sharding']
    assert j2['multi_gpu']['obs_this_rank'] < 0.6 * 100000
    assert 'host memory' in j2['config']['collective']
    assert j2['solve']['code'] == 0 and j1['solve']['code'] == 0
    assert abs(j2['solve']['sigma0'] - j1['solve']['sigma0']) < 1e-8 * j1['solve']['sigma0']
    # north_star's table: one row per rank with its shard, the HBM rate of the streaming kernels and the MFMA share of the
    # Schur kernel (round 6: printed by rank 0 so that the first real N > 1 run yields it)
    rows = j2['multi_gpu']['per_rank_roofline']
    assert [r['rank'] for r in rows] == [0, 1] and sum(r['obs'] for r in rows) == 100000
    for r in rows:
        assert r['schur_mfma_pct_of_fp64_peak_algorithmic'] > 0 and r['residual_HBM_GBs'] > 0 and r['backsub_HBM_GBs'] > 0


def test_rccl_allreduce_on_raw_device_pointer(hip):
    """parallel.Comm.allreduce_ptr (the callback the core invokes) on a raw
    device pointer and a non-default HIP stream, over the nccl (= RCCL)
    backend with a one-rank group."""
    import os
    import torch
    import torch.distributed as dist
    from dbat_amd.parallel import Comm
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29581')
    if not dist.is_initialized():
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        comm = Comm()
        x = torch.arange(1000, dtype=torch.float64, device='cuda')
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            x.mul_(2.0)
        rc = comm.allreduce_ptr(x.data_ptr(), x.numel(), st.cuda_stream)
        torch.cuda.synchronize()
        assert rc == 0 and comm.n_collectives == 1
        assert torch.equal(x.cpu(), torch.arange(1000, dtype=torch.float64) * 2)
        a = comm.allreduce_numpy(np.arange(5.0))
        assert np.array_equal(a, np.arange(5.0))
    finally:
        dist.destroy_process_group()


def _full_size_properties(name, damping):
    """Size-independent properties at a BASELINE.json size: convergence,
    sigma0 at the noise level (0.5 px noise, IP.std 1 px), monotone residual
    norms, a stationary end point (||J p|| <= 1e-5 ||r||), truth recovered to
    the noise level, and two different damping schemes agreeing on x."""
    from dbat_amd import bundle, synth, _hip
    s, truth = synth.make_scene(name)
    res, ok, iters, s0, E = bundle(s, damping, store_trace=False)
    assert ok and E.code == 0
    assert 0.49 < s0 < 0.52
    assert np.all(np.diff(E.res) <= 1e-9 * E.res[0])
    h = _hip.Handle(res)
    try:
        p, st = h.linearize_solve(h.serialize(), 0.0, True)
        assert not st['singular']
        assert np.sqrt(st['JpJp']) <= 1e-5 * np.sqrt(2 * st['f'])
    finally:
        h.close()
    # truth recovered: fixed IO to the noise level here (estimate and truth differ by the datum -- camera 0 sits
    # at its noisy initial position -- hence metres, not sigmas).  Self-calibration against the truth, in the
    # truth's datum and in units of the posterior sigma: tests/test_fullsize_parity.py (C2 and C4), which also
    # pins why the camera constants of a 0.5 px run sit 0.04 mm (C2) / 0.45 mm (C4) above the generator's values
    if not bool(np.any(s.bundle.est.IO)):
        assert np.abs(res.EO.val[:3] - truth['EO'][:3]).max() < 1.0
    other = 'gna' if damping != 'gna' else 'lmp'
    r2, ok2, it2, s02, E2 = bundle(s, other, store_trace=False)
    assert ok2 and abs(s02 - s0) < 1e-7 * s0
    assert relerr(E2.x, E.x) < 1e-6
    return res, E


def test_C2_full_size_selfcal_properties(hip):
    """BASELINE config 3: 1000 cams / 100k pts / 1M obs, self-calibrating
    Brown K1-K3, P1-P2 (one shared IO block)."""
    res, E = _full_size_properties('C2', 'lm')
    assert E.numParams == 6000 - 7 + 8 + 300000


def test_C3_full_size_properties(hip):
    """BASELINE config 4 on one GPU: 1000 cams / 1M pts / 10M obs, fixed IO."""
    res, E = _full_size_properties('C3', 'lm')
    assert E.numObs == 20_000_000 and E.numParams == 3_005_993


def test_C4_full_size_properties(hip):
    """BASELINE config 5 on one GPU: 5000 cams / 5M pts / 50M obs, 4 camera
    groups with independent self-calibrated IO (the small-scale oracle parity of
    that layout is the 'groups4' variant of the cases above)."""
    res, E = _full_size_properties('C4', 'lm')
    assert E.numParams == 30000 - 7 + 32 + 15_000_000 and E.numObs == 100_000_000
    # four independent IO blocks: four distinct camera constants (their values against the four true constants:
    # tests/test_fullsize_parity.py::test_truth_within_posterior_sigma)
    assert len(np.unique(np.round(res.IO.val[0], 12))) == 4


@pytest.mark.parametrize('variant', ['plain', 'selfcal'])
def test_mixed_tiled_and_heavy_points(hip, variant, monkeypatch):
    """Points with more cameras than a tile holds ("heavy", through k_build)
    next to tiled points (MFMA kernel) in one problem: step parity with the
    oracle and an identical bundle result."""
    from dbat_amd import bundle
    s, truth = synth_struct('small', variant)
    # thin out every second point to 4 rays; with CMAX=6 the 8-ray points are heavy
    pt, cam = s.IP.pt, s.IP.cam
    rank = np.zeros(len(pt), int)
    order = np.lexsort((cam, pt))
    first = np.r_[True, pt[order][1:] != pt[order][:-1]]
    idx = np.arange(len(pt)) - np.maximum.accumulate(np.where(first, np.arange(len(pt)), 0))
    rank[order] = idx
    keep = ~((pt % 2 == 0) & (rank >= 4))
    s.IP.val, s.IP.std = s.IP.val[:, keep], s.IP.std[:, keep]
    s.IP.cam, s.IP.pt = s.IP.cam[keep], s.IP.pt[keep]
    monkeypatch.setenv('DBAT_HIP_CMAX', '6')
    so, x0, w = oracle_setup(s)
    R = np.sqrt(w)
    r_o, K = o.brown_euler_cam4(x0, so, jac=True)
    J = (sp.diags(R) @ K).tocsc()
    p_o, *_ = o._scaled_gn(J, R * r_o)
    for heavy in ('1', '0'):                            # the matrix-core path of csrc/heavy.hpp, and the column lists it replaces
        monkeypatch.setenv('DBAT_HIP_HEAVY', heavy)
        h = hip.Handle(s)
        try:
            info = h.info()
            assert 0 < info['n_tiles'] and info['n_batches'] > 0 and (info['heavy_tasks'] > 0) == (heavy == '1')
            p_h, st = h.linearize_solve(x0, 0.0, True)
            assert relerr(p_h, p_o) < TOL_STEP
        finally:
            h.close()
    monkeypatch.delenv('DBAT_HIP_HEAVY')
    res, ok, iters, s0, E = bundle(s, 'gna')
    monkeypatch.delenv('DBAT_HIP_CMAX')
    ro, oko, ito, s0o, Eo = o.bundle(s, 'gna')
    assert ok and oko and iters == ito and relerr(E.x, Eo.x) < TOL_X


@pytest.mark.parametrize('variant', ['plain', 'selfcal'])
def test_giant_points(hip, variant, monkeypatch):
    """Object points with more observations than one batch holds (control points
    seen in every image; here 140 images, 128-observation batches, 64-thread
    workgroups so that every point takes three chunks) go through
    k_build_giant / k_backsub_giant: step parity with the oracle and an
    identical bundle result.  Distortion-free camera, so that the projections
    far outside the image format stay well defined."""
    from dbat_amd import bundle, synth
    s, truth = synth.make_scene('small', cams=140, points=500, rays=6)
    s.IO.val[5:10] = 0.0
    truth['IO'][5:10] = 0.0
    if variant == 'selfcal':
        s.bundle.est.IO[[0, 1, 2, 5, 6]] = True
    nc = s.EO.val.shape[1]
    px = float(np.ravel(s.IO.sensor.pxSize)[0])
    rng = np.random.default_rng(5)
    add_cam, add_pt = [], []
    for p in (3, 77, 250):
        have = set(s.IP.cam[s.IP.pt == p].tolist())
        for c in range(nc):
            if c not in have:
                add_cam.append(c); add_pt.append(p)
    cam = np.r_[s.IP.cam, np.array(add_cam)]; pt = np.r_[s.IP.pt, np.array(add_pt)]
    order = np.lexsort((pt, cam))                       # image-major, ascending OP
    cam, pt = cam[order], pt[order]
    uv, depth = synth.project(truth['IO'], truth['EO'], truth['OP'], cam, pt, px, nK=3, nP=2)
    assert np.all(depth < 0)
    s.IP.val = uv + rng.normal(0, 0.5, uv.shape)
    s.IP.std = np.ones_like(uv)
    s.IP.cam, s.IP.pt = cam, pt
    assert np.bincount(s.IP.pt).max() == nc
    monkeypatch.setenv('DBAT_HIP_BT', '128')
    monkeypatch.setenv('DBAT_HIP_GIANT_THREADS', '64')
    so, x0, w = oracle_setup(s)
    R = np.sqrt(w)
    r_o, K = o.brown_euler_cam4(x0, so, jac=True)
    J = (sp.diags(R) @ K).tocsc()
    p_o, *_ = o._scaled_gn(J, R * r_o)
    h = hip.Handle(s)
    try:
        p_h, st = h.linearize_solve(x0, 0.0, True)
        assert not st['singular']
        assert relerr(p_h, p_o) < TOL_STEP
        Jp = J @ p_o
        assert abs(st['JpJp'] - Jp @ Jp) <= 1e-7 * (Jp @ Jp)
        assert relerr(h.gradient(), J.T @ (R * r_o)) < 1e-10
    finally:
        h.close()
    res, ok, iters, s0, E = bundle(s, 'gna')
    ro, oko, ito, s0o, Eo = o.bundle(s, 'gna')
    assert ok and oko and iters == ito and relerr(E.x, Eo.x) < TOL_X


def _all_see_all_scene(cams, points, selfcal, groups=1, seed=3):
    """Every point in every image (the geometry of the reference's calibration demo, demo/camcaldemo.m:56-119)."""
    from dbat_amd import synth
    return synth.make_dense_scene(cams, points, selfcal, groups, seed)


@pytest.mark.parametrize('variant', ['plain', 'selfcal', 'selfcal-groups3'])
def test_giant_points_on_the_matrix_cores(hip, variant, monkeypatch):
    """csrc/heavy.hpp: points seen in all 300 images (more than a 256-observation batch holds: k_heavy_z_giant, three
    rounds of 128 threads) beside ordinary tiled points, their Schur terms by k_heavy_syrk over 38 camera groups.  Step,
    ||J p||^2 and gradient against the oracle, the bundle against the oracle's, the deterministic mode bit for bit."""
    from dbat_amd import bundle, synth
    s, truth = synth.make_scene('small', cams=300, points=400, rays=6)
    s.IO.val[5:10] = 0.0
    truth['IO'][5:10] = 0.0
    nc = s.EO.val.shape[1]
    if variant != 'plain':
        s.bundle.est.IO[[0, 1, 2, 5, 6]] = True
        if variant == 'selfcal-groups3':
            s.IO.struct.block[:] = (1 + (np.arange(nc) * 3) // nc)[None, :]
    px = float(np.ravel(s.IO.sensor.pxSize)[0])
    rng = np.random.default_rng(5)
    add_cam, add_pt = [], []
    for p in (3, 77, 250):
        have = set(s.IP.cam[s.IP.pt == p].tolist())
        for c in range(nc):
            if c not in have:
                add_cam.append(c); add_pt.append(p)
    cam = np.r_[s.IP.cam, np.array(add_cam)]; pt = np.r_[s.IP.pt, np.array(add_pt)]
    order = np.lexsort((pt, cam))
    cam, pt = cam[order], pt[order]
    uv, depth = synth.project(truth['IO'], truth['EO'], truth['OP'], cam, pt, px, nK=3, nP=2)
    assert np.all(depth < 0)
    s.IP.val = uv + rng.normal(0, 0.5, uv.shape)
    s.IP.std = np.ones_like(uv)
    s.IP.cam, s.IP.pt = cam, pt
    monkeypatch.setenv('DBAT_HIP_GIANT_THREADS', '128')
    so, x0, w = oracle_setup(s)
    R = np.sqrt(w)
    r_o, K = o.brown_euler_cam4(x0, so, jac=True)
    J = (sp.diags(R) @ K).tocsc()
    p_o, *_ = o._scaled_gn(J, R * r_o)
    h = hip.Handle(s)
    try:
        info = h.info()
        assert info['heavy_tasks'] > 0 and info['heavy_points'] == 3 and info['n_tiles'] > 0
        p_h, st = h.linearize_solve(x0, 0.0, True)
        assert not st['singular']
        assert relerr(p_h, p_o) < TOL_STEP
        Jp = J @ p_o
        assert abs(st['JpJp'] - Jp @ Jp) <= 1e-7 * (Jp @ Jp)
        assert relerr(h.gradient(), J.T @ (R * r_o)) < 1e-10
        h.set_deterministic(True)
        pd = [h.linearize_solve(x0, 0.0, True)[0] for _ in range(4)]
        assert all(np.array_equal(pd[0], q) for q in pd[1:]) and relerr(pd[0], p_h) < 1e-8
    finally:
        h.close()
    res, ok, iters, s0, E = bundle(s, 'gna')
    ro, oko, ito, s0o, Eo = o.bundle(s, 'gna')
    assert ok and oko and iters == ito and relerr(E.x, Eo.x) < TOL_X


@pytest.mark.parametrize('cams,selfcal,groups', [(21, True, 1), (30, False, 1), (40, True, 3), (9, True, 2)])
@pytest.mark.parametrize('ks', [None, '2'])
def test_every_point_in_every_image(hip, cams, selfcal, groups, ks, monkeypatch):
    """The visibility of the reference's flagship project (demo/camcaldemo.m:56-119): every point is heavy.  Against the
    oracle: the step through the row groups of csrc/heavy.hpp (3 ... 5 camera groups + the IO / right-hand-side group,
    several independent IO blocks, two k-steps per task: many tasks per pair of groups), and against the column-list
    kernels it replaces (DBAT_HIP_HEAVY=0)."""
    from dbat_amd import bundle
    if ks:
        monkeypatch.setenv('DBAT_HIP_HEAVY_KS', ks)
    monkeypatch.setenv('DBAT_HIP_CMAX', '8')            # (nine cameras: heavy because a tile holds eight)
    s, _ = _all_see_all_scene(cams, 150, selfcal, groups)
    so, x0, w = oracle_setup(s)
    R = np.sqrt(w)
    r_o, K = o.brown_euler_cam4(x0, so, jac=True)
    J = (sp.diags(R) @ K).tocsc()
    p_o, *_ = o._scaled_gn(J, R * r_o)
    h = hip.Handle(s)
    try:
        info = h.info()
        assert info['heavy_points'] == 150 and info['n_tiles'] == 0 and info['heavy_row_groups'] >= (cams + 7) // 8 + 1
        p_h, st = h.linearize_solve(x0, 0.0, True)
        assert not st['singular'] and relerr(p_h, p_o) < TOL_STEP
        assert relerr(h.gradient(), J.T @ (R * r_o)) < 1e-10
        p_lm, _ = h.linearize_solve(x0, 1e-3, False)     # damped, unscaled (levenberg_marquardt.m:119)
        JTJ = (J.T @ J).toarray()
        p_lmo = np.linalg.solve(JTJ + 1e-3 * np.eye(JTJ.shape[0]), -(J.T @ (R * r_o)))
        assert relerr(p_lm, p_lmo) < TOL_STEP
    finally:
        h.close()
    monkeypatch.setenv('DBAT_HIP_HEAVY', '0')
    h = hip.Handle(s)
    try:
        assert h.info()['heavy_tasks'] == 0
        p_l, _ = h.linearize_solve(x0, 0.0, True)
        assert relerr(p_l, p_h) < 1e-9
    finally:
        h.close()
    monkeypatch.delenv('DBAT_HIP_HEAVY')
    res, ok, iters, s0, E = bundle(s, 'gna')
    ro, oko, ito, s0o, Eo = o.bundle(s, 'gna')
    assert ok and oko and iters == ito and relerr(E.x, Eo.x) < TOL_X


@pytest.mark.parametrize('variant', ['plain', 'selfcal', 'priors'])
def test_plan_reuse_set_values(hip, variant):
    """dbat_hip_set_values: a handle re-used for other parameter values of the same structure (bundle.m:156-159 keeps the
    serial indices; deserialize.m:31-46) gives the step, the objective value and the bundle of a handle built for those
    values -- also where a FIXED interior orientation changes (the corrected image coordinates are computed again) and
    where the values and standard deviations of the prior observations change.  A different structure is refused and
    leaves the handle as it was."""
    s, truth = synth_struct('small', variant)
    rng = np.random.default_rng(9)
    t = copy.deepcopy(s)
    t.EO.val = t.EO.val + rng.normal(0, 0.01, t.EO.val.shape) * np.asarray(t.bundle.est.EO, float)
    t.OP.val = t.OP.val + rng.normal(0, 0.02, t.OP.val.shape)
    t.IO.val = t.IO.val.copy(); t.IO.val[0] *= 1.002; t.IO.val[5] *= 0.9          # cc, K1: fixed (plain / priors) or start values
    if variant == 'priors':
        t.prior.EO.val = t.prior.EO.val + 0.01; t.prior.OP.std = t.prior.OP.std * 1.5
    h = hip.Handle(s)
    fresh = hip.Handle(t)
    try:
        x_s = h.serialize()
        p_s, _ = h.linearize_solve(x_s, 0.0, True)
        h.set_values(t)
        x_t = h.serialize()
        assert np.array_equal(x_t, fresh.serialize()) and not np.array_equal(x_t, x_s)
        p_r, st_r = h.linearize_solve(x_t, 0.0, True)
        p_f, st_f = fresh.linearize_solve(x_t, 0.0, True)
        assert relerr(p_r, p_f) < 1e-10 and abs(st_r['f'] - st_f['f']) <= 1e-12 * st_f['f'] and relerr(p_r, p_s) > 1e-6
        assert np.allclose(h.residual(x_t)[0], fresh.residual(x_t)[0], rtol=0, atol=1e-12)
        opt = hip.default_options('gna')
        xr, res_r, *_ = h.solve(x_t, opt)
        xf, res_f, *_ = fresh.solve(x_t, opt)
        assert res_r.code == 0 and res_r.iters == res_f.iters and relerr(xr, xf) < 1e-9
        h.set_values(s)                                               # ... and back
        assert np.array_equal(h.serialize(), x_s)
        p_b, _ = h.linearize_solve(x_s, 0.0, True)
        assert relerr(p_b, p_s) < 1e-10
        u = copy.deepcopy(s)
        u.bundle.est.OP[0, 6] = False                                 # another structure
        with pytest.raises(hip.DbatHipError, match='structure'):
            h.set_values(u)
        assert np.array_equal(h.serialize(), x_s)
    finally:
        h.close(); fresh.close()


def test_bundle_reuses_the_cached_handle(hip, capsys):
    """bundle() -> bundle() -> bundle_cov() on one structure build ONE plan; a changed mask builds a
    new one; results are those of runs with handles of their own.  'trace' prints the solver's line per iteration."""
    from dbat_amd import bundle, bundle_cov
    s, _ = synth_struct('small', 'selfcal')
    hip.clear_cache()
    st = dict(hip.cache_stats)
    r1 = bundle(s, 'gna', 'trace')
    out = capsys.readouterr().out
    lines = [ln for ln in out.splitlines() if ln.startswith('Gauss-Newton-Armijo: iteration')]
    assert len(lines) == r1[2] + 1 and lines[0].endswith('residual norm=%.2g' % r1[4].res[0]) and 'last alpha=1' in lines[1]
    assert hip.cache_stats['misses'] == st['misses'] + 1 and hip.cache_stats['hits'] == st['hits']
    # from the result, pushed a little away from it: same structure, other values.  (LM started AT a converged point accepts or
    # rejects its first steps by the last bits of f -- with the default mode's atomic sums a coin that fell on 'iteration cap'
    # in one run of the suite out of six.)
    t = copy.deepcopy(r1[0])
    t.OP.val = t.OP.val + 1e-3 * np.cos(np.arange(t.OP.val.size, dtype=float)).reshape(t.OP.val.shape)
    t.OP.val[~np.asarray(t.bundle.est.OP, bool)] = r1[0].OP.val[~np.asarray(t.bundle.est.OP, bool)]
    r2 = bundle(t, 'lm')
    assert hip.cache_stats['hits'] == st['hits'] + 1 and hip.cache_stats['misses'] == st['misses'] + 1
    C1 = bundle_cov(r2[0], r2[4], 'CEO')
    assert hip.cache_stats['hits'] == st['hits'] + 2 and hip.cache_stats['misses'] == st['misses'] + 1
    # the same calls with handles of their own
    q1 = bundle(s, 'gna', reuse_handle=False)
    q2 = bundle(copy.deepcopy(t), 'lm', reuse_handle=False)
    assert r1[2] == q1[2] and relerr(r1[4].x, q1[4].x) < 1e-9 and r2[1] and relerr(r2[4].x, q2[4].x) < 1e-9
    hip.clear_cache()
    C2 = bundle_cov(q2[0], q2[4], 'CEO')
    assert abs(C1 - C2).max() <= 1e-8 * abs(C2).max()
    # a changed mask: new plan, and the result of the changed problem
    u = copy.deepcopy(s)
    u.bundle.est.OP[:, 11] = False
    m0 = hip.cache_stats['misses']
    r3 = bundle(u, 'gna')
    assert hip.cache_stats['misses'] == m0 + 1
    q3 = bundle(u, 'gna', reuse_handle=False)
    assert r3[1] and relerr(r3[4].x, q3[4].x) < 1e-9 and r3[4].numParams == r1[4].numParams - 3
    hip.clear_cache()


@pytest.mark.parametrize('selfcal', [False, True])
def test_heavy_points_with_priors_weights_and_fixed_coordinates(hip, selfcal):
    """The matrix-core path of the heavy points (csrc/heavy.hpp) with everything a project can hang on a point: image
    observations with their own standard deviations (no per-camera weight), prior observations of object points and of
    camera positions, fixed control points (rows and columns masked), a camera with fixed elements -- every point in
    every one of 24 images.  Step, gradient, column norms, the bundle and the posterior covariance blocks against the oracle."""
    from dbat_amd import bundle, bundle_cov
    s, truth = _all_see_all_scene(24, 120, selfcal)
    nc, npnt = s.EO.val.shape[1], s.OP.val.shape[1]
    rng = np.random.default_rng(21)
    s.IP.std = np.asfortranarray(s.IP.std * (1 + (np.arange(s.IP.std.shape[1]) % 3)[None, :] * 0.5))
    s.IP.sigmas = np.unique(s.IP.std)
    cps = np.arange(0, npnt, 17)
    s.prior.OP.use[:, cps] = True
    s.prior.OP.val[:, cps] = truth['OP'][:, cps] + rng.normal(0, 0.01, (3, len(cps)))
    s.prior.OP.std[:, cps] = np.array([[0.01], [0.01], [0.02]])
    fixed = np.arange(5, npnt, 23)
    s.OP.val[:, fixed] = truth['OP'][:, fixed]
    s.bundle.est.OP[:, fixed] = False
    s.bundle.est.OP[2, 9] = False                                   # one coordinate of a point
    cams = np.arange(1, nc, 5)
    s.prior.EO.use[0:3, cams] = True
    s.prior.EO.val[0:3, cams] = truth['EO'][0:3, cams] + rng.normal(0, 0.02, (3, len(cams)))
    s.prior.EO.std[0:3, cams] = 0.02
    so, x0, w = oracle_setup(s)
    R = np.sqrt(w)
    r_o, K = o.brown_euler_cam4(x0, so, jac=True)
    J = (sp.diags(R) @ K).tocsc()
    p_o, *_ = o._scaled_gn(J, R * r_o)
    h = hip.Handle(s)
    try:
        info = h.info()
        assert info['heavy_points'] == npnt and info['n_tiles'] == 0
        p_h, st = h.linearize_solve(x0, 0.0, True)
        assert not st['singular'] and relerr(p_h, p_o) < TOL_STEP
        assert abs(st['f'] - 0.5 * (R * r_o) @ (R * r_o)) <= 1e-11 * st['f']
        assert relerr(h.gradient(), J.T @ (R * r_o)) < 1e-10
        assert relerr(h.colnorms(), np.sqrt(np.asarray(J.multiply(J).sum(0)).ravel())) < 1e-10
        h.set_deterministic(True)
        pd = [h.linearize_solve(x0, 0.0, True)[0] for _ in range(3)]
        assert all(np.array_equal(pd[0], q) for q in pd[1:]) and relerr(pd[0], p_h) < 1e-8
    finally:
        h.close()
    res, ok, iters, s0, E = bundle(s, 'gna')
    ro, oko, ito, s0o, Eo = o.bundle(s, 'gna')
    assert ok and oko and iters == ito and relerr(E.x, Eo.x) < TOL_X and abs(s0 - s0o) < 1e-8 * s0o
    CEO, COP = bundle_cov(res, E, 'CEO', 'COP')
    CEOo, COPo = o.bundle_cov(ro, Eo, 'CEO', 'COP')
    assert abs(CEO - CEOo).max() <= 1e-6 * abs(CEOo).max() and abs(COP - COPo).max() <= 1e-6 * abs(COPo).max()


@pytest.mark.parametrize('model', [3, 5])
def test_posterior_covariance_camcal_known_answer(hip, model):
    """bundle_cov on the GPU (inv(S) and the per-point blocks from the Schur
    pieces) against the standard deviations printed in the reference's camcal
    reports, and against the oracle's dense inverse of the full normal matrix."""
    from dbat_amd import bundle, bundle_cov
    from helpers import check_camcal_cov_against_report
    exp = camcal_expected()['model%d' % model]
    res, ok, iters, s0, E = bundle(camcal_struct(model), 'gna')
    assert ok
    CIO, CEO, COP = bundle_cov(res, E, 'CIO', 'CEO', 'COP')
    check_camcal_cov_against_report(res, CIO, CEO, COP, exp)
    ro, oko, ito, s0o, Eo = o.bundle(camcal_struct(model), 'gna')
    CIOo, CEOo, COPo = o.bundle_cov(ro, Eo, 'CIO', 'CEO', 'COP')
    for A, B in ((CIO, CIOo), (CEO, CEOo), (COP, COPo)):
        assert abs(A - B).max() <= 1e-6 * abs(B).max()
    CEOF = bundle_cov(res, E, 'CEOF')
    assert abs(CEOF - o.bundle_cov(ro, Eo, 'CEOF')).max() <= 1e-6 * abs(CEOo).max()
    # 'CXX' and 'COPF' (bundle_cov.m:18-24), offered at the sizes the reference's callers use them at
    CXX, COPF = bundle_cov(res, E, 'CXX', 'COPF')
    CXXo, COPFo = o.bundle_cov(ro, Eo, 'CXX', 'COPF')
    assert CXX.shape == CXXo.shape and abs(CXX - CXXo).max() <= 1e-6 * abs(CXXo).max()
    assert abs(COPF - COPFo).max() <= 1e-6 * abs(COPFo).max()
    # ... and the diagonal blocks of the full matrix are the blocks the device computes by selected inversion
    assert abs(COPF.multiply(COP != 0) - COP).max() <= 1e-8 * abs(COP).max()


@pytest.mark.parametrize('variant', ['plain', 'selfcal', 'imagevar', 'priors', 'groups4'])
def test_posterior_covariance_synthetic(hip, variant):
    from dbat_amd import bundle, bundle_cov
    s, truth = synth_struct('tiny', variant)
    res, ok, iters, s0, E = bundle(s, 'gna')
    ro, oko, ito, s0o, Eo = o.bundle(s, 'gna')
    assert ok and oko
    got = bundle_cov(res, E, 'CIO', 'CEO', 'COP')
    want = o.bundle_cov(ro, Eo, 'CIO', 'CEO', 'COP')
    for A, B in zip(got, want):
        assert A.shape == B.shape
        if abs(B).max() > 0:
            assert abs(A - B).max() <= 1e-6 * abs(B).max()
        else:
            assert abs(A).max() == 0


@pytest.mark.parametrize('dense', [False, True])
def test_posterior_covariance_C1_sampled(hip, dense, monkeypatch):
    """100 cams / 10k pts: blocks of inv(J'J) for a sample of points and images
    from sparse direct solves with unit vectors (the oracle's dense inverse
    would need 7.5 GB) against the device's Schur-block formulation -- with inv(S) from the selected inversion of the
    compact nested-dissection factor (the default on one rank) and from the dense inverse (DBAT_HIP_COV_DENSE)."""
    import scipy.sparse.linalg as spl
    from dbat_amd import bundle, bundle_cov
    if dense:
        monkeypatch.setenv('DBAT_HIP_COV_DENSE', '1')
    s, truth = synth_struct('C1', 'plain')
    res, ok, iters, s0, E = bundle(s, 'gna')
    assert ok
    CEO, COP = bundle_cov(res, E, 'CEO', 'COP')
    so, x, w = oracle_setup(res)
    r_o, K = o.brown_euler_cam4(x, so, jac=True)
    J = (sp.diags(np.sqrt(w)) @ K).tocsc()
    lu = spl.splu((J.T @ J).tocsc())
    des = so.bundle.deserial
    def block(src):
        Eu = np.zeros((J.shape[1], len(src))); Eu[src, np.arange(len(src))] = 1.0
        return E.s0 ** 2 * lu.solve(Eu)[src]
    xOP = np.full(res.OP.val.size, -1); xOP[des.OP.dest] = des.OP.src
    for p in (0, 1234, 5000, 9999):
        src = xOP[3 * p:3 * p + 3]
        assert np.all(src >= 0)
        want = block(src)
        got = COP[3 * p:3 * p + 3, 3 * p:3 * p + 3].toarray()
        assert np.abs(got - want).max() <= 1e-6 * np.abs(want).max()
    m = res.EO.val.shape[0]
    xEO = np.full(res.EO.val.size, -1); xEO[des.EO.dest] = des.EO.src
    for c in (5, 50, 99):
        src = xEO[m * c:m * c + 6]
        keep = src >= 0
        want = block(src[keep])
        got = CEO[m * c:m * c + 6, m * c:m * c + 6].toarray()[np.ix_(keep, keep)]
        assert np.abs(got - want).max() <= 1e-6 * np.abs(want).max()


@pytest.mark.parametrize('case', ['fixed_eo', 'fixed_op', 'unobserved'])
def test_degenerate_but_valid_inputs(hip, case):
    """Only the object points estimated (spatial intersection), only the
    cameras estimated (spatial resection of every image), and object points
    without any observation (fixed, as bundle.m needs them): same results as
    the oracle."""
    from dbat_amd import bundle
    s, truth = synth_struct('tiny', 'plain')
    if case == 'fixed_eo':
        s.bundle.est.EO[:] = False
        s.EO.val[:6] = truth['EO']
    elif case == 'fixed_op':
        s.bundle.est.OP[:] = False
        s.OP.val[:] = truth['OP']
        s.bundle.est.EO[:6] = True
    else:
        # drop all observations of three points and fix them (an unobserved free point is
        # structurally rank deficient, code -4, in the reference as well)
        drop = np.isin(s.IP.pt, [0, 7, 11])
        s.IP.val, s.IP.std = s.IP.val[:, ~drop], s.IP.std[:, ~drop]
        s.IP.cam, s.IP.pt = s.IP.cam[~drop], s.IP.pt[~drop]
        s.bundle.est.OP[:, [0, 7, 11]] = False
    res, ok, iters, s0, E = bundle(s, 'gna')
    ro, oko, ito, s0o, Eo = o.bundle(s, 'gna')
    assert E.code == Eo.code
    assert ok == oko and iters == ito
    assert relerr(E.x, Eo.x) < TOL_X
    if case == 'unobserved':
        # and left free: both report the structural rank deficiency
        s.bundle.est.OP[:, 0] = True
        res, ok, iters, s0, E = bundle(s, 'gna')
        ro, oko, ito, s0o, Eo = o.bundle(s, 'gna')
        assert E.code == Eo.code == -4


def test_report_lines_hip(hip):
    """The same report lines from the GPU result (bundle + bundle_cov on the device)."""
    from dbat_amd import bundle, bundle_cov
    from dbat_amd.report import bundle_result_lines
    from helpers import check_report_lines
    res, ok, iters, s0, E = bundle(camcal_struct(3), 'gna')
    assert ok
    CIO, CEO, COP = bundle_cov(res, E, 'CIO', 'CEO', 'COP')
    lines = bundle_result_lines(res, E, CIO, CEO, COP)
    assert check_report_lines(lines) >= len(lines) - 10


@pytest.mark.parametrize('model,damping', [(3, 'gna'), (3, 'lmp'), (1, 'gna'), (2, 'gna'), (4, 'gna'), (5, 'gna')])
def test_camcal_demo_pipeline_hip(hip, model, damping):
    """The whole demo/camcaldemo.m pipeline -- EXIF camera, 3-point resection,
    forward intersection (dbat_amd.initial) -- then the bundle on the GPU.  From
    these initial values the committed report also pins the path:
    camcal-dbatreport.txt:39-43 '9 iterations', 'First error: 30873.9'."""
    from dbat_amd import bundle, bundle_cov
    from dbat_amd.report import bundle_result_lines
    from helpers import camcal_demo_struct, check_report_lines
    exp = camcal_expected()['model%d' % model]
    s = camcal_demo_struct(model)
    res, ok, iters, s0, E = bundle(s, damping)
    ro, oko, ito, s0o, Eo = o.bundle(s, damping)
    assert ok and oko and E.code == 0
    check_history(E, Eo, iters, ito, damping, s=s)
    assert relerr(E.x, Eo.x) < TOL_X
    assert relerr(E.trace, Eo.trace) < 1e-6
    check_camcal_against_report(res, s0, E, exp)
    assert abs(E.res[0] / 30873.9 - 1) < 1e-5        # tests/test_initial.py: sixth digit is resection noise
    if damping == 'gna':
        assert iters == exp['iterations'] == 9
    if damping == 'gna':
        import os
        from helpers import GOLDEN
        CIO, CEO, COP = bundle_cov(res, E, 'CIO', 'CEO', 'COP')
        lines = bundle_result_lines(res, E, CIO, CEO, COP)
        ref = os.path.join(GOLDEN, 'camcal-dbatreport.txt' if model == 3 else 'camcal-dbatreport-model%d.txt' % model)
        # all but the first error verbatim, for every lens model
        assert len(lines) >= 590 and check_report_lines(lines, ref_path=ref, demo_x0=True) >= len(lines) - 2


@pytest.mark.parametrize('kind', ['1ray', 'missing-obs', 'no-datum'])
def test_camcal_failure_demos_hip(hip, kind):
    """The failure-mode demos through bundle() on the GPU against their committed
    reports (camcal-dbatreport-{1ray,missing-obs,no-datum}.txt): code -4 with the
    structural rank and DMPERM's suspected parameters, code -2 with the
    numerical rank; sigma0 and the error at x0; the report's head line by line."""
    from dbat_amd import bundle
    from dbat_amd.report import bundle_result_lines
    from helpers import (camcal_failure_struct, camcal_failures_expected, check_failure_against_report,
                         check_report_lines)
    exp = camcal_failures_expected()[kind]
    s = camcal_failure_struct(kind)
    res, ok, iters, s0, E = bundle(s, 'gna')
    check_failure_against_report(ok, iters, s0, E, exp)
    ro, oko, ito, s0o, Eo = o.bundle(s, 'gna')
    assert E.code == Eo.code and list(E.paramTypes) == list(Eo.paramTypes)
    if kind != '1ray':                                # 1ray: x0 holds NaN (a one-ray point cannot be intersected)
        assert relerr(E.res, Eo.res) < 1e-10 and abs(s0 / s0o - 1) < 1e-10
        assert relerr(res.post.res.IP, ro.post.res.IP) < 1e-9
    assert np.array_equal(res.EO.val, s.EO.val, equal_nan=True)       # not updated (bundle.m:356-358)
    lines = [l for l in bundle_result_lines(res, E) if not l.strip().startswith(('Vector', '('))]
    check_report_lines(lines, ref_lines=exp['head'], demo_x0=True,
                       x0_lines=('First error:', 'Last error:', 'Sigma0:', 'Sigma0 (pixels):'))


def test_sxb_script_known_answer_hip(hip):
    """data/script/sxb through bundle() and bundle_cov() on the GPU: weighted
    prior observations of control points, two image-point standard deviations,
    project coordinates of 1e6 m (the f64 path has no head-room to lose there).
    Known answers of result/report.txt, and the oracle's iteration history."""
    import os
    from dbat_amd import bundle, bundle_cov
    from dbat_amd.report import bundle_result_lines
    from helpers import (sxb_struct, sxb_expected, check_sxb_against_report, check_camcal_cov_against_report,
                         check_report_lines, GOLDEN)
    exp = sxb_expected()
    s = sxb_struct()
    res, ok, iters, s0, E = bundle(s, 'gna')
    assert ok and E.code == 0
    check_sxb_against_report(res, s0, E, iters, exp)
    ro, oko, ito, s0o, Eo = o.bundle(s, 'gna')
    check_history(E, Eo, iters, ito, 'gna')
    assert relerr(E.x, Eo.x) < 1e-12                  # |x| ~ 1e6: 1e-12 relative = 1e-6 m
    assert np.abs(E.x - Eo.x).max() < 1e-5
    assert abs(s0 - s0o) < 1e-9 * s0o
    assert relerr(res.post.res.OP[res.prior.OP.use], ro.post.res.OP[ro.prior.OP.use]) < 1e-6
    CIO, CEO, COP = bundle_cov(res, E, 'CIO', 'CEO', 'COP')
    check_camcal_cov_against_report(res, CIO, CEO, COP, exp['report'])
    Co = o.bundle_cov(ro, Eo, 'CEO', 'COP')
    assert relerr(CEO.toarray() if hasattr(CEO, 'toarray') else CEO, Co[0].toarray() if hasattr(Co[0], 'toarray') else Co[0]) < 1e-5
    lines = bundle_result_lines(res, E, CIO, CEO, COP)
    n = check_report_lines(lines, ref_path=os.path.join(GOLDEN, 'sxb-report.txt'), demo_x0=True, x0_tol=1e-4)
    assert len(lines) >= 455 and n >= len(lines) - 2     # every line but the first error verbatim
    for damping in ('lm', 'lmp'):
        r2, ok2, it2, s02, E2 = bundle(s, damping)
        assert ok2 and abs(s02 / s0 - 1) < 1e-7


@pytest.mark.parametrize('variant', ['fixed', 'selfcal', 'imagevariant'])
def test_roma_demo_variants_known_answer_hip(hip, variant):
    """demo/romabundledemo{,_selfcal,_imagevariant}.m on the GPU against their
    committed reports (roma-dbatreport*.txt): sigma0 0.623075 / 0.566548 /
    0.502538, parameter counts, and for the self-calibrating runs all nine
    camera values with their posterior deviations (60 cameras, 26 321 points,
    aspect estimated; image-variant: one principal point per image)."""
    from dbat_amd import bundle, bundle_cov
    from helpers import roma_demo_struct, roma_variants_expected, check_roma_variant
    exp = roma_variants_expected()[variant]
    s = roma_demo_struct(variant)
    res, ok, iters, s0, E = bundle(s, 'gna')
    assert ok and E.code == 0
    CIO = bundle_cov(res, E, 'CIO') if variant != 'fixed' else None
    check_roma_variant(res, s0, E, exp, CIO)
    if variant == 'imagevariant':
        assert len(np.unique(np.round(res.IO.val[1], 9))) == 60 and len(np.unique(res.IO.val[0])) == 1
    for damping in ('lm', 'lmp'):                    # same minimum from the other damping schemes
        r2, ok2, it2, s02, E2 = bundle(s, damping)
        assert ok2 and abs(s02 / s0 - 1) < 1e-6


def test_roma_script_report_lines_hip(hip):
    """The committed result file of the roma script run
    (data/script/romabundledemo/result/report.txt, 1334 lines: 60 images,
    26 321 points, self-calibration, dependent datum) written from the GPU's
    bundle() and bundle_cov() results: every line the report module produces
    -- all but the 17 bookkeeping lines -- must be a verbatim line of it, in
    order: camera values, deviations, significances and correlations, 360 EO
    values with deviations and correlations, coverage, ray counts, residual
    statistics, point precision of 26 321 points, intersection angles."""
    import os
    from dbat_amd import bundle, bundle_cov
    from dbat_amd.report import bundle_result_lines
    from helpers import check_report_lines, GOLDEN
    res, ok, iters, s0, E = bundle(roma_struct(), 'gna')
    assert ok
    CIO, CEO, COP = bundle_cov(res, E, 'CIO', 'CEO', 'COP')
    lines = bundle_result_lines(res, E, CIO, CEO, COP)
    n = check_report_lines(lines, ref_path=os.path.join(GOLDEN, 'roma-report.txt'), demo_x0=True)
    assert len(lines) >= 1310 and n >= len(lines) - 2


@pytest.mark.parametrize('label', ['c1', 'c2', 's1', 's2', 's3', 's4'])
def test_prague2016_reports_hip(hip, label):
    """The six prague2016 PhotoModeler projects through bundle() / bundle_cov()
    on the GPU, line by line against the committed DBAT reports (s1: every
    object point fixed, 30 unknowns; s2/s3: control points as the only or
    almost the only points; c2/s2-s4: weighted control points)."""
    from dbat_amd import bundle, bundle_cov
    from dbat_amd.report import bundle_result_lines
    from helpers import prague_struct, check_report_lines
    s, ref = prague_struct(label)
    res, ok, iters, s0, E = bundle(s, 'gna')
    assert ok and E.code == 0
    ro, oko, ito, s0o, Eo = o.bundle(s, 'gna')
    check_history(E, Eo, iters, ito, 'gna')
    assert np.abs(E.x - Eo.x).max() < 1e-7 * max(1.0, np.abs(Eo.x).max())
    CIO, CEO, COP = bundle_cov(res, E, 'CIO', 'CEO', 'COP')
    lines = bundle_result_lines(res, E, CIO, CEO, COP)
    n = check_report_lines(lines, ref_path=ref, demo_x0=True)
    assert len(lines) >= 319 and n >= len(lines) - 1


@pytest.mark.parametrize('use_prior_eo', [False, True])
def test_sxb_prior_eo_reports_hip(hip, use_prior_eo):
    """demo/sxb_prior_eo.m on the GPU: prior observations of camera positions
    (EO prior rows) with weighted control points at 1e6-m coordinates; the
    committed reports line by line, and the oracle's iteration history."""
    from dbat_amd import bundle, bundle_cov
    from dbat_amd.report import bundle_result_lines
    from helpers import sxb_prior_eo_struct, check_report_lines
    s, ref = sxb_prior_eo_struct(use_prior_eo)
    res, ok, iters, s0, E = bundle(s, 'gna')
    assert ok and E.code == 0
    ro, oko, ito, s0o, Eo = o.bundle(s, 'gna')
    check_history(E, Eo, iters, ito, 'gna')
    assert np.abs(E.x - Eo.x).max() < 1e-5 and relerr(E.x, Eo.x) < 1e-12
    if use_prior_eo:
        assert relerr(res.post.res.EO[res.prior.EO.use[:6]], ro.post.res.EO[ro.prior.EO.use]) < 1e-6
    CIO, CEO, COP = bundle_cov(res, E, 'CIO', 'CEO', 'COP')
    lines = bundle_result_lines(res, E, CIO, CEO, COP)
    n = check_report_lines(lines, ref_path=ref, demo_x0=True, x0_tol=1e-4)
    assert len(lines) >= 430 and n >= len(lines) - 2


@pytest.mark.parametrize('name,variant', [('tiny', 'plain'), ('small', 'plain'), ('small', 'priors'), ('C1', 'plain'),
                                          ('tiny', 'selfcal'), ('small', 'selfcal'), ('tiny', 'groups4'), ('small', 'groups4')])
def test_signature_group_kernel(hip, name, variant, monkeypatch):
    """k_build_sig (signature groups: points seen by the same cameras share one dense block on
    the matrix cores) forced on for scenes whose groups are short, where it is off by default:
    step parity with the oracle's full sparse solve, same step as the tile kernel, and the
    same bundle result as the oracle.  (At C3, where it is the default, the full-size property
    test runs it.)"""
    from dbat_amd import bundle
    s, truth = synth_struct(name, variant)
    if (name, variant) == ('tiny', 'groups4'):
        # 7 of its 9 batches hold points whose cameras span more than 16 IO columns: the plan would send ALL points down the
        # matrix-core path of the heavy points (plan.hpp: a tile kernel for two batches is not worth its launch) and no
        # tile would be left to test -- the column lists keep the two tiles ('small' / groups4 runs both paths together)
        monkeypatch.setenv('DBAT_HIP_HEAVY', '0')
    so, x0, w = oracle_setup(s)
    R = np.sqrt(w)
    r_o, K = o.brown_euler_cam4(x0, so, jac=True)
    J = (sp.diags(R) @ K).tocsc()
    p_o, *_ = o._scaled_gn(J, R * r_o)
    monkeypatch.setenv('DBAT_HIP_SIG', '0')
    h = hip.Handle(s)
    try:
        assert h.build_kernel_name() != 'k_build_sig'
        p_tile, _ = h.linearize_solve(x0, 0.0, True)
    finally:
        h.close()
    monkeypatch.setenv('DBAT_HIP_SIG', '2')
    h = hip.Handle(s)
    try:
        assert h.build_kernel_name() == 'k_build_sig'
        p_h, st = h.linearize_solve(x0, 0.0, True)
        assert not st['singular']
        assert relerr(p_h, p_o) < TOL_STEP and relerr(p_h, p_tile) < 1e-9
        assert abs(st['f'] - 0.5 * (R * r_o) @ (R * r_o)) <= 1e-11 * st['f']
        Jp = J @ p_o
        assert abs(st['JpJp'] - Jp @ Jp) <= 1e-7 * (Jp @ Jp)
        assert relerr(h.gradient(), J.T @ (R * r_o)) < 1e-10
        assert relerr(h.colnorms(), np.sqrt(np.asarray(J.multiply(J).sum(0)).ravel())) < 1e-10
        JTJ = (J.T @ J).tocsc()
        lam = 1e-4 * JTJ.diagonal().sum() / J.shape[1]
        q_o, _ = o.normal_solve((JTJ + lam * sp.identity(J.shape[1])).tocsc(), -(J.T @ (R * r_o)))
        q_h, _ = h.linearize_solve(x0, lam, False)
        assert relerr(q_h, q_o) < TOL_STEP
    finally:
        h.close()
    if name != 'C1':
        for damping in ('gna', 'lm', 'lmp'):
            res, ok, iters, s0, E = bundle(s, damping)
            ro, oko, ito, s0o, Eo = o.bundle(s, damping)
            assert ok and oko and relerr(E.x, Eo.x) < TOL_X
            check_history(E, Eo, iters, ito, damping, s=s)


@pytest.mark.parametrize('variant', ['plain', 'selfcal'])
def test_tile_kernels_without_signature_groups(hip, variant, monkeypatch):
    """The tile kernels that take a scene whose signature groups are too short (real projects: irregular
    visibility): k_build_tile3 (fixed IO, two producer groups) and k_build_tile2 (self-calibration), forced here
    by DBAT_HIP_SIG=0 on a scene the signature kernel would take: step and bundle result as the oracle's."""
    from dbat_amd import bundle
    monkeypatch.setenv('DBAT_HIP_SIG', '0')
    s, truth = synth_struct('small', variant)
    so, x0, w = oracle_setup(s)
    R = np.sqrt(w)
    r_o, K = o.brown_euler_cam4(x0, so, jac=True)
    p_o, *_ = o._scaled_gn((sp.diags(R) @ K).tocsc(), R * r_o)
    h = hip.Handle(s)
    try:
        assert h.build_kernel_name() == ('k_build_tile3' if variant == 'plain' else 'k_build_tile2')
        p_h, st = h.linearize_solve(x0, 0.0, True)
    finally:
        h.close()
    assert relerr(p_h, p_o) < TOL_STEP
    res, ok, iters, s0, E = bundle(s, 'lm')
    ro, oko, ito, s0o, Eo = o.bundle(s, 'lm')
    assert ok and oko and relerr(E.x, Eo.x) < TOL_X


@pytest.mark.parametrize('name', ['camcal3', 'tiny-selfcal', 'tiny-imagevar', 'tiny-priors', 'tiny-groups4', 'small-priors'])
def test_jacobian_csc_matches_oracle(hip, name):
    """dbat_hip_jacobian_csc: the whole J (image rows + prior rows) as a CSC matrix in the reference's row and
    column order, weighted and unweighted, against the oracle's sparse J (multi_res.m:300-313, prior_obs.m)."""
    s = dict(cases())[name]()
    so, x0, w = oracle_setup(s)
    rng = np.random.default_rng(3)
    x = x0 + 1e-5 * rng.standard_normal(len(x0)) * np.maximum(1e-3, np.abs(x0))
    r_o, K = o.brown_euler_cam4(x, so, jac=True)
    h = hip.Handle(s)
    try:
        Ju, Jw = h.jacobian_csc(x, False), h.jacobian_csc(x, True)
    finally:
        h.close()
    K = K.tocsc(); K.sort_indices()
    assert Ju.shape == K.shape and Ju.has_sorted_indices
    assert abs(Ju - K).max() <= TOL_BLOCK * abs(K).max()
    assert abs(Jw - sp.diags(np.sqrt(w)) @ K).max() <= TOL_BLOCK * abs(sp.diags(np.sqrt(w)) @ K).max()
    assert Ju.nnz <= K.nnz + 1 and (Ju != 0).sum() >= (K != 0).sum() - 1          # same pattern (explicit zeros aside)
    from dbat_amd import bundle
    res, ok, iters, s0, E = bundle(s, 'gna', jacobian=True)
    assert E.final.weighted.J.shape == (E.numObs, E.numParams)
    g = E.final.weighted.J.T @ E.final.weighted.r
    assert np.linalg.norm(g) <= 1e-4 * np.linalg.norm(E.final.weighted.J.data) * np.linalg.norm(E.final.weighted.r)   # a stationary point


def test_stage_timers_of_a_solve(hip):
    """dbat_hip_result.stage_s (E.timeStages): hipEvent stage timers of the damping loop add up to the loop's
    wall time (minus the host work before the first launch), every stage of an LM solve is visited, and LM's first
    linearisation is taken for its trace alone."""
    from dbat_amd import bundle
    s, _ = synth_struct('small', 'plain')
    res, ok, iters, s0, E = bundle(s, 'lm', store_trace=False)
    assert ok
    st = E.timeStages
    assert set(st) == {'linearise', 'factor_solve', 'backsub', 'residual', 'other'}
    assert all(v >= 0 for v in st.values()) and min(st['linearise'], st['factor_solve'], st['backsub'], st['residual']) > 0
    assert 0.5 * E.time < sum(st.values()) <= 1.05 * E.time + 1e-3
    h = hip.Handle(s)
    try:
        opt = hip.default_options('lm')
        opt.store_trace = 0
        x, r, rr, damp, aux, T = h.solve(h.serialize(), opt)
        assert r.n_trace_only == 1 and r.n_linearizations == r.n_solves      # one build per solve; the last accepted point is not linearised
    finally:
        h.close()


@pytest.mark.parametrize('case,variant', [('tiny', 'plain'), ('small', 'priors'), ('small', 'selfcal'),
                                          ('small', 'imagevar'), ('small', 'groups4'), ('C1', 'plain')])
def test_lm_trace_only_pass(hip, case, variant):
    """Levenberg-Marquardt's lambda0 = c trace(J'J)/n (levenberg_marquardt.m:88-90) comes from a streaming pass that
    forms no normal equations (k_trace_cm): the same trace as the full linearisation reports, and the same as the
    oracle's J'J where that is small."""
    s, _ = synth_struct(case, variant)
    h = hip.Handle(s)
    try:
        x0 = h.serialize()
        _, st = h.linearize_solve(x0, 0.0, False)
        opt = hip.default_options('lm')
        opt.store_trace = 0
        opt.max_iter = 1
        x, r, rr, damp, aux, T = h.solve(x0, opt)
        assert r.n_trace_only == 1
        lam0 = damp[0]
        assert abs(lam0 - abs(opt.lambda0) * st['trace'] / h.n) <= 1e-12 * lam0
        if case != 'C1':
            so, x0o, w = oracle_setup(s)
            _, K = o.brown_euler_cam4(x0o, so, jac=True)
            tr = float((sp.diags(w) @ K.multiply(K)).sum())
            assert abs(lam0 - abs(opt.lambda0) * tr / h.n) <= 1e-11 * lam0
    finally:
        h.close()


@pytest.mark.parametrize('damping', ['gna', 'lm'])
def test_c_driver_solve_matches_bundle(hip, damping, tmp_path):
    """create -> solve -> final residuals -> destroy from a compiled C caller of the ABI
    (tests/abi_c_driver.c, no Python in the process): same x, sigma0 and iteration count as
    bundle() through the ctypes binding."""
    import json
    import subprocess
    from dbat_amd import bundle, _hip
    from helpers import build_abi_c_driver, dump_problem
    s, _ = synth_struct('tiny', 'priors')
    path = str(tmp_path / 'problem.bin')
    dump_problem(s, path)
    r = subprocess.run([build_abi_c_driver(), 'solve', path, str(_hip.DAMP[damping])], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout)
    res, ok, iters, s0, E = bundle(s, damping)
    assert out['code'] == E.code == 0
    assert relerr(np.array(out['x']), E.x) < 1e-9
    assert abs(out['sigma0'] - s0) < 1e-9 * s0
    assert abs(out['rtr'] - float(E.final.weighted.r @ E.final.weighted.r)) < 1e-8 * out['rtr']
    if damping != 'lm':
        assert out['iters'] == iters


@pytest.mark.parametrize('case', ['tiny', 'camcal', 'roma', 'C1'])
def test_forward_intersection_hip(hip, case):
    """dbat_hip_forwintersect (photogrammetry/forwintersect.m on the device) against the host
    restatement oracle/initial_oracle.py forwintersect: synthetic scene, the camcal demo after its
    resection (first error 30873.9 of camcal-dbatreport.txt:41 hangs on these points), the
    roma script data (26 321 points, initial EO from the table) and a 10k-point scene; also
    skipPrior, an id subset, and points with a single ray (NaN)."""
    from dbat_amd import initial
    import initial_oracle
    if case == 'camcal':
        from helpers import camcal_demo_struct
        s = camcal_demo_struct(3)
    elif case == 'roma':
        s = roma_struct()
    else:
        s = synth_struct(case, 'priors' if case == 'tiny' else 'plain')[0]
    if case == 'tiny':                                 # a one-ray point
        p = s.IP.pt[5]
        keep = ~((s.IP.pt == p) & (np.arange(len(s.IP.pt)) != 5))
        s.IP.val, s.IP.std = s.IP.val[:, keep], s.IP.std[:, keep]
        s.IP.cam, s.IP.pt = s.IP.cam[keep], s.IP.pt[keep]
    s.OP.val = s.OP.val.copy()
    for kw in (dict(), dict(skipPrior=True), dict(ids=s.OP.id[::3])):
        a = initial_oracle.forwintersect(s, **kw)
        b = initial.forwintersect(s, **kw)
        assert np.array_equal(np.isnan(a.OP.val), np.isnan(b.OP.val))
        m = ~np.isnan(a.OP.val)
        assert np.abs(a.OP.val[m] - b.OP.val[m]).max() <= 1e-9 * max(1.0, np.abs(a.OP.val[m]).max())
    if case == 'tiny':
        assert np.isnan(initial.forwintersect(s).OP.val[:, p]).all()


@pytest.mark.parametrize('name,sig', [('small', '0'), ('small', '2'), ('C1', '0'), ('C1', '2')])
def test_run_to_run_repeatability(hip, name, sig, monkeypatch):
    """The reduced system is summed with floating-point atomics (LDS and HBM), so two runs differ by
    the rounding of a different summation order -- and by nothing else: a race between workgroups
    (a tile read before it is published, a panel reused too early) shows up as run-to-run
    differences far above rounding.  20 repeats of linearise + solve, tile kernels and signature
    kernels: the step must repeat to 1e-10 relative (observed: 1e-13 ... 1e-12), the objective value
    bit for bit (one reduction order)."""
    from dbat_amd import synth
    monkeypatch.setenv('DBAT_HIP_SIG', sig)
    s, _ = synth.make_scene(name)
    h = hip.Handle(s)
    try:
        x0 = h.serialize()
        ref, f_ref, worst = None, None, 0.0
        for i in range(20):
            p, st = h.linearize_solve(x0, 0.0, True)
            f = h.residual(x0, want_r=False)
            if ref is None:
                ref, f_ref = p.copy(), f
            worst = max(worst, relerr(p, ref))
            assert f == f_ref
            assert not st['singular']
        assert worst < 1e-10, worst
    finally:
        h.close()


def _det_scene(name):
    from dbat_amd import synth
    if name == 'camcal':
        return camcal_struct(3)
    if name == 'roma':
        return roma_struct()
    if name == 'sxb':
        from helpers import sxb_struct
        return sxb_struct()
    base, _, var = name.partition('+')
    return synth.make_scene(base, selfcal=True)[0] if var == 'io' else synth.make_scene(base)[0]


@pytest.mark.parametrize('sig', ['2', '0'])
@pytest.mark.parametrize('name', ['small', 'small+io', 'C1', 'C1+io', 'camcal', 'roma', 'sxb'])
def test_deterministic_mode_repeats_bit_for_bit(hip, name, sig, monkeypatch):
    """dbat_hip_set_deterministic: what the reference has by construction, where one thread forms J'J
    (gauss_newton_armijo.m:166-174).  Round 5: no ordering of the atomics any more -- the sums are made
    exact (csrc/kernels.hpp DevProblem::deterministic), so the mode covers every build path: the signature-group kernel
    (DBAT_HIP_SIG=2) and, for scenes with irregular visibility -- the reference's own projects: camcal (every point in every
    image, nine IO unknowns: heavy points), roma (26 321 points), sxb (prior observations) -- the column-list kernel with
    its giant-point companion.  Ten linearise + solve steps and two whole bundles repeat BIT FOR BIT, and the deterministic
    step agrees with the default one to rounding."""
    from dbat_amd import bundle
    if name in ('camcal', 'roma', 'sxb') and sig == '2':
        pytest.skip('the real projects take the path the plan chooses (one run)')
    monkeypatch.setenv('DBAT_HIP_SIG', sig)
    s = _det_scene(name)
    h = hip.Handle(s)
    try:
        x0 = h.serialize()
        p_default, st0 = h.linearize_solve(x0, 0.0, True)
        h.set_deterministic(True)
        ref = None
        for i in range(10):
            p, st = h.linearize_solve(x0, 0.0, True)
            assert st['singular'] == st0['singular']
            if ref is None:
                ref = p.copy()
            assert np.array_equal(p, ref), relerr(p, ref)
        assert relerr(ref, p_default) < 1e-8
        h.set_deterministic(False)                      # ... and off again: the default mode's step
        p_off, _ = h.linearize_solve(x0, 0.0, True)
        assert relerr(p_off, p_default) < 1e-10
    finally:
        h.close()
    damping = 'gna' if name in ('camcal', 'roma', 'sxb') else 'lm'
    runs = [bundle(s, damping, deterministic=True) for _ in range(2)]
    assert runs[0][1] and np.array_equal(runs[0][4].x, runs[1][4].x) and runs[0][3] == runs[1][3]
    assert np.array_equal(np.asarray(runs[0][4].res), np.asarray(runs[1][4].res))
    ref_run = bundle(s, damping)
    # the same adjustment as the default mode's (LM's iteration count is rounding noise at convTol = 1e-6: check_history)
    assert abs(runs[0][3] - ref_run[3]) < 1e-9 * ref_run[3] and relerr(runs[0][4].x, ref_run[4].x) < 1e-8
    if damping == 'gna':
        assert runs[0][2] == ref_run[2]


@pytest.mark.parametrize('rays,groups', [(6, 1), (10, 1), (6, 4)])
def test_selfcal_io_rows_summed_per_point(hip, rays, groups, monkeypatch):
    """Self-calibration: tiles whose cameras share one IO block run the k_build_sig instantiation
    that sums a point's IO rows in pass 1 (registers) instead of LDS atomics in pass 2.  On a scene
    with long signature groups (several rounds per chunk, one and two lanes per point, chunks that
    end in a partial round) its step must agree with the instantiation with the atomics to rounding;
    with four IO blocks both instantiations run side by side."""
    from dbat_amd import synth
    monkeypatch.setenv('DBAT_HIP_SIG', '2')
    s, _ = synth.make_scene('C1', selfcal=True, rays=rays, groups=groups)
    def step():
        h = hip.Handle(s)
        try:
            p, st = h.linearize_solve(h.serialize(), 0.0, True)
            assert not st['singular']
            return p
        finally:
            h.close()
    p_new = step()
    monkeypatch.setenv('DBAT_HIP_SIG_IOS_OFF', '1')
    assert relerr(p_new, step()) < 1e-9


@pytest.mark.parametrize('name,selfcal', [('small', False), ('small', True), ('C1', False)])
def test_cholesky_split_sums(hip, name, selfcal, monkeypatch):
    """The dataflow Cholesky cuts long left-looking sums into helper tasks (partial sums in scratch
    tiles, chol_df.hpp DfJob).  Forced here on small systems -- every sum of more than two products
    split into pieces of two -- the step must agree with the default schedule to rounding."""
    from dbat_amd import synth
    s, _ = synth.make_scene(name, selfcal=selfcal) if selfcal else synth.make_scene(name)
    def step():
        h = hip.Handle(s)
        try:
            p, st = h.linearize_solve(h.serialize(), 0.0, True)
            assert not st['singular']
            return p
        finally:
            h.close()
    ref = step()
    monkeypatch.setenv('DBAT_HIP_DF_SPLIT', '2')
    monkeypatch.setenv('DBAT_HIP_DF_CHUNK', '2')
    assert relerr(step(), ref) < 1e-9


@pytest.mark.parametrize('switch,value', [('DBAT_HIP_ND_OFF', '1'), ('DBAT_HIP_ND_LEAF', '8'), ('DBAT_HIP_ND_PAD_ALL', '1'),
                                          ('DBAT_HIP_ND_JOIN_SMALL', '0'), ('DBAT_HIP_SPRANK_OFF', '1'), ('DBAT_HIP_TILE_BMIN', '1'),
                                          ('DBAT_HIP_TILE_BMIN', '6'), ('DBAT_HIP_PLAN_STATS', '2'), ('DBAT_HIP_PIVOT_STATS', '1'),
                                          ('DBAT_HIP_PLAN_THREADS', '3'), ('DBAT_HIP_DF_CHAIN', '0'), ('DBAT_HIP_DF_CHAIN_WG', '1'),
                                          ('DBAT_HIP_DF_CHAIN_WG', '3'), ('DBAT_HIP_DF_L2', '1')])
def test_product_switches_leave_the_result_alone(hip, switch, value, monkeypatch):
    """csrc/env.hpp: a product switch selects a layout or a schedule (the dissection of the camera network, the tile
    length, the threads of the host plan) or prints statistics -- the step is the same to rounding whatever its value.
    (The switches that select kernels have their own tests: SIG, SIG_IOS_OFF, CMAX, BT, GIANT_THREADS, MG_REPLICATED,
    DF_SPLIT / DF_CHUNK, HEAVY / HEAVY_KS: test_every_point_in_every_image, test_mixed_tiled_and_heavy_points.)  Fixed IO and self-calibration, 'small' (several tiles, a dissection with separators)."""
    from dbat_amd import synth
    for selfcal in (False, True):
        s, _ = synth.make_scene('small', selfcal=selfcal) if selfcal else synth.make_scene('small')
        def step():
            h = hip.Handle(s)
            try:
                p, st = h.linearize_solve(h.serialize(), 0.0, True)
                assert not st['singular'] and h.structural_rank_ok()
                return p
            finally:
                h.close()
        ref = step()
        monkeypatch.setenv(switch, value)
        assert relerr(step(), ref) < 1e-9
        monkeypatch.delenv(switch)


def test_cholesky_task_orders_in_the_measurement_build(hip):
    """Every candidate order of the factorisation's task list is a topological order and must give the same
    step.  DBAT_HIP_DF_ORDER is a measurement switch (csrc/env.hpp): the product library refuses it, so the
    orders are forced in a child process that loads libdbat_hip_prof.so (the same sources, -DDBAT_HIP_PROFILING)."""
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = os.path.join(ROOT, 'dbat_amd', 'libdbat_hip_prof.so')
    if not os.path.exists(prof):
        pytest.skip('libdbat_hip_prof.so not built (make -C dbat_amd/csrc prof)')
    code = (
        "import os, sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from dbat_amd import _hip, synth\n"
        "s, _ = synth.make_scene('small', selfcal=True)\n"
        "def step():\n"
        "    h = _hip.Handle(s)\n"
        "    try:\n"
        "        p, st = h.linearize_solve(h.serialize(), 0.0, True)\n"
        "        assert not st['singular']\n"
        "        return p\n"
        "    finally:\n"
        "        h.close()\n"
        "ref = step()\n"
        "os.environ['DBAT_HIP_DF_SPLIT'] = '2'; os.environ['DBAT_HIP_DF_CHUNK'] = '2'\n"
        "for order in ('0', '4', '6', '7'):\n"
        "    os.environ['DBAT_HIP_DF_ORDER'] = order\n"
        "    e = np.linalg.norm(step() - ref) / np.linalg.norm(ref)\n"
        "    assert e < 1e-9, (order, e)\n"
        "print('ORDERS_OK')\n" % ROOT)
    env = dict(os.environ, DBAT_AMD_LIB='prof')
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'ORDERS_OK' in r.stdout, r.stderr[-2000:]
    # ... and the product library refuses the switch
    os.environ['DBAT_HIP_DF_ORDER'] = '4'
    try:
        from dbat_amd import synth
        with pytest.raises(hip.DbatHipError) as e:
            hip.Handle(synth.make_scene('tiny')[0])
        assert 'measurement switch' in str(e.value)
    finally:
        del os.environ['DBAT_HIP_DF_ORDER']


@pytest.mark.parametrize('rays,selfcal', [(12, False), (13, False), (11, True)])
def test_signature_group_kernel_five_row_blocks(hip, rays, selfcal, monkeypatch):
    """Chunks with 11 ... 13 cameras per point need five 16-row blocks (the four-wave instantiations
    of k_build_sig, fixed IO and self-calibrating): step parity with the oracle and with the tile
    kernels, bundle result as the oracle's."""
    from dbat_amd import bundle, synth
    s, truth = synth.make_scene('small', rays=rays, selfcal=selfcal)
    so, x0, w = oracle_setup(s)
    R = np.sqrt(w)
    r_o, K = o.brown_euler_cam4(x0, so, jac=True)
    J = (sp.diags(R) @ K).tocsc()
    p_o, *_ = o._scaled_gn(J, R * r_o)
    steps = {}
    for sig in ('0', '2'):
        monkeypatch.setenv('DBAT_HIP_SIG', sig)
        h = hip.Handle(s)
        try:
            assert (h.build_kernel_name() == 'k_build_sig') == (sig == '2')
            steps[sig], st = h.linearize_solve(x0, 0.0, True)
            Jp = J @ p_o
            assert abs(st['JpJp'] - Jp @ Jp) <= 1e-7 * (Jp @ Jp)
        finally:
            h.close()
    assert relerr(steps['2'], p_o) < TOL_STEP and relerr(steps['2'], steps['0']) < 1e-9
    res, ok, iters, s0, E = bundle(s, 'gna')
    ro, oko, ito, s0o, Eo = o.bundle(s, 'gna')
    assert ok and oko and iters == ito and relerr(E.x, Eo.x) < TOL_X
