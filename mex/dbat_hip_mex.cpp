// MEX gateway: MATLAB <-> include/dbat_hip.h.
//
// Built only where MATLAB's mex.h exists (not in this repo's CI image):
//   mex -R2018a CXXFLAGS='$CXXFLAGS -std=c++17' -I../include dbat_hip_mex.cpp ...
//       -L../dbat_amd -ldbat_hip
// Conventions follow the reference's own MEX file code/test/postcov/icpc_mex.c
// (:495-611): argument validation with mexErrMsgIdAndTxt("DBAT:<fn>:<id>"),
// interleaved-complex API, a same-named .m stub that errors when the MEX file
// is missing (code/test/postcov/icpc_mex.m:14).
//
//   [x,code,iters,sigma0,res,damp,aux,T,ru,rw,time,CEO,CIO,COP,Jw,Ju] = dbat_hip_mex(P, opt)
//   id = dbat_hip_mex('commId')     128 bytes (uint8) of a fresh RCCL unique id: rank 0 of a multi-GPU run creates it
//                                   and hands it to the other MATLAB workers (labSend / a file), who pass it as opt.commId
//   dbat_hip_mex('clear')           destroy the handle the gateway keeps between calls (below) and unlock the MEX file
//
// PLAN REUSE.  BUNDLE re-enters from any s at no set-up cost (the serial indices stay in s.bundle.serial / deserial,
// bundle.m:156-159).  The gateway keeps the handle of its last one-rank call (mexLock; destroyed by mexAtExit, by
// 'clear', or when a problem of another structure arrives): a call whose structure key (dbat_hip_structure_key: sizes,
// visibility, observations, masks, blocks, prior pattern) equals the kept handle's only sends the new parameter values
// (dbat_hip_set_values) -- the second bundle_hip on a project, and every later one, skips the host plan and its uploads.
//
// P   struct with the flattened DBAT struct fields built by bundle_hip.m
// opt struct: damping (0..3), maxIter, convTol, absTerm, singularTest, trace,
//             wantJ (0: never, 1: always, 2: after a failed run, code -2 / -4 -- E.final.weighted.J, bundle.m:341-350,372-446),
//             liveTrace (print the LSA function's line per iteration while the loop runs: gauss_newton_armijo.m:119-128),
//             wantCov (posterior covariance blocks), deterministic (dbat_hip_set_deterministic),
//             device (HIP device ordinal), shardRank, shardCount (this worker's share of the object points,
//             dbat_hip.h "several GPUs"), commId (uint8 [128] from dbat_hip_mex('commId'); required when shardCount > 1)
#include <cstring>
#include <vector>

#include "mex.h"
#include "dbat_hip.h"

static const mxArray *field(const mxArray *s, const char *name) {
    const mxArray *f = mxGetField(s, 0, name);
    if (!f) mexErrMsgIdAndTxt("DBAT:dbat_hip_mex:badInput", "missing field %s", name);
    return f;
}
static double scalar(const mxArray *s, const char *name) { return mxGetScalar(field(s, name)); }

// J (dbat_hip_jacobian_csc) as a MATLAB sparse matrix, as [r,J]=resFun(x) returns it (brown_euler_cam4.m:163-182)
// nullptr on failure (dbat_hip_last_error says why): the caller releases the handle BEFORE it raises -- mexErrMsgIdAndTxt
// does not return, and a handle that is still alive then would leak its device memory inside the MATLAB process.
static mxArray *sparse_jacobian(dbat_hip_handle *h, const double *x, int weighted, int64_t m, int64_t n) {
    int64_t nnz = 0;
    if (dbat_hip_jacobian_csc(h, x, weighted, &nnz, nullptr, nullptr, nullptr) != DBAT_HIP_OK) return nullptr;
    mxArray *J = mxCreateSparse((mwSize)m, (mwSize)n, (mwSize)(nnz > 0 ? nnz : 1), mxREAL);
    std::vector<int64_t> colptr((size_t)n + 1), rowidx((size_t)nnz);
    if (dbat_hip_jacobian_csc(h, x, weighted, &nnz, colptr.data(), rowidx.data(), mxGetDoubles(J)) != DBAT_HIP_OK) {
        mxDestroyArray(J);
        return nullptr;
    }
    mwIndex *jc = mxGetJc(J), *ir = mxGetIr(J);
    for (int64_t c = 0; c <= n; ++c) jc[c] = (mwIndex)colptr[(size_t)c];
    for (int64_t e = 0; e < nnz; ++e) ir[e] = (mwIndex)rowidx[(size_t)e];
    return J;
}

// A MATLAB function handle as a dbat_hip_term_fn / dbat_hip_veto_fn: feval(fh, args...) with errors trapped, so that
// the handle is released before MATLAB unwinds.
struct MatlabFn { const mxArray *fh; mxArray *error; };
static int32_t call_handle(MatlabFn *f, int nargs, mxArray **args) {
    if (f->error) return 1;
    mxArray *in[3] = {const_cast<mxArray *>(f->fh), args[0], nargs > 1 ? args[1] : nullptr};
    mxArray *out = nullptr;
    f->error = mexCallMATLABWithTrap(1, &out, nargs + 1, in, "feval");
    for (int i = 0; i < nargs; ++i) mxDestroyArray(args[i]);
    if (f->error) return 1;                            // stop (termFun) / reject (vetoFun); rethrown after the solve
    const int32_t v = out && !mxIsEmpty(out) && mxGetScalar(out) != 0;
    if (out) mxDestroyArray(out);
    return v;
}
static mxArray *column(const double *p, int64_t k) {
    mxArray *a = mxCreateDoubleMatrix((mwSize)k, 1, mxREAL);
    std::memcpy(mxGetDoubles(a), p, sizeof(double) * (size_t)k);
    return a;
}
static int32_t call_term(void *user, const double *Jp, const double *r, int64_t m) {
    mxArray *args[2] = {column(Jp, m), column(r, m)};
    return call_handle(static_cast<MatlabFn *>(user), 2, args);
}
static int32_t call_veto(void *user, const double *x, int64_t n) {
    mxArray *args[1] = {column(x, n)};
    return call_handle(static_cast<MatlabFn *>(user), 1, args);
}

// 'trace': the line the LSA function prints, while the loop runs (gauss_newton_armijo.m:119-128, gauss_markov.m:74-76,
// levenberg_marquardt.m:138-147, levenberg_marquardt_powell.m:160-164)
static void print_trace(void *, int32_t damping, int32_t n, double res, double damp, int32_t step, double rho) {
    static const char *step_str[3] = {"GN", "IP", "CP"};
    if (damping == DBAT_HIP_DAMP_GNA) {
        if (damp != damp) mexPrintf("Gauss-Newton-Armijo: iteration %d, residual norm=%.2g\n", (int)n, res);
        else if (damp == 1.0) mexPrintf("Gauss-Newton-Armijo: iteration %d, residual norm=%.2g, last alpha=1\n", (int)n, res);
        else mexPrintf("Gauss-Newton-Armijo: iteration %d, residual norm=%.2g, last alpha=1/%.0f\n", (int)n, res, 1.0 / damp);   // rats(2^-k)
    } else if (damping == DBAT_HIP_DAMP_LM) {
        if (damp != damp) mexPrintf("Levenberg-Marquardt: iteration %d, residual norm=%.2g\n", (int)n, res);
        else mexPrintf("Levenberg-Marquardt: iteration %d, residual norm=%.2g, lambda=%.2g\n", (int)n, res, damp);
    } else if (damping == DBAT_HIP_DAMP_LMP) {
        mexPrintf("Levenberg-Marquardt-Powell: iteration %d, residual norm=%.2g, delta=%.2g, step=%s, rho=%.1f\n", (int)n, res, damp,
                  step >= 0 && step < 3 ? step_str[step] : "?", rho);
    } else mexPrintf("Gauss-Markov: iteration %d, residual norm=%.2g\n", (int)n, res);
}

// the handle kept between calls (one-rank problems; see PLAN REUSE above)
static dbat_hip_handle *g_kept = nullptr;
static void drop_kept() {
    if (g_kept) { dbat_hip_destroy(g_kept); g_kept = nullptr; }
}
// the handle for pb: the kept one with pb's values if the structure is the same, else a new one (nullptr: dbat_hip_last_error)
static dbat_hip_handle *acquire(const dbat_hip_problem &pb) {
    dbat_hip_handle *h = g_kept;
    g_kept = nullptr;                              // (in use: an error path destroys it instead of keeping a half-used handle)
    if (h && pb.shard_count == 1) {
        uint64_t kp[2], kh[2];
        if (dbat_hip_structure_key(&pb, kp) == DBAT_HIP_OK && dbat_hip_handle_key(h, kh) == DBAT_HIP_OK &&
            kp[0] == kh[0] && kp[1] == kh[1] && dbat_hip_set_values(h, &pb) == DBAT_HIP_OK)
            return h;
    }
    if (h) dbat_hip_destroy(h);
    h = nullptr;
    return dbat_hip_create(&pb, &h) == DBAT_HIP_OK ? h : nullptr;
}
static void release(dbat_hip_handle *h, bool one_rank) {
    if (!one_rank) { dbat_hip_destroy(h); return; }          // (a sharded handle owns a communicator: its life is the call's)
    dbat_hip_set_deterministic(h, 0);
    g_kept = h;
    if (!mexIsLocked()) { mexLock(); mexAtExit(drop_kept); }
}

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    if (nrhs == 1 && mxIsChar(prhs[0])) {          // dbat_hip_mex('commId') / dbat_hip_mex('clear')
        char what[16] = {0};
        mxGetString(prhs[0], what, sizeof(what));
        if (std::strcmp(what, "clear") == 0) {
            drop_kept();
            if (mexIsLocked()) mexUnlock();
            return;
        }
        if (std::strcmp(what, "commId") != 0) mexErrMsgIdAndTxt("DBAT:dbat_hip_mex:badInput", "unknown request %s", what);
        plhs[0] = mxCreateNumericMatrix(1, DBAT_HIP_UNIQUE_ID_BYTES, mxUINT8_CLASS, mxREAL);
        if (dbat_hip_comm_unique_id((uint8_t *)mxGetData(plhs[0])) != DBAT_HIP_OK)
            mexErrMsgIdAndTxt("DBAT:bundle:internal", "%s", dbat_hip_last_error());
        return;
    }
    if (nrhs != 2 || !mxIsStruct(prhs[0]) || !mxIsStruct(prhs[1]))
        mexErrMsgIdAndTxt("DBAT:dbat_hip_mex:badInput", "usage: dbat_hip_mex(P, opt) or dbat_hip_mex('commId')");
    const mxArray *P = prhs[0], *O = prhs[1];
    dbat_hip_problem pb;
    std::memset(&pb, 0, sizeof(pb));
    pb.abi_version = DBAT_HIP_ABI_VERSION;
    pb.n_images = (int32_t)scalar(P, "nImages");
    pb.n_points = (int32_t)scalar(P, "nOP");
    pb.n_obs = (int64_t)scalar(P, "nIP");
    pb.dist_model = (int32_t)scalar(P, "distModel");
    pb.nK = (int32_t)scalar(P, "nK");
    pb.nP = (int32_t)scalar(P, "nP");
    // int32 / uint8 arrays are prepared (0-based, column-major) by bundle_hip.m
    pb.ip_cam = (const int32_t *)mxGetData(field(P, "ipCam"));
    pb.ip_pt = (const int32_t *)mxGetData(field(P, "ipPt"));
    pb.ip_val = mxGetDoubles(field(P, "ipVal"));
    pb.ip_std = mxGetDoubles(field(P, "ipStd"));
    pb.IO_val = mxGetDoubles(field(P, "IO"));
    pb.px_size = mxGetDoubles(field(P, "pxSize"));
    pb.EO_val = mxGetDoubles(field(P, "EO"));
    pb.OP_val = mxGetDoubles(field(P, "OP"));
    pb.est_IO = (const uint8_t *)mxGetData(field(P, "estIO"));
    pb.est_EO = (const uint8_t *)mxGetData(field(P, "estEO"));
    pb.est_OP = (const uint8_t *)mxGetData(field(P, "estOP"));
    pb.IO_block = (const int32_t *)mxGetData(field(P, "IOblock"));
    pb.EO_block = (const int32_t *)mxGetData(field(P, "EOblock"));
    pb.prior_IO_use = (const uint8_t *)mxGetData(field(P, "useIO"));
    pb.prior_IO_val = mxGetDoubles(field(P, "priorIO"));
    pb.prior_IO_std = mxGetDoubles(field(P, "stdIO"));
    pb.prior_EO_use = (const uint8_t *)mxGetData(field(P, "useEO"));
    pb.prior_EO_val = mxGetDoubles(field(P, "priorEO"));
    pb.prior_EO_std = mxGetDoubles(field(P, "stdEO"));
    pb.prior_OP_use = (const uint8_t *)mxGetData(field(P, "useOP"));
    pb.prior_OP_val = mxGetDoubles(field(P, "priorOP"));
    pb.prior_OP_std = mxGetDoubles(field(P, "stdOP"));
    pb.device = (int32_t)scalar(O, "device");
    pb.shard_rank = (int32_t)scalar(O, "shardRank");
    pb.shard_count = (int32_t)scalar(O, "shardCount");
    const mxArray *cid = field(O, "commId");
    if (pb.shard_count < 1 || pb.shard_rank < 0 || pb.shard_rank >= pb.shard_count)
        mexErrMsgIdAndTxt("DBAT:bundle:badInput", "shardRank %d outside shardCount %d", (int)pb.shard_rank, (int)pb.shard_count);
    if (pb.shard_count > 1 && (!mxIsUint8(cid) || mxGetNumberOfElements(cid) != DBAT_HIP_UNIQUE_ID_BYTES))
        mexErrMsgIdAndTxt("DBAT:bundle:badInput", "shardCount > 1 needs opt.commId: the %d bytes of dbat_hip_mex('commId') from rank 0",
                          DBAT_HIP_UNIQUE_ID_BYTES);

    const bool one_rank = pb.shard_count == 1;
    dbat_hip_handle *h = acquire(pb);
    if (!h) mexErrMsgIdAndTxt("DBAT:bundle:badInput", "%s", dbat_hip_last_error());
    // several workers, one GPU each: join the communicator (collective: every worker is here at the same time)
    if (pb.shard_count > 1 && dbat_hip_comm_init(h, (const uint8_t *)mxGetData(cid)) != DBAT_HIP_OK) {
        dbat_hip_destroy(h);
        mexErrMsgIdAndTxt("DBAT:bundle:internal", "%s", dbat_hip_last_error());
    }
    if ((int)scalar(O, "deterministic") && dbat_hip_set_deterministic(h, 1) != DBAT_HIP_OK) {
        dbat_hip_destroy(h);
        mexErrMsgIdAndTxt("DBAT:bundle:badInput", "%s", dbat_hip_last_error());
    }
    dbat_hip_options opt;
    dbat_hip_default_options((int32_t)scalar(O, "damping"), &opt);
    opt.max_iter = (int32_t)scalar(O, "maxIter");
    opt.conv_tol = scalar(O, "convTol");
    opt.abs_term = (int32_t)scalar(O, "absTerm");
    opt.singular_test = (int32_t)scalar(O, "singularTest");
    opt.store_trace = (int32_t)scalar(O, "trace");
    // the caller's own termFun(Jp,r) / vetoFun(x) (function handles; [] = the tests bundle.m:168-192 builds): called back on
    // this thread, from inside dbat_hip_solve
    MatlabFn term_cb{field(O, "termFun"), nullptr}, veto_cb{field(O, "vetoFun"), nullptr};
    if (!mxIsEmpty(term_cb.fh)) { opt.term_fun = call_term; opt.term_user = &term_cb; }
    if (!mxIsEmpty(veto_cb.fh)) { opt.veto_fun = call_veto; opt.veto_user = &veto_cb; }
    if ((int)scalar(O, "liveTrace") && pb.shard_rank == 0) opt.trace_fun = print_trace;
    for (const MatlabFn *f : {&term_cb, &veto_cb})
        if (!mxIsEmpty(f->fh) && !mxIsClass(f->fh, "function_handle")) {
            dbat_hip_destroy(h);
            mexErrMsgIdAndTxt("DBAT:bundle:badInput", "opt.termFun / opt.vetoFun: a function handle or []");
        }
    const int64_t n = dbat_hip_num_params(h), m = dbat_hip_num_residuals(h);
    const int mi = opt.max_iter;
    plhs[0] = mxCreateDoubleMatrix(n, 1, mxREAL);
    double *x = mxGetDoubles(plhs[0]);
    dbat_hip_serialize(h, x);
    std::vector<double> res(mi + 3), damp(2 * mi + 4), aux(2 * mi + 4);
    mxArray *T = mxCreateDoubleMatrix(opt.store_trace ? n : 0, opt.store_trace ? mi + 2 : 0, mxREAL);
    dbat_hip_result r;
    const int rc = dbat_hip_solve(h, &opt, x, &r, res.data(), damp.data(), aux.data(),
                                  opt.store_trace ? mxGetDoubles(T) : nullptr);
    if (rc != DBAT_HIP_OK) {
        dbat_hip_destroy(h);
        mexErrMsgIdAndTxt("DBAT:bundle:internal", "%s", dbat_hip_last_error());
    }
    for (MatlabFn *f : {&term_cb, &veto_cb})
        if (f->error) {                                // an error inside a callback ended the run: hand it on
            dbat_hip_destroy(h);
            mxArray *rethrow_in[1] = {f->error};
            mexCallMATLAB(0, nullptr, 1, rethrow_in, "rethrow");
        }
    auto vec = [](const double *p, int k) {
        mxArray *a = mxCreateDoubleMatrix(1, k, mxREAL);
        std::memcpy(mxGetDoubles(a), p, sizeof(double) * k);
        return a;
    };
    if (nlhs > 1) plhs[1] = mxCreateDoubleScalar(r.code);
    if (nlhs > 2) plhs[2] = mxCreateDoubleScalar(r.iters);
    if (nlhs > 3) plhs[3] = mxCreateDoubleScalar(r.sigma0);
    if (nlhs > 4) plhs[4] = vec(res.data(), r.n_res);
    if (nlhs > 5) plhs[5] = vec(damp.data(), r.n_damp);
    if (nlhs > 6) plhs[6] = vec(aux.data(), 2 * mi + 4);
    if (nlhs > 7) { mxSetN(T, r.n_trace); plhs[7] = T; } else mxDestroyArray(T);
    if (nlhs > 8) {
        plhs[8] = mxCreateDoubleMatrix(m, 1, mxREAL);
        mxArray *rw = mxCreateDoubleMatrix(m, 1, mxREAL);
        if (r.code != -4) dbat_hip_final_residuals(h, mxGetDoubles(plhs[8]), mxGetDoubles(rw));
        if (nlhs > 9) plhs[9] = rw; else mxDestroyArray(rw);
    }
    if (nlhs > 10) {            // [total, linearise, factor+solve, back-substitution, residual evaluations, other] seconds
        const double tm[6] = {r.time_s, r.stage_s[0], r.stage_s[1], r.stage_s[2], r.stage_s[3], r.stage_s[4]};
        plhs[10] = vec(tm, 6);
    }
    if (nlhs > 11) {
        // posterior covariance blocks (bundle_cov.m 'CEO','CIO','COP') at the result:
        // plhs[11] 6 x 6 x nImages, plhs[12] nIOu x nIOu (IO unknowns in x order), plhs[13] 3 x 3 x nOP
        const mwSize dE[3] = {6, 6, (mwSize)pb.n_images}, dP[3] = {3, 3, (mwSize)pb.n_points};
        int64_t inf[24];
        dbat_hip_info(h, inf);
        const mwSize nIOu = (mwSize)(inf[0] - 6 * (int64_t)pb.n_images);
        plhs[11] = mxCreateNumericArray(3, dE, mxDOUBLE_CLASS, mxREAL);
        mxArray *cio = mxCreateDoubleMatrix(nIOu, nIOu, mxREAL), *cop = mxCreateNumericArray(3, dP, mxDOUBLE_CLASS, mxREAL);
        if (r.code == 0 && (int)scalar(O, "wantCov") &&
            dbat_hip_posterior_cov(h, x, r.sigma0, mxGetDoubles(plhs[11]), nIOu ? mxGetDoubles(cio) : nullptr,
                                   mxGetDoubles(cop), nullptr) != DBAT_HIP_OK)
            mexWarnMsgIdAndTxt("DBAT:bundle_cov:notPD", "%s", dbat_hip_last_error());
        if (nlhs > 12) plhs[12] = cio; else mxDestroyArray(cio);
        if (nlhs > 13) plhs[13] = cop; else mxDestroyArray(cop);
    }
    if (nlhs > 14) {
        // E.final.weighted.J / E.final.unweighted.J at the point the solver returned (bundle.m:341-350): on request, and
        // for the post-mortem of a failed run (bundle.m:372-446 analyses J).  One-rank handles only (dbat_hip.h).
        const int want = (int)scalar(O, "wantJ");
        const bool give = pb.shard_count == 1 && (want == 1 || (want == 2 && (r.code == -2 || r.code == -4)));
        plhs[14] = give ? sparse_jacobian(h, x, 1, m, n) : mxCreateSparse((mwSize)0, (mwSize)0, (mwSize)1, mxREAL);
        if (nlhs > 15) plhs[15] = give ? sparse_jacobian(h, x, 0, m, n) : mxCreateSparse((mwSize)0, (mwSize)0, (mwSize)1, mxREAL);
        if (!plhs[14] || (nlhs > 15 && !plhs[15])) {
            dbat_hip_destroy(h);
            mexErrMsgIdAndTxt("DBAT:bundle:internal", "%s", dbat_hip_last_error());
        }
    }
    release(h, one_rank);
}
