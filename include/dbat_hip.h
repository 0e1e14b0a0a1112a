/*
 * dbat_hip.h -- C ABI of the MI355X-native damped bundle-adjustment core.
 *
 * Drop-in boundary for the hot path of niclasborlin/dbat (MATLAB):
 *   bundle()  ->  lsa/{gauss_markov,gauss_newton_armijo,levenberg_marquardt,
 *                      levenberg_marquardt_powell}.m
 *             ->  resFun = brown_euler_cam4(x,s)  (residual + sparse Jacobian)
 *             ->  (J'*J [+lambda*I]) \ (-J'*r)
 *
 * The reference has no FFI for this path; the seam it replaces is the MATLAB
 * function-handle contract between bundle.m and lsa/ *.m plus the DBAT struct
 * (SURVEY.md section 8(b)).  Every entry point cites the reference interface
 * it stands in for (paths relative to /root/reference/code/).
 *
 * Conventions
 *   - plain C types only; all arrays are caller-owned HOST memory unless a
 *     name ends in _dev; double arrays are column-major like MATLAB's;
 *     indices are 0-based int32 (int64 where a count can exceed 2^31).
 *   - every function returns 0 on success or a negative DBAT_HIP_E* code;
 *     dbat_hip_last_error() gives the message.  Numerical failure of an
 *     adjustment is NOT an error return: it is reported through
 *     dbat_hip_result.code (0,-1,-2,-3,-4) exactly as the reference's
 *     solvers do (gauss_newton_armijo.m:38-46).
 *   - one handle per calling thread; a handle owns its device memory and its
 *     HIP stream.  No call is re-entrant on the same handle.
 *   - all device arithmetic is IEEE double.
 */
#ifndef DBAT_HIP_H
#define DBAT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DBAT_HIP_ABI_VERSION 4   /* 4: dbat_hip_options grew (trace_fun); dbat_hip_structure_key / dbat_hip_handle_key / dbat_hip_set_values; dbat_hip_bench_step ms[16], dbat_hip_info [24]; 3: dbat_hip_options grew (term_fun, veto_fun); 2: dbat_hip_result grew (stage_s, n_trace_only); dbat_hip_info [16]; dbat_hip_bench_step ms[12] */

/* error returns */
#define DBAT_HIP_OK            0
#define DBAT_HIP_EINVAL       -101  /* bad argument / inconsistent struct (DBAT:bundle:badInput) */
#define DBAT_HIP_EUNSUPPORTED -102  /* struct uses a feature outside the hot path */
#define DBAT_HIP_EDEVICE      -103  /* HIP / rocSOLVER failure */
#define DBAT_HIP_ENOMEM       -104

/* damping names of bundle.m:98-101 */
#define DBAT_HIP_DAMP_GM   0   /* 'none' / 'gm'  -> lsa/gauss_markov.m */
#define DBAT_HIP_DAMP_GNA  1   /* 'gna'           -> lsa/gauss_newton_armijo.m */
#define DBAT_HIP_DAMP_LM   2   /* 'lm'            -> lsa/levenberg_marquardt.m */
#define DBAT_HIP_DAMP_LMP  3   /* 'lmp'           -> lsa/levenberg_marquardt_powell.m */

typedef struct dbat_hip_handle dbat_hip_handle;

/*
 * The DBAT struct fields read by the hot path (misc/prob2dbatstruct.m:12-186;
 * list in SURVEY.md 8(b)).  nIOrows = 5+nK+nP.
 */
typedef struct dbat_hip_problem {
    int32_t abi_version;        /* DBAT_HIP_ABI_VERSION */
    int32_t n_images;           /* size(s.EO.val,2) */
    int32_t n_points;           /* size(s.OP.val,2) */
    int64_t n_obs;              /* size(s.IP.val,2) */
    int32_t dist_model;         /* unique(s.IO.model.distModel): 2,3,4 or 5 (brown_euler_cam4.m:122-130) */
    int32_t nK, nP;             /* s.IO.model.nK, .nP */

    /* image observations, image-major, ascending OP inside an image
     * (prob2dbatstruct.m:343-365; multi_res.m:126-135) */
    const int32_t *ip_cam;      /* s.IP.cam - 1          [n_obs] */
    const int32_t *ip_pt;       /* OP column of each IP column (find(s.IP.vis)) [n_obs] */
    const double  *ip_val;      /* s.IP.val  2 x n_obs, pixels */
    const double  *ip_std;      /* s.IP.std  2 x n_obs, pixels (buildweightmatrix.m:16-20) */

    const double  *IO_val;      /* s.IO.val  nIOrows x n_images */
    const double  *px_size;     /* s.IO.sensor.pxSize 2 x n_images */
    const double  *EO_val;      /* s.EO.val(1:6,:) 6 x n_images */
    const double  *OP_val;      /* s.OP.val  3 x n_points */

    const uint8_t *est_IO;      /* s.bundle.est.IO nIOrows x n_images */
    const uint8_t *est_EO;      /* s.bundle.est.EO 6 x n_images */
    const uint8_t *est_OP;      /* s.bundle.est.OP 3 x n_points */
    const int32_t *IO_block;    /* s.IO.struct.block nIOrows x n_images */
    const int32_t *EO_block;    /* s.EO.struct.block 6 x n_images (shared elements -- camera stations -- take the column-list kernels) */

    /* prior observations (lsa/prior_obs.m:26-72; buildweightmatrix.m:25-29) */
    const uint8_t *prior_IO_use; const double *prior_IO_val; const double *prior_IO_std;
    const uint8_t *prior_EO_use; const double *prior_EO_val; const double *prior_EO_std;
    const uint8_t *prior_OP_use; const double *prior_OP_val; const double *prior_OP_std;

    int32_t device;             /* HIP device ordinal this handle runs on */
    int32_t shard_rank;         /* object-point shard owned by this handle ... */
    int32_t shard_count;        /* ... of shard_count (1 = whole problem) */
} dbat_hip_problem;

/* options = the varargin of bundle() (bundle.m:78-132) plus the constants
 * bundle.m hard-codes for each damping scheme (:281-283, :301-304, :321-325). */
/* The caller's own tests, as bundle() hands them to the lsa solvers (bundle.m:168-192):
 *   termFun(Jp, r) -> logical  (gauss_newton_armijo.m:187-191, levenberg_marquardt.m:217, levenberg_marquardt_powell.m:134-140):
 *       Jp = J*p and r, both weighted, n_residuals rows in the reference's row order; != 0 ends the iteration (code 0);
 *   vetoFun(x) -> logical  (gauss_newton_armijo.m:268-271, levenberg_marquardt.m:170-173, levenberg_marquardt_powell.m:146-166):
 *       x = the trial point (n_params, reference order); != 0 rejects it.
 * NULL (the default): the built-in tests of bundle.m:186-192 on ||Jp|| and ||r|| -- no vectors leave the device -- and no veto.
 * With a termFun every termination test costs one pass for J*p, one for r and their copies to the host. */
typedef int32_t (*dbat_hip_term_fn)(void *user, const double *Jp, const double *r, int64_t n_residuals);
typedef int32_t (*dbat_hip_veto_fn)(void *user, const double *x, int64_t n_params);
/* 'trace' (bundle.m:82): called INSIDE the damping loop, at the statement where the lsa solver prints its line
 * (gauss_newton_armijo.m:119-128, gauss_markov.m:74-76, levenberg_marquardt.m:138-147, levenberg_marquardt_powell.m:160-164):
 * iteration n, residual norm rr(end), the damping quantity the line shows (GNA: the last alpha, LM: lambda, LMP: delta;
 * NaN where the reference prints none: GNA's and LM's iteration 0, Gauss-Markov), and for LMP the step type
 * (0 GN, 1 IP, 2 CP; -1 otherwise) and the gain ratio rho (NaN otherwise).  NULL: nothing is reported before the loop returns. */
typedef void (*dbat_hip_trace_fn)(void *user, int32_t damping, int32_t iter, double res_norm, double damp, int32_t step_type, double rho);

typedef struct dbat_hip_options {
    int32_t damping;        /* DBAT_HIP_DAMP_* ; default GNA (bundle.m:79) */
    int32_t max_iter;       /* 20   (bundle.m:78) */
    double  conv_tol;       /* 1e-6 (bundle.m:86) */
    int32_t abs_term;       /* 'absterm' (bundle.m:185-192) */
    int32_t singular_test;  /* 'singulartest' (bundle.m:81) */
    int32_t store_trace;    /* keep every iterate (E.trace); costs n*(iters+1) doubles of host memory */
    double  mu;             /* GNA Armijo constant 0.1 (bundle.m:281) */
    double  alpha_min;      /* GNA shortest step 1e-9 (bundle.m:283) */
    double  lambda0;        /* LM: -1e-10 => 1e-10*trace(J'J)/n (bundle.m:301; levenberg_marquardt.m:88-95) */
    double  lambda_min;     /* LM: = lambda0 (bundle.m:304) */
    double  rho_bad;        /* LMP 0.25 (bundle.m:321) */
    double  rho_good;       /* LMP 0.75 (bundle.m:322) */
    double  delta0;         /* LMP: <=0 => norm(x0) (bundle.m:325) */
    dbat_hip_term_fn term_fun;  /* NULL: bundle.m:186-192 */
    void   *term_user;
    dbat_hip_veto_fn veto_fun;  /* NULL: no veto (bundle.m:168-172; the reference's own 'chirality' is undefined) */
    void   *veto_user;
    dbat_hip_trace_fn trace_fun; /* NULL: no live trace lines */
    void   *trace_user;
} dbat_hip_options;

/* what the lsa solvers return: [x,code,n,final,T,rr,extra...]
 * (gauss_newton_armijo.m:1-2, levenberg_marquardt.m:1-2,
 *  levenberg_marquardt_powell.m:1-2, gauss_markov.m:1) */
typedef struct dbat_hip_result {
    int32_t code;           /* 0 ok, -1 too many iterations, -2 singular normal matrix,
                               -3 no alpha found, -4 structurally rank deficient */
    int32_t iters;          /* n */
    int32_t n_res;          /* #entries written to res[]  (rr) */
    int32_t n_damp;         /* #entries written to damp[] (alphas / lambdas / deltas) */
    int32_t n_trace;        /* #columns written to trace (T) */
    double  sigma0;         /* sqrt(r'*r/(m-n)) at the last linearisation point (bundle.m:476-483) */
    double  time_s;         /* wall seconds inside the damping loop (E.time, bundle.m:287-294) */
    int32_t n_residual_evals;   /* residual-only evaluations (line search / trial points) */
    int32_t n_linearizations;   /* residual+Jacobian+normal-equation builds */
    int32_t n_solves;           /* reduced-system factorisations */
    int32_t n_trace_only;       /* linearisations taken for trace(J'J) alone (LM's first: no reduced system formed) */
    /* Where the time went (the E.time of bundle.m:287-294, by stage): seconds of the handle's stream between
     * hipEvents recorded where a stage is enqueued -- each interval runs to the next stage's first kernel, so
     * it includes whatever wait for the host follows the stage.  [0] linearisation (residual + Jacobian blocks
     * + J'J + Schur complement), [1] factorisation + solve of the reduced system, [2] back-substitution +
     * step norms, [3] residual-only evaluations (line search / trial points), [4] everything else inside the
     * loop (iterate gathers for the trace, set-up copies). */
    double  stage_s[5];
} dbat_hip_result;

const char *dbat_hip_last_error(void);
int  dbat_hip_abi_version(void);

/* default options for a damping scheme (bundle.m:78-86,281-283,301-304,321-325) */
int  dbat_hip_default_options(int32_t damping, dbat_hip_options *opt);

/* ---- set-up: replaces buildserialindices / serialize / buildweightmatrix --- */

/* Host-only part of create(): index maps and sizes, no GPU needed.
 * misc/buildserialindices.m:69-159 (x order [IO;EO;OP], leading elements of
 * parameter blocks, residual row ranges). */
int  dbat_hip_plan(const dbat_hip_problem *prob, int64_t *n_params, int64_t *n_residuals,
                   int64_t *n_io, int64_t *n_eo, int64_t *n_op,
                   int64_t *shard_pt_lo, int64_t *shard_pt_hi);

/* Host-only: the structural rank test of the damping loops at iteration 0
 * (sprank(J) < size(J,2) => code -4; gauss_newton_armijo.m:132-142,
 * levenberg_marquardt.m:126-135, levenberg_marquardt_powell.m:113-122), decided
 * by a maximum matching of the unknowns to the rows of J.  No GPU needed. */
int  dbat_hip_plan_structural_rank_ok(const dbat_hip_problem *prob, int32_t *ok);

/* Host-only: owner[p] = rank (0..prob->shard_count-1) whose shard holds object
 * point p (contiguous ranges of the spatially sorted processing order,
 * balanced by observation count).  prob->shard_rank is ignored. */
int  dbat_hip_plan_point_owner(const dbat_hip_problem *prob, int32_t *owner /*[n_points]*/);

/* Host-only x0 = serialize(s) straight from the problem description
 * (misc/serialize.m:14-18 over the indices of buildserialindices.m); x0 has
 * n_params entries (see dbat_hip_plan).  No GPU needed. */
int  dbat_hip_plan_serialize(const dbat_hip_problem *prob, double *x0);

/* Build the device problem: uploads observations, builds the point-major
 * batches, weights (buildweightmatrix.m:13-43) and index maps.
 * Replaces bundle.m:156-175. */
int  dbat_hip_create(const dbat_hip_problem *prob, dbat_hip_handle **out);
void dbat_hip_destroy(dbat_hip_handle *h);

/* Plan reuse.  The reference re-enters bundle() from any s at no set-up cost: the indices of
 * buildserialindices.m are kept in s.bundle.serial / deserial and only rebuilt when missing (bundle.m:156-159),
 * deserialize.m:31-46 puts any x back into the same struct.  Here the set-up is the plan and its uploads
 * (dbat_hip_create); a handle is re-used for another problem of the SAME STRUCTURE:
 *   dbat_hip_structure_key   host only: 128 bits over everything a plan depends on -- sizes, lens model, ip_cam, ip_pt,
 *                            ip_val, ip_std, px_size, the est_* masks, IO_block / EO_block, the prior_*_use masks, shard_rank /
 *                            shard_count, device and the DBAT_HIP_* switches of the environment.  NOT part of it: IO_val,
 *                            EO_val, OP_val, prior_*_val, prior_*_std.
 *   dbat_hip_handle_key      the key of the problem a handle was created from.
 *   dbat_hip_set_values      new IO_val / EO_val / OP_val and prior values / standard deviations into the handle (its
 *                            serialize() then returns the new x0; a fixed interior orientation with new values has its
 *                            corrected image coordinates recomputed); every state of an earlier solve is dropped.
 *                            DBAT_HIP_EINVAL, and the handle untouched, if the key of prob differs from the handle's:
 *                            a changed mask, block, visibility or observation needs a new handle.
 * One-rank handles and the handles of a sharded run alike (the communicator stays attached). */
int  dbat_hip_structure_key(const dbat_hip_problem *prob, uint64_t *key /*[2]*/);
int  dbat_hip_handle_key(const dbat_hip_handle *h, uint64_t *key /*[2]*/);
int  dbat_hip_set_values(dbat_hip_handle *h, const dbat_hip_problem *prob);

int64_t dbat_hip_num_params(const dbat_hip_handle *h);      /* s.bundle.serial.n */
int64_t dbat_hip_num_residuals(const dbat_hip_handle *h);   /* s.post.res.ix.n */

/* x0 = serialize(s)  (misc/serialize.m:14-18) */
int  dbat_hip_serialize(const dbat_hip_handle *h, double *x /*[n]*/);
/* s = deserialize(s,x)  (misc/deserialize.m:28-30): full IO/EO/OP arrays */
int  dbat_hip_deserialize(const dbat_hip_handle *h, const double *x,
                          double *IO /*nIOrows x nc*/, double *EO /*6 x nc*/, double *OP /*3 x np*/);
/* structural rank test done by the solvers at n==0 (gauss_newton_armijo.m:132-142):
 * 1 = full structural rank, 0 = deficient (code -4). Host only. */
int  dbat_hip_structural_rank_ok(const dbat_hip_handle *h, int32_t *ok);

/* ---- resFun: r = resFun(x) and [r,J] = resFun(x) ------------------------ */

/* r = brown_euler_cam4(x,s) (brown_euler_cam4.m:122-148; multi_res.m:20-55;
 * prior_obs.m:26-43).  r_unweighted [n_residuals] in the reference row order
 * [image rows; IO priors; EO priors; OP priors] (may be NULL);
 * f = 0.5*r'*W*r (gauss_newton_armijo.m:253).  Collective on a sharded handle
 * (every rank calls it and receives the whole vector and the whole f). */
int  dbat_hip_residual(dbat_hip_handle *h, const double *x, double *r_unweighted, double *f);

/* [r,J] = resFun(x), J returned as the per-observation blocks the sparse J is
 * made of (multi_res.m:138-294): for IP column k (reference order)
 *   JEO[12k..] = d r_k / d EO(1:6,cam)   2x6 column-major
 *   JOP[6k..]  = d r_k / d OP(:,pt)      2x3
 *   JIO[2*nIOrows*k..] = d r_k / d IO(:,cam)  2 x nIOrows
 * unweighted, fixed parameters included (the caller applies est masks and
 * block sharing).  Any output may be NULL.  For bundle_cov / parity tests. */
int  dbat_hip_jacobian_blocks(dbat_hip_handle *h, const double *x,
                              double *JEO, double *JOP, double *JIO);

/* ---- normal equations + solve: (J'*J [+lambda*I]) \ (-J'*r) ------------- */

/* One linearisation at x and one solve:
 *   scale_columns=1 : p = D*((D*J'*J*D) \ -(D*J'*r)), D = diag(1/||J(:,j)||)
 *                     (gauss_newton_armijo.m:166-174; levenberg_marquardt_powell.m:267-279)
 *   scale_columns=0 : p = (J'*J + lambda*I) \ (-J'*r)   (levenberg_marquardt.m:119; gauss_markov.m:79)
 * computed by Schur-complement elimination of the object-point blocks.
 * p [n] in x order.  stats[8] = { f=0.5 r'r, ||J p||^2, r'Jp, ||p||^2,
 * trace(J'J), singular flag, the estimate (min pivot / max pivot)^2 the flag is taken from (CHOLMOD's rcond;
 * 0 after a failed factorisation), the factorisation's info (> 0: first non-positive pivot, index in
 * the factorised order) }.  Jp/stats may be NULL. */
int  dbat_hip_linearize_solve(dbat_hip_handle *h, const double *x, double lambda,
                              int32_t scale_columns, double *p, double *stats);

/* g = J'*r at the last linearisation point, x order (levenberg_marquardt.m:82) */
int  dbat_hip_gradient(dbat_hip_handle *h, double *g);
/* Jn = sqrt(sum(J.^2,1)) at the last linearisation point (gauss_newton_armijo.m:166) */
int  dbat_hip_colnorms(dbat_hip_handle *h, double *Jn);
/* ||J*v||^2 at the last linearisation point (Jp, dog-leg g'J'Jg: lmp.m:304-311) */
int  dbat_hip_jtimes_sqnorm(dbat_hip_handle *h, const double *v, double *sqnorm);
/* J*v itself at the last linearisation point: n_residuals weighted rows in the reference's row order (image rows, then the
 * prior rows) -- the vector termFun(Jp, r) of bundle.m:186-192 / gauss_newton_armijo.m:187 receives; a caller with a
 * termination test of its own runs the loop through dbat_hip_linearize_solve and evaluates it on this (the built-in loops
 * only need ||Jp||: dbat_hip_jtimes_sqnorm).  One-rank handles only. */
int  dbat_hip_jtimes(dbat_hip_handle *h, const double *v, double *Jv);

/* The same for a SAMPLE of the image observations -- n IP columns ip_col[i] (0-based, reference order) -- so that
 * the model of a problem with 10^7 ... 10^8 observations can be checked without exporting 12 doubles for each:
 * res[2n] the unweighted residual [mm] (multi_res.m:20-55), JEO[12n], JOP[6n], JIO[2*nIOrows*n] laid out as in
 * dbat_hip_jacobian_blocks but indexed by i.  One-rank handles only. */
int  dbat_hip_jacobian_sample(dbat_hip_handle *h, const double *x, int64_t n, const int64_t *ip_col, double *res,
                              double *JEO, double *JOP, double *JIO);

/* J = [image rows; IO prior rows; EO prior rows; OP prior rows] at x as a compressed-sparse-column matrix of
 * n_residuals x n_params -- what [r,J]=resFun(x) returns (brown_euler_cam4.m:163-182, multi_res.m:300-313) and
 * bundle() hands on as E.final.weighted.J / E.final.unweighted.J (bundle.m:341-350; bundle_cov.m:18-23,68 reads
 * it for 'CXX' / 'COPF').  On request only: the solver itself never forms J.  The 2x6 / 2x3 / 2xnIO blocks come
 * from the device (the kernel of dbat_hip_jacobian_blocks), the assembly is a counting sort on the host.
 * weighted != 0: rows scaled by 1/sigma (chol(W) J, gauss_newton_armijo.m:104,116).
 * Two calls: with colptr == NULL only *nnz is set; then colptr [n_params+1] (int64), rowidx [nnz] (int64, ascending
 * inside a column), val [nnz].  One-rank handles only (DBAT_HIP_EUNSUPPORTED on a sharded one). */
int  dbat_hip_jacobian_csc(dbat_hip_handle *h, const double *x, int32_t weighted, int64_t *nnz,
                           int64_t *colptr, int64_t *rowidx, double *val);

/* ---- the damping loops (host control flow, device arithmetic) ----------- */

/* [x,code,n,final,T,rr,extra] = <solver>(resFun,vetoFun,x0,W,maxIter,termFun,...)
 * dispatched on opt->damping as bundle.m:267-338 does.
 *   x      [n]  in: x0, out: final estimate
 *   res    [max_iter+3]   rr
 *   damp   [2*max_iter+4] alphas (GNA) / lambdas (LM) / deltas (LMP)
 *   aux    [2*max_iter+4] LMP: rhos then (offset max_iter+2) step types; may be NULL
 *   trace  [n*(max_iter+2)] iterates as columns if opt->store_trace, else may be NULL
 */
int  dbat_hip_solve(dbat_hip_handle *h, const dbat_hip_options *opt, double *x,
                    dbat_hip_result *result, double *res, double *damp, double *aux,
                    double *trace);

/* Post-processing of bundle.m:449-460: unweighted residuals at the last
 * linearisation point of dbat_hip_solve, reference row order. */
int  dbat_hip_final_residuals(dbat_hip_handle *h, double *r_unweighted, double *r_weighted);

/* ---- multi-GPU: one handle per rank, object points sharded -------------- */

/* The reference is one MATLAB thread; the sharded path has no
 * counterpart there.  One handle per rank (prob->shard_rank of
 * prob->shard_count, one GPU each); object points and their observations are
 * sharded.  DOMAIN SHARDING (the default): the first levels of the nested
 * dissection of the camera network give every rank a domain of images; an object point
 * goes to the rank whose domain holds its interior images, so a rank builds AND factors
 * its domain of the reduced camera system alone.  What crosses xGMI, all as RCCL
 * all-reduces on the handle's stream: per linearisation the vectors [J_c'r | diag | sums]
 * (2 NS + 8 doubles); per factorisation the ranks' shares of the TOP-SEPARATOR tiles of
 * the factor, one contiguous buffer (16 MB at 1000 images / 8 ranks, where a
 * reduce-scatter of the whole reduced system would carry 288 MB), after which every rank
 * factors the top separators and substitutes back; per solve 8 + 4*shard_count doubles
 * of scalar sums and pivot extremes; per objective value one double.
 * DBAT_HIP_MG_REPLICATED (or a problem without a usable dissection: shared EO blocks):
 * the envelope of the reduced system [S | J_c'r | diag] is summed per linearisation
 * instead and every rank factors all of it.  With a
 * communicator every entry point below "the damping loops", dbat_hip_residual,
 * dbat_hip_final_residuals, dbat_hip_gradient/_colnorms and
 * dbat_hip_posterior_cov is COLLECTIVE: all ranks call it with the same
 * arguments and all receive the complete result (x, residual rows, blocks). */
#define DBAT_HIP_UNIQUE_ID_BYTES 128
/* rank 0: a fresh RCCL unique id (ncclGetUniqueId); the caller hands the 128
 * bytes to the other ranks by whatever channel it has (MPI, a store, a file) */
int  dbat_hip_comm_unique_id(uint8_t *id /*[128]*/);
/* every rank: join the communicator (ncclCommInitRank with the handle's
 * shard_rank / shard_count on the handle's device).  Collective. */
int  dbat_hip_comm_init(dbat_hip_handle *h, const uint8_t *id /*[128]*/);
/* helper for the host side of a multi-rank driver: all-reduce `count` host
 * doubles over the handle's communicator; op 0 = sum, 1 = max, 2 = min
 * (a barrier is a reduce of one double).  Identity without a communicator. */
int  dbat_hip_comm_allreduce_host(dbat_hip_handle *h, double *buf, int64_t count, int32_t op);

/* Deterministic mode: on != 0 makes every later linearisation of this handle give the same bits in every run -- as the
 * reference does by construction, where one MATLAB thread forms J'J (gauss_newton_armijo.m:166-174).  The default
 * mode's atomics reorder f64 sums (steps repeat to 1e-13).  The sums are not ordered but made EXACT: the camera side is
 * summed chunk by chunk in a fixed order; from its column norms every element of the reduced system gets a power-of-two
 * grid on which every Schur contribution is rounded (half a unit in the last place of the element's Cauchy-Schwarz bound
 * sqrt(U_ii U_jj)), and sums of grid multiples below 2^53 grid units carry no rounding, whatever their order.
 * Covers every build path on one rank (signature-group kernel: + 7 % at C3; scenes with irregular visibility run the
 * column-list kernel over all their batches).  The step differs from the default mode's by cond(S) x 1e-15 (C1: 2e-10,
 * C3: 5e-8); converged estimates and sigma0 agree to 1e-9 -- the right-hand side is rounded on the scale of its own
 * summation error.  DBAT_HIP_EUNSUPPORTED for several ranks, shared EO blocks (camera stations), more than nine estimated
 * IO columns per camera. */
int  dbat_hip_set_deterministic(dbat_hip_handle *h, int32_t on);

/* Test hook (gloo on CPU boxes, two shards on one GPU through the host):
 * sum-all-reduce of `count` doubles at device address `buf_dev`, enqueued on
 * `stream` (a hipStream_t), used in place of RCCL when no communicator is set. */
typedef int (*dbat_hip_allreduce_fn)(void *user, void *buf_dev, int64_t count, void *stream);
int  dbat_hip_set_allreduce(dbat_hip_handle *h, dbat_hip_allreduce_fn fn, void *user);

/* Host only: the domain of every image under domain sharding with prob->shard_count ranks (nested dissection of
 * the co-visibility graph, csrc/nd.hpp): cam_owner[n_images] = the rank whose domain the image belongs to, or -1
 * for an image of a top separator (replicated on every rank).  An object point is owned by the rank whose domain
 * holds its interior images (dbat_hip_plan_point_owner); the invariant the scheme rests on -- no point sees
 * interior images of two domains -- is checked by tests/test_parallel_cpu.py.  *subtree = 1 if this problem
 * is sharded that way (0: contiguous point ranges, the whole reduced system summed and factored by every rank:
 * one rank, shared EO blocks, or the environment asked for it). */
int  dbat_hip_plan_domain_map(const dbat_hip_problem *prob, int32_t *cam_owner, int32_t *subtree);

/* mask[n_params]: 1 where this handle's shard owns the x entry (its object
 * points; rank 0 also owns IO and EO).  Host only; results returned by the
 * library are already complete on every rank. */
int  dbat_hip_owned_mask(const dbat_hip_handle *h, uint8_t *mask);

/* ---- initial values ------------------------------------------------------ */

/* s = forwintersect(s0, ids, skipPrior) (photogrammetry/forwintersect.m:27-46;
 * pm_multilenscorr1.m:45-69, pm_multiforwintersect.m:41, pm_forwintersect3.m:55-82): object
 * points by forward intersection of their lens-corrected image rays with the IO / EO of x
 * (the OP part of x is not used).  OP [3*n_points] in: current values, out: intersected
 * points; skip[p] != 0 (may be NULL) leaves point p as it is; a point with fewer than two
 * rays becomes NaN.  Collective on a sharded handle. */
int  dbat_hip_forwintersect(dbat_hip_handle *h, const double *x, const uint8_t *skip, double *OP);

/* [s,rms,fail] = resect(s0, cams, cpId, n, v, chkId) (photogrammetry/resect.m:42-131) with the per-camera work on
 * the device: for every camera the candidate triangles of control points are solved by the three-point resection of
 * pm_resect_3pt.m:27-147 (Grunert's quartic, up to four poses each, camera behind the image plane as resect.m:105
 * asks for) and scored by the rms reprojection error over the camera's check points; the best pose wins, a later
 * triangle only if strictly better.  No handle: the inputs are what resect.m has after its MATLAB-side preparation.
 *   pt_start  [n_images+1]  range of every camera's check points in X / xn
 *   X         [3*total]     object coordinates of the check points, per camera
 *   xn        [2*total]     their lens-corrected, normalised image coordinates K \ [x; y; 1] (resect.m:95-99)
 *   tri_start [n_images+1]  range of every camera's candidate triangles in tri
 *   tri       [3*total]     three indices into the camera's point range per triangle, in trial order
 *   P         [12*n_images] out: best 3 x 4 camera matrix (column-major), NaN if the camera has no pose
 *   rms       [n_images]    out: its rms error, Inf if none
 * Returns DBAT_HIP_EDEVICE without a HIP device (no CPU path). */
int  dbat_hip_resect(int32_t device, int32_t n_images, const int64_t *pt_start, const double *X, const double *xn,
                     const int64_t *tri_start, const int32_t *tri, double *P, double *rms);

/* ---- measurement hooks -------------------------------------------------- */

/* One benchmark step = one Levenberg-Marquardt iteration's device work at
 * the current point: J'J build + Schur solve (+ back-substitution) and one
 * residual-only evaluation at the trial point.  x is not advanced, so every
 * step does identical work.  From HIP events on the handle's stream, ms[16]:
 * phases { linearize+Schur build, factor+solve, back-substitution, trial
 * residual } then single kernels { the Schur kernel alone (k_build_sig, or the
 * tile kernel k_build_tile3 / k_build_tile2 where the signature groups are too
 * short, or k_build when nothing is tiled; dbat_hip_build_kernel_name says which),
 * k_chol_df incl. its flag reset, the back-substitution kernels, k_residual_cm },
 * then (several ranks, domain sharding; else 0) { factorisation of the rank's own domain
 * + its shares of the top tiles, the all-reduce of the top tiles as the stream sees it,
 * top separators + backward substitution, 0 }, then (heavy / giant points on the matrix cores, csrc/heavy.hpp;
 * else 0) { camera side of their observations + k_heavy_z / k_heavy_z_giant, k_heavy_syrk, 0, 0 }.  ms[16]. */
int  dbat_hip_bench_step(dbat_hip_handle *h, double lambda, int32_t scale_columns, double *ms);
/* load x into the handle (device resident) before bench steps */
int  dbat_hip_set_x(dbat_hip_handle *h, const double *x);
/* sizes of the internal layout, for roofline accounting:
 * info[0]=NS (reduced system order) info[1]=#batches info[2]=max obs per point
 * info[3]=obs in this shard info[4]=points in this shard info[5]=batch size
 * info[6]=max camera-side columns per observation info[7]=#tiles
 * several ranks: info[8]=1 domain sharding (0: replicated factorisation) info[9]=doubles of the reduced
 * system summed per factorisation (top tiles / the envelope) info[10]=doubles summed as vectors per
 * linearisation info[11]=images in the top separators info[12]=tile rows of the factor
 * info[13], info[14]=tasks of the two launches of the factorisation
 * info[15]=v_mfma_f64_16x16x4_f64 instructions (2048 flops each) one launch of the tile kernel executes (its symmetric
 * products; the algorithmic count of the roofline is the full product)
 * heavy / giant points on the matrix cores (csrc/heavy.hpp; all 0 when the column-list kernels take them):
 * info[16]=tasks of k_heavy_syrk info[17]=its v_mfma_f64_16x16x4_f64 instructions per launch info[18]=row groups
 * info[19]=bytes of the scratch array Zs info[20]=points info[21]=observations info[22]=k-steps per task at most
 * info[23]=algorithmic flops of their Schur terms, sum of 108 k + 216 k^2 */
int  dbat_hip_info(const dbat_hip_handle *h, int64_t *info /*[24]*/);

/* Host only (no GPU): statistics of the layout the plan gives this problem (this shard), so that a test
 * can tell which code path of the signature kernel a scene exercises.  st[16]:
 * [0] tiles [1] batches [2] tiled batches [3] signature groups [4] their points [5] chunks
 * [6..9] chunks of 1-8 / 9-16 / 17-32 / 33-64 points (8, 4, 2, 1 lanes per point in pass 1)
 * [10] chunks that need more than one round of pass 2 [11] most cameras per chunk
 * [12] most rows of a chunk [13] 1 = k_build_sig selected [14] 1 = k_backsub_sig selected
 * [15] tasks of k_heavy_syrk (0: heavy / giant points, if any, take the column-list kernels) */
int  dbat_hip_plan_layout_stats(const dbat_hip_problem *prob, int64_t *st /*[16]*/);

/* name of the kernel that builds the Schur complement of the tiled points in this handle
 * (the dominant kernel of a step; the one the bench's roofline entry is about) */
int  dbat_hip_build_kernel_name(const dbat_hip_handle *h, char *buf, int32_t buf_len);

/* Schedule of the Cholesky of the reduced system (measurement only): st[0] order of the
 * factorised system incl. block padding, st[1] tile tasks, st[2] 64x64x64 tile products of
 * the update phase, st[3] tile rows, st[4] 1 = nested-dissection order, st[5] 1 = dataflow kernel. */
int  dbat_hip_chol_stats(const dbat_hip_handle *h, int64_t *st /*[6]*/);

/* Posterior covariance blocks at x: sigma0^2 * blocks of inv(J'J), J the
 * weighted Jacobian incl. prior rows.  Replaces bundle/bundle_cov.m:63-214
 * ('CIO','CEO','COP'; the native code it stands in for is
 * test/postcov/icpc_mex.c) -- computed from the Schur blocks instead of a
 * Cholesky factor of the full normal matrix:
 *   inv(J'J)[cams,cams] = inv(S),  inv(J'J)[p,p] = V_p^-1 + (W_p V_p^-1)' inv(S) (W_p V_p^-1).
 * CEO  [36*n_images]  6x6 block per image, column-major, rows/columns of fixed elements zero
 * CIO  [nIOu*nIOu]    the IO unknowns in the order of the IO part of x (buildserialindices.m:11-17)
 * COP  [9*n_points]   3x3 block per object point, rows/columns of fixed coordinates zero
 * Sinv [NS*NS]        optional: inv(S) itself (lower triangle valid; NOT scaled by sigma0^2),
 *                     order [EO of image 0..n-1 | IO unknowns], for 'CEOF'/'CIOF'
 * Any output may be NULL.  On a sharded handle (collective) inv(S) is computed on every rank,
 * the per-point blocks by the owning rank, and all ranks receive all blocks. */
int  dbat_hip_posterior_cov(dbat_hip_handle *h, const double *x, double sigma0, double *CEO, double *CIO,
                            double *COP, double *Sinv);

#ifdef __cplusplus
}
#endif
#endif /* DBAT_HIP_H */
