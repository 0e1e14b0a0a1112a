// HIP kernels of the bundle hot path (gfx950, FP64, 64-lane wavefronts).
//
// Layout: observations are stored point-major ("processing order") in
// batches of whole object points, at most BT observations per batch; one
// workgroup (BT threads) handles one batch, one lane handles one observation.
// The Jacobian is never materialised: every pass recomputes the 2x6 / 2x3 /
// 2xnIO blocks of its observation from (u,v), the camera record and the
// object point (F5-F9 of SURVEY.md 8(a)).
//
//   k_cam_prep / k_envelope_cams / k_axpby_cams   per camera: rotation, its derivatives, IO fan-out
//                (K0), alone or riding in the first launch of a linearisation / of a trial point
//   k_residual_cm, k_residual   residual only, 0.5*r'r (camera-major; point-major for exports)   (K2)
//   k_cam_normal6, k_cam_normal camera side of the tiled observations: J_c'J_c, J_c'r
//   k_build, k_build_giant      residual + Jacobian blocks + point blocks V, g_p, V^-1 and the Schur
//                complement by the column lists (heavy points, shared EO)             (K1,K3,K4,K5)
//   k_build_tile2, k_build_tile3  the same on the f64 matrix cores by dense 128-row tiles, where the
//                signature groups are too short for sig.hpp's k_build_sig
//   k_finish, k_envelope_op     priors, damping, fixed rows, column scaling on the envelope (F11)
//   k_backsub, k_backsub_giant  dp = -V^-1 (g_p + W' dc), J*p sums                     (K7,K8,K9)
//   k_build_tail, k_prior_sq, k_prior_jv   the tails of a linearisation / an objective value / a
//                solve: one launch each, grid sum finished by the last block (grid_sum)
//   k_jtimes     ||J v||^2, r'Jv                                                       (K8)
//   k_cov_*, k_forwintersect*   posterior covariance blocks, forward intersection
#pragma once
#include <hip/hip_runtime.h>

#include "model.hpp"

namespace dbat {

struct DevProblem {
    // sizes
    int nc, np, nIOrows, nK, nP, nIOu, ncolmax, BT;
    int64_t NS, NZ, nobs, nb;
    int64_t ldS;                    // leading dimension of S (>= NS+1: row NS holds the right-hand side)
    // camera static data
    const int32_t *cam_ncol, *cam_col, *cam_iorow;
    const uint32_t *cam_eo_est;
    const int32_t *io_src;          // [nIOrows*nc]
    const double *io_fixed;         // [nIOrows*nc]
    const double *px;               // [2*nc]
    const double *cam_w;            // [2*nc]
    // z metadata
    const uint8_t *z_est, *z_mine;
    const double *z_prw, *z_prv;
    // observations (processing order)
    const int32_t *o_cam, *o_pt;
    const double *o_uv;             // image coordinates -- or, uv_pre, the corrected coordinates rhs (k_uv_to_rhs)
    const double *o_uv_raw;         // always the measured (u, v): Jacobian export, forward intersection
    int uv_pre;                     // fixed interior orientation: o_uv, cm_uv, sg_uv hold rhs
    const double *o_w;              // may be null (uniform)
    const uint32_t *o_seg;
    const int64_t *o_row;
    const int64_t *batch_start;
    // tiles (fixed-IO path): runs of batches touching at most CMAX cameras
    int CMAX, ntiles;
    int ablate;                     // measurement builds only (DBAT_HIP_ABLATE, read through DBAT_ABLATE): phases off, phase clocks
    int trace_only;                 // this linearisation serves trace(J'J) alone (levenberg_marquardt.m:88-95): no Schur complement
    // deterministic mode (dbat_hip_set_deterministic): two runs give the same bits.  No ordering of the atomics -- the sums
    // are made EXACT instead, and exact sums do not depend on their order:
    //   * the camera side (J_c'J_c, J_c'r, the squared column norms) leaves one partial per camera-major chunk and is
    //     summed camera by camera in chunk order (k_det_cam_reduce, k_det_io_reduce);
    //   * from the column norms every row r of the reduced system gets a power of two u_r >= sqrt(U_rr), and element
    //     (i, j) the grid q_ij = u_i u_j 2^-51.  Every contribution to the Schur complement is a piece of a Gram matrix
    //     whose diagonal is bounded by U (W V^-1 W' <= U), so |sum of the absolute contributions to (i, j)| <= u_i u_j
    //     (Cauchy-Schwarz): rounded to multiples of q_ij they add up without any rounding -- in LDS and in HBM, by the
    //     same atomics as in the default mode.  The right-hand side likewise with u_i u_f, u_f^2 >= r'r.
    //   The rounding is half a unit in the last place of the BOUND of an element, where an ordinary sum rounds to half a
    //   unit of the running sum: the result differs from the default mode's by a few 1e-16 of sqrt(S_ii S_jj).
    int deterministic;
    int any_prior;                  // 0: no prior observation anywhere (z_prw is all zero): the sums over the unknowns skip that array
    double *det_cam_part;           // [camera-major chunks][DET_CP] the chunks' Gram matrices
    double *det_io_part;            // [nc][DET_IOP] self-calibration: the cameras' IO x IO blocks and IO gradient entries
    double *det_rr;                 // [nc] r'r of the cameras' observations; [nc]: their sum
    double *det_u;                  // [NS + 1] u_r; [NS] = u_f
    const int32_t *det_cam_chunks;  // [2 nc + 2] chunk ranges of every camera: tiled part, untiled part (camera-major copy)
    const uint8_t *o_lc, *o_pidx;
    const int32_t *tile_batch, *tile_cam_start, *tile_cams;
    const int32_t *tile_order;                      // launch index -> tile (longest first)
    const int32_t *tile_io_start, *tile_iocols;     // IO columns (IOu indices) of every tile
    const uint8_t *tile_io_simple;                  // per tile: one IO block, identity row map (k_build_sig); may be null
    const uint8_t *tile_cam_io;                     // [#tile cams][16] local IO row of a camera's j-th IO column
    // "giant" points: more observations than a batch holds; one workgroup per point
    int ngiant;
    const int64_t *giant_start;                     // [ngiant+1] first observation (processing order)
    double *giant_W;                                // scratch [n giant observations][3*ncolmax]  W = E'B
};

// ablation bits exist in measurement builds only (-DDBAT_HIP_PROFILING); the product build compiles them out
#ifdef DBAT_HIP_PROFILING
#define DBAT_ABLATE(d, bits) (((d).ablate & (bits)) != 0)
#else
#define DBAT_ABLATE(d, bits) (0)
#endif

__device__ __forceinline__ void atomic_add_f64(double *p, double v) {
    unsafeAtomicAdd(p, v);          // global_atomic_add_f64 / ds_add_f64 on gfx950
}

// deterministic mode: v rounded to a multiple of q = ui uj 2^-51 (ui, uj powers of two; |v| <= ui uj): the classic
// (v + M) - M with M = 1.5 * 2^52 q = 3 ui uj.  (The empty asm keeps the compiler from folding the pair.)
constexpr int DET_CP = 256;         // doubles per chunk partial (16 x 16 Gram matrix; fixed IO uses the first 28)
constexpr int DET_IOP = 128;        // doubles per camera in det_io_part: (ncol-6)^2 <= 81 block entries, then <= 9 gradient entries
__device__ __forceinline__ double det_round(double v, double ui, double uj) {
    const double M = 3.0 * ui * uj;
    double t = v + M;
    asm volatile("" : "+v"(t));
    return t - M;
}
// smallest power of two u with u * u >= a (a >= 0); 0 for a == 0 (a row nothing contributes to)
__device__ __forceinline__ double det_pow2_sqrt(double a) {
    if (!(a > 0.0)) return 0.0;
    int e;
    const double m = frexp(a, &e);                  // a = m 2^e, m in [0.5, 1)
    (void)m;
    return ldexp(1.0, (e + 1) >> 1);                // 2^ceil(e / 2) >= sqrt(a)
}

// ---------------------------------------------------------------- K0 ----
// The record of camera c from the z-vector given by the accessor zv(index) (a plain array, or a
// trial point x + alpha p that is only being written by the same launch).
template <class ZV>
__device__ __forceinline__ void cam_prep_one(const DevProblem &d, ZV zv, int c, CamRec *__restrict__ cams) {
    // straight into the record in global memory: a local CamRec (400 bytes, indexed by loops) lives in scratch
    CamRec &r = cams[c];
    // the six EO values through the column list: a shared element lives in the slot of its leading entry
    const int32_t *ec = d.cam_col + (int64_t)c * MAXCOL;
    r.c[0] = zv(ec[0]); r.c[1] = zv(ec[1]); r.c[2] = zv(ec[2]);
    const double ang[3] = {zv(ec[3]), zv(ec[4]), zv(ec[5])};
    double Mt[9], sk, ck;
    cam_rotation(ang, Mt, sk, ck);
#pragma unroll
    for (int k = 0; k < 9; ++k) r.Mt[k] = Mt[k];
    r.sk = sk; r.ck = ck;
    auto io_at = [&](int k) -> double {
        if (k >= d.nIOrows) return 0.0;
        const int32_t s = d.io_src[(int64_t)c * d.nIOrows + k];
        return s >= 0 ? zv(6 * (int64_t)d.nc + s) : d.io_fixed[(int64_t)c * d.nIOrows + k];
    };
    r.f = io_at(0); r.pp[0] = io_at(1); r.pp[1] = io_at(2); r.b[0] = io_at(3); r.b[1] = io_at(4);
#pragma unroll
    for (int k = 0; k < MAXK; ++k) r.K[k] = k < d.nK ? io_at(5 + k) : 0.0;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) r.P[k] = k < d.nP ? io_at(5 + d.nK + k) : 0.0;
    r.sz = d.px[2 * c];
    r.w[0] = d.cam_w[2 * c]; r.w[1] = d.cam_w[2 * c + 1];
    const int ncol = d.cam_ncol[c];
    r.ncol = ncol;
#pragma unroll
    for (int k = 0; k < MAXCOL; ++k) r.col[k] = ec[k];
    // bit 8: the camera's estimated IO rows are cc, px, py, K1-K3, P1, P2 in this order (the usual
    // self-calibration): the kernels then pick the IO columns without a select chain
    constexpr int std8[8] = {0, 1, 2, 5, 6, 7, 8, 9};
    bool is8 = ncol == 14 && d.nK == 3 && d.nP == 2;
#pragma unroll
    for (int k = 0; k < MAXIO; ++k) {
        const int32_t row = d.cam_iorow[(int64_t)c * MAXIO + k];
        r.iorow[k] = row;
        if (k < 8) is8 = is8 && row == std8[k];
    }
    r.eo_est = d.cam_eo_est[c] | (is8 ? 0x100u : 0u);
}
__global__ void k_cam_prep(DevProblem d, const double *__restrict__ z, CamRec *__restrict__ cams) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < d.nc) cam_prep_one(d, [z](int64_t i) { return z[i]; }, c, cams);
}

// Fixed interior orientation: rhs = T_post * brown(T_pre(x(u, v))) of an observation (model.hpp image_side) depends on
// (u, v) and the camera's interior orientation only -- not on anything that is estimated -- so it is computed ONCE per
// handle and the kernels of the iteration read rhs where they would read (u, v) (obs_eval<.., PRE>): the residual is
// lhs - rhs with the same arithmetic as before, the Jacobian blocks A, B never needed (u, v).  Three copies of the
// image coordinates exist: point-major (cam[] = o_cam), camera-major (one camera per chunk), slot-major of the
// signature groups (sig.hpp; one camera per slot of a chunk).
template <int MODEL>
__global__ __launch_bounds__(256) void k_uv_to_rhs(int nK, int nP, const CamRec *__restrict__ cams, int64_t n,
                                                   const int32_t *__restrict__ cam, const double *__restrict__ uv,
                                                   double *__restrict__ rhs) {
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n) return;
    ImgSide im;
    image_side<MODEL, false>(cams[cam[o]], nK, nP, uv[2 * o], uv[2 * o + 1], im);
    rhs[2 * o] = im.rhs[0]; rhs[2 * o + 1] = im.rhs[1];
}
template <int MODEL>
__global__ __launch_bounds__(256) void k_uv_to_rhs_cm(int nK, int nP, const CamRec *__restrict__ cams,
                                                      const int32_t *__restrict__ chunk_cam,
                                                      const int64_t *__restrict__ chunk_start, double *__restrict__ cm_uv) {
    const CamRec &C = cams[chunk_cam[blockIdx.x]];
    for (int64_t q = chunk_start[blockIdx.x] + threadIdx.x; q < chunk_start[blockIdx.x + 1]; q += 256) {
        ImgSide im;
        image_side<MODEL, false>(C, nK, nP, cm_uv[2 * q], cm_uv[2 * q + 1], im);
        cm_uv[2 * q] = im.rhs[0]; cm_uv[2 * q + 1] = im.rhs[1];
    }
}
template <int MODEL>
__global__ __launch_bounds__(64) void k_uv_to_rhs_sig(int nK, int nP, const CamRec *__restrict__ cams,
                                                      const int32_t *__restrict__ sg_chunk,
                                                      const int32_t *__restrict__ sg_gcam, double *__restrict__ sg_uv) {
    const int32_t *cd = sg_chunk + 8 * (int64_t)blockIdx.x;
    const int npts = cd[1], k = cd[2], gm = cd[4], gi0 = cd[5], uv0 = cd[6];
    for (int j = 0; j < k; ++j) {
        const CamRec &C = cams[sg_gcam[16 * (int64_t)blockIdx.x + j]];
        for (int i = threadIdx.x; i < npts; i += 64) {
            const int64_t q = uv0 + (int64_t)j * gm + gi0 + i;
            ImgSide im;
            image_side<MODEL, false>(C, nK, nP, sg_uv[2 * q], sg_uv[2 * q + 1], im);
            sg_uv[2 * q] = im.rhs[0]; sg_uv[2 * q + 1] = im.rhs[1];
        }
    }
}

// block-wide sum of NV values per thread; result valid in thread 0
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double *sh /* >= NV*nwaves */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double x = v[i];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if (lane == 0) sh[i * nw + wave] = x;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            double s = 0;
            for (int w = 0; w < nw; ++w) s += sh[i * nw + w];
            v[i] = s;
        }
    }
    __syncthreads();
}

// Grid-wide sum without a second launch.  Every block publishes its NV sums (agent-scope stores:
// the L2 of the eight XCDs are not coherent among themselves), waits for them, and takes a ticket;
// the block that draws the last ticket adds all the blocks' sums up in a FIXED order (thread t takes
// blocks t, t + T, ...; then the block reduction), so the result does not depend on which block
// finishes last.  Returns true in that block, with the totals in v[] of thread 0.  *ctr is zero on
// entry and zero again on exit.  extra/n_extra: NV-interleaved partial sums of an EARLIER launch
// (plain loads) that the last block adds on top: element i < NX of each goes to v[XO + i].  Atomics
// of the calling kernel that the last block is to read must be waited for by their own waves.
template <int NV, int NX = 0, int XO = 0>
__device__ __forceinline__ bool grid_sum(double (&v)[NV], double *sh /* >= NV*nwaves */, double *__restrict__ partial,
                                         unsigned *__restrict__ ctr, const double *__restrict__ extra = nullptr,
                                         int64_t n_extra = 0, int extra_stride = 1) {
    __shared__ int s_last;
    block_sum<NV>(v, sh);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
            __hip_atomic_store(partial + (int64_t)NV * blockIdx.x + i, v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);               // the stores (and this block's other atomics) have been performed
        s_last = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return false;
    double acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.0;
    for (int64_t b = threadIdx.x; b < gridDim.x; b += blockDim.x)
#pragma unroll
        for (int i = 0; i < NV; ++i)
            acc[i] += __hip_atomic_load(partial + NV * b + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if constexpr (NX > 0) {
        for (int64_t b = threadIdx.x; b < n_extra; b += blockDim.x)
#pragma unroll
            for (int i = 0; i < NX; ++i) acc[XO + i] += extra[extra_stride * b + i];
    }
    __syncthreads();                                 // sh is reused
    block_sum<NV>(acc, sh);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = acc[i];
        __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return true;
}

// ---------------------------------------------------------------- K2 ----
// Residual only.  partial[blockIdx] = sum of squared weighted residuals.
// r_w (weighted, processing order) and r_unw (mm, reference row order) optional.
template <int MODEL, bool PRE>
__global__ __launch_bounds__(256) void k_residual(DevProblem d, const double *__restrict__ z,
                                                  const CamRec *__restrict__ cams,
                                                  double *__restrict__ partial,
                                                  double *__restrict__ r_w, double *__restrict__ r_unw) {
    __shared__ double sh[8];
    double acc[1] = {0.0};
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < d.nobs; o += stride) {
        const int cam = d.o_cam[o], pt = d.o_pt[o];
        const CamRec &C = cams[cam];
        const double *q = z + d.NS + 3 * (int64_t)pt;
        const double Q[3] = {q[0], q[1], q[2]};
        double r[2];
        double(*nil6)[6] = nullptr; double(*nil3)[3] = nullptr; double(*nilc)[MAXIO] = nullptr;
        obs_eval<MODEL, false, false, PRE>(C, d.nK, d.nP, Q, d.o_uv[2 * o], d.o_uv[2 * o + 1], r, nil6, nil3, nilc);
        if (r_unw) { const int64_t row = d.o_row[o]; r_unw[2 * row] = r[0]; r_unw[2 * row + 1] = r[1]; }
        const double w0 = d.o_w ? d.o_w[2 * o] : C.w[0], w1 = d.o_w ? d.o_w[2 * o + 1] : C.w[1];
        r[0] *= w0; r[1] *= w1;
        if (r_w) { r_w[2 * o] = r[0]; r_w[2 * o + 1] = r[1]; }
        acc[0] = fma2(acc[0], r[0], r[0], r[1], r[1]);
    }
    block_sum<1>(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc[0];
}

__device__ __forceinline__ void mailbox_done(double *mailbox, unsigned long long seq);      // (below)

// Residual only, camera-major: one workgroup per chunk of one camera's observations (camera
// record uniform, (u,v) and point index coalesced).  partial[blockIdx] = sum of squared
// weighted residuals.  Used for every objective value the damping loops compare.
template <int MODEL, bool PRE>
__global__ __launch_bounds__(256) void k_residual_cm(DevProblem d, const double *__restrict__ z,
                                                     const CamRec *__restrict__ cams,
                                                     const int32_t *__restrict__ cm_pt,
                                                     const double *__restrict__ cm_uv, const double *__restrict__ cm_w,
                                                     const int32_t *__restrict__ chunk_cam,
                                                     const int64_t *__restrict__ chunk_start,
                                                     double *__restrict__ partial, unsigned *__restrict__ tail_ctr = nullptr,
                                                     double *__restrict__ out = nullptr, double *__restrict__ mailbox = nullptr,
                                                     unsigned long long seq = 0) {
    __shared__ double sh[8];
    __shared__ int s_last;
    const CamRec &C = cams[chunk_cam[blockIdx.x]];
    const int64_t q0 = chunk_start[blockIdx.x], q1 = chunk_start[blockIdx.x + 1];
    double acc[1] = {0.0};
    // a chunk holds at most 2048 observations: eight per thread.  Four at a time: their point indices and image
    // coordinates are requested together, then the four gathers of the object points -- two memory latencies per
    // four observations instead of two per observation (the kernel is bound by them, not by its 20 B per observation)
    const double2 *uvp = reinterpret_cast<const double2 *>(cm_uv), *wp = reinterpret_cast<const double2 *>(cm_w);
    for (int64_t qb = q0 + threadIdx.x; qb < q1; qb += 4 * 256) {
        int pt[4]; double2 uv[4], ww[4]; double Q[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t q = qb + 256 * u;
            const bool on = q < q1;
            pt[u] = on ? cm_pt[q] : -1;
            uv[u] = on ? uvp[q] : double2{0, 0};
            ww[u] = (on && cm_w) ? wp[q] : double2{C.w[0], C.w[1]};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double *p = z + d.NS + 3 * (int64_t)(pt[u] < 0 ? 0 : pt[u]);
            Q[u][0] = p[0]; Q[u][1] = p[1]; Q[u][2] = p[2];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            double r[2];
            double(*nil6)[6] = nullptr; double(*nil3)[3] = nullptr; double(*nilc)[MAXIO] = nullptr;
            obs_eval<MODEL, false, false, PRE>(C, d.nK, d.nP, Q[u], uv[u].x, uv[u].y, r, nil6, nil3, nilc);
            r[0] *= ww[u].x; r[1] *= ww[u].y;
            if (pt[u] >= 0) acc[0] += r[0] * r[0] + r[1] * r[1];
        }
    }
    block_sum<1>(acc, sh);
    if (!tail_ctr) {
        if (threadIdx.x == 0) partial[blockIdx.x] = acc[0];  // summed by k_prior_sq (thousands of tickets on one address cost more)
        return;
    }
    // Small projects without prior observations (tail_ctr: a few hundred blocks at most): the block that finishes last adds
    // the partial sums up -- in index order, whichever block it is -- and tells the host; no k_prior_sq launch behind this
    // one (10 us of a 370 us step at C1 and of the reference's own projects).  *tail_ctr is zero on entry and on exit.
    if (threadIdx.x == 0) {
        __hip_atomic_store(partial + blockIdx.x, acc[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);
        s_last = __hip_atomic_fetch_add(tail_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;
    double tot[1] = {0.0};
    for (unsigned b = threadIdx.x; b < gridDim.x; b += blockDim.x)
        tot[0] += __hip_atomic_load(partial + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();                                     // sh is reused
    block_sum<1>(tot, sh);
    if (threadIdx.x == 0) {
        out[0] = tot[0];
        __hip_atomic_store(tail_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (mailbox) { mailbox[0] = tot[0]; mailbox_done(mailbox, seq); }
    }
}

// The last word a kernel says to the host: after its results are in the pinned mailbox it stores the
// ticket of the host's current wait in slot 63 (system-scope release).  The host spins on that slot
// instead of sleeping in hipStreamSynchronize -- kernels of a stream finish in order, so the ticket
// of the last one means everything before it is done.
__device__ __forceinline__ void mailbox_done(double *mailbox, unsigned long long seq) {
    __atomic_store_n(reinterpret_cast<unsigned long long *>(mailbox) + 63, seq, __ATOMIC_RELEASE);
}

// prior-observation rows (prior_obs.m:26-43): sum over owned z of w*(z-prior)^2, plus the n_res partial
// sums res_partial of the residual kernel launched before.  The total goes to out[0] and
// (mailbox != null) to the host's pinned mailbox.  Few, large blocks: a grid sum costs one
// ticket per block on a single address.
__global__ __launch_bounds__(1024) void k_prior_sq(DevProblem d, const double *__restrict__ z,
                                                   double *__restrict__ partial, unsigned *__restrict__ ctr,
                                                   const double *__restrict__ res_partial, int64_t n_res,
                                                   double *__restrict__ out, double *__restrict__ mailbox,
                                                   unsigned long long seq) {
    __shared__ double sh[16];
    double acc[1] = {0.0};
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (d.any_prior)
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.NZ; i += stride) {
            const double w = d.z_prw[i];
            if (w > 0 && d.z_mine[i]) { const double e = z[i] - d.z_prv[i]; acc[0] += w * e * e; }
        }
    if (grid_sum<1, 1>(acc, sh, partial, ctr, res_partial, n_res) && threadIdx.x == 0) {
        out[0] = acc[0];
        if (mailbox) { mailbox[0] = acc[0]; mailbox_done(mailbox, seq); }
    }
}

// trace(J'J) and nothing else -- levenberg_marquardt.m:76-95 linearises at x0 only to set lambda0 = c trace(J'J) / n.
// One streaming pass over the camera-major copy: the squared weighted, masked Jacobian columns of every observation
// (partial[chunk]); k_trace_tail adds the prior weights of the estimated unknowns and hands the total to the mailbox.
// (defined with the other evaluation wrappers below)
template <int MODEL, int NCX>
__device__ __forceinline__ void eval_obs_pre(const DevProblem &d, const CamRec &C, const double Q[3], double u,
                                             double v, double w0, double w1, unsigned est, double r[2],
                                             double E[2][NCX], double B[2][3]);

template <int MODEL, int NCX>
__global__ __launch_bounds__(256) void k_trace_cm(DevProblem d, const double *__restrict__ z,
                                                  const CamRec *__restrict__ cams,
                                                  const int32_t *__restrict__ cm_pt,
                                                  const double *__restrict__ cm_uv, const double *__restrict__ cm_w,
                                                  const int32_t *__restrict__ chunk_cam,
                                                  const int64_t *__restrict__ chunk_start,
                                                  double *__restrict__ partial) {
    __shared__ double sh[8];
    const CamRec &C = cams[chunk_cam[blockIdx.x]];
    const int64_t q0 = chunk_start[blockIdx.x], q1 = chunk_start[blockIdx.x + 1];
    const int ncol = NCX > 6 ? min(C.ncol, NCX) : 6;
    double acc[1] = {0.0};
    for (int64_t q = q0 + threadIdx.x; q < q1; q += 256) {
        const int64_t zp = d.NS + 3 * (int64_t)cm_pt[q];
        const double Q[3] = {z[zp], z[zp + 1], z[zp + 2]};
        const unsigned est = (d.z_est[zp] ? 1u : 0u) | (d.z_est[zp + 1] ? 2u : 0u) | (d.z_est[zp + 2] ? 4u : 0u);
        const double w0 = cm_w ? cm_w[2 * q] : C.w[0], w1 = cm_w ? cm_w[2 * q + 1] : C.w[1];
        double r[2], E[2][NCX], B[2][3];
        eval_obs_pre<MODEL, NCX>(d, C, Q, cm_uv[2 * q], cm_uv[2 * q + 1], w0, w1, est, r, E, B);
        double s = 0.0;
#pragma unroll
        for (int a = 0; a < NCX; ++a) if (a < ncol) s += E[0][a] * E[0][a] + E[1][a] * E[1][a];
#pragma unroll
        for (int c = 0; c < 3; ++c) s += B[0][c] * B[0][c] + B[1][c] * B[1][c];
        acc[0] += s;
    }
    block_sum<1>(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc[0];
}
__global__ __launch_bounds__(1024) void k_trace_tail(DevProblem d, double *__restrict__ partial, unsigned *__restrict__ ctr,
                                                     const double *__restrict__ obs_partial, int64_t n_obs_partial,
                                                     double *__restrict__ mailbox, unsigned long long seq) {
    __shared__ double sh[16];
    double acc[1] = {0.0};
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.NZ; i += stride) {
        const double w = d.z_prw[i];
        if (w > 0 && d.z_est[i] && d.z_mine[i]) acc[0] += w;
    }
    if (grid_sum<1, 1>(acc, sh, partial, ctr, obs_partial, n_obs_partial) && threadIdx.x == 0) {
        mailbox[32] = 0.0; mailbox[33] = acc[0]; mailbox[34] = 0.0;       // (f of the linearisation point: not computed here)
        mailbox_done(mailbox, seq);
    }
}

// After the build kernels: {sum of their npart residual partials + the prior rows' squares,
// owned squared column norms of the point columns} -> out[0], out[1].  One launch.
__global__ __launch_bounds__(1024) void k_build_tail(DevProblem d, const double *__restrict__ z,
                                                     const double *__restrict__ build_partial, int64_t npart,
                                                     const double *__restrict__ jn2p, double *__restrict__ partial,
                                                     unsigned *__restrict__ ctr, double *__restrict__ out,
                                                     double *__restrict__ zcopy /* null or the copy of z to keep (zlin) */) {
    __shared__ double sh[32];
    double acc[2] = {0.0, 0.0};
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.NZ; i += stride) {
        const double w = d.any_prior ? d.z_prw[i] : 0.0;
        const bool mine = d.z_mine[i] != 0;
        const double zi = z[i];
        if (zcopy) zcopy[i] = zi;
        if (w > 0 && mine) { const double e = zi - d.z_prv[i]; acc[0] += w * e * e; }
        if (i >= d.NS && mine) acc[1] += jn2p[i - d.NS];
    }
    if (grid_sum<2, 1>(acc, sh, partial, ctr, build_partial, npart) && threadIdx.x == 0) { out[0] = acc[0]; out[1] = acc[1]; }
}

// sum npart partials (NV interleaved values each) into out[NV]; single block
template <int NV>
__global__ __launch_bounds__(1024) void k_sum_partials(const double *__restrict__ partial, int64_t npart,
                                                       double *__restrict__ out, int accumulate) {
    __shared__ double sh[NV * 16];
    double acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0;
    for (int64_t k = threadIdx.x; k < npart; k += blockDim.x)
#pragma unroll
        for (int i = 0; i < NV; ++i) acc[i] += partial[k * NV + i];
    block_sum<NV>(acc, sh);
    if (threadIdx.x == 0)
#pragma unroll
        for (int i = 0; i < NV; ++i) out[i] = accumulate ? out[i] + acc[i] : acc[i];
}

// 3x3 SPD inverse via the adjugate (symmetric storage a00 a01 a02 a11 a12 a22)
__device__ __forceinline__ bool inv3_sym(const double a[6], double inv[6]) {
    const double c00 = a[3] * a[5] - a[4] * a[4];
    const double c01 = a[2] * a[4] - a[1] * a[5];
    const double c02 = a[1] * a[4] - a[2] * a[3];
    const double det = a[0] * c00 + a[1] * c01 + a[2] * c02;
    const double id = 1.0 / det;
    inv[0] = c00 * id; inv[1] = c01 * id; inv[2] = c02 * id;
    inv[3] = (a[0] * a[5] - a[2] * a[2]) * id;
    inv[4] = (a[1] * a[2] - a[0] * a[4]) * id;
    inv[5] = (a[0] * a[3] - a[1] * a[1]) * id;
    return det > 0 && c00 >= 0 && (a[0] > 0);
}

// E(:, 6+j) = w .* Cf(:, iorow[j]): the estimated IO rows of the camera as camera-side columns
template <int NCX>
__device__ __forceinline__ void io_columns(const CamRec &C, const double (*Cf)[MAXIO], double w0, double w1,
                                           double (*E)[NCX]) {
    if (NCX >= 14 && (C.eo_est & 0x100u)) {          // cc px py K1 K2 K3 P1 P2
        constexpr int std8[8] = {0, 1, 2, 5, 6, 7, 8, 9};
#pragma unroll
        for (int j = 0; j < NCX - 6; ++j) {
            E[0][6 + j] = j < 8 ? Cf[0][std8[j < 8 ? j : 0]] * w0 : 0.0;
            E[1][6 + j] = j < 8 ? Cf[1][std8[j < 8 ? j : 0]] * w1 : 0.0;
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < NCX - 6; ++j) {
        double c0 = 0, c1 = 0;
        if (6 + j < C.ncol) {
            const int row = C.iorow[j];
#pragma unroll
            for (int rr = 0; rr < MAXIO; ++rr) if (rr == row) { c0 = Cf[0][rr]; c1 = Cf[1][rr]; }
        }
        E[0][6 + j] = c0 * w0; E[1][6 + j] = c1 * w1;
    }
}

// Evaluate one observation, weight it, mask fixed parameters and gather the
// camera-side columns E = [A | C(:,estimated IO rows)].  NCX = 6 (fixed IO) or
// the capacity of E (6 + up to NCX-6 estimated IO rows).
template <int MODEL, int NCX>
__device__ __forceinline__ void eval_obs_cols_n(const DevProblem &d, const CamRec &C, const double *z,
                                                int64_t o, int pt, double r[2], double E[2][NCX], double B[2][3]) {
    constexpr bool WITH_IO = NCX > 6;
    const double *q = z + d.NS + 3 * (int64_t)pt;
    const double Q[3] = {q[0], q[1], q[2]};
    double A[2][6];
    double Cf[2][MAXIO];                       // untouched (and optimised away) when !WITH_IO
    obs_eval<MODEL, true, WITH_IO, !WITH_IO>(C, d.nK, d.nP, Q, d.o_uv[2 * o], d.o_uv[2 * o + 1], r, A, B, Cf);   // fixed IO: rhs is precomputed
    const double w0 = d.o_w ? d.o_w[2 * o] : C.w[0], w1 = d.o_w ? d.o_w[2 * o + 1] : C.w[1];
    r[0] *= w0; r[1] *= w1;
    const uint8_t *pe = d.z_est + d.NS + 3 * (int64_t)pt;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double m = pe[k] ? 1.0 : 0.0;
        B[0][k] *= w0 * m; B[1][k] *= w1 * m;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double m = ((C.eo_est >> k) & 1u) ? 1.0 : 0.0;
        E[0][k] = A[0][k] * w0 * m; E[1][k] = A[1][k] * w1 * m;
    }
    if constexpr (WITH_IO) io_columns<NCX>(C, Cf, w0, w1, E);
}

// Variant with the operands already in registers (prefetched one batch ahead by
// k_build_tile2): Q = object point, (u,v), weights, est = bit k set if point
// coordinate k is estimated.  NCX = 6 (fixed IO) or the capacity of E.
template <int MODEL, int NCX>
__device__ __forceinline__ void eval_obs_pre(const DevProblem &d, const CamRec &C, const double Q[3], double u,
                                             double v, double w0, double w1, unsigned est, double r[2],
                                             double E[2][NCX], double B[2][3]) {
    constexpr bool WITH_IO = NCX > 6;
    double A[2][6];
    double Cf[2][MAXIO];
    obs_eval<MODEL, true, WITH_IO, !WITH_IO>(C, d.nK, d.nP, Q, u, v, r, A, B, Cf);   // fixed IO: (u, v) hold rhs
    r[0] *= w0; r[1] *= w1;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double m = ((est >> k) & 1u) ? 1.0 : 0.0;
        B[0][k] *= w0 * m; B[1][k] *= w1 * m;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double m = ((C.eo_est >> k) & 1u) ? 1.0 : 0.0;
        E[0][k] = A[0][k] * w0 * m; E[1][k] = A[1][k] * w1 * m;
    }
    if constexpr (WITH_IO) io_columns<NCX>(C, Cf, w0, w1, E);
}

// Sum of x over the 64 lanes of the wave (result valid in every lane): inclusive row sums by
// DPP row shifts (VALU only, no LDS), then the four row totals by v_readlane.
__device__ __forceinline__ double wave_sum_f64(double x) {
#define DBAT_DPP_ADD(CTRL)                                                                          \
    {                                                                                               \
        const long long b_ = __double_as_longlong(x);                                               \
        const int lo_ = __builtin_amdgcn_update_dpp(0, (int)(b_ & 0xffffffffll), CTRL, 0xf, 0xf, true); \
        const int hi_ = __builtin_amdgcn_update_dpp(0, (int)(b_ >> 32), CTRL, 0xf, 0xf, true);      \
        x += __longlong_as_double(((long long)hi_ << 32) | (unsigned int)lo_);                      \
    }
    DBAT_DPP_ADD(0x111) DBAT_DPP_ADD(0x112) DBAT_DPP_ADD(0x114) DBAT_DPP_ADD(0x118)   // row_shr:1,2,4,8
#undef DBAT_DPP_ADD
    const long long b = __double_as_longlong(x);
    const int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
    auto rl = [&](int lane) {
        return __longlong_as_double(((long long)__builtin_amdgcn_readlane(hi, lane) << 32) |
                                    (unsigned int)__builtin_amdgcn_readlane(lo, lane));
    };
    return (rl(15) + rl(31)) + (rl(47) + rl(63));
}

// 1/x by v_rcp_f64 and two Newton steps (no division sequence on the path)
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = r * (2.0 - x * r);
    return r * (2.0 - x * r);
}

// Elimination of an object point's 3 x 3 block V (symmetric storage a00 a01 a02 a11 a12 a22), stably: V = U D U'
// with U UNIT UPPER triangular (the last coordinate is eliminated first), hence V^-1 = R R' with
// R = U^-T D^-1/2 LOWER triangular -- the Cholesky factor of V^-1, obtained from V itself and not from an
// explicitly formed inverse.  Z = W R then gives the point's Schur term Z Z' = W V^-1 W' with the backward error
// of a Cholesky factorisation of the full normal matrix (a small RELATIVE perturbation of V).  The adjugate
// inverse used until round 3 loses cond(V) eps there: a point seen under a narrow angle next to a camera with
// very large derivatives could push a pivot of the reduced system below zero while the full factorisation
// (MATLAB's, the oracle's) still goes through (bench/fuzz_solve.py, start values far from the solution).
// R = {r00, r10, r20, r11, r21, r22}; inv = R R' (for the back-substitution).  V not positive definite: NaN.
__device__ __forceinline__ void point_block_factor(const double (&V)[6], double (&R)[6], double (&inv)[6]) {
    const double s2 = fast_rcp(sqrt(V[5])), i2 = s2 * s2;
    const double u12 = V[4] * i2, u02 = V[2] * i2;
    const double d1 = V[3] - u12 * V[4];
    const double s1 = fast_rcp(sqrt(d1)), i1 = s1 * s1;
    const double u01 = (V[1] - u02 * V[4]) * i1;
    const double d0 = V[0] - u01 * u01 * d1 - u02 * V[2];
    const double s0 = fast_rcp(sqrt(d0));
    R[0] = s0; R[1] = -u01 * s0; R[2] = (u01 * u12 - u02) * s0;
    R[3] = s1; R[4] = -u12 * s1; R[5] = s2;
    inv[0] = R[0] * R[0]; inv[1] = R[0] * R[1]; inv[2] = R[0] * R[2];
    inv[3] = R[1] * R[1] + R[3] * R[3]; inv[4] = R[1] * R[2] + R[3] * R[4];
    inv[5] = R[2] * R[2] + R[4] * R[4] + R[5] * R[5];
}
// The point arrays keep R (six doubles per point, where V^-1 used to be): p = -V^-1 s = -R (R' s), through the factor
__device__ __forceinline__ void point_block_solve_neg(const double *__restrict__ R, double s0, double s1, double s2,
                                                      double &p0, double &p1, double &p2) {
    const double y0 = R[0] * s0 + R[1] * s1 + R[2] * s2, y1 = R[3] * s1 + R[4] * s2, y2 = R[5] * s2;
    p0 = -(R[0] * y0);
    p1 = -(R[1] * y0 + R[3] * y1);
    p2 = -(R[2] * y0 + R[4] * y1 + R[5] * y2);
}
// ... and V^-1 = R R' itself (posterior covariance of the object points)
__device__ __forceinline__ void point_block_inverse(const double *__restrict__ R, double (&inv)[6]) {
    inv[0] = R[0] * R[0]; inv[1] = R[0] * R[1]; inv[2] = R[0] * R[2];
    inv[3] = R[1] * R[1] + R[3] * R[3]; inv[4] = R[1] * R[2] + R[3] * R[4];
    inv[5] = R[2] * R[2] + R[4] * R[4] + R[5] * R[5];
}

template <int MODEL, bool WITH_IO>
__device__ __forceinline__ void eval_obs_cols(const DevProblem &d, const CamRec &C, const double *z,
                                              int64_t o, int pt, double r[2],
                                              double E[2][WITH_IO ? MAXCOL : 6], double B[2][3]) {
    eval_obs_cols_n<MODEL, (WITH_IO ? MAXCOL : 6)>(d, C, z, o, pt, r, E, B);
}

// ---------------------------------------------------------------- K1 ----
// One workgroup per batch; lane t <-> observation batch_start[b]+t.
// Outputs: S (lower triangle, NS x NS column-major, atomically accumulated),
// g_c (J_c' r), g_red (g_c - W V^-1 g_p), diagU (sum of squared camera-side
// Jacobian columns), per point R[6] (V^-1 = R R': point_block_factor; the array is still called Vinv), gp[3], jn2[3] (squared column norms),
// r_w (weighted residuals), partial sums of r'r.
template <int MODEL, bool WITH_IO>
__global__ __launch_bounds__(256) void k_build(DevProblem d, const double *__restrict__ z,
                                               const CamRec *__restrict__ cams, double lambda, int scale,
                                               double *__restrict__ S, double *__restrict__ g_c,
                                               double *__restrict__ g_red, double *__restrict__ diagU,
                                               double *__restrict__ Vinv, double *__restrict__ gp,
                                               double *__restrict__ jn2p,
                                               double *__restrict__ partial,
                                               unsigned long long *__restrict__ pivmm, int batch0) {
    constexpr int NCX = WITH_IO ? MAXCOL : 6;
    extern __shared__ double smem[];
    const int BT = blockDim.x;
    const int strideW = d.ncolmax * 3;
    double *Wl = smem;                               // [BT][strideW]
    double *red = Wl + (size_t)BT * strideW;         // [BT][9]  B'B (6) | B'r (3)
    double *pinfo = red + (size_t)BT * 9;            // [BT][9]  R (6: V^-1 = R R') | y = R' g (3)
    __shared__ double sh[8];

    const int t = threadIdx.x;
    const int64_t o0 = d.batch_start[batch0 + blockIdx.x];
    const int nobs = (int)(d.batch_start[batch0 + blockIdx.x + 1] - o0);
    const bool active = t < nobs;
    const int64_t o = o0 + t;
    // deterministic mode (DevProblem::deterministic): the camera side of EVERY observation comes from the camera-major
    // kernels (summed in chunk order); here only the Schur terms are added, each onto the grid of its element -- exact sums
    const bool det = d.deterministic != 0;
    const double *U = d.det_u;
    const double uf = det ? U[d.NS] : 0.0;

    double r[2] = {0, 0};
    double E[2][NCX];
    double B[2][3];
    int cam = 0, pt = 0, seg_start = 0, seg_len = 0, ncol = 6;
    const CamRec *C = cams;
    if (active) {
        cam = d.o_cam[o]; pt = d.o_pt[o];
        const uint32_t sg = d.o_seg[o];
        seg_start = sg & 0xFFFF; seg_len = sg >> 16;
        C = cams + cam;
        ncol = WITH_IO ? C->ncol : 6;
        eval_obs_cols<MODEL, WITH_IO>(d, *C, z, o, pt, r, E, B);
        double *rd = red + (size_t)t * 9;
        rd[0] = B[0][0] * B[0][0] + B[1][0] * B[1][0];
        rd[1] = B[0][0] * B[0][1] + B[1][0] * B[1][1];
        rd[2] = B[0][0] * B[0][2] + B[1][0] * B[1][2];
        rd[3] = B[0][1] * B[0][1] + B[1][1] * B[1][1];
        rd[4] = B[0][1] * B[0][2] + B[1][1] * B[1][2];
        rd[5] = B[0][2] * B[0][2] + B[1][2] * B[1][2];
        rd[6] = B[0][0] * r[0] + B[1][0] * r[1];
        rd[7] = B[0][1] * r[0] + B[1][1] * r[1];
        rd[8] = B[0][2] * r[0] + B[1][2] * r[1];
    }
    __syncthreads();
    // ---- per point: V, g_p, damping, priors, V^-1 (K3,K4)
    double pmin = 1e300, pmax = 0.0;       // Cholesky pivots of the estimated point coordinates
    if (active && t == seg_start) {
        double V[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
        for (int j = 0; j < seg_len; ++j) {
            const double *rd = red + (size_t)(t + j) * 9;
#pragma unroll
            for (int k = 0; k < 6; ++k) V[k] += rd[k];
            g[0] += rd[6]; g[1] += rd[7]; g[2] += rd[8];
        }
        const int64_t zp = d.NS + 3 * (int64_t)pt;
        const int dix[3] = {0, 3, 5};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double pw = d.z_prw[zp + k];
            if (pw > 0) { V[dix[k]] += pw; g[k] += pw * (z[zp + k] - d.z_prv[zp + k]); }
            jn2p[3 * (int64_t)pt + k] = V[dix[k]];
            if (d.z_est[zp + k]) V[dix[k]] += lambda; else V[dix[k]] = 1.0;
        }
        double inv[6], Rpb[6];
        point_block_factor(V, Rpb, inv);
        {   // diag of chol(V): the leading pivots of the full normal-matrix factor
            const double d0 = sqrt(V[0]), l10 = V[1] / d0, l20 = V[2] / d0;
            const double d1 = sqrt(V[3] - l10 * l10), l21 = (V[4] - l20 * l10) / d1;
            const double d2 = sqrt(V[5] - l20 * l20 - l21 * l21);
            const double dd[3] = {d0, d1, d2};
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (d.z_est[zp + k]) {
                    // column scaling D multiplies the pivots by 1/||J(:,k)||
                    double v = scale ? dd[k] / sqrt(jn2p[3 * (int64_t)pt + k]) : dd[k];
                    v = v == v ? v : 0.0;                            // NaN pivot => 0
                    pmin = fmin(pmin, v); pmax = fmax(pmax, v);
                }
        }
        double *pi = pinfo + (size_t)t * 9;
#pragma unroll
        for (int k = 0; k < 6; ++k) { pi[k] = Rpb[k]; Vinv[6 * (int64_t)pt + k] = Rpb[k]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) gp[3 * (int64_t)pt + k] = g[k];
        pi[6] = Rpb[0] * g[0] + Rpb[1] * g[1] + Rpb[2] * g[2];      // y = R' g
        pi[7] = Rpb[3] * g[1] + Rpb[4] * g[2];
        pi[8] = Rpb[5] * g[2];
    }
    __syncthreads();
    // ---- W = E'B, Z = W R (V^-1 = R R': point_block_factor), reduced right-hand side -Z y
    double Vi[6] = {0, 0, 0, 0, 0, 0}, gpt[3] = {0, 0, 0};
    double W[WITH_IO ? 1 : 6][3], Y[WITH_IO ? 1 : 6][3];
    if (active) {
        const double *pi = pinfo + (size_t)seg_start * 9;
#pragma unroll
        for (int k = 0; k < 6; ++k) Vi[k] = pi[k];
        gpt[0] = pi[6]; gpt[1] = pi[7]; gpt[2] = pi[8];
        double *wl = Wl + (size_t)t * strideW;
#pragma unroll
        for (int a = 0; a < NCX; ++a) {
            if (a < ncol) {
                const double w0 = E[0][a] * B[0][0] + E[1][a] * B[1][0];
                const double w1 = E[0][a] * B[0][1] + E[1][a] * B[1][1];
                const double w2 = E[0][a] * B[0][2] + E[1][a] * B[1][2];
                // Z = W R; the scratch rows and Y hold Z: the point's Schur term is the Gram matrix Z Z'
                const double y0 = w0 * Vi[0] + w1 * Vi[1] + w2 * Vi[2];
                const double y1 = w1 * Vi[3] + w2 * Vi[4];
                const double y2 = w2 * Vi[5];
                wl[3 * a] = y0; wl[3 * a + 1] = y1; wl[3 * a + 2] = y2;
                if constexpr (!WITH_IO) { W[a][0] = w0; W[a][1] = w1; W[a][2] = w2; Y[a][0] = y0; Y[a][1] = y1; Y[a][2] = y2; }
                if (WITH_IO && a >= 6) continue;     // IO columns: below, summed over the wave where the lanes share them
                const int col = C->col[a];
                if (det) { atomic_add_f64(g_red + col, det_round(-(y0 * gpt[0] + y1 * gpt[1] + y2 * gpt[2]), U[col], uf)); continue; }
                const double ga = E[0][a] * r[0] + E[1][a] * r[1];
                atomic_add_f64(g_c + col, ga);
                atomic_add_f64(g_red + col, ga - (y0 * gpt[0] + y1 * gpt[1] + y2 * gpt[2]));
                atomic_add_f64(diagU + col, E[0][a] * E[0][a] + E[1][a] * E[1][a]);
            }
        }
    }
    // An estimated IO parameter is usually shared by many images (one camera: by all of them), so every observation
    // of the batch adds to the SAME elements -- 2 000 observations of the camcal demo queue up on nine addresses
    // of g_c and on the 45 of the IO x IO block (11.7 ms per linearisation in round 3, where 0.1 would do).  In
    // converged control flow the lanes compare their targets: if all of them agree, the wave adds up first
    // (DPP row sums, no LDS) and one lane issues the atomic; otherwise every lane goes its own way as before.
    // (ua, ub: deterministic mode -- the grid of the target; the wave's sum is formed in a fixed order and rounded once)
    auto add_shared = [&](double *base, int64_t idx, bool valid, double val, double ua = 0.0, double ub = 0.0) {
        const unsigned long long vm = __ballot(valid);
        if (!vm) return;
        const int fl = __ffsll((long long)vm) - 1;
        const int64_t idx0 = __shfl(idx, fl, 64);
        if (__ballot(valid && idx != idx0) == 0) {
            double sum = wave_sum_f64(valid ? val : 0.0);
            if (det) sum = det_round(sum, __shfl(ua, fl, 64), __shfl(ub, fl, 64));
            if ((int)(threadIdx.x & 63) == fl) atomic_add_f64(base + idx0, sum);
        } else if (valid) atomic_add_f64(base + idx, det ? det_round(val, ua, ub) : val);
    };
    if constexpr (WITH_IO) {
#pragma unroll
        for (int a = 6; a < NCX; ++a) {
            if (__ballot(active && a < ncol) == 0) break;
            const bool valid = active && a < ncol;
            double ea0 = 0, ea1 = 0;
#pragma unroll
            for (int q = 6; q < NCX; ++q) if (q == a) { ea0 = E[0][q]; ea1 = E[1][q]; }
            const double *wl = Wl + (size_t)t * strideW;
            const int64_t col = valid ? C->col[a] : 0;
            const double ga = (valid && !det) ? ea0 * r[0] + ea1 * r[1] : 0.0;
            const double gr = valid ? ga - (wl[3 * a] * gpt[0] + wl[3 * a + 1] * gpt[1] + wl[3 * a + 2] * gpt[2]) : 0.0;
            if (!det) add_shared(g_c, col, valid, ga);
            add_shared(g_red, col, valid, gr, (det && valid) ? U[col] : 0.0, uf);
            if (!det) add_shared(diagU, col, valid, ea0 * ea0 + ea1 * ea1);
        }
    }
    // Shared IO columns again: if every observation of a point sees the SAME IO columns (one camera, or one IO block
    // per point), the pair terms that involve an IO column collapse -- with ZU = sum_j Z_io,j over the point's
    // observations, sum_j Y_i W_j'(EO x IO) = Z_eo,i ZU' and sum_i sum_j (IO x IO) = ZU ZU': 54 atomics per
    // observation and 45 per point instead of 54 k and 45 k^2.  The point's leader finds out (its observations compare
    // their column lists with its own) and sums ZU into its own scratch row.
    __shared__ int same_io[256];
    bool blocksum = false;
    if constexpr (WITH_IO) {
        if (t < BT) same_io[t] = 1;
        __syncthreads();
        if (active && t != seg_start) {
            const CamRec *CL = cams + d.o_cam[o0 + seg_start];
            bool same = CL->ncol == ncol;
            for (int a = 6; a < NCX; ++a) if (a < ncol && same) same = CL->col[a] == C->col[a];
            if (!same) same_io[seg_start] = 0;
        }
        __syncthreads();
        blocksum = active && ncol > 6 && same_io[seg_start] != 0;
        if (blocksum && t == seg_start) {
            double *wl = Wl + (size_t)t * strideW;
            for (int jj = seg_start + 1; jj < seg_start + seg_len; ++jj) {
                const double *wj = Wl + (size_t)jj * strideW;
                for (int q = 18; q < 3 * ncol; ++q) wl[q] += wj[q];
            }
        }
    }
    __syncthreads();
    // ---- Schur complement (K5): S(rows of obs j, cols of obs i) += [i==j] E'E - Y_i W_j'
    if (active) {
        if constexpr (!WITH_IO) {
            // cameras ascend inside a point, so partners j>=i give lower-triangle blocks
            const int cbase = 6 * cam;
            for (int jj = t; jj < seg_start + seg_len; ++jj) {
                const double *wj = Wl + (size_t)jj * strideW;
                const int rbase = 6 * d.o_cam[o0 + jj];
#pragma unroll
                for (int b = 0; b < 6; ++b) {
                    const double wb0 = wj[3 * b], wb1 = wj[3 * b + 1], wb2 = wj[3 * b + 2];
#pragma unroll
                    for (int a = 0; a < 6; ++a) {
                        double val = -(Y[a][0] * wb0 + Y[a][1] * wb1 + Y[a][2] * wb2);
                        if (jj == t) {
                            if (b < a) continue;
                            if (!det) val = fma2(val, E[0][a], E[0][b], E[1][a], E[1][b]);
                        }
                        if (det) val = det_round(val, U[cbase + a], U[rbase + b]);
                        atomic_add_f64(S + (int64_t)(cbase + a) * d.ldS + (rbase + b), val);
                    }
                }
            }
        } else {
            const double *wi = Wl + (size_t)t * strideW;
            const double *zu = Wl + (size_t)seg_start * strideW;      // (block sums: the leader's IO part holds ZU)
            for (int a = 0; a < NCX; ++a) {
                if (a >= ncol || (blocksum && a >= 6)) break;
                const int gcol = C->col[a];
                const double y0 = wi[3 * a], y1 = wi[3 * a + 1], y2 = wi[3 * a + 2];      // (the scratch rows hold Z)
                double ea0 = 0, ea1 = 0;
#pragma unroll
                for (int q = 0; q < NCX; ++q) if (q == a) { ea0 = E[0][q]; ea1 = E[1][q]; }
                if (blocksum) {                          // EO column a x the point's IO columns, once
#pragma unroll
                    for (int b = 6; b < NCX; ++b) {
                        if (b >= ncol) break;
                        const double pz = -(y0 * zu[3 * b] + y1 * zu[3 * b + 1] + y2 * zu[3 * b + 2]);
                        atomic_add_f64(S + (int64_t)gcol * d.ldS + C->col[b],
                                       det ? det_round(pz, U[gcol], U[C->col[b]]) : ea0 * E[0][b] + ea1 * E[1][b] + pz);
                    }
                }
                for (int jj = seg_start; jj < seg_start + seg_len; ++jj) {
                    const CamRec *Cj = cams + d.o_cam[o0 + jj];
                    const double *wj = Wl + (size_t)jj * strideW;
                    const int ncj = blocksum ? 6 : Cj->ncol;
                    for (int b = 0; b < ncj; ++b) {
                        if (a >= 6 && b >= 6) break;     // IO x IO: below
                        const int grow = Cj->col[b];
                        if (grow < gcol) continue;
                        double val = -(y0 * wj[3 * b] + y1 * wj[3 * b + 1] + y2 * wj[3 * b + 2]);
                        if (jj == t && !det) {
                            double eb0 = 0, eb1 = 0;
#pragma unroll
                            for (int q = 0; q < NCX; ++q) if (q == b) { eb0 = E[0][q]; eb1 = E[1][q]; }
                            val = fma2(val, ea0, eb0, ea1, eb1);
                        }
                        if (det) val = det_round(val, U[gcol], U[grow]);
                        atomic_add_f64(S + (int64_t)gcol * d.ldS + grow, val);
                    }
                }
            }
        }
    }
    if constexpr (WITH_IO) {
        // IO x IO elements of the pair terms (and of E'E on the diagonal pair), all lanes in step: own IO column a,
        // partner j of the point, partner's IO column b
        // (a lane of a block-sum point takes part once, j = 0: its own E_io'E_io, and -- the leader -- ZU ZU')
        int maxlen = active ? (blocksum ? 1 : seg_len) : 0;
        for (int off = 32; off > 0; off >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off, 64));
        const double *wi = Wl + (size_t)t * strideW;
        const bool leader = t == seg_start;
        for (int a = 6; a < NCX; ++a) {
            if (__ballot(active && a < ncol) == 0) break;
            const bool va = active && a < ncol;
            const int64_t gcol = va ? C->col[a] : 0;
            double ea0 = 0, ea1 = 0;
#pragma unroll
            for (int q = 6; q < NCX; ++q) if (q == a) { ea0 = E[0][q]; ea1 = E[1][q]; }
            // block sums: the pair term comes from the leader alone, who holds ZU in its scratch row
            const bool pairs = va && (!blocksum || leader);
            const double y0 = pairs ? wi[3 * a] : 0.0, y1 = pairs ? wi[3 * a + 1] : 0.0, y2 = pairs ? wi[3 * a + 2] : 0.0;
            for (int j = 0; j < maxlen; ++j) {
                const bool vj = va && j < (blocksum ? 1 : seg_len);
                const int jj = (vj && !blocksum) ? seg_start + j : t;
                const CamRec *Cj = cams + ((vj && !blocksum) ? d.o_cam[o0 + jj] : cam);
                const double *wj = Wl + (size_t)jj * strideW;
                const int ncj = vj ? Cj->ncol : 0;
                for (int b = 6; b < NCX; ++b) {
                    if (__ballot(vj && b < ncj) == 0) break;
                    const int64_t grow = (vj && b < ncj) ? Cj->col[b] : -1;
                    const bool valid = vj && b < ncj && grow >= gcol;
                    double val = 0.0;
                    if (valid) {
                        val = -(y0 * wj[3 * b] + y1 * wj[3 * b + 1] + y2 * wj[3 * b + 2]);
                        if (jj == t && !det) {
                            double eb0 = 0, eb1 = 0;
#pragma unroll
                            for (int q = 6; q < NCX; ++q) if (q == b) { eb0 = E[0][q]; eb1 = E[1][q]; }
                            val = fma2(val, ea0, eb0, ea1, eb1);
                        }
                    }
                    add_shared(S, gcol * d.ldS + grow, valid, val, (det && valid) ? U[gcol] : 0.0, (det && valid) ? U[grow] : 0.0);
                }
            }
        }
    }
    double acc[1] = {r[0] * r[0] + r[1] * r[1]};
    block_sum<1>(acc, sh);
    if (t == 0) partial[blockIdx.x] = acc[0];
    for (int off = 32; off > 0; off >>= 1) {
        pmin = fmin(pmin, __shfl_down(pmin, off, 64));
        pmax = fmax(pmax, __shfl_down(pmax, off, 64));
    }
    if ((t & 63) == 0 && pmax > 0.0) {      // positive doubles order like their bit patterns
        atomicMin(pivmm, (unsigned long long)__double_as_longlong(pmin));
        atomicMax(pivmm + 1, (unsigned long long)__double_as_longlong(pmax));
    }
}

// ---------------------------------------------------------------- K1g ---
// k_build for an object point with more observations than one batch holds
// (a control point seen in hundreds of images): one workgroup per point, the
// observations in chunks of blockDim.  Pass 1 evaluates every observation
// (residual, E'E, gradient pieces, W = E'B to scratch) and sums V = B'B, B'r
// over the whole point; then V^-1; pass 2 removes (W V^-1) g_p from the reduced
// right-hand side and adds the k^2 pair terms  -Y_i W_j'  with global atomics.
// Same outputs as k_build.
template <int MODEL, bool WITH_IO>
__global__ __launch_bounds__(256) void k_build_giant(DevProblem d, const double *__restrict__ z,
                                                     const CamRec *__restrict__ cams, double lambda, int scale,
                                                     double *__restrict__ S, double *__restrict__ g_c,
                                                     double *__restrict__ g_red, double *__restrict__ diagU,
                                                     double *__restrict__ Vinv, double *__restrict__ gp,
                                                     double *__restrict__ jn2p,
                                                     double *__restrict__ partial,
                                                     unsigned long long *__restrict__ pivmm) {
    constexpr int NCX = WITH_IO ? MAXCOL : 6;
    __shared__ double sh[9 * 4];
    __shared__ double pin[9];
    const int t = threadIdx.x, BT = blockDim.x;
    const int64_t o0 = d.giant_start[blockIdx.x], o1 = d.giant_start[blockIdx.x + 1];
    const int k = (int)(o1 - o0);
    const int strideW = d.ncolmax * 3;
    double *Wg = d.giant_W + (o0 - d.giant_start[0]) * strideW;
    const int pt = d.o_pt[o0];
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, rr = 0.0;
    for (int i = t; i < k; i += BT) {                // ---- pass 1
        const int64_t o = o0 + i;
        const CamRec &C = cams[d.o_cam[o]];
        const int ncol = WITH_IO ? C.ncol : 6;
        double r[2], E[2][NCX], B[2][3];
        eval_obs_cols<MODEL, WITH_IO>(d, C, z, o, pt, r, E, B);
        rr = fma2(rr, r[0], r[0], r[1], r[1]);
        acc[0] = fma2(acc[0], B[0][0], B[0][0], B[1][0], B[1][0]);
        acc[1] = fma2(acc[1], B[0][0], B[0][1], B[1][0], B[1][1]);
        acc[2] = fma2(acc[2], B[0][0], B[0][2], B[1][0], B[1][2]);
        acc[3] = fma2(acc[3], B[0][1], B[0][1], B[1][1], B[1][1]);
        acc[4] = fma2(acc[4], B[0][1], B[0][2], B[1][1], B[1][2]);
        acc[5] = fma2(acc[5], B[0][2], B[0][2], B[1][2], B[1][2]);
        acc[6] = fma2(acc[6], B[0][0], r[0], B[1][0], r[1]);
        acc[7] = fma2(acc[7], B[0][1], r[0], B[1][1], r[1]);
        acc[8] = fma2(acc[8], B[0][2], r[0], B[1][2], r[1]);
        double *wl = Wg + (size_t)i * strideW;
        for (int a = 0; a < NCX; ++a) {
            if (a >= ncol) break;
            double ea0 = 0, ea1 = 0;
#pragma unroll
            for (int q = 0; q < NCX; ++q) if (q == a) { ea0 = E[0][q]; ea1 = E[1][q]; }
            wl[3 * a] = ea0 * B[0][0] + ea1 * B[1][0];
            wl[3 * a + 1] = ea0 * B[0][1] + ea1 * B[1][1];
            wl[3 * a + 2] = ea0 * B[0][2] + ea1 * B[1][2];
            if (d.deterministic) continue;           // (the camera side comes from the camera-major kernels, DevProblem::deterministic)
            const int gcol = C.col[a];
            const double ga = ea0 * r[0] + ea1 * r[1];
            atomic_add_f64(g_c + gcol, ga);
            atomic_add_f64(g_red + gcol, ga);
            atomic_add_f64(diagU + gcol, ea0 * ea0 + ea1 * ea1);
            for (int b = 0; b < ncol; ++b) {         // E'E of this observation, lower triangle
                const int grow = C.col[b];
                if (grow < gcol) continue;
                double eb0 = 0, eb1 = 0;
#pragma unroll
                for (int q = 0; q < NCX; ++q) if (q == b) { eb0 = E[0][q]; eb1 = E[1][q]; }
                atomic_add_f64(S + (int64_t)gcol * d.ldS + grow, ea0 * eb0 + ea1 * eb1);
            }
        }
    }
    block_sum<9>(acc, sh);
    double pmin = 1e300, pmax = 0.0;
    if (t == 0) {                                    // ---- V, damping, priors, V^-1 (as k_build P2)
        double V[6] = {acc[0], acc[1], acc[2], acc[3], acc[4], acc[5]}, g[3] = {acc[6], acc[7], acc[8]};
        const int64_t zp = d.NS + 3 * (int64_t)pt;
        const int dix[3] = {0, 3, 5};
        double jn[3];
        for (int q = 0; q < 3; ++q) {
            const double pw = d.z_prw[zp + q];
            if (pw > 0) { V[dix[q]] += pw; g[q] += pw * (z[zp + q] - d.z_prv[zp + q]); }
            jn[q] = V[dix[q]];
            jn2p[3 * (int64_t)pt + q] = jn[q];
            if (d.z_est[zp + q]) V[dix[q]] += lambda; else V[dix[q]] = 1.0;
        }
        double inv[6], Rpb[6];
        point_block_factor(V, Rpb, inv);
        const double d0 = sqrt(V[0]), l10 = V[1] / d0, l20 = V[2] / d0;
        const double d1 = sqrt(V[3] - l10 * l10), l21 = (V[4] - l20 * l10) / d1;
        const double d2 = sqrt(V[5] - l20 * l20 - l21 * l21);
        const double dd[3] = {d0, d1, d2};
        for (int q = 0; q < 3; ++q)
            if (d.z_est[zp + q]) {
                double v = scale ? dd[q] / sqrt(jn[q]) : dd[q];
                v = v == v ? v : 0.0;
                pmin = fmin(pmin, v); pmax = fmax(pmax, v);
            }
        for (int q = 0; q < 6; ++q) { pin[q] = Rpb[q]; Vinv[6 * (int64_t)pt + q] = Rpb[q]; }
        for (int q = 0; q < 3; ++q) gp[3 * (int64_t)pt + q] = g[q];
        pin[6] = Rpb[0] * g[0] + Rpb[1] * g[1] + Rpb[2] * g[2];     // y = R' g
        pin[7] = Rpb[3] * g[1] + Rpb[4] * g[2];
        pin[8] = Rpb[5] * g[2];
        if (pmax > 0.0) {
            atomicMin(pivmm, (unsigned long long)__double_as_longlong(pmin));
            atomicMax(pivmm + 1, (unsigned long long)__double_as_longlong(pmax));
        }
    }
    __threadfence_block();
    __syncthreads();                                 // pin and the W scratch of the whole point are visible
    const double v0 = pin[0], v1 = pin[1], v2 = pin[2], v3 = pin[3], v4 = pin[4], v5 = pin[5];      // R: V^-1 = R R'
    const double g0 = pin[6], g1 = pin[7], g2 = pin[8];                                               // y = R' g
    for (int i = t; i < k; i += BT) {                // the scratch rows W -> Z = W R: the Schur term is the Gram matrix Z Z'
        const int nci = WITH_IO ? cams[d.o_cam[o0 + i]].ncol : 6;
        double *wi = Wg + (size_t)i * strideW;
        for (int a = 0; a < nci; ++a) {
            const double w0 = wi[3 * a], w1 = wi[3 * a + 1], w2 = wi[3 * a + 2];
            wi[3 * a] = w0 * v0 + w1 * v1 + w2 * v2; wi[3 * a + 1] = w1 * v3 + w2 * v4; wi[3 * a + 2] = w2 * v5;
        }
    }
    __threadfence_block();
    __syncthreads();
    for (int i = t; i < k; i += BT) {                // ---- pass 2
        const CamRec &Ci = cams[d.o_cam[o0 + i]];
        const int nci = WITH_IO ? Ci.ncol : 6;
        const double *wi = Wg + (size_t)i * strideW;
        for (int a = 0; a < nci; ++a) {
            const int gcol = Ci.col[a];
            const double y0 = wi[3 * a], y1 = wi[3 * a + 1], y2 = wi[3 * a + 2];
            const bool det = d.deterministic != 0;   // (every term onto the grid of its element: exact sums)
            const double gy = -(y0 * g0 + y1 * g1 + y2 * g2);
            atomic_add_f64(g_red + gcol, det ? det_round(gy, d.det_u[gcol], d.det_u[d.NS]) : gy);
            // partners: fixed IO -> cameras ascend inside a point, j >= i covers the lower triangle;
            // with IO columns every ordered pair whose row is not above the column
            for (int j = WITH_IO ? 0 : i; j < k; ++j) {
                const CamRec &Cj = cams[d.o_cam[o0 + j]];
                const int ncj = WITH_IO ? Cj.ncol : 6;
                const double *wj = Wg + (size_t)j * strideW;
                for (int b = 0; b < ncj; ++b) {
                    const int grow = Cj.col[b];
                    if (grow < gcol) continue;
                    if (!WITH_IO && j == i && b < a) continue;
                    const double pz = -(y0 * wj[3 * b] + y1 * wj[3 * b + 1] + y2 * wj[3 * b + 2]);
                    atomic_add_f64(S + (int64_t)gcol * d.ldS + grow, det ? det_round(pz, d.det_u[gcol], d.det_u[grow]) : pz);
                }
            }
        }
    }
    double accr[1] = {rr};
    block_sum<1>(accr, sh);
    if (t == 0) partial[blockIdx.x] = accr[0];
}

// Back-substitution for a giant point (k_backsub for one point per workgroup).
template <int MODEL, bool WITH_IO>
__global__ __launch_bounds__(256) void k_backsub_giant(DevProblem d, const double *__restrict__ z,
                                                       const CamRec *__restrict__ cams,
                                                       const double *__restrict__ Vinv, const double *__restrict__ gp,
                                                 double *__restrict__ dz,
                                                       double *__restrict__ partial /* [ngiant][2] */) {
    constexpr int NCX = WITH_IO ? MAXCOL : 6;
    __shared__ double sh[3 * 4];
    __shared__ double dps[3];
    const int t = threadIdx.x, BT = blockDim.x;
    const int64_t o0 = d.giant_start[blockIdx.x], o1 = d.giant_start[blockIdx.x + 1];
    const int k = (int)(o1 - o0);
    const int pt = d.o_pt[o0];
    double s[3] = {0, 0, 0};
    for (int i = t; i < k; i += BT) {
        const int64_t o = o0 + i;
        const CamRec &C = cams[d.o_cam[o]];
        const int ncol = WITH_IO ? C.ncol : 6;
        double r[2], E[2][NCX], B[2][3], tt[2] = {0, 0};
        eval_obs_cols<MODEL, WITH_IO>(d, C, z, o, pt, r, E, B);
#pragma unroll
        for (int a = 0; a < NCX; ++a)
            if (a < ncol) { const double dc = dz[C.col[a]]; tt[0] += E[0][a] * dc; tt[1] += E[1][a] * dc; }
        s[0] = fma2(s[0], B[0][0], tt[0], B[1][0], tt[1]);
        s[1] = fma2(s[1], B[0][1], tt[0], B[1][1], tt[1]);
        s[2] = fma2(s[2], B[0][2], tt[0], B[1][2], tt[1]);
    }
    block_sum<3>(s, sh);
    if (t == 0) {
        for (int q = 0; q < 3; ++q) s[q] += gp[3 * (int64_t)pt + q];
        double p0, p1, p2;
        point_block_solve_neg(Vinv + 6 * (int64_t)pt, s[0], s[1], s[2], p0, p1, p2);
        const int64_t zp = d.NS + 3 * (int64_t)pt;
        dps[0] = d.z_est[zp] ? p0 : 0.0; dps[1] = d.z_est[zp + 1] ? p1 : 0.0; dps[2] = d.z_est[zp + 2] ? p2 : 0.0;
        dz[zp] = dps[0]; dz[zp + 1] = dps[1]; dz[zp + 2] = dps[2];
    }
    __syncthreads();
    double acc[2] = {0, 0};
    for (int i = t; i < k; i += BT) {
        const int64_t o = o0 + i;
        const CamRec &C = cams[d.o_cam[o]];
        const int ncol = WITH_IO ? C.ncol : 6;
        double r[2], E[2][NCX], B[2][3], tt[2] = {0, 0};
        eval_obs_cols<MODEL, WITH_IO>(d, C, z, o, pt, r, E, B);
#pragma unroll
        for (int a = 0; a < NCX; ++a)
            if (a < ncol) { const double dc = dz[C.col[a]]; tt[0] += E[0][a] * dc; tt[1] += E[1][a] * dc; }
        const double j0 = tt[0] + B[0][0] * dps[0] + B[0][1] * dps[1] + B[0][2] * dps[2];
        const double j1 = tt[1] + B[1][0] * dps[0] + B[1][1] * dps[1] + B[1][2] * dps[2];
        acc[0] = fma2(acc[0], j0, j0, j1, j1);
    }
    block_sum<2>(acc, sh);
    if (t == 0) { partial[2 * blockIdx.x] = acc[0]; partial[2 * blockIdx.x + 1] = acc[1]; }
}

// ---------------------------------------------------------------- K1t ---
// MFMA tile kernels (k_build_tile2 / k_build_tile3 below, k_build_sig in sig.hpp): one workgroup
// per TILE -- a run of batches whose observations touch at most CMAX <= 21 cameras, i.e. at most
// 126 rows of the reduced system -- accumulates the tile's share of the Schur complement on the
// f64 matrix cores and flushes it to HBM once.
typedef double mfma_d4 __attribute__((ext_vector_type(4)));
// Row stride of the LDS operand panels: 144 doubles = 288 dwords == 32 (mod 64
// banks), so the four k-rows one ds_read_b64 wave-instruction touches fall on
// disjoint bank halves (128 would make lanes l and l+16 collide).
constexpr int TILE_LD = 144;

// all LDS traffic of this wave has completed
__device__ __forceinline__ void lds_fence() {           // all LDS traffic of this wave has completed
    __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0)
    asm volatile("" ::: "memory");
}

// ---------------------------------------------------------------- K3c ---
// J_c'J_c and J_c'r of the tiled observations, camera by camera: one workgroup
// per chunk of ONE camera's observations (camera-major copy of the plan,
// coalesced reads of (u,v) and the point index; the camera record is uniform
// for the whole workgroup).  Every lane evaluates one observation and writes its
// two rows [E | r] (16 columns) into the wave's LDS panel; the 16 x 16 Gram matrix
//   G = sum_obs [E r]' [E r]
// accumulates on the f64 matrix cores (v_mfma_f64_16x16x4_f64, A = B = panel
// fragment).  G holds the camera's 6 x 6 block, its camera-IO and IO-IO blocks,
// E'r and the squared column norms at once; it is added to S, g_c, g_red, diagU
// with a handful of atomics per chunk -- instead of ~40 (fixed IO) to ~150
// (self-calibration) LDS atomics per observation in the tile kernel.
template <int MODEL, int NCX>
__global__ __launch_bounds__(256, 3) void k_cam_normal(DevProblem d, const double *__restrict__ z,
                                                       const CamRec *__restrict__ cams,
                                                       const int32_t *__restrict__ cm_pt, const double *__restrict__ cm_uv,
                                                       const double *__restrict__ cm_w,
                                                       const int32_t *__restrict__ chunk_cam,
                                                       const int64_t *__restrict__ chunk_start, double *__restrict__ S,
                                                       double *__restrict__ g_c, double *__restrict__ g_red,
                                                       double *__restrict__ diagU) {
    constexpr bool IO = NCX > 6;
    // Round 5: the wave's panel holds the rows of 32 observations, [16 columns][64 rows], column stride 66 doubles;
    // the two halves of the wave take turns.  34 KB per workgroup instead of 75: three workgroups per CU (three waves per
    // SIMD, 168 registers) where two stood -- the kernel spent a quarter of its wave cycles waiting for the gathers with
    // nobody to take the SIMD (profiles/r05_c4_summary.md).  Row h of lane l's observation is panel row 32 h + (l & 31):
    // consecutive lanes write consecutive doubles (the 2 l + h of round 4 put lanes l and l + 16 on one bank: 34 % of the
    // LDS cycles were conflicts); the Gram matrix does not care about the order of the rows.
    constexpr int GLD = 66;
    __shared__ double Gl[4 * 16 * GLD];              // (the four partial Gram matrices at the end go here as well)
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int cam = chunk_cam[blockIdx.x];
    const int64_t q0 = chunk_start[blockIdx.x], q1 = chunk_start[blockIdx.x + 1];
    const CamRec &C = cams[cam];
    const int ncol = IO ? min(C.ncol, NCX) : 6;
    double *Gw = Gl + w * 16 * GLD;
    // columns that stay zero
    for (int c = NCX + 1; c < 16; ++c) Gw[c * GLD + lane] = 0.0;
    mfma_d4 acc = {0, 0, 0, 0};
    // The loads of a round are a dependent pair (point index, then the gather of the point): the index and image
    // coordinates travel two rounds ahead and the point one.
    {
        auto load_obs = [&](int64_t q, int &pt, double2 &uv, double2 &ww) {
            pt = 0; uv = double2{0, 0}; ww = double2{C.w[0], C.w[1]};
            if (q < q1) {
                pt = cm_pt[q];
                uv = reinterpret_cast<const double2 *>(cm_uv)[q];
                if (cm_w) ww = reinterpret_cast<const double2 *>(cm_w)[q];
            }
        };
        const int64_t qw = q0 + 64 * w + lane;
        int pt_a, pt_b;
        double2 uv_a, uv_b, w_a, w_b;
        load_obs(qw, pt_a, uv_a, w_a);
        load_obs(qw + 256, pt_b, uv_b, w_b);
        double Qn[3];
        { const int64_t zp = d.NS + 3 * (int64_t)pt_a; Qn[0] = z[zp]; Qn[1] = z[zp + 1]; Qn[2] = z[zp + 2]; }
        const int prow = lane & 31;
        const double *ga = Gw + (lane & 15) * GLD + (lane >> 4);
        for (int64_t base = q0 + 64 * w; base < q1; base += 256) {
            const int64_t q = base + lane;
            const double Q[3] = {Qn[0], Qn[1], Qn[2]};
            const double2 uv = uv_a, ww = w_a;
            pt_a = pt_b; uv_a = uv_b; w_a = w_b;
            { const int64_t zp = d.NS + 3 * (int64_t)pt_a; Qn[0] = z[zp]; Qn[1] = z[zp + 1]; Qn[2] = z[zp + 2]; }   // (index 0 beyond the chunk)
            load_obs(q + 512, pt_b, uv_b, w_b);
            double r[2] = {0, 0}, E[2][NCX];
#pragma unroll
            for (int c = 0; c < NCX; ++c) { E[0][c] = 0.0; E[1][c] = 0.0; }
            if (q < q1) {
                double B[2][3];
                eval_obs_pre<MODEL, NCX>(d, C, Q, uv.x, uv.y, ww.x, ww.y, 7u, r, E, B);
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if ((lane >> 5) == half) {
                    if (ncol == NCX) {                   // (uniform; the usual case: no selects)
#pragma unroll
                        for (int c = 0; c < NCX; ++c) { Gw[c * GLD + prow] = E[0][c]; Gw[c * GLD + 32 + prow] = E[1][c]; }
                    } else {
#pragma unroll
                        for (int c = 0; c < NCX; ++c) {
                            const bool on = c < ncol;
                            Gw[c * GLD + prow] = on ? E[0][c] : 0.0;
                            Gw[c * GLD + 32 + prow] = on ? E[1][c] : 0.0;
                        }
                    }
                    Gw[NCX * GLD + prow] = r[0]; Gw[NCX * GLD + 32 + prow] = r[1];
                }
                lds_fence();                             // the panel is private to the wave
                if (base + 32 * half < q1) {             // (wave-uniform: the second half of a chunk's last round may be empty)
                    double av[16];                       // all operand fragments first, then the products back to back
#pragma unroll
                    for (int kk = 0; kk < 16; ++kk) av[kk] = ga[4 * kk];
#pragma unroll
                    for (int kk = 0; kk < 16; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], av[kk], acc, 0, 0, 0);
                }
                lds_fence();
            }
        }
    }
    double *Gs = Gl;
    __syncthreads();                                 // every wave is done with its panel
    // G(i, j): register e of lane l holds i = (l>>4) + 4e, j = l&15.  Sum the four waves.
#pragma unroll
    for (int e = 0; e < 4; ++e) Gs[w * 256 + ((lane >> 4) + 4 * e) * 16 + (lane & 15)] = acc[e];
    __syncthreads();
    const int i = t >> 4, j = t & 15;
    const double g = (Gs[t] + Gs[256 + t]) + (Gs[512 + t] + Gs[768 + t]);
    // deterministic mode: the chunk's Gram matrix as it is; k_det_cam_reduce adds the chunks of a camera in their order
    if (d.deterministic) { d.det_cam_part[(int64_t)blockIdx.x * DET_CP + t] = g; return; }
    if (i < ncol && g != 0.0) {
        const int64_t ri = C.col[i];
        if (j < ncol) {
            const int64_t rj = C.col[j];
            // every unordered pair once: local i > j, or the diagonal
            if (i > j) atomic_add_f64(S + (ri >= rj ? rj * d.ldS + ri : ri * d.ldS + rj), g);
            else if (i == j) { atomic_add_f64(S + ri * d.ldS + ri, g); atomic_add_f64(diagU + ri, g); }
        } else if (j == NCX) {
            atomic_add_f64(g_c + ri, g);
            atomic_add_f64(g_red + ri, g);
        }
    }
}

// Fixed interior orientation: the Gram matrix of [E | r] is 7 x 7 -- 27 useful sums.  On the matrix
// cores that is one 16 x 16 x 4 product per two observation rows with a fifth of its outputs
// used (2 048 cycles per 64 observations, more than their evaluation).  Here every lane keeps the 27
// sums of ITS observations (a chunk has up to eight per lane) in registers -- 54 FMAs per
// observation -- and the workgroup adds them up once per chunk (DPP wave sums, four partials in LDS).
template <int MODEL>
__global__ __launch_bounds__(256) void k_cam_normal6(DevProblem d, const double *__restrict__ z,
                                                     const CamRec *__restrict__ cams,
                                                     const int32_t *__restrict__ cm_pt, const double *__restrict__ cm_uv,
                                                     const double *__restrict__ cm_w,
                                                     const int32_t *__restrict__ chunk_cam,
                                                     const int64_t *__restrict__ chunk_start, double *__restrict__ S,
                                                     double *__restrict__ g_c, double *__restrict__ g_red,
                                                     double *__restrict__ diagU) {
    __shared__ double Gs[4 * 28];
    __shared__ double Rd[4][14 * 65];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int cam = chunk_cam[blockIdx.x];
    const int64_t q0 = chunk_start[blockIdx.x], q1 = chunk_start[blockIdx.x + 1];
    const CamRec &C = cams[cam];
    double G[28];                                    // 21: E'E lower triangle by rows (i >= j), then 6: E'r, then r'r
#pragma unroll
    for (int i = 0; i < 28; ++i) G[i] = 0.0;
    // four observations at a time: their point indices and coordinates are requested together, then the four
    // gathers of the object points (a chunk has at most eight observations per thread)
    const double2 *uvp = reinterpret_cast<const double2 *>(cm_uv), *wp = reinterpret_cast<const double2 *>(cm_w);
    for (int64_t qb = q0 + t; qb < q1; qb += 4 * 256) {
        int pt[4]; double2 uv[4], ww[4]; double Qv[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t q = qb + 256 * u;
            const bool on = q < q1;
            pt[u] = on ? cm_pt[q] : -1;
            uv[u] = on ? uvp[q] : double2{0, 0};
            ww[u] = (on && cm_w) ? wp[q] : double2{C.w[0], C.w[1]};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double *p = z + d.NS + 3 * (int64_t)(pt[u] < 0 ? 0 : pt[u]);
            Qv[u][0] = p[0]; Qv[u][1] = p[1]; Qv[u][2] = p[2];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (pt[u] < 0) continue;
            // residual and weighted camera block of the observation, unmasked (the masks of the camera's fixed elements are
            // applied to the sums), nothing else: the kernel is bound by the vector instructions it issues (obs_eval,
            // PRE: res_euler_brown_*.m with the image side precomputed, eulerpinhole2.m, pinhole.m:54-66)
            const double d0 = Qv[u][0] - C.c[0], d1 = Qv[u][1] - C.c[1], d2 = Qv[u][2] - C.c[2];
            const double X0 = C.Mt[0] * d0 + C.Mt[1] * d1 + C.Mt[2] * d2;
            const double X1 = C.Mt[3] * d0 + C.Mt[4] * d1 + C.Mt[5] * d2;
            const double X2 = C.Mt[6] * d0 + C.Mt[7] * d1 + C.Mt[8] * d2;
            const double iz = recip(X2);
            const double ph0 = X0 * iz, ph1 = X1 * iz;
            const double nf = -C.f, sc = nf * iz, s0 = sc * ww[u].x, s1 = sc * ww[u].y;
            const double r0 = (nf * ph0 - uv[u].x) * ww[u].x, r1 = (nf * ph1 - uv[u].y) * ww[u].y;
            double y[3][3];
            angle_terms(C, d0, d1, d2, X0, X1, X2, y);
            double e0[6], e1[6];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                e0[k] = -s0 * (C.Mt[k] - ph0 * C.Mt[6 + k]);      e1[k] = -s1 * (C.Mt[3 + k] - ph1 * C.Mt[6 + k]);
                e0[3 + k] = s0 * (y[k][0] - ph0 * y[k][2]);       e1[3 + k] = s1 * (y[k][1] - ph1 * y[k][2]);
            }
            int n = 0;
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j, ++n) G[n] = __builtin_fma(e0[i], e0[j], __builtin_fma(e1[i], e1[j], G[n]));
#pragma unroll
            for (int i = 0; i < 6; ++i) G[21 + i] = __builtin_fma(e0[i], r0, __builtin_fma(e1[i], r1, G[21 + i]));
            G[27] = __builtin_fma(r0, r0, __builtin_fma(r1, r1, G[27]));     // (the deterministic mode's bound of the right-hand side)
        }
    }
    // The 27 sums over the wave's lanes through LDS, 14 at a time: every lane leaves its values, lane l then adds value
    // l % 16 over the 16 lanes of quarter l / 16, and two exchanges join the quarters -- about 110 instructions where 27
    // DPP wave sums took 400 (a chunk has eight observations per lane: that was a quarter of the kernel's instructions).
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int v = 0; v < 14; ++v)
            Rd[w][v * 65 + lane] = G[14 * h + v];
        lds_fence();
        __builtin_amdgcn_wave_barrier();
        const int v = lane & 15, q4 = lane >> 4;
        double sum = 0.0;
        if (v < 14) {
#pragma unroll
            for (int i = 0; i < 16; ++i) sum += Rd[w][v * 65 + 16 * q4 + i];
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        if (lane < 14) Gs[w * 28 + 14 * h + lane] = sum;
        lds_fence();
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // deterministic mode: the chunk's 28 sums as they are; k_det_cam_reduce adds the chunks of a camera in their order
    if (d.deterministic) {
        if (t < 28) d.det_cam_part[(int64_t)blockIdx.x * DET_CP + t] = (Gs[t] + Gs[28 + t]) + (Gs[56 + t] + Gs[84 + t]);
        return;
    }
    if (t < 27) {
        double g = (Gs[t] + Gs[28 + t]) + (Gs[56 + t] + Gs[84 + t]);
        int i = t - 21, j = i;
        if (t < 21) {
            i = 0;
            while ((i + 1) * (i + 2) / 2 <= t) ++i;
            j = t - i * (i + 1) / 2;
        }
        if (!((C.eo_est >> i) & (C.eo_est >> j) & 1u)) g = 0.0;      // a camera element that is not estimated: no row, no column
        if (g != 0.0) {
            if (t < 21) {
                const int64_t ri = C.col[i], rj = C.col[j];
                if (i == j) { atomic_add_f64(S + ri * d.ldS + ri, g); atomic_add_f64(diagU + ri, g); }
                else atomic_add_f64(S + (ri >= rj ? rj * d.ldS + ri : ri * d.ldS + rj), g);
            } else {
                const int64_t ri = C.col[t - 21];
                atomic_add_f64(g_c + ri, g);
                atomic_add_f64(g_red + ri, g);
            }
        }
    }
}

// ---------------------------------------------------------------- deterministic mode, camera side
// One workgroup per camera: the Gram matrices of its camera-major chunks (tiled part, then untiled part) are added in chunk
// order -- one thread per element -- and go where k_cam_normal / k_cam_normal6 put them with atomics: the camera's own
// block and its camera x IO block of S, g_c, g_red, diagU (one writer each).  What several cameras share (the IO x IO
// block, the IO entries of the gradient and of the column norms) is left per camera in det_io_part for k_det_io_reduce.
template <int NCX>
__global__ __launch_bounds__(256) void k_det_cam_reduce(DevProblem d, const CamRec *__restrict__ cams, double *__restrict__ S,
                                                        double *__restrict__ g_c, double *__restrict__ g_red,
                                                        double *__restrict__ diagU) {
    const int cam = blockIdx.x, t = threadIdx.x;
    const CamRec &C = cams[cam];
    const int nel = NCX == 6 ? 28 : 256;
    double g = 0.0;
    if (t < nel) {
        for (int part = 0; part < 2; ++part) {
            const int q0 = d.det_cam_chunks[2 * cam + part * (2 * d.nc + 1)], q1 = d.det_cam_chunks[2 * cam + 1 + part * (2 * d.nc + 1)];
            for (int q = q0; q < q1; ++q) g += d.det_cam_part[(int64_t)q * DET_CP + t];
        }
    }
    if constexpr (NCX == 6) {
        if (t == 27) d.det_rr[cam] = g;
        if (t < 27) {
            int i = t - 21, j = i;
            if (t < 21) {
                i = 0;
                while ((i + 1) * (i + 2) / 2 <= t) ++i;
                j = t - i * (i + 1) / 2;
            }
            if (!((C.eo_est >> i) & (C.eo_est >> j) & 1u)) g = 0.0;      // a camera element that is not estimated: no row, no column
            if (t < 21) {
                const int64_t ri = C.col[i], rj = C.col[j];
                if (((C.eo_est >> i) & (C.eo_est >> j) & 1u)) {
                    S[(ri >= rj ? rj * d.ldS + ri : ri * d.ldS + rj)] = g;
                    if (i == j) diagU[ri] = g;
                }
            } else if ((C.eo_est >> i) & 1u) {
                const int64_t ri = C.col[i];
                g_c[ri] = g; g_red[ri] = g;
            }
        }
    } else {
        const int ncol = min(C.ncol, NCX);
        const int i = t >> 4, j = t & 15;
        if (i == NCX && j == NCX) d.det_rr[cam] = g;
        double *iop = d.det_io_part + (int64_t)cam * DET_IOP;
        if (i < ncol) {
            const int64_t ri = C.col[i];
            if (j < ncol && i >= j) {
                if (j >= 6) iop[(i - 6) * 9 + (j - 6)] = g;              // IO x IO: shared by the cameras of the block
                else {
                    const int64_t rj = C.col[j];
                    S[(ri >= rj ? rj * d.ldS + ri : ri * d.ldS + rj)] = g;
                    if (i == j) diagU[ri] = g;
                }
            } else if (j == NCX) {
                if (i >= 6) iop[81 + (i - 6)] = g;
                else { g_c[ri] = g; g_red[ri] = g; }
            }
        }
    }
}

// The IO unknowns' block: one workgroup per element (p >= q) of the nio x nio block and per gradient entry.  Thread t adds
// the cameras t, t + 256, ... in index order (a camera contributes where both columns are on its list), the 256 partial
// sums are joined by a fixed tree: the same bits whatever the hardware does.
template <int NCX>
__global__ __launch_bounds__(256) void k_det_io_reduce(DevProblem d, const CamRec *__restrict__ cams, int nio, double *__restrict__ S,
                                                       double *__restrict__ g_c, double *__restrict__ g_red,
                                                       double *__restrict__ diagU) {
    __shared__ double sh[256];
    const int npair = nio * (nio + 1) / 2, e = blockIdx.x, t = threadIdx.x;
    int p, q;
    if (e < npair) { p = 0; while ((p + 1) * (p + 2) / 2 <= e) ++p; q = e - p * (p + 1) / 2; }
    else { p = e - npair; q = -1; }
    const int64_t rp = 6 * (int64_t)d.nc + p, rq = 6 * (int64_t)d.nc + q;
    double s = 0.0;
    for (int c = t; c < d.nc; c += 256) {
        const CamRec &C = cams[c];
        const int ncol = min(C.ncol, NCX);
        int a = -1, b = -1;
        for (int k = 6; k < ncol; ++k) { if (C.col[k] == rp) a = k; if (C.col[k] == rq) b = k; }
        if (a < 0) continue;
        const double *iop = d.det_io_part + (int64_t)c * DET_IOP;
        if (q < 0) s += iop[81 + (a - 6)];
        else if (b >= 0) s += iop[(max(a, b) - 6) * 9 + (min(a, b) - 6)];
    }
    sh[t] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (t < w) sh[t] += sh[t + w];
        __syncthreads();
    }
    if (t == 0) {
        if (q < 0) { g_c[rp] = sh[0]; g_red[rp] = sh[0]; }
        else { S[rq * d.ldS + rp] = sh[0]; if (p == q) diagU[rp] = sh[0]; }
    }
}

// u_r of every row of the reduced system from its squared column norm, and u_f from r'r (summed in camera order by one
// thread: nc additions); then the camera-side entries of S and of g_red onto their grids, so that every later addition to
// them is exact (one thread per camera element, as in k_det_cam_reduce; block nc: the IO x IO block).
// Deterministic mode, the bound of the right-hand side: the gradient that enters -(W V^-1) g_p carries the weighted
// residuals of the PRIOR observations beside those of the image observations (g_p += pw (z - prior)), so
// u_f^2 >= r'r + sum pw (z - prior)^2 (with r'r of the image rows alone, prior offsets a few times the image
// residual norm broke |sum| <= u_i u_f and with it the exactness of the sums).  Block b sums its slice of z in a fixed
// order into det_rr[nc + 1 + b]; k_det_rows adds the DET_PRIOR_PARTS slices.
constexpr int DET_PRIOR_PARTS = 256;
__global__ __launch_bounds__(256) void k_det_prior_sq(DevProblem d, const double *__restrict__ z) {
    __shared__ double sh[256];
    const int t = threadIdx.x;
    const int64_t per = (d.NZ + DET_PRIOR_PARTS - 1) / DET_PRIOR_PARTS;
    const int64_t lo = per * blockIdx.x, hi = lo + per < d.NZ ? lo + per : d.NZ;
    double s = 0.0;
    for (int64_t i = lo + t; i < hi; i += 256) {
        const double pw = d.z_prw[i];
        if (pw > 0) { const double e = z[i] - d.z_prv[i]; s += pw * e * e; }
    }
    sh[t] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (t < w) sh[t] += sh[t + w];
        __syncthreads();
    }
    if (t == 0) d.det_rr[d.nc + 1 + blockIdx.x] = sh[0];
}
__global__ __launch_bounds__(256) void k_det_rows(DevProblem d, const double *__restrict__ diagU) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < d.NS) d.det_u[r] = det_pow2_sqrt(diagU[r]);
    if (blockIdx.x == 0) {                          // r'r: thread t adds the cameras t, t + 256, ...; fixed tree over the threads
        __shared__ double sh[256];
        const int t = threadIdx.x;
        double s = d.any_prior ? d.det_rr[d.nc + 1 + t] : 0.0;      // (+ the prior rows' squares: k_det_prior_sq)
        for (int c = t; c < d.nc; c += 256) s += d.det_rr[c];
        sh[t] = s;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if (t < w) sh[t] += sh[t + w];
            __syncthreads();
        }
        if (t == 0) { d.det_rr[d.nc] = sh[0]; d.det_u[d.NS] = det_pow2_sqrt(sh[0]); }
    }
}
template <int NCX>
__global__ __launch_bounds__(256) void k_det_round_cam(DevProblem d, const CamRec *__restrict__ cams, int nio, double *__restrict__ S,
                                                       double *__restrict__ g_red) {
    const int t = threadIdx.x;
    const double *u = d.det_u, uf = d.det_u[d.NS];
    if ((int)blockIdx.x == d.nc) {                   // the IO x IO block and the IO entries of the right-hand side
        const int npair = nio * (nio + 1) / 2;
        for (int e = t; e < npair + nio; e += blockDim.x) {
            if (e < npair) {
                int p = 0; while ((p + 1) * (p + 2) / 2 <= e) ++p;
                const int q = e - p * (p + 1) / 2;
                const int64_t rp = 6 * (int64_t)d.nc + p, rq = 6 * (int64_t)d.nc + q;
                S[rq * d.ldS + rp] = det_round(S[rq * d.ldS + rp], u[rp], u[rq]);
            } else {
                const int64_t rp = 6 * (int64_t)d.nc + (e - npair);
                g_red[rp] = det_round(g_red[rp], u[rp], uf);
            }
        }
        return;
    }
    const CamRec &C = cams[blockIdx.x];
    const int ncol = NCX == 6 ? 6 : min(C.ncol, NCX);
    const int i = t >> 4, j = t & 15;
    if (i < ncol && j < 6 && i >= j) {               // the camera's own block and its rows of the camera x IO block
        const int64_t ri = C.col[i], rj = C.col[j];
        double *e = S + (ri >= rj ? rj * d.ldS + ri : ri * d.ldS + rj);
        *e = det_round(*e, u[ri], u[rj]);
    }
    if (i < 6 && j == 15) { const int64_t ri = C.col[i]; g_red[ri] = det_round(g_red[ri], u[ri], uf); }
}

// ---------------------------------------------------------------- K1t2 --
// Wave-specialised tile kernel for the fixed-IO path: 512 threads.  Waves 0-3
// ("producers") evaluate the observations of the tile's batches exactly as
// k_build_tile does (P1-P3, without the camera-side products J_c'J_c and
// J_c'r: those are k_cam_normal's) and scatter the Y / W fragments of chunks of
// TILE2_PC points into one of TWO operand panels in LDS; waves 4-7
// ("consumers") run the 128 x 128 x 3*PC contraction of a filled panel on the
// f64 matrix cores and hand the panel back.  Producer w and consumer w+4 share
// a SIMD, and the hand-over of batch b-1's chunks is interleaved with P1, P2
// and P3 of batch b (its W and V^-1 stay in registers meanwhile), so the matrix
// pipe works under the residual/Jacobian arithmetic, the global-load latencies
// and the LDS atomics instead of after them.
// Hand-over is by LDS counters (full/done/freed per panel): the producers fill
// a zeroed panel and signal `full`; a consumer wave signals `done` after its
// products, waits for the other three, zeroes its quarter of the panel and
// signals `freed`.  With four panels in flight the producers of batch b+1 run
// under the matrix work of batch b.  The producers synchronise among themselves
// (P1 -> P2 -> P3) with a counter barrier; the hardware s_barrier is only used
// where all eight waves take part.
// Every spin has a cap that poisons the objective value instead of hanging.
constexpr int TILE2_PC = 16;                     // fixed IO; self-calibration: 8 (LDS also holds the IO blocks)
constexpr int TILE2_NBUF = 2;                    // operand panels in flight between producers and consumers

// sum_p Y_p W_p' with Y = W V^-1 is the symmetric product Z Z' for Z = W R, V^-1 = R R'
// (R = lower Cholesky factor of the 3x3 inverse): ONE operand panel instead of two.  Wave w
// owns row tiles w and 7-w of the lower triangle; both its A fragments are among the B
// fragments it loads anyway.
template <int WV>
__device__ __forceinline__ void tile_syrk_steps(const double *Zt, int lane, int ksteps, mfma_d4 (&acc)[9]) {
    constexpr int LD = TILE_LD, RA = WV, RB = 7 - WV;      // RA <= RB
    if (ksteps <= 0) return;
    const double *zr = Zt + (lane >> 4) * LD + (lane & 15);
    double wa[RB + 1];
#pragma unroll
    for (int c = 0; c <= RB; ++c) wa[c] = zr[16 * c];
    for (int kk = 0; kk < ksteps; ++kk) {
        const int kn = kk + 1 < ksteps ? kk + 1 : kk;      // last step re-reads itself (harmless)
        const double *zn = zr + 4 * kn * LD;
        double nwa[RB + 1];
#pragma unroll
        for (int c = 0; c <= RB; ++c) nwa[c] = zn[16 * c];
#pragma unroll
        for (int c = 0; c <= RA; ++c)
            acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[RA], wa[c], acc[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c <= RB; ++c)
            acc[RA + 1 + c] = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[RB], wa[c], acc[RA + 1 + c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c <= RB; ++c) wa[c] = nwa[c];
    }
}
constexpr int TILE2_SPIN_CAP = 1 << 24;

__device__ unsigned long long g_tile2_prof[16];    // DBAT_HIP_ABLATE & 32: phase times (10 ns ticks), summed over tiles

struct Tile2Sync { int full[4], done[4], freed[4], ks[4], pbar, abort_, npts[2]; };

__device__ __forceinline__ bool lds_wait_ge(int *cnt, int target, int *abort_) {
    int spins = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
        __builtin_amdgcn_s_sleep(0);
        if ((++spins & 1023) == 0) {
            if (spins > TILE2_SPIN_CAP) __hip_atomic_store(abort_, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (__hip_atomic_load(abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return false;
        }
    }
    asm volatile("" ::: "memory");
    return true;
}
__device__ __forceinline__ void lds_signal(int *cnt) {   // one count per wave
    lds_fence();
    if ((threadIdx.x & 63) == 0) atomicAdd(cnt, 1);
}

// NCX = 6: fixed IO.  NCX > 6: self-calibration -- the estimated IO columns of the tile's
// cameras are extra rows of the tile (as in k_build_tile); an IO row is shared by the
// observations of a point, so its panel entries are summed with LDS atomics into the
// zeroed panel.  PC points per chunk, NBUF panels.
template <int MODEL, int NCX, int PC, int NBUF>
__global__ __launch_bounds__(512) void k_build_tile2(DevProblem d, const double *__restrict__ z,
                                                     const CamRec *__restrict__ cams, double lambda, int scale,
                                                     double *__restrict__ S, double *__restrict__ g_c,
                                                     double *__restrict__ g_red, double *__restrict__ diagU,
                                                     double *__restrict__ Vinv, double *__restrict__ gp,
                                                     double *__restrict__ jn2p,
                                                     double *__restrict__ partial,
                                                     unsigned long long *__restrict__ pivmm) {
    constexpr int KC = 3 * PC, LD = TILE_LD, PANEL = KC * LD;
    constexpr bool IO = NCX > 6;
    constexpr int NPROD = 256;                       // producer threads = batch size
    extern __shared__ double smem[];
    double *pan = smem;                              // [NBUF][KC*LD]  Z = W R panels
    double *red = pan + NBUF * PANEL;                // [2][NPROD/2][9]  sum of B'B | B'r per point, even/odd batch
    double *pinv = red + (size_t)NPROD * 9;          // [NPROD/2][15] V^-1 | g_p | R per point of the batch
    double *vt = pinv + (size_t)(NPROD / 2) * 15;    // [LD]  -(W V^-1 g_p) by local row (the rest of the
                                                     // camera side, J_c'J_c and J_c'r, is k_cam_normal's)
    __shared__ double sh[16];
    __shared__ Tile2Sync sy;
    __shared__ int64_t bs_sh[64];                    // batch_start of this tile's first 64 batches (a tile is capped at 48)
    const int t = threadIdx.x, lane = t & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(t >> 6);
    const bool producer = wave8 < 4;
    const int wave = wave8 & 3;
    const int tile = d.tile_order[blockIdx.x];
    const int b0 = d.tile_batch[tile], b1 = d.tile_batch[tile + 1];
    const int c0 = d.tile_cam_start[tile];
    const int ncam = d.tile_cam_start[tile + 1] - c0;
    const int io0 = IO ? d.tile_io_start[tile] : 0;
    const int nio = IO ? d.tile_io_start[tile + 1] - io0 : 0;
    const int iobase = 6 * ncam;                     // first IO row of the tile-local system
    const int nrows = iobase + nio;
    for (int i = t; i <= b1 - b0 && i < 64; i += 512) bs_sh[i] = d.batch_start[b0 + i];
    for (int i = t; i < NBUF * PANEL + NPROD * 9; i += 512) pan[i] = 0.0;     // panels and the point sums
    for (int i = t; i < LD; i += 512) vt[i] = 0.0;
    if (t < NBUF) { sy.full[t] = 0; sy.done[t] = 0; sy.freed[t] = 0; sy.ks[t] = 0; }
    if (t == 0) { sy.pbar = 0; sy.abort_ = 0; sy.npts[0] = sy.npts[1] = 0; }
    mfma_d4 acc[9];
#pragma unroll
    for (int s = 0; s < 9; ++s) acc[s] = mfma_d4{0, 0, 0, 0};
    double pmin = 1e300, pmax = 0.0, rr = 0.0;
    __syncthreads();
    const bool prof = DBAT_ABLATE(d, 32) && lane == 0 && wave == 0;
    long long tp[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = prof ? wall_clock64() : 0;
    auto lap = [&](int i) { if (prof) { const long long now = wall_clock64(); tp[i] += now - tlast; tlast = now; } };
    if (producer) {
        int nchunk = 0;                              // chunks handed over so far (panel = nchunk % NBUF)
        int pbar_gen = 0;
        bool ok = true;
        auto bstart = [&](int bb) -> int64_t { return bb - b0 < 64 ? bs_sh[bb - b0] : d.batch_start[bb]; };
        auto pbarrier = [&]() -> bool {
            lds_signal(&sy.pbar); ++pbar_gen;
            return lds_wait_ge(&sy.pbar, 4 * pbar_gen, &sy.abort_);
        };
        // observation header of the next batch and the object point of its observations are
        // fetched one batch ahead, so their HBM latency hides behind the current batch
        int hn_cam = 0, hn_pt = 0, hn_lc = 0, hn_pidx = 0; uint32_t hn_sg = 0; double hn_u = 0, hn_v = 0;
        double qn[3] = {0, 0, 0}, pwn[3] = {0, 0, 0};
        unsigned estr0 = 0, estr1 = 0, estr2 = 0;   // raw flags: combining them here would wait for the loads
        auto fetch_header = [&](int bb) {
            if (bb >= b1) return;
            const int64_t oo0 = bstart(bb);
            if (t < (int)(bstart(bb + 1) - oo0)) {
                const int64_t oo = oo0 + t;
                hn_cam = d.o_cam[oo]; hn_pt = d.o_pt[oo]; hn_lc = d.o_lc[oo]; hn_pidx = d.o_pidx[oo];
                hn_sg = d.o_seg[oo]; hn_u = d.o_uv[2 * oo]; hn_v = d.o_uv[2 * oo + 1];
            }
        };
        auto fetch_point = [&](int bb) {
            if (bb >= b1) return;
            if (t < (int)(bstart(bb + 1) - bstart(bb))) {
                const int64_t zp = d.NS + 3 * (int64_t)hn_pt;
                qn[0] = z[zp]; qn[1] = z[zp + 1]; qn[2] = z[zp + 2];
                estr0 = d.z_est[zp]; estr1 = d.z_est[zp + 1]; estr2 = d.z_est[zp + 2];
                if (t == (int)(hn_sg & 0xFFFF)) { pwn[0] = d.z_prw[zp]; pwn[1] = d.z_prw[zp + 1]; pwn[2] = d.z_prw[zp + 2]; }
            }
        };
        fetch_header(b0);
        fetch_point(b0);
        for (int b = b0; b < b1 && ok; ++b) {
            const int64_t o0 = bstart(b);
            const int nobs = (int)(bstart(b + 1) - o0);
            const bool active = t < nobs;
            const int64_t o = o0 + t;
            int *npts_sh = &sy.npts[b & 1];
            double r[2] = {0, 0};
            const int hn_lc_cur = hn_lc;
            double E[2][NCX];
            double B[2][3];
            int ncol = 6;
            uint32_t ciop[4] = {0, 0, 0, 0};         // local IO rows of this camera's IO columns (1 byte each)
            auto lrow = [&](int a) -> int {          // tile-local row of camera-side column a
                return a < 6 ? 6 * hn_lc_cur + a : iobase + (int)((ciop[(a - 6) >> 2] >> (8 * ((a - 6) & 3))) & 255u);
            };
            const int cam = hn_cam, pt = hn_pt, lc = hn_lc, pidx = hn_pidx;
            const int seg_start = hn_sg & 0xFFFF;
            const double uu = hn_u, vv = hn_v;
            const double Q[3] = {qn[0], qn[1], qn[2]};
            const double pw3[3] = {pwn[0], pwn[1], pwn[2]};
            const unsigned est = (estr0 ? 1u : 0u) | (estr1 ? 2u : 0u) | (estr2 ? 4u : 0u);
            if (t == 0) *npts_sh = 0;
            fetch_header(b + 1);
            lap(9);
            if (active) {                            // ---- P1
                const CamRec &C = cams[cam];
                const double w0 = d.o_w ? d.o_w[2 * o] : C.w[0], w1 = d.o_w ? d.o_w[2 * o + 1] : C.w[1];
                if constexpr (IO) {
                    ncol = min(C.ncol, NCX);
                    const uint32_t *cp = (const uint32_t *)(d.tile_cam_io + (size_t)(c0 + lc) * 16);
                    ciop[0] = cp[0]; ciop[1] = cp[1]; ciop[2] = cp[2]; ciop[3] = cp[3];
                }
                eval_obs_pre<MODEL, NCX>(d, C, Q, uu, vv, w0, w1, est, r, E, B);
                rr = fma2(rr, r[0], r[0], r[1], r[1]);
                // per-point sums of B'B and B'r with LDS atomics (the LDS unit, not the FP64 pipe the
                // matrix work of the consumer wave on this SIMD is using)
                double *ps = red + (size_t)(b & 1) * (NPROD / 2) * 9 + (size_t)pidx * 9;
                atomic_add_f64(ps + 0, B[0][0] * B[0][0] + B[1][0] * B[1][0]);
                atomic_add_f64(ps + 1, B[0][0] * B[0][1] + B[1][0] * B[1][1]);
                atomic_add_f64(ps + 2, B[0][0] * B[0][2] + B[1][0] * B[1][2]);
                atomic_add_f64(ps + 3, B[0][1] * B[0][1] + B[1][1] * B[1][1]);
                atomic_add_f64(ps + 4, B[0][1] * B[0][2] + B[1][1] * B[1][2]);
                atomic_add_f64(ps + 5, B[0][2] * B[0][2] + B[1][2] * B[1][2]);
                atomic_add_f64(ps + 6, B[0][0] * r[0] + B[1][0] * r[1]);
                atomic_add_f64(ps + 7, B[0][1] * r[0] + B[1][1] * r[1]);
                atomic_add_f64(ps + 8, B[0][2] * r[0] + B[1][2] * r[1]);
            }
            fetch_point(b + 1);
            lap(1);
            if (!pbarrier()) { ok = false; break; }
            lap(2);
            {   // the sums of the next batch start from zero: clear the other buffer now (its last readers,
                // the leaders of batch b-1, finished before that batch's second barrier)
                double *pz = red + (size_t)((b + 1) & 1) * (NPROD / 2) * 9;
                for (int i = t; i < (NPROD / 2) * 9; i += NPROD) pz[i] = 0.0;
            }
            if (active && t == seg_start) {          // ---- P2
                atomicMax(npts_sh, pidx + 1);
                const double *ps = red + (size_t)(b & 1) * (NPROD / 2) * 9 + (size_t)pidx * 9;
                double V[6] = {ps[0], ps[1], ps[2], ps[3], ps[4], ps[5]}, g[3] = {ps[6], ps[7], ps[8]};
                const int64_t zp = d.NS + 3 * (int64_t)pt;
                const int dix[3] = {0, 3, 5};
                double jn[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const double pw = pw3[k];
                    if (pw > 0) { V[dix[k]] += pw; g[k] += pw * (Q[k] - d.z_prv[zp + k]); }
                    jn[k] = V[dix[k]];
                    jn2p[3 * (int64_t)pt + k] = jn[k];
                    if ((est >> k) & 1u) V[dix[k]] += lambda; else V[dix[k]] = 1.0;
                }
                double inv[6], Rpb[6];
                {   // the block's factor (point_block_factor) and the SQUARED pivots of chol(V) (the square roots are
                    // taken once per wave at the end of the kernel)
                    point_block_factor(V, Rpb, inv);
                    const double r0 = fast_rcp(V[0]);
                    const double d1s = V[3] - V[1] * V[1] * r0;
                    const double tt = V[4] - V[2] * V[1] * r0;
                    const double d2s = V[5] - V[2] * V[2] * r0 - tt * tt * fast_rcp(d1s);
                    const double dd[3] = {V[0], d1s, d2s};
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                        if ((est >> k) & 1u) {
                            double v = scale ? dd[k] * fast_rcp(jn[k]) : dd[k];
                            v = v > 0.0 ? v : 0.0;                  // NaN or negative pivot => 0
                            pmin = fmin(pmin, v); pmax = fmax(pmax, v);
                        }
                }
                double *pi = pinv + (size_t)pidx * 15;
#pragma unroll
                for (int k = 0; k < 6; ++k) { pi[k] = Rpb[k]; Vinv[6 * (int64_t)pt + k] = Rpb[k]; }
#pragma unroll
                for (int k = 0; k < 3; ++k) gp[3 * (int64_t)pt + k] = g[k];
                pi[6] = Rpb[0] * g[0] + Rpb[1] * g[1] + Rpb[2] * g[2];      // y = R' g
                pi[7] = Rpb[3] * g[1] + Rpb[4] * g[2];
                pi[8] = Rpb[5] * g[2];
                {   // V^-1 = R R', R lower triangular
                    const double r00 = Rpb[0], r10 = Rpb[1], r20 = Rpb[2], r11 = Rpb[3], r21 = Rpb[4], r22 = Rpb[5];
                    pi[9] = r00; pi[10] = r10; pi[11] = r20; pi[12] = r11; pi[13] = r21; pi[14] = r22;
                }
            }
            lap(3);
            if (!pbarrier()) { ok = false; break; }
            lap(2);
            const int npts = *npts_sh;
            double v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0, v5 = 0;
            double nr0 = 0, nr1 = 0, nr2 = 0, nr3 = 0, nr4 = 0, nr5 = 0;
            double g0 = 0, g1 = 0, g2 = 0;
            // Self-calibration: the IO rows of a shared camera are hit by every observation.  If all
            // lanes of the wave map their IO columns to the same tile rows (one camera / IO block --
            // the usual case), the IO x IO block and the IO entries of the vectors are summed over the
            // wave in registers and added once; otherwise every lane uses LDS atomics.
            bool io_uniform = false;
            if constexpr (IO) {
                const uint32_t c0f = __builtin_amdgcn_readfirstlane(ciop[0]), c1f = __builtin_amdgcn_readfirstlane(ciop[1]);
                const uint32_t c2f = __builtin_amdgcn_readfirstlane(ciop[2]), c3f = __builtin_amdgcn_readfirstlane(ciop[3]);
                const int ncf = __builtin_amdgcn_readfirstlane(ncol);
                io_uniform = __all(!active || (ciop[0] == c0f && ciop[1] == c1f && ciop[2] == c2f && ciop[3] == c3f && ncol == ncf)) &&
                             __builtin_amdgcn_readfirstlane(active ? 1 : 0) == 1;
                if (io_uniform) { ciop[0] = c0f; ciop[1] = c1f; ciop[2] = c2f; ciop[3] = c3f; }   // idle lanes too: uniform row lookups
            }
            if (active) {                            // ---- P3: E'E, gradient pieces
                const double *pi = pinv + (size_t)pidx * 15;
                v0 = pi[0]; v1 = pi[1]; v2 = pi[2]; v3 = pi[3]; v4 = pi[4]; v5 = pi[5];
                nr0 = pi[9]; nr1 = pi[10]; nr2 = pi[11]; nr3 = pi[12]; nr4 = pi[13]; nr5 = pi[14];
                g0 = pi[6]; g1 = pi[7]; g2 = pi[8];
#pragma unroll
                for (int a = 0; a < NCX; ++a) {
                    if (a >= ncol || DBAT_ABLATE(d, 2)) continue;       // ablate: profiling only
                    if (IO && a >= 6 && io_uniform) continue;        // summed over the wave below
                    const double w0 = E[0][a] * B[0][0] + E[1][a] * B[1][0];
                    const double w1 = E[0][a] * B[0][1] + E[1][a] * B[1][1];
                    const double w2 = E[0][a] * B[0][2] + E[1][a] * B[1][2];
                    const double y0 = w0 * v0 + w1 * v1 + w2 * v2;               // z = w R, and -(W V^-1 g) = -z (R' g)
                    const double y1 = w1 * v3 + w2 * v4;
                    const double y2 = w2 * v5;
                    atomic_add_f64(vt + lrow(a), -(y0 * g0 + y1 * g1 + y2 * g2));
                }
            }
            if constexpr (IO) {
                if (io_uniform && !DBAT_ABLATE(d, 2)) {
                    const int ncw = __builtin_amdgcn_readfirstlane(ncol);
#pragma unroll
                    for (int a = 6; a < NCX; ++a) {
                        if (a >= ncw) break;                         // wave-uniform
                        // idle lanes hold undefined E and B: they must contribute exact zeros
                        double gr = 0.0;
                        if (active) {
                            const double e0 = E[0][a], e1 = E[1][a];
                            const double w0 = e0 * B[0][0] + e1 * B[1][0];
                            const double w1 = e0 * B[0][1] + e1 * B[1][1];
                            const double w2 = e0 * B[0][2] + e1 * B[1][2];
                            const double y0 = w0 * v0 + w1 * v1 + w2 * v2;
                            const double y1 = w1 * v3 + w2 * v4;
                            const double y2 = w2 * v5;
                            gr = -(y0 * g0 + y1 * g1 + y2 * g2);
                        }
                        const double s_r = wave_sum_f64(gr);
                        if (lane == 0) atomic_add_f64(vt + lrow(a), s_r);
                    }
                }
            }
            lap(4);
            // ---- P4: Z = W R of this batch, chunk by chunk, into panels the consumers have
            // released AND zeroed: no restore pass, no barrier between the producers
            for (int p0 = 0; p0 < npts && ok; p0 += PC, ++nchunk) {
                const int s = nchunk % NBUF, u = nchunk / NBUF;
                if (u > 0 && !lds_wait_ge(&sy.freed[s], 4 * u, &sy.abort_)) { ok = false; break; }
                lap(0);
                double *Zt = pan + s * PANEL;
                if (active && pidx >= p0 && pidx < p0 + PC) {
                    const int kb = 3 * (pidx - p0);
#pragma unroll
                    for (int a = 0; a < NCX; ++a) {
                        if (a >= ncol) continue;
                        const int row = lrow(a);
                        const double w0 = E[0][a] * B[0][0] + E[1][a] * B[1][0];
                        const double w1 = E[0][a] * B[0][1] + E[1][a] * B[1][1];
                        const double w2 = E[0][a] * B[0][2] + E[1][a] * B[1][2];
                        const double z0 = w0 * nr0 + w1 * nr1 + w2 * nr2, z1 = w1 * nr3 + w2 * nr4, z2 = w2 * nr5;
                        if (a < 6) {                 // a camera row belongs to one observation of the point
                            Zt[(kb + 0) * LD + row] = z0; Zt[(kb + 1) * LD + row] = z1; Zt[(kb + 2) * LD + row] = z2;
                        } else {                     // an IO row is shared by the point's observations: sum
                            atomic_add_f64(Zt + (kb + 0) * LD + row, z0); atomic_add_f64(Zt + (kb + 1) * LD + row, z1);
                            atomic_add_f64(Zt + (kb + 2) * LD + row, z2);
                        }
                    }
                }
                if (t == 0) sy.ks[s] = (3 * min(PC, npts - p0) + 3) >> 2;
                lds_signal(&sy.full[s]);
                lap(5);
            }
            if (!ok) break;
        }
        {   // terminating chunk
            const int s = nchunk % NBUF, u = nchunk / NBUF;
            if (ok && u > 0) ok = lds_wait_ge(&sy.freed[s], 4 * u, &sy.abort_);
            if (t == 0) sy.ks[s] = -1;
            lds_signal(&sy.full[s]);
        }
        if (!ok) rr = __longlong_as_double(0x7ff8000000000000ll);      // poison the objective value
    } else {
        // ---- consumers: this wave's nine 16x16 tiles of the lower triangle
        const bool full_tile = 16 * (7 - wave) < nrows;
        int yoff[9], woff[9];
        bool ton[9];
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            const int rt = s <= wave ? wave : 7 - wave;
            const int ct = s <= wave ? s : s - wave - 1;
            yoff[s] = 16 * rt; woff[s] = 16 * ct; ton[s] = 16 * rt < nrows;
        }
        for (int n = 0;; ++n) {
            const int s = n % NBUF, u = n / NBUF;
            if (!lds_wait_ge(&sy.full[s], 4 * (u + 1), &sy.abort_)) break;
            lap(6);
            int ksteps = __hip_atomic_load(&sy.ks[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (ksteps < 0) break;
            if DBAT_ABLATE(d, 8) ksteps = 0;                // profiling: hand-over without the matrix work
            const double *Zt = pan + s * PANEL;
            if (full_tile) {
                switch (wave) {
                    case 0: tile_syrk_steps<0>(Zt, lane, ksteps, acc); break;
                    case 1: tile_syrk_steps<1>(Zt, lane, ksteps, acc); break;
                    case 2: tile_syrk_steps<2>(Zt, lane, ksteps, acc); break;
                    default: tile_syrk_steps<3>(Zt, lane, ksteps, acc); break;
                }
            } else {
                for (int kk = 0; kk < ksteps; ++kk) {
                    const int krow = 4 * kk + (lane >> 4);
                    const double *yr = Zt + krow * LD + (lane & 15);
                    const double *wr = yr;
#pragma unroll
                    for (int q = 0; q < 9; ++q)
                        if (ton[q])
                            acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(yr[yoff[q]], wr[woff[q]], acc[q], 0, 0, 0);
                }
            }
            lap(10);
            // every consumer wave has read the panel -> zero it (a quarter per wave) -> hand it back
            lds_signal(&sy.done[s]);
            if (!lds_wait_ge(&sy.done[s], 4 * (u + 1), &sy.abort_)) break;
            lap(11);
            {   // 16 bytes per lane and instruction: half as many trips through the LDS queue
                static_assert((PANEL / 4) % 2 == 0 && (PANEL * sizeof(double)) % 64 == 0, "panel quarters are 16-byte aligned");
                double2 *Zw = reinterpret_cast<double2 *>(pan + s * PANEL + wave * (PANEL / 4));
                for (int i = lane; i < PANEL / 8; i += 64) Zw[i] = double2{0.0, 0.0};
            }
            lds_signal(&sy.freed[s]);
            lap(7);
        }
        // ---- flush this wave's part of the tile:  S -= sum_p Z Z'
        auto grow = [&](int lr) -> int64_t {         // tile-local row -> row of the reduced system
            if (lr < iobase) return 6 * (int64_t)d.tile_cams[c0 + lr / 6] + lr % 6;
            return 6 * (int64_t)d.nc + d.tile_iocols[io0 + lr - iobase];
        };
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            if (ton[s]) {
                const int lcol = woff[s] + (lane & 15);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int lr = yoff[s] + (lane >> 4) + 4 * e;
                    const double v = acc[s][e];
                    if (lr < nrows && lcol <= lr && v != 0.0)
                        atomic_add_f64(S + grow(lcol) * d.ldS + grow(lr), -v);
                }
            }
        }
    }
    lap(8);
    __syncthreads();
    if (prof) {
        const int base = producer ? 0 : 0;
#pragma unroll
        for (int i = 0; i < 12; ++i) if (tp[i]) atomicAdd(&g_tile2_prof[base + i], (unsigned long long)tp[i]);
    }
    // ---- reduced right-hand side: -(W V^-1 g_p) of this tile (all 512 threads)
    for (int i = t; i < nrows; i += 512) {
        const int64_t col = i < iobase ? 6 * (int64_t)d.tile_cams[c0 + i / 6] + i % 6
                                       : 6 * (int64_t)d.nc + d.tile_iocols[io0 + i - iobase];
        atomic_add_f64(g_red + col, vt[i]);
    }
    double accr[1] = {rr};
    block_sum<1>(accr, sh);
    if (t == 0) partial[blockIdx.x] = accr[0];
    pmin = pmin < 1e300 ? sqrt(pmin) : pmin; pmax = sqrt(pmax);   // squared pivots were tracked
    for (int off = 32; off > 0; off >>= 1) {
        pmin = fmin(pmin, __shfl_down(pmin, off, 64));
        pmax = fmax(pmax, __shfl_down(pmax, off, 64));
    }
    if ((t & 63) == 0 && pmax > 0.0) {
        atomicMin(pivmm, (unsigned long long)__double_as_longlong(pmin));
        atomicMax(pivmm + 1, (unsigned long long)__double_as_longlong(pmax));
    }
}

// ---------------------------------------------------------------- K1t3 --
// Fixed-IO variant of k_build_tile2 with TWO producer groups (waves 0-3 and 4-7) in front of
// the four consumer waves (8-11): group g evaluates the batches b0+g, b0+g+2, ... of the tile,
// so two batches are in flight per CU and the latency chain of one (camera gathers, the
// evaluation, the 3x3 inverses) hides behind the other.  The chunk numbers are fixed in
// advance -- chunk0[b] = chunks of the tile's earlier batches, from the point counts of the
// batches -- so both groups fill the same ring of NBUF panels in a fixed order and the consumers
// take the chunks in that order.  Per group: its own point sums (cleared by the leaders that read
// them), V^-1 g | R records and barrier counter.
#ifndef DBAT_TILE3_PC
#define DBAT_TILE3_PC 16
#endif
#ifndef DBAT_TILE3_NBUF
#define DBAT_TILE3_NBUF 2
#endif
constexpr int TILE3_PC = DBAT_TILE3_PC, TILE3_NBUF = DBAT_TILE3_NBUF;   // points per chunk, panels in the ring
struct Tile3Sync { int full[4], done[4], freed[4], ks[4], pbar[2], abort_, chunk0[65], npts[65]; };

template <int MODEL, int PC, int NBUF>
__global__ __launch_bounds__(768) void k_build_tile3(DevProblem d, const double *__restrict__ z,
                                                     const CamRec *__restrict__ cams, double lambda, int scale,
                                                     double *__restrict__ S, double *__restrict__ g_red,
                                                     double *__restrict__ Vinv, double *__restrict__ gp,
                                                     double *__restrict__ jn2p,
                                                     double *__restrict__ partial,
                                                     unsigned long long *__restrict__ pivmm) {
    constexpr int NCX = 6;
    constexpr int KC = 3 * PC, LD = TILE_LD, PANEL = KC * LD;
    constexpr int NPROD = 256, NT = 768, PW = 9;     // PW: V^-1 g (3) | R (6) per point
    extern __shared__ double smem[];
    double *pan = smem;                              // [NBUF][KC*LD]  Z = W R panels
    double *red = pan + NBUF * PANEL;                // [2 groups][NPROD/2][9]  sum of B'B | B'r per point
    double *pinv = red + (size_t)NPROD * 9;          // [2 groups][NPROD/2][PW]
    double *vt = pinv + (size_t)NPROD * PW;          // [LD]  -(W V^-1 g_p) by local row
    __shared__ double sh[16];
    __shared__ Tile3Sync sy;
    __shared__ int64_t bs_sh[66];
    const int t = threadIdx.x, lane = t & 63;
    const int wave12 = __builtin_amdgcn_readfirstlane(t >> 6);
    const bool producer = wave12 < 8;
    const int grp = wave12 >> 2;                     // producer group 0/1 (consumers: 2)
    const int wave = wave12 & 3;
    const int tl = t & 255;                          // lane of the batch
    const int tile = d.tile_order[blockIdx.x];
    const int b0 = d.tile_batch[tile], b1 = d.tile_batch[tile + 1];
    const int c0 = d.tile_cam_start[tile];
    const int ncam = d.tile_cam_start[tile + 1] - c0;
    const int nrows = 6 * ncam;
    const int nbt = b1 - b0;                         // plan: at most 64 batches per tile for this kernel
    for (int i = t; i <= nbt && i <= 65; i += NT) bs_sh[i] = d.batch_start[b0 + i];
    for (int i = t; i < NBUF * PANEL + NPROD * 9; i += NT) pan[i] = 0.0;
    for (int i = t; i < LD; i += NT) vt[i] = 0.0;
    if (t < NBUF) { sy.full[t] = 0; sy.done[t] = 0; sy.freed[t] = 0; sy.ks[t] = 0; }
    if (t == 0) { sy.pbar[0] = sy.pbar[1] = 0; sy.abort_ = 0; }
    __syncthreads();
    if (t == 0) {                                    // chunk numbering of the tile
        int c = 0;
        for (int i = 0; i < nbt; ++i) {
            sy.chunk0[i] = c;
            const int npts_i = d.o_pidx[bs_sh[i + 1] - 1] + 1;      // points of the batch = last observation's point slot + 1
            sy.npts[i] = npts_i;
            c += (npts_i + PC - 1) / PC;
        }
        sy.chunk0[nbt] = c;
    }
    mfma_d4 acc[9];
#pragma unroll
    for (int s = 0; s < 9; ++s) acc[s] = mfma_d4{0, 0, 0, 0};
    double pmin = 1e300, pmax = 0.0, rr = 0.0;
    __syncthreads();
    if (producer) {
        double *redg = red + (size_t)grp * (NPROD / 2) * 9;
        double *pinvg = pinv + (size_t)grp * (NPROD / 2) * PW;
        int pbar_gen = 0;
        bool ok = true;
        auto pbarrier = [&]() -> bool {
            lds_signal(&sy.pbar[grp]); ++pbar_gen;
            return lds_wait_ge(&sy.pbar[grp], 4 * pbar_gen, &sy.abort_);
        };
        int hn_cam = 0, hn_pt = 0, hn_lc = 0, hn_pidx = 0; uint32_t hn_sg = 0; double hn_u = 0, hn_v = 0;
        double qn[3] = {0, 0, 0}, pwn[3] = {0, 0, 0};
        unsigned estr0 = 0, estr1 = 0, estr2 = 0;
        auto fetch_header = [&](int bb) {
            if (bb >= b1) return;
            const int64_t oo0 = bs_sh[bb - b0];
            if (tl < (int)(bs_sh[bb + 1 - b0] - oo0)) {
                const int64_t oo = oo0 + tl;
                hn_cam = d.o_cam[oo]; hn_pt = d.o_pt[oo]; hn_lc = d.o_lc[oo]; hn_pidx = d.o_pidx[oo];
                hn_sg = d.o_seg[oo]; hn_u = d.o_uv[2 * oo]; hn_v = d.o_uv[2 * oo + 1];
            }
        };
        auto fetch_point = [&](int bb) {
            if (bb >= b1) return;
            if (tl < (int)(bs_sh[bb + 1 - b0] - bs_sh[bb - b0])) {
                const int64_t zp = d.NS + 3 * (int64_t)hn_pt;
                qn[0] = z[zp]; qn[1] = z[zp + 1]; qn[2] = z[zp + 2];
                estr0 = d.z_est[zp]; estr1 = d.z_est[zp + 1]; estr2 = d.z_est[zp + 2];
                if (tl == (int)(hn_sg & 0xFFFF)) { pwn[0] = d.z_prw[zp]; pwn[1] = d.z_prw[zp + 1]; pwn[2] = d.z_prw[zp + 2]; }
            }
        };
        fetch_header(b0 + grp);
        fetch_point(b0 + grp);
        for (int b = b0 + grp; b < b1 && ok; b += 2) {
            const int64_t o0 = bs_sh[b - b0];
            const int nobs = (int)(bs_sh[b + 1 - b0] - o0);
            const bool active = tl < nobs;
            const int64_t o = o0 + tl;
            double r[2] = {0, 0};
            double E[2][NCX];
            double B[2][3];
            const int cam = hn_cam, pt = hn_pt, lc = hn_lc, pidx = hn_pidx;
            const int seg_start = hn_sg & 0xFFFF;
            const double uu = hn_u, vv = hn_v;
            const double Q[3] = {qn[0], qn[1], qn[2]};
            const double pw3[3] = {pwn[0], pwn[1], pwn[2]};
            const unsigned est = (estr0 ? 1u : 0u) | (estr1 ? 2u : 0u) | (estr2 ? 4u : 0u);
            const int npts = sy.npts[b - b0];
            fetch_header(b + 2);
            if (active) {                            // ---- P1
                const CamRec &C = cams[cam];
                const double w0 = d.o_w ? d.o_w[2 * o] : C.w[0], w1 = d.o_w ? d.o_w[2 * o + 1] : C.w[1];
                eval_obs_pre<MODEL, NCX>(d, C, Q, uu, vv, w0, w1, est, r, E, B);
                rr = fma2(rr, r[0], r[0], r[1], r[1]);
                double *ps = redg + (size_t)pidx * 9;
                atomic_add_f64(ps + 0, B[0][0] * B[0][0] + B[1][0] * B[1][0]);
                atomic_add_f64(ps + 1, B[0][0] * B[0][1] + B[1][0] * B[1][1]);
                atomic_add_f64(ps + 2, B[0][0] * B[0][2] + B[1][0] * B[1][2]);
                atomic_add_f64(ps + 3, B[0][1] * B[0][1] + B[1][1] * B[1][1]);
                atomic_add_f64(ps + 4, B[0][1] * B[0][2] + B[1][1] * B[1][2]);
                atomic_add_f64(ps + 5, B[0][2] * B[0][2] + B[1][2] * B[1][2]);
                atomic_add_f64(ps + 6, B[0][0] * r[0] + B[1][0] * r[1]);
                atomic_add_f64(ps + 7, B[0][1] * r[0] + B[1][1] * r[1]);
                atomic_add_f64(ps + 8, B[0][2] * r[0] + B[1][2] * r[1]);
            }
            fetch_point(b + 2);
            if (!pbarrier()) { ok = false; break; }
            if (active && tl == seg_start) {         // ---- P2
                double *ps = redg + (size_t)pidx * 9;
                double V[6] = {ps[0], ps[1], ps[2], ps[3], ps[4], ps[5]}, g[3] = {ps[6], ps[7], ps[8]};
#pragma unroll
                for (int k = 0; k < 9; ++k) ps[k] = 0.0;     // the group's next batch starts from zero
                const int64_t zp = d.NS + 3 * (int64_t)pt;
                const int dix[3] = {0, 3, 5};
                double jn[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const double pw = pw3[k];
                    if (pw > 0) { V[dix[k]] += pw; g[k] += pw * (Q[k] - d.z_prv[zp + k]); }
                    jn[k] = V[dix[k]];
                    jn2p[3 * (int64_t)pt + k] = jn[k];
                    if ((est >> k) & 1u) V[dix[k]] += lambda; else V[dix[k]] = 1.0;
                }
                double inv[6], Rpb[6];
                {
                    point_block_factor(V, Rpb, inv);
                    const double r0 = fast_rcp(V[0]);
                    const double d1s = V[3] - V[1] * V[1] * r0;
                    const double tt = V[4] - V[2] * V[1] * r0;
                    const double d2s = V[5] - V[2] * V[2] * r0 - tt * tt * fast_rcp(d1s);
                    const double dd[3] = {V[0], d1s, d2s};
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                        if ((est >> k) & 1u) {
                            double v = scale ? dd[k] * fast_rcp(jn[k]) : dd[k];
                            v = v > 0.0 ? v : 0.0;
                            pmin = fmin(pmin, v); pmax = fmax(pmax, v);
                        }
                }
                double *pi = pinvg + (size_t)pidx * PW;
#pragma unroll
                for (int k = 0; k < 6; ++k) Vinv[6 * (int64_t)pt + k] = Rpb[k];
#pragma unroll
                for (int k = 0; k < 3; ++k) gp[3 * (int64_t)pt + k] = g[k];
                {   // V^-1 = R R', R lower triangular; h = V^-1 g = R (R' g), through the factor (see point_block_factor)
                    const double r00 = Rpb[0], r10 = Rpb[1], r20 = Rpb[2], r11 = Rpb[3], r21 = Rpb[4], r22 = Rpb[5];
                    const double y0 = r00 * g[0] + r10 * g[1] + r20 * g[2], y1 = r11 * g[1] + r21 * g[2], y2 = r22 * g[2];
                    pi[0] = r00 * y0;
                    pi[1] = r10 * y0 + r11 * y1;
                    pi[2] = r20 * y0 + r21 * y1 + r22 * y2;
                    pi[3] = r00; pi[4] = r10; pi[5] = r20; pi[6] = r11; pi[7] = r21; pi[8] = r22;
                }
            }
            if (!pbarrier()) { ok = false; break; }
            double nr0 = 0, nr1 = 0, nr2 = 0, nr3 = 0, nr4 = 0, nr5 = 0;
            if (active) {                            // ---- P3: -(W V^-1 g)
                const double *pi = pinvg + (size_t)pidx * PW;
                const double h0 = pi[0], h1 = pi[1], h2 = pi[2];
                nr0 = pi[3]; nr1 = pi[4]; nr2 = pi[5]; nr3 = pi[6]; nr4 = pi[7]; nr5 = pi[8];
#pragma unroll
                for (int a = 0; a < NCX; ++a) {
                    const double w0 = E[0][a] * B[0][0] + E[1][a] * B[1][0];
                    const double w1 = E[0][a] * B[0][1] + E[1][a] * B[1][1];
                    const double w2 = E[0][a] * B[0][2] + E[1][a] * B[1][2];
                    atomic_add_f64(vt + 6 * lc + a, -(w0 * h0 + w1 * h1 + w2 * h2));
                }
            }
            // ---- P4: Z = W R, chunk by chunk, into the ring of panels in the tile's chunk order
            int nchunk = sy.chunk0[b - b0];
            for (int p0 = 0; p0 < npts && ok; p0 += PC, ++nchunk) {
                const int s = nchunk % NBUF, u = nchunk / NBUF;
                double *Zt = pan + s * PANEL;
                if (u > 0) {
                    // the panel's previous chunk has been multiplied by all four consumer waves:
                    // this group clears it (the consumers no longer do) and then fills it
                    if (!lds_wait_ge(&sy.freed[s], 4 * u, &sy.abort_)) { ok = false; break; }
                    double2 *Zw = reinterpret_cast<double2 *>(Zt);
                    for (int i = tl; i < PANEL / 2; i += NPROD) Zw[i] = double2{0.0, 0.0};
                    if (!pbarrier()) { ok = false; break; }
                }
                if (active && pidx >= p0 && pidx < p0 + PC) {
                    const int kb = 3 * (pidx - p0);
#pragma unroll
                    for (int a = 0; a < NCX; ++a) {
                        const int row = 6 * lc + a;
                        const double w0 = E[0][a] * B[0][0] + E[1][a] * B[1][0];
                        const double w1 = E[0][a] * B[0][1] + E[1][a] * B[1][1];
                        const double w2 = E[0][a] * B[0][2] + E[1][a] * B[1][2];
                        Zt[(kb + 0) * LD + row] = w0 * nr0 + w1 * nr1 + w2 * nr2;
                        Zt[(kb + 1) * LD + row] = w1 * nr3 + w2 * nr4;
                        Zt[(kb + 2) * LD + row] = w2 * nr5;
                    }
                }
                if (tl == 0) sy.ks[s] = (3 * min(PC, npts - p0) + 3) >> 2;
                lds_signal(&sy.full[s]);
            }
            if (!ok) break;
        }
        // terminating chunk: by the group that owns the tile's last batch (after its last chunk)
        if (((nbt - 1) & 1) == grp) {
            const int nchunk = sy.chunk0[nbt];
            const int s = nchunk % NBUF, u = nchunk / NBUF;
            if (ok && u > 0) ok = lds_wait_ge(&sy.freed[s], 4 * u, &sy.abort_);
            if (tl == 0) sy.ks[s] = -1;
            lds_signal(&sy.full[s]);
        }
        if (!ok) rr = __longlong_as_double(0x7ff8000000000000ll);
    } else {
        // ---- consumers (waves 8-11): as in k_build_tile2
        const bool full_tile = 16 * (7 - wave) < nrows;
        int yoff[9], woff[9];
        bool ton[9];
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            const int rt = s <= wave ? wave : 7 - wave;
            const int ct = s <= wave ? s : s - wave - 1;
            yoff[s] = 16 * rt; woff[s] = 16 * ct; ton[s] = 16 * rt < nrows;
        }
        for (int n = 0;; ++n) {
            const int s = n % NBUF, u = n / NBUF;
            if (!lds_wait_ge(&sy.full[s], 4 * (u + 1), &sy.abort_)) break;
            const int ksteps = __hip_atomic_load(&sy.ks[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (ksteps < 0) break;
            const double *Zt = pan + s * PANEL;
            if (full_tile) {
                switch (wave) {
                    case 0: tile_syrk_steps<0>(Zt, lane, ksteps, acc); break;
                    case 1: tile_syrk_steps<1>(Zt, lane, ksteps, acc); break;
                    case 2: tile_syrk_steps<2>(Zt, lane, ksteps, acc); break;
                    default: tile_syrk_steps<3>(Zt, lane, ksteps, acc); break;
                }
            } else {
                for (int kk = 0; kk < ksteps; ++kk) {
                    const int krow = 4 * kk + (lane >> 4);
                    const double *yr = Zt + krow * LD + (lane & 15);
                    const double *wr = yr;
#pragma unroll
                    for (int q = 0; q < 9; ++q)
                        if (ton[q])
                            acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(yr[yoff[q]], wr[woff[q]], acc[q], 0, 0, 0);
                }
            }
            lds_signal(&sy.freed[s]);                // read; the next producer group clears and refills it
        }
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            if (ton[s]) {
                const int lcol = woff[s] + (lane & 15);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int lr = yoff[s] + (lane >> 4) + 4 * e;
                    const double v = acc[s][e];
                    if (lr < nrows && lcol <= lr && v != 0.0)
                        atomic_add_f64(S + (6 * (int64_t)d.tile_cams[c0 + lcol / 6] + lcol % 6) * d.ldS
                                         + 6 * (int64_t)d.tile_cams[c0 + lr / 6] + lr % 6, -v);
                }
            }
        }
    }
    __syncthreads();
    for (int i = t; i < nrows; i += NT)
        atomic_add_f64(g_red + 6 * (int64_t)d.tile_cams[c0 + i / 6] + i % 6, vt[i]);
    double accr[1] = {rr};
    block_sum<1>(accr, sh);
    if (t == 0) partial[blockIdx.x] = accr[0];
    pmin = pmin < 1e300 ? sqrt(pmin) : pmin; pmax = sqrt(pmax);
    for (int off = 32; off > 0; off >>= 1) {
        pmin = fmin(pmin, __shfl_down(pmin, off, 64));
        pmax = fmax(pmax, __shfl_down(pmax, off, 64));
    }
    if ((t & 63) == 0 && pmax > 0.0) {
        atomicMin(pivmm, (unsigned long long)__double_as_longlong(pmin));
        atomicMax(pivmm + 1, (unsigned long long)__double_as_longlong(pmax));
    }
}

// ------------------------------------------------------- posterior cov ---
// bundle_cov.m: blocks of C = s0^2 inv(J'J).  With the points eliminated,
//   inv(J'J)[cams, cams] = inv(S)                         (CEO, CIO)
//   inv(J'J)[p, p]       = V_p^-1 + Y_p' inv(S) Y_p,  Y_p = W_p V_p^-1  (COP)
// inv(S) comes in one of two forms (SinvView):
//   dense   the lower triangle of inv(S), column-major with leading dimension ld (rocsolver_dpotri on the in-place factor:
//           several ranks, shared EO blocks, or a caller who wants the whole inverse);
//   tiles   the entries of inv(S) ON THE PATTERN OF THE FACTOR only -- compact 64 x 64 tiles in the nested-dissection order
//           of the dataflow Cholesky, from the selected inversion of chol_df.hpp (DataflowChol::selected_inverse).  Everything
//           the covariance blocks need lies on that pattern: the cameras' own blocks, the IO block, and the blocks of camera
//           pairs that see a common object point.
struct SinvView {
    const double *dense = nullptr; int64_t ld = 0;
    const double *tiles = nullptr; const int64_t *toff = nullptr; const int *perm = nullptr; int nT = 0;
};
__device__ __forceinline__ double sym_at(const SinvView &V, int64_t /*unused*/, int r, int c) {
    if (V.dense) return r >= c ? V.dense[(int64_t)c * V.ld + r] : V.dense[(int64_t)r * V.ld + c];
    const int pr = V.perm[r], pc = V.perm[c];
    const int i = max(pr, pc), j = min(pr, pc);
    const int64_t off = V.toff[(int64_t)(i >> 6) * V.nT + (j >> 6)];
    return off < 0 ? 0.0 : V.tiles[off + (int64_t)(j & 63) * 64 + (i & 63)];      // (diagonal tiles hold both triangles)
}

// CEO: 6x6 block per image (zero rows/columns for fixed elements); CIO: the
// nIOu x nIOu block of the IO unknowns.
__global__ void k_cov_cam(DevProblem d, const SinvView Sinv, double s02, double *__restrict__ CEO,
                          double *__restrict__ CIO) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t nE = 36 * (int64_t)d.nc, nI = (int64_t)d.nIOu * d.nIOu;
    if (i < nE) {
        if (!CEO) return;
        const int c = (int)(i / 36), e = (int)(i % 36), a = e % 6, b = e / 6;
        const int ra = d.cam_col[(int64_t)c * MAXCOL + a], rb = d.cam_col[(int64_t)c * MAXCOL + b];   // (shared elements: the leader's)
        CEO[i] = (d.z_est[ra] && d.z_est[rb]) ? s02 * sym_at(Sinv, d.ldS, ra, rb) : 0.0;
    } else if (i < nE + nI) {
        if (!CIO) return;
        const int64_t q = i - nE;
        const int a = (int)(q % d.nIOu), b = (int)(q / d.nIOu);
        CIO[q] = s02 * sym_at(Sinv, d.ldS, 6 * d.nc + a, 6 * d.nc + b);
    }
}

// COP for the points of one batch (same layout as k_build: lane t <-> observation).
template <int MODEL, bool WITH_IO>
__global__ __launch_bounds__(256) void k_cov_points(DevProblem d, const double *__restrict__ z,
                                                    const CamRec *__restrict__ cams,
                                                    const double *__restrict__ Vinv,
                                                    const SinvView Sinv, double s02,
                                                    double *__restrict__ COP) {
    constexpr int NCX = WITH_IO ? MAXCOL : 6;
    extern __shared__ double smem[];
    const int BT = blockDim.x;
    const int strideW = d.ncolmax * 3;
    double *Yl = smem;                               // [BT][strideW]  Y = W V^-1
    double *red = Yl + (size_t)BT * strideW;         // [BT][6]
    const int t = threadIdx.x;
    const int64_t o0 = d.batch_start[blockIdx.x];
    const int nobs = (int)(d.batch_start[blockIdx.x + 1] - o0);
    const bool active = t < nobs;
    const int64_t o = o0 + t;
    int pt = 0, seg_start = 0, seg_len = 0, ncol = 6;
    const CamRec *C = cams;
    double vi[6] = {0, 0, 0, 0, 0, 0};
    if (active) {
        pt = d.o_pt[o];
        const uint32_t sg = d.o_seg[o];
        seg_start = sg & 0xFFFF; seg_len = sg >> 16;
        C = cams + d.o_cam[o];
        ncol = WITH_IO ? C->ncol : 6;
        double r[2], E[2][NCX], B[2][3];
        eval_obs_cols<MODEL, WITH_IO>(d, *C, z, o, pt, r, E, B);
        point_block_inverse(Vinv + 6 * (int64_t)pt, vi);
        double *yl = Yl + (size_t)t * strideW;
#pragma unroll
        for (int a = 0; a < NCX; ++a)
            if (a < ncol) {
                const double w0 = E[0][a] * B[0][0] + E[1][a] * B[1][0];
                const double w1 = E[0][a] * B[0][1] + E[1][a] * B[1][1];
                const double w2 = E[0][a] * B[0][2] + E[1][a] * B[1][2];
                yl[3 * a] = w0 * vi[0] + w1 * vi[1] + w2 * vi[2];
                yl[3 * a + 1] = w0 * vi[1] + w1 * vi[3] + w2 * vi[4];
                yl[3 * a + 2] = w0 * vi[2] + w1 * vi[4] + w2 * vi[5];
            }
    }
    __syncthreads();
    if (active) {
        // c = Y_i' (sum_j Sinv[cols_i, cols_j] Y_j), symmetric part xx xy xz yy yz zz
        double c6[6] = {0, 0, 0, 0, 0, 0};
        const double *yi = Yl + (size_t)t * strideW;
        for (int a = 0; a < ncol; ++a) {
            const int ra = C->col[a];
            double r0 = 0, r1 = 0, r2 = 0;
            for (int jj = seg_start; jj < seg_start + seg_len; ++jj) {
                const CamRec *Cj = cams + d.o_cam[o0 + jj];
                const int ncj = WITH_IO ? Cj->ncol : 6;
                const double *yj = Yl + (size_t)jj * strideW;
                for (int b = 0; b < ncj; ++b) {
                    const double sv = sym_at(Sinv, d.ldS, ra, Cj->col[b]);
                    r0 += sv * yj[3 * b]; r1 += sv * yj[3 * b + 1]; r2 += sv * yj[3 * b + 2];
                }
            }
            const double y0 = yi[3 * a], y1 = yi[3 * a + 1], y2 = yi[3 * a + 2];
            c6[0] += y0 * r0; c6[1] += 0.5 * (y0 * r1 + y1 * r0); c6[2] += 0.5 * (y0 * r2 + y2 * r0);
            c6[3] += y1 * r1; c6[4] += 0.5 * (y1 * r2 + y2 * r1); c6[5] += y2 * r2;
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) red[6 * t + q] = c6[q];
    }
    __syncthreads();
    if (active && t == seg_start) {
        double c6[6] = {vi[0], vi[1], vi[2], vi[3], vi[4], vi[5]};
        for (int j = 0; j < seg_len; ++j)
#pragma unroll
            for (int q = 0; q < 6; ++q) c6[q] += red[6 * (t + j) + q];
        const int64_t zp = d.NS + 3 * (int64_t)pt;
        const bool e0 = d.z_est[zp], e1 = d.z_est[zp + 1], e2 = d.z_est[zp + 2];
        double *out = COP + 9 * (int64_t)pt;
        out[0] = e0 ? s02 * c6[0] : 0.0;
        out[1] = out[3] = (e0 && e1) ? s02 * c6[1] : 0.0;
        out[2] = out[6] = (e0 && e2) ? s02 * c6[2] : 0.0;
        out[4] = e1 ? s02 * c6[3] : 0.0;
        out[5] = out[7] = (e1 && e2) ? s02 * c6[4] : 0.0;
        out[8] = e2 ? s02 * c6[5] : 0.0;
    }
}

// COP of a giant point: one workgroup per point, Y in the W scratch.
template <int MODEL, bool WITH_IO>
__global__ __launch_bounds__(256) void k_cov_giant(DevProblem d, const double *__restrict__ z,
                                                   const CamRec *__restrict__ cams,
                                                   const double *__restrict__ Vinv, const SinvView Sinv,
                                                   double s02, double *__restrict__ COP) {
    constexpr int NCX = WITH_IO ? MAXCOL : 6;
    __shared__ double sh[6 * 4];
    const int t = threadIdx.x, BT = blockDim.x;
    const int64_t o0 = d.giant_start[blockIdx.x], o1 = d.giant_start[blockIdx.x + 1];
    const int k = (int)(o1 - o0);
    const int strideW = d.ncolmax * 3;
    double *Yg = d.giant_W + (o0 - d.giant_start[0]) * strideW;
    const int pt = d.o_pt[o0];
    double vi[6];
    point_block_inverse(Vinv + 6 * (int64_t)pt, vi);
    for (int i = t; i < k; i += BT) {
        const int64_t o = o0 + i;
        const CamRec &C = cams[d.o_cam[o]];
        const int ncol = WITH_IO ? C.ncol : 6;
        double r[2], E[2][NCX], B[2][3];
        eval_obs_cols<MODEL, WITH_IO>(d, C, z, o, pt, r, E, B);
        double *yl = Yg + (size_t)i * strideW;
        for (int a = 0; a < NCX; ++a) {
            if (a >= ncol) break;
            double ea0 = 0, ea1 = 0;
#pragma unroll
            for (int q = 0; q < NCX; ++q) if (q == a) { ea0 = E[0][q]; ea1 = E[1][q]; }
            const double w0 = ea0 * B[0][0] + ea1 * B[1][0];
            const double w1 = ea0 * B[0][1] + ea1 * B[1][1];
            const double w2 = ea0 * B[0][2] + ea1 * B[1][2];
            yl[3 * a] = w0 * vi[0] + w1 * vi[1] + w2 * vi[2];
            yl[3 * a + 1] = w0 * vi[1] + w1 * vi[3] + w2 * vi[4];
            yl[3 * a + 2] = w0 * vi[2] + w1 * vi[4] + w2 * vi[5];
        }
    }
    __threadfence_block();
    __syncthreads();
    double c6[6] = {0, 0, 0, 0, 0, 0};
    for (int i = t; i < k; i += BT) {
        const CamRec &Ci = cams[d.o_cam[o0 + i]];
        const int nci = WITH_IO ? Ci.ncol : 6;
        const double *yi = Yg + (size_t)i * strideW;
        for (int a = 0; a < nci; ++a) {
            const int ra = Ci.col[a];
            double r0 = 0, r1 = 0, r2 = 0;
            for (int j = 0; j < k; ++j) {
                const CamRec &Cj = cams[d.o_cam[o0 + j]];
                const int ncj = WITH_IO ? Cj.ncol : 6;
                const double *yj = Yg + (size_t)j * strideW;
                for (int b = 0; b < ncj; ++b) {
                    const double sv = sym_at(Sinv, d.ldS, ra, Cj.col[b]);
                    r0 += sv * yj[3 * b]; r1 += sv * yj[3 * b + 1]; r2 += sv * yj[3 * b + 2];
                }
            }
            const double y0 = yi[3 * a], y1 = yi[3 * a + 1], y2 = yi[3 * a + 2];
            c6[0] += y0 * r0; c6[1] += 0.5 * (y0 * r1 + y1 * r0); c6[2] += 0.5 * (y0 * r2 + y2 * r0);
            c6[3] += y1 * r1; c6[4] += 0.5 * (y1 * r2 + y2 * r1); c6[5] += y2 * r2;
        }
    }
    block_sum<6>(c6, sh);
    if (t == 0) {
#pragma unroll
        for (int q = 0; q < 6; ++q) c6[q] += vi[q];
        const int64_t zp = d.NS + 3 * (int64_t)pt;
        const bool e0 = d.z_est[zp], e1 = d.z_est[zp + 1], e2 = d.z_est[zp + 2];
        double *out = COP + 9 * (int64_t)pt;
        out[0] = e0 ? s02 * c6[0] : 0.0;
        out[1] = out[3] = (e0 && e1) ? s02 * c6[1] : 0.0;
        out[2] = out[6] = (e0 && e2) ? s02 * c6[2] : 0.0;
        out[4] = e1 ? s02 * c6[3] : 0.0;
        out[5] = out[7] = (e1 && e2) ? s02 * c6[4] : 0.0;
        out[8] = e2 ? s02 * c6[5] : 0.0;
    }
}

// Envelope of S <-> contiguous buffer (multi-GPU: only the envelope of the
// reduced system travels through the all-reduce).  Column c owns the rows
// [c, col_bend[c]) of the co-visibility band and the dense tail rows
// [max(tail0, col_bend[c]), NS); its packed run starts at col_off[c].
// to_packed = 1: S -> buf, 0: buf -> S.
__global__ __launch_bounds__(256) void k_pack_envelope(double *__restrict__ S, int64_t ldS, int NS, int tail0,
                                                       const int *__restrict__ col_bend,
                                                       const int64_t *__restrict__ col_off,
                                                       double *__restrict__ buf, int to_packed) {
    const int c = blockIdx.x;
    const int be = col_bend[c];
    const int nband = be - c;
    const int t0 = max(tail0, be);
    const int ntot = nband + (NS - t0);
    double *col = S + (int64_t)c * ldS;
    double *pk = buf + col_off[c];
    for (int i = threadIdx.x; i < ntot; i += 256) {
        const int r = i < nband ? c + i : t0 + (i - nband);
        if (to_packed) pk[i] = col[r]; else col[r] = pk[i];
    }
}

// Zero (scale != 0: scale by d_i d_j) the envelope of the lower triangle of S only: column c
// owns the rows [c, col_bend[c]) of the co-visibility band and the dense tail rows
// [max(tail0, col_bend[c]), NS) -- nothing outside it is ever written by the build kernels or read
// by the factorisation, and at C4 the dense 30 032^2 array is 7.2 GB.
__device__ __forceinline__ void envelope_body(double *__restrict__ S, int64_t ldS, int NS, int tail0,
                                              const int *__restrict__ col_bend, const double *__restrict__ ds,
                                              double *__restrict__ vec, unsigned long long *__restrict__ pivmm) {
    const int c = blockIdx.x;
    if (vec && threadIdx.x < 3) vec[(int64_t)threadIdx.x * NS + c] = 0.0;
    if (c == 0 && vec && threadIdx.x >= 64 && threadIdx.x < 72) vec[(int64_t)3 * NS + (threadIdx.x - 64)] = 0.0;
    if (c == 0 && pivmm && threadIdx.x >= 128 && threadIdx.x < 132)       // {min, max} x {points, cameras}
        pivmm[threadIdx.x - 128] = (threadIdx.x & 1) ? 0ull : (unsigned long long)__double_as_longlong(1e300);
    const int be = col_bend[c];
    const int nband = be - c;
    const int t0 = max(tail0, be);
    const int ntot = nband + (NS - t0);
    double *col = S + (int64_t)c * ldS;
    const double dc = ds ? ds[c] : 0.0;
    for (int i = threadIdx.x; i < ntot; i += 256) {
        const int r = i < nband ? c + i : t0 + (i - nband);
        col[r] = ds ? col[r] * dc * ds[r] : 0.0;
    }
}
__global__ __launch_bounds__(256) void k_envelope_op(double *__restrict__ S, int64_t ldS, int NS, int tail0,
                                                     const int *__restrict__ col_bend,
                                                     const double *__restrict__ ds /* null: zero */,
                                                     double *__restrict__ vec = nullptr /* [3 NS + 8] zeroed with S */,
                                                     unsigned long long *__restrict__ pivmm = nullptr /* reset */) {
    envelope_body(S, ldS, NS, tail0, col_bend, ds, vec, pivmm);
}
// The first launch of a linearisation: envelope of S, the vectors behind it and the pivot extremes
// cleared, and the camera records at z (block c < nc builds camera c) -- instead of k_cam_prep + this.
__global__ __launch_bounds__(256) void k_envelope_cams(DevProblem d, const double *__restrict__ z, CamRec *__restrict__ cams,
                                                       double *__restrict__ S, int64_t ldS, int NS, int tail0,
                                                       const int *__restrict__ col_bend, double *__restrict__ vec,
                                                       unsigned long long *__restrict__ pivmm) {
    if (threadIdx.x == 255 && (int)blockIdx.x < d.nc) cam_prep_one(d, [z](int64_t i) { return z[i]; }, (int)blockIdx.x, cams);
    envelope_body(S, ldS, NS, tail0, col_bend, nullptr, vec, pivmm);
}

// ---------------------------------------------------------------- F11 ---
// Camera/IO side: priors, damping, fixed rows, column scaling factors.
//   jn2c[i] = diagU[i] + prior weight  (squared column norm of J)
//   dscale[i] = 1/sqrt(jn2c) if scaling, estimated and >0 ; else 1
//   S(NS, i) = -dscale[i]*g_red[i]  (0 for fixed): the right-hand side, stored as
//              the extra row below the matrix (see chol.hpp)
__global__ void k_finish(DevProblem d, const double *__restrict__ z, double lambda, int scale,
                         double *__restrict__ S, double *__restrict__ g_c, double *__restrict__ g_red,
                         const double *__restrict__ diagU, double *__restrict__ jn2c,
                         double *__restrict__ dscale, double *__restrict__ partial, unsigned *__restrict__ ctr,
                         const double *__restrict__ red_scal, double *__restrict__ out,
                         double *__restrict__ mailbox, const uint8_t *__restrict__ mine = nullptr,
                         int *__restrict__ df_info = nullptr, int *__restrict__ df_ctl = nullptr,
                         unsigned long long *__restrict__ df_q = nullptr, int df_nq = 0) {
    // (df_*: what the factorisation's own reset launch would do -- k_df_reset, chol_df.hpp -- taken along here, one
    // launch less per solve: 10 us of a 380 us step at C1 and of the reference's own projects)
    if (df_ctl) {
        const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (g == 0) { *df_info = 0; for (int q = 0; q < 8; ++q) df_ctl[q] = 0; }
        for (int64_t j = g; j < df_nq; j += (int64_t)gridDim.x * blockDim.x) df_q[j] = 0xFFFFFFFFFFFFFFFFull;
    }
    // mine (several ranks, domain sharding): S and g_red hold THIS rank's share of the reduced system and are
    // summed over the ranks later (the top separators) or never (the rank's own domain): the terms that enter
    // once -- prior, damping, the unit diagonal of a fixed element -- are added where the rank owns the column.
    // g_c and diagU are complete on every rank (summed before this kernel), and so are jn2c and dscale.
    __shared__ double sh[8];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double acc[1] = {0.0};                           // trace(J'J), camera/IO part: estimated columns
    if (i < d.NS) {
        const bool own = !mine || mine[i];
        const double pw = d.z_prw[i];
        double add = 0, g = g_red[i], jn2 = diagU[i];
        if (pw > 0) {
            // (the prior residual needs this rank's z[i] to be current: true for the owner; g_c is only ever
            // read at owned entries, the others' share of this term is dropped with them)
            const double e = pw * (z[i] - d.z_prv[i]);
            jn2 += pw;
            if (own) { g_c[i] += e; add += pw; g += e; g_red[i] = g; }
        }
        jn2c[i] = jn2;
        const bool est = d.z_est[i] != 0;
        double ds = 1.0;
        if (est) {
            if (own) add += lambda;
            if (scale && jn2 > 0) ds = 1.0 / sqrt(jn2);
            S[i * d.ldS + i] += add;
            S[i * d.ldS + d.NS] = -ds * g;
            acc[0] = jn2;
        } else {
            S[i * d.ldS + i] = own ? 1.0 : 0.0;
            S[i * d.ldS + d.NS] = 0.0;
        }
        dscale[i] = ds;
    }
    if (grid_sum<1>(acc, sh, partial, ctr) && threadIdx.x == 0) {
        out[0] = acc[0];
        if (mailbox) { mailbox[34] = acc[0]; mailbox[32] = red_scal[0]; mailbox[33] = red_scal[1]; }   // slots no other kernel writes
    }
}

// out = in where this rank owns the entry, 0 elsewhere (sum over ranks = the full vector)
__global__ void k_mask_owned(int64_t n, const uint8_t *__restrict__ mine, const double *__restrict__ in,
                             double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = mine[i] ? in[i] : 0.0;
}

// weighted image rows from the unweighted ones (reference row order o_row), this shard's observations
__global__ void k_weight_rows(DevProblem d, const double *__restrict__ r_unw, double *__restrict__ r_wgt) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= d.nobs) return;
    const int64_t row = d.o_row[k];
    const int cam = d.o_cam[k];
    const double w0 = d.o_w ? d.o_w[2 * k] : d.cam_w[2 * cam], w1 = d.o_w ? d.o_w[2 * k + 1] : d.cam_w[2 * cam + 1];
    r_wgt[2 * row] = r_unw[2 * row] * w0;
    r_wgt[2 * row + 1] = r_unw[2 * row + 1] * w1;
}

__global__ void k_unscale(int64_t NS, const double *__restrict__ q, const double *__restrict__ ds,
                          double *__restrict__ dz) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < NS) dz[i] = ds[i] * q[i];
}

// ---------------------------------------------------------------- K7 ----
// Back-substitution dp = -V^-1 (g_p + W' dc) and the sums ||Jp||^2, r'Jp over
// the image rows.  dz[0..NS) holds dc on entry; dz[NS..) receives dp.
// NCXT: capacity of the camera-side column list (6 = fixed IO, 14, MAXCOL).
template <int MODEL, int NCXT>
__global__ __launch_bounds__(256) void k_backsub(DevProblem d, const double *__restrict__ z,
                                                 const CamRec *__restrict__ cams,
                                                 const double *__restrict__ Vinv, const double *__restrict__ gp,
                                                 double *__restrict__ dz,
                                                 double *__restrict__ partial /* [nb][2] */, int batch0) {
    constexpr int NCX = NCXT;
    constexpr bool WITH_IO = NCXT > 6;
    extern __shared__ double smem[];
    double *red = smem;                 // [BT][3]  B't
    double *dpl = red + (size_t)blockDim.x * 3;   // [BT][3]  dp at leader slot
    __shared__ double sh[16];
    const int t = threadIdx.x;
    const int batch = batch0 + blockIdx.x;
    const int64_t o0 = d.batch_start[batch];
    const int nobs = (int)(d.batch_start[batch + 1] - o0);
    const bool active = t < nobs;
    const int64_t o = o0 + t;
    double r[2] = {0, 0}, tt[2] = {0, 0};
    double E[2][NCX];
    double B[2][3] = {{0, 0, 0}, {0, 0, 0}};
    int pt = 0, seg_start = 0, seg_len = 0;
    if (active) {
        const int cam = d.o_cam[o]; pt = d.o_pt[o];
        const uint32_t sg = d.o_seg[o];
        seg_start = sg & 0xFFFF; seg_len = sg >> 16;
        const CamRec &C = cams[cam];
        const int ncol = WITH_IO ? min(C.ncol, NCX) : 6;
        eval_obs_cols_n<MODEL, NCX>(d, C, z, o, pt, r, E, B);
#pragma unroll
        for (int a = 0; a < NCX; ++a)
            if (a < ncol) { const double dc = dz[C.col[a]]; tt[0] += E[0][a] * dc; tt[1] += E[1][a] * dc; }
        red[3 * t] = B[0][0] * tt[0] + B[1][0] * tt[1];
        red[3 * t + 1] = B[0][1] * tt[0] + B[1][1] * tt[1];
        red[3 * t + 2] = B[0][2] * tt[0] + B[1][2] * tt[1];
    }
    __syncthreads();
    if (active && t == seg_start) {
        double s[3] = {gp[3 * (int64_t)pt], gp[3 * (int64_t)pt + 1], gp[3 * (int64_t)pt + 2]};
        for (int j = 0; j < seg_len; ++j) { s[0] += red[3 * (t + j)]; s[1] += red[3 * (t + j) + 1]; s[2] += red[3 * (t + j) + 2]; }
        double p0, p1, p2;
        point_block_solve_neg(Vinv + 6 * (int64_t)pt, s[0], s[1], s[2], p0, p1, p2);
        const int64_t zp = d.NS + 3 * (int64_t)pt;
        const double q0 = d.z_est[zp] ? p0 : 0.0, q1 = d.z_est[zp + 1] ? p1 : 0.0, q2 = d.z_est[zp + 2] ? p2 : 0.0;
        dz[zp] = q0; dz[zp + 1] = q1; dz[zp + 2] = q2;
        dpl[3 * t] = q0; dpl[3 * t + 1] = q1; dpl[3 * t + 2] = q2;
    }
    __syncthreads();
    double acc[2] = {0, 0};
    if (active) {
        const double *dp = dpl + 3 * seg_start;
        const double j0 = tt[0] + B[0][0] * dp[0] + B[0][1] * dp[1] + B[0][2] * dp[2];
        const double j1 = tt[1] + B[1][0] * dp[0] + B[1][1] * dp[1] + B[1][2] * dp[2];
        acc[0] = j0 * j0 + j1 * j1;
        // r'Jp = g'p is summed over the unknowns in k_prior_jv (no residual needed here)
    }
    block_sum<2>(acc, sh);
    if (t == 0) { partial[2 * (int64_t)batch] = acc[0]; partial[2 * (int64_t)batch + 1] = acc[1]; }
}

// ------------------------------------------------- forward intersection ---
// photogrammetry/forwintersect.m:27-46 (+ cammodel/pm_multilenscorr1.m:45-69,
// pm_multiforwintersect.m, pm_forwintersect3.m:55-82): the object point that minimises the
// summed squared distance to its lens-corrected image rays,
//     (sum_j (I - d_j d_j')) Q = sum_j (I - d_j d_j') c_j,
// d_j = M_j K^-1 [x; y; 1] normalised, c_j the projection centre.  One lane per observation of
// a batch (the point-major layout of k_backsub), 3 x 3 system per point at its leader lane.
// OPout (processing order): the point, or NaN with fewer than two rays.
__device__ __forceinline__ void fwd_ray_terms(const CamRec &C, int nK, int nP, double u, double v, double out[9]) {
    const double qx = C.sz * u, qy = -C.sz * v;                  // mm, y up
    const double xb = qx - C.pp[0], yb = qy - C.pp[1];
    const double r2 = xb * xb + yb * yb;
    double Kr = 0, pw = 1;
    for (int j = 0; j < MAXK; ++j) if (j < nK) { pw *= r2; Kr += C.K[j] * pw; }
    double dx = xb * Kr, dy = yb * Kr;
    if (nP >= 2) {
        const double P1 = C.P[0], P2 = C.P[1], P3 = nP > 2 ? 1.0 + C.P[2] : 1.0;
        dx += (P1 * (r2 + 2 * xb * xb) + 2 * P2 * xb * yb) * P3;
        dy += (P2 * (r2 + 2 * yb * yb) + 2 * P1 * xb * yb) * P3;
    }
    const double dc[3] = {qx - dx - C.pp[0], qy - dy - C.pp[1], -C.f};
    double dw[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) dw[i] = C.Mt[i] * dc[0] + C.Mt[3 + i] * dc[1] + C.Mt[6 + i] * dc[2];     // M dc, Mt = M'
    const double inn = 1.0 / sqrt(dw[0] * dw[0] + dw[1] * dw[1] + dw[2] * dw[2]);
    const double d0 = dw[0] * inn, d1 = dw[1] * inn, d2 = dw[2] * inn;
    const double p00 = 1 - d0 * d0, p01 = -d0 * d1, p02 = -d0 * d2, p11 = 1 - d1 * d1, p12 = -d1 * d2, p22 = 1 - d2 * d2;
    out[0] = p00; out[1] = p01; out[2] = p02; out[3] = p11; out[4] = p12; out[5] = p22;
    out[6] = p00 * C.c[0] + p01 * C.c[1] + p02 * C.c[2];
    out[7] = p01 * C.c[0] + p11 * C.c[1] + p12 * C.c[2];
    out[8] = p02 * C.c[0] + p12 * C.c[1] + p22 * C.c[2];
}
__device__ __forceinline__ void fwd_solve(const double a[9], int rays, double *op) {
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    double inv[6];
    if (rays < 2) { op[0] = op[1] = op[2] = nan; return; }
    inv3_sym(a, inv);
    op[0] = inv[0] * a[6] + inv[1] * a[7] + inv[2] * a[8];
    op[1] = inv[1] * a[6] + inv[3] * a[7] + inv[4] * a[8];
    op[2] = inv[2] * a[6] + inv[4] * a[7] + inv[5] * a[8];
}
__global__ __launch_bounds__(256) void k_forwintersect(DevProblem d, const CamRec *__restrict__ cams,
                                                       double *__restrict__ OPout) {
    extern __shared__ double smem[];                 // [BT][9]
    const int t = threadIdx.x;
    const int64_t o0 = d.batch_start[blockIdx.x];
    const int nobs = (int)(d.batch_start[blockIdx.x + 1] - o0);
    int pt = 0, seg_start = 0, seg_len = 0;
    if (t < nobs) {
        const int64_t o = o0 + t;
        pt = d.o_pt[o];
        const uint32_t sg = d.o_seg[o];
        seg_start = sg & 0xFFFF; seg_len = sg >> 16;
        fwd_ray_terms(cams[d.o_cam[o]], d.nK, d.nP, d.o_uv_raw[2 * o], d.o_uv_raw[2 * o + 1], smem + 9 * t);
    }
    __syncthreads();
    if (t < nobs && t == seg_start) {
        double a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int j = 0; j < seg_len; ++j)
#pragma unroll
            for (int q = 0; q < 9; ++q) a[q] += smem[9 * (t + j) + q];
        fwd_solve(a, seg_len, OPout + 3 * (int64_t)pt);
    }
}
// points with more observations than a batch holds: one workgroup per point
__global__ __launch_bounds__(256) void k_forwintersect_giant(DevProblem d, const CamRec *__restrict__ cams,
                                                             double *__restrict__ OPout) {
    __shared__ double sh[9 * 4];
    const int t = threadIdx.x;
    const int64_t o0 = d.giant_start[blockIdx.x], o1 = d.giant_start[blockIdx.x + 1];
    double a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t o = o0 + t; o < o1; o += blockDim.x) {
        double q[9];
        fwd_ray_terms(cams[d.o_cam[o]], d.nK, d.nP, d.o_uv_raw[2 * o], d.o_uv_raw[2 * o + 1], q);
#pragma unroll
        for (int i = 0; i < 9; ++i) a[i] += q[i];
    }
    block_sum<9>(a, sh);
    if (t == 0) fwd_solve(a, (int)(o1 - o0), OPout + 3 * (int64_t)d.o_pt[o0]);
}

// ---------------------------------------------------------------- K8 ----
// partial[2*blk] += ||J v||^2, partial[2*blk+1] += r'Jv over image rows
// (grid-stride; v in z layout).
template <int MODEL, int NCXT>
__global__ __launch_bounds__(256) void k_jtimes(DevProblem d, const double *__restrict__ z,
                                                const CamRec *__restrict__ cams, const double *__restrict__ v,
                                                double *__restrict__ partial) {
    constexpr int NCX = NCXT;
    constexpr bool WITH_IO = NCXT > 6;
    __shared__ double sh[16];
    double acc[2] = {0, 0};
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < d.nobs; o += stride) {
        const int cam = d.o_cam[o], pt = d.o_pt[o];
        const CamRec &C = cams[cam];
        const int ncol = WITH_IO ? min(C.ncol, NCX) : 6;
        double r[2], E[2][NCX], B[2][3];
        eval_obs_cols_n<MODEL, NCX>(d, C, z, o, pt, r, E, B);
        const double *vp = v + d.NS + 3 * (int64_t)pt;
        double j0 = B[0][0] * vp[0] + B[0][1] * vp[1] + B[0][2] * vp[2];
        double j1 = B[1][0] * vp[0] + B[1][1] * vp[1] + B[1][2] * vp[2];
#pragma unroll
        for (int a = 0; a < NCX; ++a)
            if (a < ncol) { const double vc = v[C.col[a]]; j0 += E[0][a] * vc; j1 += E[1][a] * vc; }
        acc[0] = fma2(acc[0], j0, j0, j1, j1);
    }
    block_sum<2>(acc, sh);
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = acc[0]; partial[2 * blockIdx.x + 1] = acc[1]; }
}

// J v of the image rows as a VECTOR (weighted, chol(W) J as the solvers see it), in the reference's row order
// (multi_res.m:143-144,297: image-major, ascending object point, x / y interleaved): what termFun(Jp, r) of
// bundle.m:186-192 / gauss_newton_armijo.m:187 receives.
template <int MODEL, int NCXT>
__global__ __launch_bounds__(256) void k_jtimes_vec(DevProblem d, const double *__restrict__ z, const CamRec *__restrict__ cams,
                                                    const double *__restrict__ v, double *__restrict__ Jv) {
    constexpr int NCX = NCXT;
    constexpr bool WITH_IO = NCXT > 6;
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= d.nobs) return;
    const int cam = d.o_cam[o], pt = d.o_pt[o];
    const CamRec &C = cams[cam];
    const int ncol = WITH_IO ? min(C.ncol, NCX) : 6;
    double r[2], E[2][NCX], B[2][3];
    eval_obs_cols_n<MODEL, NCX>(d, C, z, o, pt, r, E, B);
    const double *vp = v + d.NS + 3 * (int64_t)pt;
    double j0 = B[0][0] * vp[0] + B[0][1] * vp[1] + B[0][2] * vp[2];
    double j1 = B[1][0] * vp[0] + B[1][1] * vp[1] + B[1][2] * vp[2];
#pragma unroll
    for (int a = 0; a < NCX; ++a)
        if (a < ncol) { const double vc = v[C.col[a]]; j0 += E[0][a] * vc; j1 += E[1][a] * vc; }
    const int64_t row = d.o_row[o];
    Jv[2 * row] = j0; Jv[2 * row + 1] = j1;
}

// {prior rows' share of ||Jv||^2, r'Jv of ALL rows, ||v||^2 (owned)}: out[3] per block.
// r'Jv = (J'r)'v = g'v with the gradient of the last linearisation (g_c incl. the camera/IO
// priors, g_p incl. the point priors): no pass over the observations, no stored residual.
// The tail of a solve in one launch: the sums above -> out[4..6]; the back-substitution kernels'
// (or k_jtimes') nbs partial sum pairs over the image rows -> out[0], out[1]; the extremes
// of the Cholesky pivots ldiag of the reduced system (estimated entries) -> pivmm[2..3]; and,
// with a mailbox, everything the host reads after a solve: out[0..7] -> mailbox[0..7],
// pivmm[0..3] -> mailbox[40..43], *info -> mailbox[44].
__global__ __launch_bounds__(1024) void k_prior_jv(DevProblem d, const double *__restrict__ z,
                                                  const double *__restrict__ v, const double *__restrict__ g_c,
                                                  const double *__restrict__ gp, double *__restrict__ partial,
                                                  unsigned *__restrict__ ctr, const double *__restrict__ bs_partial,
                                                  int64_t nbs, const double *__restrict__ ldiag,
                                                  unsigned long long *__restrict__ pivmm, const int *__restrict__ info,
                                                  double *__restrict__ out, double *__restrict__ mailbox,
                                                  unsigned long long seq, const uint8_t *__restrict__ piv_have = nullptr) {
    __shared__ double sh[80];
    double acc[5] = {0, 0, 0, 0, 0};                 // [3], [4]: the sums over the image rows, from bs_partial
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    double pmin = 1e300, pmax = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.NZ; i += stride) {
        if (ldiag && i < d.NS && d.z_est[i] && (!piv_have || piv_have[i])) {      // (piv_have: the pivots this rank computed)
            double p = ldiag[i];                     // diag(L) by natural index (the factorisation may be permuted)
            p = p == p ? p : 0.0;
            pmin = fmin(pmin, p); pmax = fmax(pmax, p);
        }
        if (!d.z_mine[i]) continue;
        const double w = d.any_prior ? d.z_prw[i] : 0.0, vi = d.z_est[i] ? v[i] : 0.0;
        if (w > 0) acc[0] += w * vi * vi;
        acc[1] += (i < d.NS ? g_c[i] : gp[i - d.NS]) * vi;
        acc[2] += vi * vi;
    }
    if (ldiag && (int64_t)blockIdx.x * blockDim.x < d.NS) {      // blocks that saw pivots
        for (int off = 32; off > 0; off >>= 1) {
            pmin = fmin(pmin, __shfl_down(pmin, off, 64));
            pmax = fmax(pmax, __shfl_down(pmax, off, 64));
        }
        if ((threadIdx.x & 63) == 0 && pmax > 0.0) {
            atomicMin(pivmm + 2, (unsigned long long)__double_as_longlong(fmax(pmin, 0.0)));
            atomicMax(pivmm + 3, (unsigned long long)__double_as_longlong(pmax));
        }
        __builtin_amdgcn_s_waitcnt(0);               // performed before this block's ticket
    }
    if (grid_sum<5, 2, 3>(acc, sh, partial, ctr, bs_partial, nbs, 2) && threadIdx.x == 0) {
        out[0] = acc[3]; out[1] = acc[4]; out[4] = acc[0]; out[5] = acc[1]; out[6] = acc[2];
        // a failed factorisation on ANY rank must be seen by all of them (out is summed over the ranks)
        out[7] = info && __hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ? 1.0 : 0.0;
        out[2] = 0.0; out[3] = 0.0;
        if (mailbox) {
            mailbox[0] = acc[3]; mailbox[1] = acc[4]; mailbox[4] = acc[0]; mailbox[5] = acc[1]; mailbox[6] = acc[2];
            for (int q = 0; q < 4; ++q)
                mailbox[40 + q] = __longlong_as_double((long long)__hip_atomic_load(pivmm + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            const int hi = __hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            long long w = hi;
            mailbox[44] = __longlong_as_double(w);
            mailbox_done(mailbox, seq);
        }
    }
}

// ---------------------------------------------------------------- K9 ----
// generic owned dot products: out = sum_{mine} a_i*b_i
__global__ __launch_bounds__(256) void k_dot(int64_t n, const uint8_t *__restrict__ mine,
                                             const double *__restrict__ a, const double *__restrict__ b /* null => 1 */,
                                             double *__restrict__ partial) {
    __shared__ double sh[8];
    double acc[1] = {0};
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        if (!mine || mine[i]) acc[0] += a[i] * (b ? b[i] : 1.0);
    block_sum<1>(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc[0];
}

// y = a*x + b*y2   (any of the pointers may alias y)
__global__ void k_axpby(int64_t n, double a, const double *__restrict__ x, double b,
                        const double *__restrict__ y2, double *__restrict__ y) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = a * x[i] + b * y2[i];
}
// The same (y must not alias x or y2) and, in the same launch, the camera records at y: block c < nc
// builds camera c from the entries it computes itself (another block may still be writing them).
__global__ void k_axpby_cams(DevProblem d, double a, const double *__restrict__ x, double b,
                             const double *__restrict__ y2, double *__restrict__ y, CamRec *__restrict__ cams) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < d.NZ) y[i] = a * x[i] + b * y2[i];
    if (threadIdx.x == blockDim.x - 1 && (int)blockIdx.x < d.nc)
        cam_prep_one(d, [=](int64_t q) { return a * x[q] + b * y2[q]; }, (int)blockIdx.x, cams);
}

// gradient in z layout: g[0..NS) = g_c, g[NS..) = g_p (0 for fixed / unowned)
__global__ void k_gradient(DevProblem d, const double *__restrict__ g_c, const double *__restrict__ gp,
                           double *__restrict__ g) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.NZ) return;
    double v = 0;
    if (d.z_est[i]) v = i < d.NS ? g_c[i] : (d.z_mine[i] ? gp[i - d.NS] : 0.0);
    g[i] = v;
}

// squared column norms in z layout
__global__ void k_jn2(DevProblem d, const double *__restrict__ jn2c, const double *__restrict__ jn2p,
                      double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.NZ) return;
    double v = 0;
    if (d.z_est[i]) v = i < d.NS ? jn2c[i] : (d.z_mine[i] ? jn2p[i - d.NS] : 0.0);
    out[i] = v;
}

// x (reference order) <-> z (internal order)
__global__ void k_scatter_x(int64_t n, const int64_t *__restrict__ x2z, const double *__restrict__ x,
                            double *__restrict__ z) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) z[x2z[i]] = x[i];
}
__global__ void k_gather_x(int64_t n, const int64_t *__restrict__ x2z, const double *__restrict__ z,
                           double *__restrict__ x) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = z[x2z[i]];
}

// per-observation Jacobian blocks in reference order (dbat_hip_jacobian_blocks)
template <int MODEL>
__global__ __launch_bounds__(256) void k_jac_blocks(DevProblem d, const double *__restrict__ z,
                                                    const CamRec *__restrict__ cams,
                                                    double *__restrict__ JEO, double *__restrict__ JOP,
                                                    double *__restrict__ JIO) {
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= d.nobs) return;
    const int cam = d.o_cam[o], pt = d.o_pt[o];
    const CamRec &C = cams[cam];
    const double *q = z + d.NS + 3 * (int64_t)pt;
    const double Q[3] = {q[0], q[1], q[2]};
    double r[2], A[2][6], B[2][3], Cf[2][MAXIO];
    obs_eval<MODEL, true, true>(C, d.nK, d.nP, Q, d.o_uv_raw[2 * o], d.o_uv_raw[2 * o + 1], r, A, B, Cf);
    const int64_t k = d.o_row[o];
    if (JEO) for (int c = 0; c < 6; ++c) { JEO[12 * k + 2 * c] = A[0][c]; JEO[12 * k + 2 * c + 1] = A[1][c]; }
    if (JOP) for (int c = 0; c < 3; ++c) { JOP[6 * k + 2 * c] = B[0][c]; JOP[6 * k + 2 * c + 1] = B[1][c]; }
    if (JIO) for (int c = 0; c < d.nIOrows; ++c) {
        JIO[2 * (int64_t)d.nIOrows * k + 2 * c] = Cf[0][c];
        JIO[2 * (int64_t)d.nIOrows * k + 2 * c + 1] = Cf[1][c];
    }
}


// ... and for a SAMPLE of the observations (dbat_hip_jacobian_sample): pos[i] = processing-order position of the
// i-th requested IP column; the results are indexed by i.  Residual unweighted [mm], as k_residual exports it.
template <int MODEL>
__global__ __launch_bounds__(256) void k_jac_sample(DevProblem d, const double *__restrict__ z,
                                                    const CamRec *__restrict__ cams, int64_t n,
                                                    const int64_t *__restrict__ pos, double *__restrict__ res,
                                                    double *__restrict__ JEO, double *__restrict__ JOP,
                                                    double *__restrict__ JIO) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t o = pos[i];
    const CamRec &C = cams[d.o_cam[o]];
    const double *q = z + d.NS + 3 * (int64_t)d.o_pt[o];
    const double Q[3] = {q[0], q[1], q[2]};
    double r[2], A[2][6], B[2][3], Cf[2][MAXIO];
    obs_eval<MODEL, true, true>(C, d.nK, d.nP, Q, d.o_uv_raw[2 * o], d.o_uv_raw[2 * o + 1], r, A, B, Cf);
    res[2 * i] = r[0]; res[2 * i + 1] = r[1];
    for (int c = 0; c < 6; ++c) { JEO[12 * i + 2 * c] = A[0][c]; JEO[12 * i + 2 * c + 1] = A[1][c]; }
    for (int c = 0; c < 3; ++c) { JOP[6 * i + 2 * c] = B[0][c]; JOP[6 * i + 2 * c + 1] = B[1][c]; }
    for (int c = 0; c < d.nIOrows; ++c) {
        JIO[2 * (int64_t)d.nIOrows * i + 2 * c] = Cf[0][c];
        JIO[2 * (int64_t)d.nIOrows * i + 2 * c + 1] = Cf[1][c];
    }
}

}  // namespace dbat
