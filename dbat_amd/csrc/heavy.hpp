// Heavy and giant object points on the f64 matrix cores (K1, K3 point side, K4, K5 of SURVEY 8(a) for the points
// that do not fit a tile: seen by more cameras than a tile holds -- every point of DBAT's camera-calibration demo,
// demo/camcaldemo.m:56-119 -- or by more than a batch holds).
//
// Until round 5 these points went through k_build / k_build_giant: every observation looped over its point's k
// partners and issued 36 (... NCX^2) global f64 atomics per pair -- a million atomics onto 18 000 addresses for the
// 2 074 observations of the demo.  Here the pair terms are a block-sparse symmetric product on
// v_mfma_f64_16x16x4_f64, and the only atomics are ONE flush per task:
//
//   plan (plan.hpp build_heavy_plan)   the rows of the reduced system these points touch are cut into ROW GROUPS of at
//       most 48 rows (eight cameras = three 16-row blocks; the estimated IO columns and the right-hand side form the last
//       groups).  A point has a slot in every group it touches: 3 k-columns x (16 x row blocks) doubles of Zs.
//   k_heavy_z / k_heavy_z_giant        lane = observation: residual, Jacobian blocks, V = sum B'B, g = sum B'r per point
//       (summed in observation order: no atomics, the same bits every run), R with V^-1 = R R' (point_block_factor),
//       then the observation's rows of Z = (E'B) R into its camera's slot; the IO rows of a point (sums over its
//       observations, again in order) and y = R'g into the slots of the last groups.  Nothing else is written: the
//       camera side (J_c'J_c, J_c'r, column norms) of these observations comes from the camera-major kernels.
//   k_heavy_syrk                        one wave per task = (group i, group j <= i, a run of k-steps of the points in
//       both): S(Gi, Gj) -= sum_p Z_p(Gi) Z_p(Gj)', up to 3 x 3 accumulator blocks in registers, operands straight
//       from Zs (L2: a slot is read once per partner group), the flush by global atomics.  The row of y gives
//       -(W V^-1 g_p) for the reduced right-hand side in the same product.
// Work: 3/4 k-step per point and group pair, nb_i x nb_j products each -- for the demo 45 blocks x 75 k-steps; traffic:
// the flush, 8 bytes per element of a pair's blocks and task.
#pragma once
#include "kernels.hpp"

namespace dbat {

struct HeavyDev {
    const int32_t *obs_dst;          // [untiled observations] Zs index of (slot, k-column 0, first EO row of the camera)
    const uint8_t *obs_ld;           // ... stride between the k-columns of that group
    const uint8_t *obs_ioloc;        // [untiled observations][HV_NIOC] slot of the camera's j-th IO column in its point's IO list
    const int32_t *pt_io0;           // [points + 1]
    const int32_t *io_dst;           // [IO slots]
    const uint8_t *io_ld;
    const int32_t *io_pt;            // [IO slots] point (index from pt0)
    const int32_t *pt_y;             // [points][2]
    const int32_t *grp_nb, *grp_row; // [groups], [groups][48]
    const int32_t *task;             // [tasks][4]
    const int32_t *ops;              // [k-steps][4][2]
    int64_t obs0;
    int32_t pt0, ntasks, max_batch_slots;
};
constexpr int HV_NIOC = 9;

// dynamic LDS of k_heavy_z: [256][9] per-observation terms | [128][9] per-point blocks | the IO rows of the batch's points
// (slots x 3, self-calibration) | deterministic mode: [256][3 (NCX - 6)] the observations' shares of the IO rows
__host__ __device__ constexpr size_t heavy_z_lds_bytes(int ncx, int max_batch_slots, bool deterministic) {
    return ((size_t)256 * 9 + (size_t)128 * 9 + (ncx > 6 ? (size_t)3 * max_batch_slots + (deterministic ? (size_t)256 * 3 * (ncx - 6) : 0) : 0)) * sizeof(double);
}

// The point's block: priors, squared column norms, damping, R (V^-1 = R R'), pivots, y = R'g -- as k_build.
// Vg: V (6) | g (3) on entry; R (6) | y (3) on return.
__device__ __forceinline__ void heavy_point_block(const DevProblem &d, const double *__restrict__ z, int pt, double lambda, int scale,
                                                  double (&Vg)[9], double *__restrict__ Vinv, double *__restrict__ gp,
                                                  double *__restrict__ jn2p, double &pmin, double &pmax) {
    double V[6] = {Vg[0], Vg[1], Vg[2], Vg[3], Vg[4], Vg[5]}, g[3] = {Vg[6], Vg[7], Vg[8]};
    const int64_t zp = d.NS + 3 * (int64_t)pt;
    const int dix[3] = {0, 3, 5};
    double jn[3];
    // (everything the block needs from memory is requested at once: a handful of lanes do this while the workgroup
    // waits, and three dependent round trips per coordinate -- weight, then value and prior -- were 6 of its 8 us)
    double pwv[3], zv[3], pvv[3];
    bool estv[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { pwv[k] = d.any_prior ? d.z_prw[zp + k] : 0.0; zv[k] = z[zp + k]; pvv[k] = d.any_prior ? d.z_prv[zp + k] : 0.0; estv[k] = d.z_est[zp + k] != 0; }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double pw = pwv[k];
        if (pw > 0) { V[dix[k]] += pw; g[k] += pw * (zv[k] - pvv[k]); }
        jn[k] = V[dix[k]];
        jn2p[3 * (int64_t)pt + k] = jn[k];
        if (estv[k]) V[dix[k]] += lambda; else V[dix[k]] = 1.0;
    }
    double inv[6], R[6];
    point_block_factor(V, R, inv);
    {   // diag of chol(V): the leading pivots of the full normal-matrix factor
        const double d0 = sqrt(V[0]), l10 = V[1] / d0, l20 = V[2] / d0;
        const double d1 = sqrt(V[3] - l10 * l10), l21 = (V[4] - l20 * l10) / d1;
        const double d2 = sqrt(V[5] - l20 * l20 - l21 * l21);
        const double dd[3] = {d0, d1, d2};
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (estv[k]) {
                double v = scale ? dd[k] / sqrt(jn[k]) : dd[k];
                v = v == v ? v : 0.0;
                pmin = fmin(pmin, v); pmax = fmax(pmax, v);
            }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) { Vg[k] = R[k]; Vinv[6 * (int64_t)pt + k] = R[k]; }
#pragma unroll
    for (int k = 0; k < 3; ++k) gp[3 * (int64_t)pt + k] = g[k];
    Vg[6] = R[0] * g[0] + R[1] * g[1] + R[2] * g[2];
    Vg[7] = R[3] * g[1] + R[4] * g[2];
    Vg[8] = R[5] * g[2];
}

// One workgroup per batch of whole points (at most 256 observations); lane t <-> observation batch_start[b] + t.
template <int MODEL, int NCX>
__global__ __launch_bounds__(256) void k_heavy_z(DevProblem d, HeavyDev hv, const double *__restrict__ z,
                                                 const CamRec *__restrict__ cams, double lambda, int scale,
                                                 double *__restrict__ Zs, double *__restrict__ Vinv, double *__restrict__ gp,
                                                 double *__restrict__ jn2p, double *__restrict__ partial,
                                                 unsigned long long *__restrict__ pivmm, int batch0) {
    constexpr int NQ = NCX - 6;                      // IO columns of one camera at most
    extern __shared__ double smem[];
    double *red = smem;                              // [256][9]  B'B (6) | B'r (3) of every observation
    double *pinfo = red + 256 * 9;                   // [128][9]  per point: V | g, then R | y
    double *zsum = pinfo + 128 * 9;                  // [IO slots of the batch][3]  the IO rows of Z
    constexpr int WS = 3 * (NQ > 0 ? NQ : 1);
    __shared__ int pseg[128];                        // per point of the batch: first lane | observations << 16
    __shared__ unsigned char lio[256][NQ > 0 ? NQ : 1];
    __shared__ double sh[8];
    const int t = threadIdx.x;
    const int64_t o0 = d.batch_start[batch0 + blockIdx.x];
    const int nobs = (int)(d.batch_start[batch0 + blockIdx.x + 1] - o0);
    const bool active = t < nobs;
    const int64_t o = o0 + t;
    const int npb = nobs > 0 ? (int)d.o_pidx[o0 + nobs - 1] + 1 : 0;      // points of the batch
    // DBAT_HIP_ABLATE & 32 (measurement build): phase clocks of thread 0 (100 MHz ticks), summed over the batches into g_tile2_prof[8 ..]
    const bool prof = DBAT_ABLATE(d, 32) && t == 0;
    long long tlast = prof ? wall_clock64() : 0;
    auto lap = [&](int i) { if (prof) { const long long now = wall_clock64(); atomicAdd(&g_tile2_prof[8 + i], (unsigned long long)(now - tlast)); tlast = now; } };
    double r[2] = {0, 0}, E[2][NCX], B[2][3];
    int pt = 0, seg_start = 0, pidx = -1, ncol = 6;
    int32_t zdst = 0; int zld = 16, ioslot0 = 0;
    if (active) {
        const int cam = d.o_cam[o];
        pt = d.o_pt[o];
        const uint32_t sg = d.o_seg[o];
        seg_start = sg & 0xFFFF;
        pidx = d.o_pidx[o];
        const CamRec &C = cams[cam];
        ncol = NQ > 0 ? min(C.ncol, NCX) : 6;
        eval_obs_cols_n<MODEL, NCX>(d, C, z, o, pt, r, E, B);
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
        if (t == seg_start) pseg[pidx] = (int)sg;
        zdst = hv.obs_dst[o - hv.obs0]; zld = hv.obs_ld[o - hv.obs0];
        if constexpr (NQ > 0) {
            const uint8_t *il = hv.obs_ioloc + (size_t)(o - hv.obs0) * HV_NIOC;
#pragma unroll
            for (int j = 0; j < NQ; ++j) lio[t][j] = il[j];
            ioslot0 = hv.pt_io0[pt - hv.pt0];
        }
    }
    __syncthreads();
    lap(0);
    // V and g of every point: its observations' terms added in their order (one thread per point and element)
    for (int idx = t; idx < npb * 9; idx += 256) {
        const int pi = idx / 9, v = idx - 9 * pi;
        const int s0 = pseg[pi] & 0xFFFF, len = (unsigned)pseg[pi] >> 16;
        double sum = 0.0;
        for (int j = 0; j < len; ++j) sum += red[(size_t)(s0 + j) * 9 + v];
        pinfo[pi * 9 + v] = sum;
    }
    __syncthreads();
    lap(1);
    double pmin = 1e300, pmax = 0.0;
    if (active && t == seg_start) {
        double Vg[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) Vg[k] = pinfo[pidx * 9 + k];
        heavy_point_block(d, z, pt, lambda, scale, Vg, Vinv, gp, jn2p, pmin, pmax);
#pragma unroll
        for (int k = 0; k < 9; ++k) pinfo[pidx * 9 + k] = Vg[k];
        const int32_t *py = hv.pt_y + 2 * (size_t)(pt - hv.pt0);
        Zs[py[0]] = Vg[6]; Zs[py[0] + py[1]] = Vg[7]; Zs[py[0] + 2 * py[1]] = Vg[8];
    }
    __syncthreads();
    lap(2);
    // the observation's rows of Z = (E'B) R: the six EO rows into its camera's slot
    double r00 = 0, r10 = 0, r20 = 0, r11 = 0, r21 = 0, r22 = 0;
    if (active) {
        const double *pi = pinfo + pidx * 9;
        r00 = pi[0]; r10 = pi[1]; r20 = pi[2]; r11 = pi[3]; r21 = pi[4]; r22 = pi[5];
        const int32_t dst = zdst;
        const int ld = zld;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const double w0 = E[0][a] * B[0][0] + E[1][a] * B[1][0];
            const double w1 = E[0][a] * B[0][1] + E[1][a] * B[1][1];
            const double w2 = E[0][a] * B[0][2] + E[1][a] * B[1][2];
            Zs[dst + a] = w0 * r00 + w1 * r10 + w2 * r20; Zs[dst + ld + a] = w1 * r11 + w2 * r21; Zs[dst + 2 * ld + a] = w2 * r22;
        }
    }
    lap(3);
    if constexpr (NQ > 0) {
        // IO rows: sums over a point's observations, Z_io(slot) = sum_j (E_io,j' B_j) R.
        //   default: every observation adds its terms to the point's slots in LDS (ds_add_f64; the lanes of a point hit the
        //     same addresses, which the LDS serialises) -- no staging array, 30 KB of LDS per workgroup where the ordered sums
        //     need 85, and every lane works.  (Adding up inside the wave first -- segmented sums by lane shuffles -- took
        //     as long as the contended atomics: 12.4 against 11.0 us per batch of the dense scene; not kept.);
        //   deterministic mode: the terms are staged per observation and one thread per (slot, k-column) adds the point's
        //     observations in their order (the same bits every run; the launch asks for the staging array).
        const int hp_first = d.o_pt[o0] - hv.pt0;
        const int gs0 = hv.pt_io0[hp_first], gs1 = hv.pt_io0[hp_first + npb];
        const int nout = (gs1 - gs0) * 3;
        const bool det = d.deterministic != 0;
        double *wio = zsum + 3 * (size_t)hv.max_batch_slots;      // [256][3 NQ], deterministic mode only
        if (!det) {
            for (int idx = t; idx < nout; idx += 256) zsum[idx] = 0.0;
            __syncthreads();
        }
        if (det) {
            if (active) {
#pragma unroll
                for (int a = 6; a < NCX; ++a)
                    if (a < ncol) {
                        const double w0 = E[0][a] * B[0][0] + E[1][a] * B[1][0];
                        const double w1 = E[0][a] * B[0][1] + E[1][a] * B[1][1];
                        const double w2 = E[0][a] * B[0][2] + E[1][a] * B[1][2];
                        double *wl = wio + (size_t)t * WS + 3 * (a - 6);
                        wl[0] = w0 * r00 + w1 * r10 + w2 * r20; wl[1] = w1 * r11 + w2 * r21; wl[2] = w2 * r22;
                    }
            }
        } else if (active) {
#pragma unroll
            for (int a = 6; a < NCX; ++a)
                if (a < ncol) {
                    const double w0 = E[0][a] * B[0][0] + E[1][a] * B[1][0];
                    const double w1 = E[0][a] * B[0][1] + E[1][a] * B[1][1];
                    const double w2 = E[0][a] * B[0][2] + E[1][a] * B[1][2];
                    double *zs = zsum + 3 * (ioslot0 - gs0 + (int)lio[t][a - 6]);
                    atomic_add_f64(zs, w0 * r00 + w1 * r10 + w2 * r20); atomic_add_f64(zs + 1, w1 * r11 + w2 * r21); atomic_add_f64(zs + 2, w2 * r22);
                }
        }
        __syncthreads();
        for (int idx = t; idx < nout; idx += 256) {
            const int gs = gs0 + idx / 3, c = idx % 3;
            double sum;
            if (!det) sum = zsum[idx];
            else {
                const int hp = hv.io_pt[gs], pi = hp - hp_first, s = gs - hv.pt_io0[hp];
                const int s0 = pseg[pi] & 0xFFFF, len = (unsigned)pseg[pi] >> 16;
                sum = 0.0;
                for (int j = 0; j < len; ++j) {
                    const int l = s0 + j;
                    int jj = -1;
                    if (s < NQ && lio[l][s < NQ ? s : 0] == s) jj = s;      // (the usual case: every camera of the point has the same IO columns)
                    else {
#pragma unroll
                        for (int q = 0; q < NQ; ++q) if (lio[l][q] == s) jj = q;
                    }
                    if (jj >= 0) sum += wio[(size_t)l * WS + 3 * jj + c];
                }
            }
            Zs[hv.io_dst[gs] + c * (int)hv.io_ld[gs]] = sum;
        }
    }
    lap(4);
    double acc[1] = {r[0] * r[0] + r[1] * r[1]};
    block_sum<1>(acc, sh);
    if (t == 0) partial[blockIdx.x] = acc[0];
    lap(5);
    for (int off = 32; off > 0; off >>= 1) {
        pmin = fmin(pmin, __shfl_down(pmin, off, 64));
        pmax = fmax(pmax, __shfl_down(pmax, off, 64));
    }
    if ((t & 63) == 0 && pmax > 0.0) {
        atomicMin(pivmm, (unsigned long long)__double_as_longlong(pmin));
        atomicMax(pivmm + 1, (unsigned long long)__double_as_longlong(pmax));
    }
}

// The same for an object point with more observations than a batch holds: one workgroup per point, the observations in
// rounds of blockDim.  W = E'B of every observation waits in the scratch rows of k_build_giant (DevProblem::giant_W)
// for the point's factor.
template <int MODEL, int NCX>
__global__ __launch_bounds__(256) void k_heavy_z_giant(DevProblem d, HeavyDev hv, const double *__restrict__ z,
                                                       const CamRec *__restrict__ cams, double lambda, int scale,
                                                       double *__restrict__ Zs, double *__restrict__ Vinv, double *__restrict__ gp,
                                                       double *__restrict__ jn2p, double *__restrict__ partial,
                                                       unsigned long long *__restrict__ pivmm) {
    constexpr int NQ = NCX - 6;
    __shared__ double sh[9 * 4];
    __shared__ double pin[9];
    const int t = threadIdx.x, BT = blockDim.x;
    const int64_t o0 = d.giant_start[blockIdx.x], o1 = d.giant_start[blockIdx.x + 1];
    const int k = (int)(o1 - o0);
    const int strideW = d.ncolmax * 3;
    double *Wg = d.giant_W + (o0 - d.giant_start[0]) * strideW;
    const int pt = d.o_pt[o0];
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, rr = 0.0;
    for (int i = t; i < k; i += BT) {
        const int64_t o = o0 + i;
        const CamRec &C = cams[d.o_cam[o]];
        const int ncol = NQ > 0 ? min(C.ncol, NCX) : 6;
        double r[2], E[2][NCX], B[2][3];
        eval_obs_cols_n<MODEL, NCX>(d, C, z, o, pt, r, E, B);
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
#pragma unroll
        for (int a = 0; a < NCX; ++a)
            if (a < ncol) {
                wl[3 * a] = E[0][a] * B[0][0] + E[1][a] * B[1][0];
                wl[3 * a + 1] = E[0][a] * B[0][1] + E[1][a] * B[1][1];
                wl[3 * a + 2] = E[0][a] * B[0][2] + E[1][a] * B[1][2];
            }
    }
    block_sum<9>(acc, sh);
    if (t == 0) {
        double pmin = 1e300, pmax = 0.0;
        heavy_point_block(d, z, pt, lambda, scale, acc, Vinv, gp, jn2p, pmin, pmax);
#pragma unroll
        for (int q = 0; q < 9; ++q) pin[q] = acc[q];
        const int32_t *py = hv.pt_y + 2 * (size_t)(pt - hv.pt0);
        Zs[py[0]] = acc[6]; Zs[py[0] + py[1]] = acc[7]; Zs[py[0] + 2 * py[1]] = acc[8];
        if (pmax > 0.0) {
            atomicMin(pivmm, (unsigned long long)__double_as_longlong(pmin));
            atomicMax(pivmm + 1, (unsigned long long)__double_as_longlong(pmax));
        }
    }
    __threadfence_block();
    __syncthreads();                                 // pin and the scratch rows of the whole point are visible
    const double r00 = pin[0], r10 = pin[1], r20 = pin[2], r11 = pin[3], r21 = pin[4], r22 = pin[5];
    for (int i = t; i < k; i += BT) {                // W -> Z = W R: EO rows to the camera's slot, IO rows stay in the scratch row
        const int64_t o = o0 + i;
        const int nci = NQ > 0 ? min(cams[d.o_cam[o]].ncol, NCX) : 6;
        double *wi = Wg + (size_t)i * strideW;
        const int32_t dst = hv.obs_dst[o - hv.obs0];
        const int ld = hv.obs_ld[o - hv.obs0];
        for (int a = 0; a < nci; ++a) {
            const double w0 = wi[3 * a], w1 = wi[3 * a + 1], w2 = wi[3 * a + 2];
            const double z0 = w0 * r00 + w1 * r10 + w2 * r20, z1 = w1 * r11 + w2 * r21, z2 = w2 * r22;
            if (a < 6) { Zs[dst + a] = z0; Zs[dst + ld + a] = z1; Zs[dst + 2 * ld + a] = z2; }
            else { wi[3 * a] = z0; wi[3 * a + 1] = z1; wi[3 * a + 2] = z2; }
        }
    }
    if constexpr (NQ > 0) {
        __threadfence_block();
        __syncthreads();
        // IO rows: one wave per (slot, k-column) in turn; every lane adds its observations in their order, then the wave
        // (a fixed tree: the same bits every run)
        const int hp = pt - hv.pt0;
        const int gs0 = hv.pt_io0[hp], nsl = hv.pt_io0[hp + 1] - gs0;
        const int wave = t >> 6, lane = t & 63, nw = BT >> 6;
        for (int idx = wave; idx < nsl * 3; idx += nw) {
            const int s = idx / 3, c = idx - 3 * s;
            double sum = 0.0;
            for (int i = lane; i < k; i += 64) {
                const uint8_t *il = hv.obs_ioloc + (size_t)(o0 + i - hv.obs0) * HV_NIOC;
                int jj = -1;
#pragma unroll
                for (int q = 0; q < NQ; ++q) if (il[q] == s) jj = q;
                if (jj >= 0) sum += Wg[(size_t)i * strideW + 3 * (6 + jj) + c];
            }
            sum = wave_sum_f64(sum);
            if (lane == 0) Zs[hv.io_dst[gs0 + s] + c * (int)hv.io_ld[gs0 + s]] = sum;
        }
    }
    double accr[1] = {rr};
    __syncthreads();
    block_sum<1>(accr, sh);
    if (t == 0) partial[blockIdx.x] = accr[0];
}

// One wave per task.  Lane l supplies row (l & 15) of a 16-row block and k-column (l >> 4) of the k-step; the k-step's
// four k-columns are consecutive columns of consecutive points of the pair list (hv.ops: their places in Zs, or -1).
// Output element e of lane l of block (r1, r2): row 16 r1 + (l >> 4) + 4 e of group i, row 16 r2 + (l & 15) of group j.
__global__ __launch_bounds__(256) void k_heavy_syrk(DevProblem d, HeavyDev hv, const double *__restrict__ Zs,
                                                    double *__restrict__ S, double *__restrict__ g_red) {
    const int lane = threadIdx.x & 63;
    // Workgroups are dealt round-robin over the eight XCDs (b and b + 8 share one: observed, for speed only).  The task
    // list is ordered so that neighbours read the same slots (plan.hpp): XCD x takes a CONTIGUOUS run of it, and a slot
    // comes into that XCD's L2 once for all its partner groups.
    const int nwg = gridDim.x, xcd = blockIdx.x & 7;
    const int wg = xcd * (nwg >> 3) + min(xcd, nwg & 7) + (blockIdx.x >> 3);
    const int task = __builtin_amdgcn_readfirstlane(wg * 4 + (threadIdx.x >> 6));
    if (task >= hv.ntasks) return;
    const int gi = hv.task[4 * task], gj = hv.task[4 * task + 1];
    const int ks0 = hv.task[4 * task + 2], nks = hv.task[4 * task + 3];
    const int nbi = hv.grp_nb[gi], nbj = hv.grp_nb[gj];
    const bool diag = gi == gj;
    const int row = lane & 15, kk = lane >> 4;
    const int2 *ops = reinterpret_cast<const int2 *>(hv.ops) + (int64_t)ks0 * 4 + kk;
    mfma_d4 acc[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) acc[q] = mfma_d4{0, 0, 0, 0};
    // operands of k-step ks: three row blocks of group i, three of group j (the diagonal pairs reuse the former)
    auto load = [&](int2 op, double (&a)[3], double (&b)[3]) {
        const double *pa = Zs + (op.x >= 0 ? op.x : 0) + row, *pb = Zs + (op.y >= 0 ? op.y : 0) + row;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            a[q] = (q < nbi && op.x >= 0) ? pa[16 * q] : 0.0;
            b[q] = (!diag && q < nbj && op.y >= 0) ? pb[16 * q] : 0.0;
        }
    };
    const int2 none = int2{-1, -1};
    // the places two k-steps ahead, the operands one k-step ahead of the products
    int2 op1 = nks > 1 ? ops[4] : none, op2 = nks > 2 ? ops[8] : none;
    double a0[3], b0[3], a1[3], b1[3];
    load(ops[0], a0, b0);
    for (int ks = 0; ks < nks; ks += 2) {
        load(op1, a1, b1);
        op1 = ks + 3 < nks ? ops[4 * (ks + 3)] : none;
#pragma unroll
        for (int r1 = 0; r1 < 3; ++r1)
            if (r1 < nbi) {
#pragma unroll
                for (int r2 = 0; r2 < 3; ++r2)
                    if (diag ? r2 <= r1 : r2 < nbj)
                        acc[3 * r1 + r2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[r1], diag ? a0[r2] : b0[r2], acc[3 * r1 + r2], 0, 0, 0);
            }
        load(op2, a0, b0);
        op2 = ks + 4 < nks ? ops[4 * (ks + 4)] : none;
        if (ks + 1 < nks) {
#pragma unroll
            for (int r1 = 0; r1 < 3; ++r1)
                if (r1 < nbi) {
#pragma unroll
                    for (int r2 = 0; r2 < 3; ++r2)
                        if (diag ? r2 <= r1 : r2 < nbj)
                            acc[3 * r1 + r2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[r1], diag ? a1[r2] : b1[r2], acc[3 * r1 + r2], 0, 0, 0);
                }
        }
    }
    // flush: S(row_i, row_j) -= acc, reduced right-hand side -= (row of y) . Z
    const bool det = d.deterministic != 0;
    const int32_t *rowi = hv.grp_row + (size_t)gi * 48, *rowj = hv.grp_row + (size_t)gj * 48;
    const int NSi = (int)d.NS;
#pragma unroll
    for (int r2 = 0; r2 < 3; ++r2) {
        if (r2 >= nbj) break;
        const int cj = rowj[16 * r2 + row];
        if (cj < 0 || cj == NSi) continue;           // padding; (y, y) is not an element of anything
        const double uj = det ? d.det_u[cj] : 0.0;
#pragma unroll
        for (int r1 = 0; r1 < 3; ++r1) {
            if (r1 >= nbi || (diag && r1 < r2)) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int li = 16 * r1 + kk + 4 * e;
                if (diag && li < 16 * r2 + row) continue;         // lower triangle of the diagonal pair
                const int ri = rowi[li];
                double v = -acc[3 * r1 + r2][e];
                if (ri < 0 || v == 0.0) continue;
                if (det) v = det_round(v, d.det_u[ri], uj);
                if (ri == NSi) atomic_add_f64(g_red + cj, v);
                else atomic_add_f64(S + (int64_t)cj * d.ldS + ri, v);
            }
        }
    }
}

}  // namespace dbat
