// k_build_sig: Schur complement of the tiled object points by SIGNATURE GROUPS
// (K1, K3 point side, K4, K5 of SURVEY 8(a); fixed interior orientation).
//
// Object points that are seen by exactly the same k cameras update the same
// 6k x 6k block of the reduced camera system.  The plan (plan.hpp) orders the
// points of a tile so that such points follow each other and cuts the runs into
// chunks of at most 64 points.  For a chunk the product
//       S_chunk = sum_p Z_p Z_p',   Z_p = W_p R_p  (6k x 3),  V_p^-1 = R_p R_p'
// is DENSE over its own 6k rows: the wave that owns the chunk keeps it in the
// accumulators of ceil((6k+1)/16) row blocks on the f64 matrix cores
// (v_mfma_f64_16x16x4_f64, A = B = the same operand registers: lower triangle
// only) -- nothing is multiplied that the sparse product does not contain, apart
// from the padding of 6k to a multiple of 16 and of 3 columns per point to k-steps
// of 4.  Row 6k of the operand holds y_p = R_p' g_p, so the same product also
// yields -(W V^-1 g_p) for the reduced right-hand side.
//
// One workgroup (four waves, one per SIMD) per tile:
//   * the tile's camera records are staged in LDS once;
//   * the tile's block of S (at most 21 cameras, lower triangle packed, 64 KB)
//     lives in LDS; a finished chunk is added to it with ds_add_f64 and the tile
//     goes to HBM once, with global f64 atomics, as in the other tile kernels;
//   * a wave takes chunks from the tile's list through an LDS counter.
// A chunk in two passes:
//   pass 1, lane = object point: loop over the k cameras (uniform per chunk:
//     broadcast reads of the staged record, coalesced (u,v) from the slot-major
//     copy), residual r and point block B per observation; V = sum B'B (+ prior,
//     + lambda), g = sum B'r in registers -- no atomics, no cross-lane traffic;
//     V^-1, its Cholesky factor R, the pivots; V^-1, g and the squared column
//     norms go to HBM for the back-substitution, R | R'g | Q to LDS;
//   pass 2, lane = observation, rounds of floor(64/k) <= 6 points: camera-side
//     block E, W = E'B, Z = W R into the wave's operand panel [rows][18 (+2)
//     k-columns], then ceil(3 n/4) k-steps of the block products.
// LDS reads of the operands are conflict free (row stride 22 doubles).
#pragma once
#include "kernels.hpp"

namespace dbat {

constexpr int SIG_LDK = 22;          // k-columns per panel row: 18 used (6 points), stride = 2 mod 4 doubles
constexpr int SIG_PPR = 6;           // points per round of pass 2
constexpr int SIG_PV = 13;           // per point in LDS: R (6) | R'g (3) | Q (3) | est bits
constexpr int SIG_CAMW = 58;         // doubles of a CamRec that the fixed-IO evaluation reads (.. w[2]) + eo_est
constexpr int SIG_STILE = 8064;      // 126*127/2 = 8001 packed lower triangle of the tile, padded

struct SigLds {                      // static part
    int next_chunk, abort_;
    int lc[4][16];                   // tile-local camera of every slot of the wave's chunk
};

__host__ __device__ constexpr size_t sig_lds_bytes(int RB) {
    return ((size_t)SIG_STILE + 128 + 21 * SIG_CAMW + 4 * ((size_t)RB * 16 * SIG_LDK + 64 * SIG_PV)) * sizeof(double);
}

template <int MODEL, int RB>
__global__ __launch_bounds__(256) void k_build_sig(DevProblem d, const double *__restrict__ z,
                                                   const CamRec *__restrict__ cams, double lambda, int scale,
                                                   double *__restrict__ S, double *__restrict__ g_red,
                                                   double *__restrict__ Vinv, double *__restrict__ gp,
                                                   double *__restrict__ jn2p, double *__restrict__ r_w,
                                                   double *__restrict__ partial, unsigned long long *__restrict__ pivmm,
                                                   const int32_t *__restrict__ sg_chunk,
                                                   const int32_t *__restrict__ sg_tile_chunk0,
                                                   const uint8_t *__restrict__ sg_lc, const double *__restrict__ sg_uv,
                                                   const double *__restrict__ sg_w) {
    constexpr int NBLK = RB * (RB + 1) / 2, PROWS = RB * 16, LDK = SIG_LDK;
    extern __shared__ double smem[];
    double *stile = smem;                            // packed lower triangle of the tile's block of S (negated sum)
    double *vt = stile + SIG_STILE;                  // [128] -(W V^-1 g) by tile row
    double *camw = vt + 128;                         // [21][SIG_CAMW]
    double *wave_base = camw + 21 * SIG_CAMW;
    __shared__ SigLds sy;
    __shared__ double sh[8];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    double *pan = wave_base + (size_t)wave * (PROWS * LDK + 64 * SIG_PV);   // [PROWS][LDK]
    double *pv = pan + PROWS * LDK;                                        // [64][SIG_PV]
    const int tile = d.tile_order[blockIdx.x];
    const int c0 = d.tile_cam_start[tile];
    const int ncam = d.tile_cam_start[tile + 1] - c0;
    const int nrows = 6 * ncam;
    const int ch0 = sg_tile_chunk0[tile], ch1 = sg_tile_chunk0[tile + 1];
    for (int i = t; i < SIG_STILE + 128; i += 256) stile[i] = 0.0;
    for (int i = t; i < 4 * (PROWS * LDK + 64 * SIG_PV); i += 256) wave_base[i] = 0.0;
    for (int i = t; i < ncam * SIG_CAMW; i += 256) {
        const int c = i / SIG_CAMW, f = i - c * SIG_CAMW;
        const CamRec &C = cams[d.tile_cams[c0 + c]];
        camw[i] = f < SIG_CAMW - 1 ? reinterpret_cast<const double *>(&C)[f] : (double)C.eo_est;
    }
    if (t == 0) { sy.next_chunk = ch0; sy.abort_ = 0; }
    __syncthreads();
    double pmin = 1e300, pmax = 0.0, rr = 0.0;
    const int dix[3] = {0, 3, 5};
    for (;;) {
        int ch = 0;
        if (lane == 0) ch = atomicAdd(&sy.next_chunk, 1);
        ch = __builtin_amdgcn_readfirstlane(ch);
        if (ch >= ch1) break;
        const int32_t *cd = sg_chunk + 8 * (int64_t)ch;
        const int pt0 = cd[0], npts = cd[1], k = cd[2], obs0 = cd[3], gm = cd[4], gi0 = cd[5], uv0 = cd[6], lc0 = cd[7];
        if (lane < k) sy.lc[wave][lane] = sg_lc[lc0 + lane];
        for (int i = lane; i < PROWS * LDK; i += 64) pan[i] = 0.0;     // rows / k-columns this chunk does not write
        __builtin_amdgcn_wave_barrier();
        // ------------------------------------------------------------ pass 1: lane = object point
        {
            const bool act = lane < npts;
            const int pt = pt0 + (act ? lane : 0);
            const int64_t zp = d.NS + 3 * (int64_t)pt;
            double Q[3] = {z[zp], z[zp + 1], z[zp + 2]};
            const unsigned est = (d.z_est[zp] ? 1u : 0u) | (d.z_est[zp + 1] ? 2u : 0u) | (d.z_est[zp + 2] ? 4u : 0u);
            double V[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
            for (int j = 0; j < k; ++j) {
                const int lc = sy.lc[wave][j];
                const CamRec &C = *reinterpret_cast<const CamRec *>(camw + lc * SIG_CAMW);
                const int64_t q = uv0 + (int64_t)j * gm + gi0 + (act ? lane : 0);
                const double uu = sg_uv[2 * q], vv = sg_uv[2 * q + 1];
                const double w0 = sg_w ? sg_w[2 * q] : C.w[0], w1 = sg_w ? sg_w[2 * q + 1] : C.w[1];
                double r[2], A[2][6], B[2][3], Cf[2][MAXIO];
                obs_eval<MODEL, true, false>(C, d.nK, d.nP, Q, uu, vv, r, A, B, Cf);    // A is dead code here
                r[0] *= w0; r[1] *= w1;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const double m = ((est >> c) & 1u) ? 1.0 : 0.0;
                    B[0][c] *= w0 * m; B[1][c] *= w1 * m;
                }
                if (act) {
                    const int64_t o = obs0 + (int64_t)lane * k + j;
                    r_w[2 * o] = r[0]; r_w[2 * o + 1] = r[1];
                    rr += r[0] * r[0] + r[1] * r[1];
                }
                V[0] += B[0][0] * B[0][0] + B[1][0] * B[1][0];
                V[1] += B[0][0] * B[0][1] + B[1][0] * B[1][1];
                V[2] += B[0][0] * B[0][2] + B[1][0] * B[1][2];
                V[3] += B[0][1] * B[0][1] + B[1][1] * B[1][1];
                V[4] += B[0][1] * B[0][2] + B[1][1] * B[1][2];
                V[5] += B[0][2] * B[0][2] + B[1][2] * B[1][2];
                g[0] += B[0][0] * r[0] + B[1][0] * r[1];
                g[1] += B[0][1] * r[0] + B[1][1] * r[1];
                g[2] += B[0][2] * r[0] + B[1][2] * r[1];
            }
            if (act) {
                double jn[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const double pw = d.z_prw[zp + c];
                    if (pw > 0) { V[dix[c]] += pw; g[c] += pw * (Q[c] - d.z_prv[zp + c]); }
                    jn[c] = V[dix[c]];
                    jn2p[3 * (int64_t)pt + c] = jn[c];
                    if ((est >> c) & 1u) V[dix[c]] += lambda; else V[dix[c]] = 1.0;
                }
                double inv[6];
                const double c00 = V[3] * V[5] - V[4] * V[4];
                const double c01 = V[2] * V[4] - V[1] * V[5];
                const double c02 = V[1] * V[4] - V[2] * V[3];
                const double det = V[0] * c00 + V[1] * c01 + V[2] * c02;
                const double id = fast_rcp(det);
                inv[0] = c00 * id; inv[1] = c01 * id; inv[2] = c02 * id;
                inv[3] = (V[0] * V[5] - V[2] * V[2]) * id;
                inv[4] = (V[1] * V[2] - V[0] * V[4]) * id;
                inv[5] = (V[0] * V[3] - V[1] * V[1]) * id;
                {   // pivots of the point block (CHOLMOD's rcond estimate)
                    const double r0 = fast_rcp(V[0]);
                    const double d1s = V[3] - V[1] * V[1] * r0;
                    const double tt = V[4] - V[2] * V[1] * r0;
                    const double d2s = V[5] - V[2] * V[2] * r0 - tt * tt * fast_rcp(d1s);
                    const double dd[3] = {V[0], d1s, d2s};
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        if ((est >> c) & 1u) {
                            double v = scale ? dd[c] * fast_rcp(jn[c]) : dd[c];
                            v = v > 0.0 ? v : 0.0;
                            pmin = fmin(pmin, v); pmax = fmax(pmax, v);
                        }
                }
#pragma unroll
                for (int c = 0; c < 6; ++c) Vinv[6 * (int64_t)pt + c] = inv[c];
#pragma unroll
                for (int c = 0; c < 3; ++c) gp[3 * (int64_t)pt + c] = g[c];
                // V^-1 = R R', R lower triangular
                const double r00 = sqrt(inv[0]), ir00 = fast_rcp(r00);
                const double r10 = inv[1] * ir00, r20 = inv[2] * ir00;
                const double r11 = sqrt(inv[3] - r10 * r10);
                const double r21 = (inv[4] - r20 * r10) * fast_rcp(r11);
                const double r22 = sqrt(inv[5] - r20 * r20 - r21 * r21);
                double *pp = pv + lane * SIG_PV;
                pp[0] = r00; pp[1] = r10; pp[2] = r20; pp[3] = r11; pp[4] = r21; pp[5] = r22;
                pp[6] = r00 * g[0] + r10 * g[1] + r20 * g[2];                  // y = R' g
                pp[7] = r11 * g[1] + r21 * g[2];
                pp[8] = r22 * g[2];
                pp[9] = Q[0]; pp[10] = Q[1]; pp[11] = Q[2];
                pp[12] = (double)est;
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ------------------------------------------------------------ pass 2: lane = observation
        mfma_d4 acc[NBLK];
#pragma unroll
        for (int s = 0; s < NBLK; ++s) acc[s] = mfma_d4{0, 0, 0, 0};
        const int ppr = min(SIG_PPR, 64 / k);        // points per round
        const int ir = lane / k, j = lane - ir * k;
        const bool lane_on = ir < ppr;
        const int rbk = (6 * k + 1 + 15) >> 4;       // row blocks this chunk needs
        const int lcj = lane_on ? sy.lc[wave][j] : 0;
        const CamRec &C = *reinterpret_cast<const CamRec *>(camw + lcj * SIG_CAMW);
        const unsigned eo_est = (unsigned)camw[lcj * SIG_CAMW + SIG_CAMW - 1];
        for (int p0 = 0; p0 < npts; p0 += ppr) {
            const int i = p0 + ir;
            const bool on = lane_on && i < npts;
            double Zr[6][3];
#pragma unroll
            for (int a = 0; a < 6; ++a) Zr[a][0] = Zr[a][1] = Zr[a][2] = 0.0;
            double y3[3] = {0, 0, 0};
            if (on) {
                const double *pp = pv + i * SIG_PV;
                const double Q[3] = {pp[9], pp[10], pp[11]};
                const unsigned est = (unsigned)pp[12];
                const int64_t q = uv0 + (int64_t)j * gm + gi0 + i;
                const double uu = sg_uv[2 * q], vv = sg_uv[2 * q + 1];
                const double w0 = sg_w ? sg_w[2 * q] : C.w[0], w1 = sg_w ? sg_w[2 * q + 1] : C.w[1];
                double r[2], A[2][6], B[2][3], Cf[2][MAXIO];
                obs_eval<MODEL, true, false>(C, d.nK, d.nP, Q, uu, vv, r, A, B, Cf);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const double m = ((est >> c) & 1u) ? 1.0 : 0.0;
                    B[0][c] *= w0 * m; B[1][c] *= w1 * m;
                }
                const double r00 = pp[0], r10 = pp[1], r20 = pp[2], r11 = pp[3], r21 = pp[4], r22 = pp[5];
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    const double m = ((eo_est >> a) & 1u) ? 1.0 : 0.0;
                    const double e0 = A[0][a] * w0 * m, e1 = A[1][a] * w1 * m;
                    const double wa0 = e0 * B[0][0] + e1 * B[1][0];
                    const double wa1 = e0 * B[0][1] + e1 * B[1][1];
                    const double wa2 = e0 * B[0][2] + e1 * B[1][2];
                    Zr[a][0] = wa0 * r00 + wa1 * r10 + wa2 * r20;
                    Zr[a][1] = wa1 * r11 + wa2 * r21;
                    Zr[a][2] = wa2 * r22;
                }
                y3[0] = pp[6]; y3[1] = pp[7]; y3[2] = pp[8];
            }
            if (lane_on) {                           // lanes of missing points overwrite the previous round with zeros
                double *pr = pan + (6 * j) * LDK + 3 * ir;
#pragma unroll
                for (int a = 0; a < 6; ++a) { pr[a * LDK] = Zr[a][0]; pr[a * LDK + 1] = Zr[a][1]; pr[a * LDK + 2] = Zr[a][2]; }
                if (j == 0) { double *py = pan + (6 * k) * LDK + 3 * ir; py[0] = y3[0]; py[1] = y3[1]; py[2] = y3[2]; }
            }
            lds_fence();
            __builtin_amdgcn_wave_barrier();
            const int ksteps = (3 * min(ppr, npts - p0) + 3) >> 2;
            const double *zr = pan + (lane & 15) * LDK + (lane >> 4);
            for (int ks = 0; ks < ksteps; ++ks) {
                double op[RB];
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) op[rb] = rb < rbk ? zr[rb * 16 * LDK + 4 * ks] : 0.0;
#pragma unroll
                for (int r1 = 0; r1 < RB; ++r1)
                    if (r1 < rbk) {
#pragma unroll
                        for (int r2 = 0; r2 <= r1; ++r2)
                            acc[r1 * (r1 + 1) / 2 + r2] =
                                __builtin_amdgcn_mfma_f64_16x16x4f64(op[r1], op[r2], acc[r1 * (r1 + 1) / 2 + r2], 0, 0, 0);
                    }
            }
            lds_fence();
            __builtin_amdgcn_wave_barrier();
        }
        // ------------------------------------------------------------ chunk -> tile (LDS atomics)
#pragma unroll
        for (int r1 = 0; r1 < RB; ++r1)
            if (r1 < rbk) {
#pragma unroll
                for (int r2 = 0; r2 <= r1; ++r2) {
                    const int lcol = 16 * r2 + (lane & 15);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int lr = 16 * r1 + (lane >> 4) + 4 * e;
                        const double v = acc[r1 * (r1 + 1) / 2 + r2][e];
                        if (lcol <= lr && lr <= 6 * k && lcol < 6 * k && v != 0.0) {
                            const int sc = lcol / 6, tc = 6 * sy.lc[wave][sc] + (lcol - 6 * sc);
                            if (lr == 6 * k) atomic_add_f64(vt + tc, -v);      // row of y: W V^-1 g
                            else {
                                const int sr = lr / 6, tr = 6 * sy.lc[wave][sr] + (lr - 6 * sr);
                                atomic_add_f64(stile + tr * (tr + 1) / 2 + tc, v);
                            }
                        }
                    }
                }
            }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // ---------------------------------------------------------------- tile -> HBM
    for (int tc = wave; tc < nrows; tc += 4) {       // one column per wave: consecutive lanes, consecutive rows
        const int64_t gcol = 6 * (int64_t)d.tile_cams[c0 + tc / 6] + tc % 6;
        for (int tr = tc + lane; tr < nrows; tr += 64) {
            const double v = stile[tr * (tr + 1) / 2 + tc];
            if (v != 0.0) atomic_add_f64(S + gcol * d.ldS + 6 * (int64_t)d.tile_cams[c0 + tr / 6] + tr % 6, -v);
        }
    }
    for (int i = t; i < nrows; i += 256)
        if (vt[i] != 0.0) atomic_add_f64(g_red + 6 * (int64_t)d.tile_cams[c0 + i / 6] + i % 6, vt[i]);
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

}  // namespace dbat
