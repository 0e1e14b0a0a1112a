// k_build_sig: Schur complement of the tiled object points by SIGNATURE GROUPS
// (K1, K3 point side, K4, K5 of SURVEY 8(a); fixed interior orientation and, NCX = 14,
// self-calibration: the tile's IO columns are further rows of every chunk).
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
// One workgroup per tile -- eight waves (two per SIMD: one evaluates while the other multiplies) for
// fixed IO with at most ten cameras per point, four waves with 512 registers for 11 ... 13 cameras
// and for self-calibration:
//   * the tile's camera records are staged in LDS once;
//   * the tile's block of S (at most 21 cameras, lower triangle packed, 64 KB)
//     lives in LDS; a finished chunk is added to it with ds_add_f64 and the tile
//     goes to HBM once, with global f64 atomics, as in the other tile kernels;
//   * a wave takes chunks from the tile's list through an LDS counter (longest first; the next
//     chunk's descriptor is fetched while the current one is worked on).
// A chunk in two passes:
//   pass 1, lane = object point (two, four or eight lanes per point when the chunk has at most 32, 16
//     or 8 points): loop over the
//     k cameras (uniform per chunk: broadcast reads of the staged record, coalesced (u,v) from the
//     slot-major copy), residual r and point block B per observation; V = sum B'B (+ prior,
//     + lambda), g = sum B'r in registers -- no atomics, no cross-lane traffic; V^-1, its
//     Cholesky factor R, the pivots; V^-1, g and the squared column norms go to HBM for the
//     back-substitution, R | R'g | Q stay in the lane's registers (pass 2 fetches them by
//     ds_bpermute).  Self-calibration, tiles with one IO block: the point's IO rows
//     Z_io = (sum_j E_io,j' B_j) R are summed here too;
//   pass 2, lane = observation, rounds of floor(64/k) <= 6 points: camera-side
//     block E, W = E'B, Z = W R into the wave's operand panel [rows][18 k-columns],
//     then ceil(3 n/4) k-steps of the block products (operands of the next k-step are read
//     while one multiplies).
// (row stride of the operand panel: 18 doubles; the reads of a k-step still conflict -- SQ_LDS_BANK_CONFLICT is 17 ... 23 %
// of SQ_LDS_IDX_ACTIVE at C3 / C4, profiles/r05_c{3,4}_summary.md -- see DESIGN.md 4)
#pragma once
#include <cstddef>
#include <type_traits>

#include "kernels.hpp"

namespace dbat {

constexpr int SIG_LDK = 18;          // k-columns per panel row: 6 points of a round; stride = 2 mod 4 doubles
constexpr int SIG_PPR = 6;           // points per round of pass 2
constexpr int SIG_KMAXR = 13;        // cameras per group at most (Plan::SG_KMAX)
constexpr int SIG_NW = 8;            // waves per workgroup (two per SIMD: one evaluates while the other multiplies)
constexpr int SIG_CAMW = (int)(offsetof(CamRec, ncol) / 8) + 1;   // doubles of a CamRec that the fixed-IO evaluation reads (.. w[2]) + eo_est
constexpr int SIG_CAMW_IO = (int)((sizeof(CamRec) + 7) / 8);   // self-calibration: the whole record (column lists)
constexpr int SIG_STILE = 8064;      // 126*127/2 = 8001 packed lower triangle of the tile, padded
// Self-calibration at EIGHT waves per workgroup (two per SIMD; -DDBAT_SIG_IO_WAVES=8, a compile-time experiment): the eight
// operand panels (92 KB) leave room for a tile of 16 cameras + 16 IO columns = 112 rows (the plan's DBAT_HIP_CMAX default
// follows: Plan::SIG_IO_CMAX).  Measured in round 6 (profiles/r06_selfcal_sig.md): under the 256-register cap the kernel
// spills 364 ... 888 bytes per lane and takes 15 % (C4) to 39 % (C2) LONGER than the four-wave kernel (and 16-camera tiles
// alone cost 6 ... 10 %): the default stays four waves, one per SIMD.
#ifndef DBAT_SIG_IO_WAVES
#define DBAT_SIG_IO_WAVES 4
#endif
constexpr int SIG_IO_CAMS = DBAT_SIG_IO_WAVES == 8 ? 16 : 21;
__host__ __device__ constexpr int sig_stile(bool io) { return io && DBAT_SIG_IO_WAVES == 8 ? 6336 : SIG_STILE; }      // 112*113/2 = 6328
__host__ __device__ constexpr int sig_cams(bool io) { return io ? SIG_IO_CAMS : 21; }

struct SigLds {                      // static part
    int next_chunk, abort_;
    int flush_turn;                  // deterministic mode: the chunk (tile-local index) whose turn it is to add to the tile
    double urow[128];                // deterministic mode: u_r of every row of the tile (DevProblem::det_u), [127 + 1]: u_f
    double uf;
    int grow[128];                   // row of the reduced system of every row of the tile
    int lc[SIG_NW][16];              // tile-local camera of every slot of the wave's chunk
    short tmap[SIG_NW][80];          // tile row of every row of the wave's chunk
    int toff[SIG_NW][80];            // ... and the offset of that row in the packed lower triangle of the tile
    unsigned char camio[21][16];     // self-calibration: tile IO row of a camera's q-th IO column
};

// eight waves (two per SIMD, 256 registers each) where the kernel fits them; the variants with five row
// blocks or with the IO Jacobian need more registers: four waves with 512 each
__host__ __device__ constexpr int sig_waves(int RB, bool io) { return RB <= 4 && !io ? SIG_NW : (io ? DBAT_SIG_IO_WAVES : 4); }
__host__ __device__ constexpr size_t sig_lds_bytes(int RB, bool io) {
    return ((size_t)sig_stile(io) + 128 + sig_cams(io) * (io ? SIG_CAMW_IO : SIG_CAMW) + sig_waves(RB, io) * ((size_t)RB * 16 * SIG_LDK)) * sizeof(double);
}

// value of x in lane `src` (ds_bpermute_b32 on both halves)
__device__ __forceinline__ double lane_get(double x, int src) {
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_ds_bpermute(4 * src, (int)(b & 0xffffffffll));
    const int hi = __builtin_amdgcn_ds_bpermute(4 * src, (int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Fixed interior orientation: the image side of an observation (the corrected image coordinates rhs) does not change
// between iterations and is precomputed (k_uv_to_rhs), the Jacobian blocks A, B do not depend on (u, v) at all.
// Pass 1 -- residual and WEIGHTED point block (res_euler_brown_*.m with eulerpinhole2.m, pinhole.m:54-66): 38 operations.
__device__ __forceinline__ void sig_eval_rB(const CamRec &C, const double (&Q)[3], double rhs0, double rhs1, double w0,
                                            double w1, double (&r)[2], double (&B)[2][3]) {
    const double d0 = Q[0] - C.c[0], d1 = Q[1] - C.c[1], d2 = Q[2] - C.c[2];
    const double X0 = C.Mt[0] * d0 + C.Mt[1] * d1 + C.Mt[2] * d2;
    const double X1 = C.Mt[3] * d0 + C.Mt[4] * d1 + C.Mt[5] * d2;
    const double X2 = C.Mt[6] * d0 + C.Mt[7] * d1 + C.Mt[8] * d2;
    const double iz = recip(X2);
    const double ph0 = X0 * iz, ph1 = X1 * iz;
    const double nf = -C.f, s = nf * iz, s0 = s * w0, s1 = s * w1;
    r[0] = (nf * ph0 - rhs0) * w0;
    r[1] = (nf * ph1 - rhs1) * w1;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        B[0][k] = s0 * (C.Mt[k] - ph0 * C.Mt[6 + k]);
        B[1][k] = s1 * (C.Mt[3 + k] - ph1 * C.Mt[6 + k]);
    }
}
// Pass 2 -- Z = E' (B R) of one observation, E = weighted camera block [dQ0 | dA], R = the (row-masked) factor of the
// point's V^-1: 6 x 3 values, about 100 operations (angle_terms: no derivative matrices).
__device__ __forceinline__ void sig_eval_Z(const CamRec &C, const double (&Q)[3], double w0, double w1,
                                           const double (&R)[6], double (&Z)[6][3]) {
    const double d0 = Q[0] - C.c[0], d1 = Q[1] - C.c[1], d2 = Q[2] - C.c[2];
    const double X0 = C.Mt[0] * d0 + C.Mt[1] * d1 + C.Mt[2] * d2;
    const double X1 = C.Mt[3] * d0 + C.Mt[4] * d1 + C.Mt[5] * d2;
    const double X2 = C.Mt[6] * d0 + C.Mt[7] * d1 + C.Mt[8] * d2;
    const double iz = recip(X2);
    const double ph0 = X0 * iz, ph1 = X1 * iz;
    const double s = -C.f * iz, s0 = s * w0, s1 = s * w1;
    double t0[3], t1[3];                             // B = [s0 t0 ; s1 t1]
#pragma unroll
    for (int k = 0; k < 3; ++k) { t0[k] = C.Mt[k] - ph0 * C.Mt[6 + k]; t1[k] = C.Mt[3 + k] - ph1 * C.Mt[6 + k]; }
    const double r00 = R[0], r10 = R[1], r20 = R[2], r11 = R[3], r21 = R[4], r22 = R[5];
    const double br0[3] = {s0 * (t0[0] * r00 + t0[1] * r10 + t0[2] * r20), s0 * (t0[1] * r11 + t0[2] * r21), s0 * (t0[2] * r22)};
    const double br1[3] = {s1 * (t1[0] * r00 + t1[1] * r10 + t1[2] * r20), s1 * (t1[1] * r11 + t1[2] * r21), s1 * (t1[2] * r22)};
    double y[3][3];
    angle_terms(C, d0, d1, d2, X0, X1, X2, y);
    double e0[6], e1[6];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        e0[k] = -(s0 * t0[k]); e1[k] = -(s1 * t1[k]);                 // world2cam.m:82  dQ0 = -dQ
        e0[3 + k] = s0 * (y[k][0] - ph0 * y[k][2]); e1[3 + k] = s1 * (y[k][1] - ph1 * y[k][2]);
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        Z[a][0] = e0[a] * br0[0] + e1[a] * br1[0];
        Z[a][1] = e0[a] * br0[1] + e1[a] * br1[1];
        Z[a][2] = e0[a] * br0[2] + e1[a] * br1[2];
    }
}

// Back-substitution, fixed IO -- t = E dc of one observation and its weighted point block B, without E: the step dc moves
// the camera-frame point by dX = -M dc[0..2] + sum_a y_a dc[3+a] (world2cam.m:82, angle_terms), and the projection
// takes that to t = (s0 (dX0 - ph0 dX2), s1 (dX1 - ph1 dX2)).  65 operations where the 2 x 6 block E and its product
// with dc took about 110 (the kernel is bound by the vector instructions it issues: SQ_INSTS_VALU, profiles/r04_c3_summary.md).
// dc: 0 for camera elements that are not estimated (masked once per camera); B: not masked by the point's fixed
// coordinates -- V^-1 decouples them (build: V(c,c) = 1, row and column 0) and their step is set to zero at the end.
__device__ __forceinline__ void sig_step_dot6(const CamRec &C, const double (&Q)[3], double w0, double w1,
                                              const double *dc, double &t0, double &t1, double (&B)[2][3]) {
    const double d0 = Q[0] - C.c[0], d1 = Q[1] - C.c[1], d2 = Q[2] - C.c[2];
    const double X0 = C.Mt[0] * d0 + C.Mt[1] * d1 + C.Mt[2] * d2;
    const double X1 = C.Mt[3] * d0 + C.Mt[4] * d1 + C.Mt[5] * d2;
    const double X2 = C.Mt[6] * d0 + C.Mt[7] * d1 + C.Mt[8] * d2;
    const double iz = recip(X2);
    const double ph0 = X0 * iz, ph1 = X1 * iz;
    const double s = -C.f * iz, s0 = s * w0, s1 = s * w1;
    double y[3][3];
    angle_terms(C, d0, d1, d2, X0, X1, X2, y);
    const double c0 = dc[0], c1 = dc[1], c2 = dc[2], a0 = dc[3], a1 = dc[4], a2 = dc[5];
    const double n0 = -c0, n1 = -c1, n2 = -c2;       // (wave-uniform: the negations cost nothing per observation)
    const double e0 = __builtin_fma(C.Mt[0], n0, __builtin_fma(C.Mt[1], n1, __builtin_fma(C.Mt[2], n2,
                      __builtin_fma(y[0][0], a0, __builtin_fma(y[1][0], a1, y[2][0] * a2)))));
    const double e1 = __builtin_fma(C.Mt[3], n0, __builtin_fma(C.Mt[4], n1, __builtin_fma(C.Mt[5], n2,
                      __builtin_fma(y[0][1], a0, __builtin_fma(y[1][1], a1, y[2][1] * a2)))));
    const double e2 = __builtin_fma(C.Mt[6], n0, __builtin_fma(C.Mt[7], n1, __builtin_fma(C.Mt[8], n2,
                      __builtin_fma(y[0][2], a0, y[1][2] * a1))));
    t0 = s0 * (e0 - ph0 * e2);
    t1 = s1 * (e1 - ph1 * e2);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        B[0][k] = s0 * (C.Mt[k] - ph0 * C.Mt[6 + k]);
        B[1][k] = s1 * (C.Mt[3 + k] - ph1 * C.Mt[6 + k]);
    }
}

// NCX = 6: fixed IO.  NCX = 14: self-calibration -- the estimated IO columns of the tile's cameras are
// nio further rows of every chunk (after its 6k camera rows, before the row of y): an IO row is
// shared by the k observations of a point, so its k-columns are summed with ds_add_f64 into the
// zeroed panel rows.
// IOS = 1, 2: self-calibration, tiles whose cameras belong to one or two IO blocks with the identity row
// map inside each block: the IO rows of a point are summed in pass 1 (see below).  One launch serves both kinds of tiles
// (k_build_sig branches per workgroup on the plan's flag), so they share the longest-first order.
// PW: observations with their own weights (sg_w; else the camera's two weights from its record in LDS).  A template
// parameter since round 5: as a run-time test the compiler turned "LDS value or global value" into a select of ADDRESSES and
// read the weights of pass 1 with flat loads, each trip waiting for vmcnt(0) lgkmcnt(0) right behind them.
template <int MODEL, int RB, int NCX, int IOS, bool PW>
__device__ __forceinline__ void build_sig_tile(const DevProblem &d, const int tile, const int part_slot, const double *__restrict__ z,
                                               const CamRec *__restrict__ cams, double lambda, int scale,
                                               double *__restrict__ S, double *__restrict__ g_red,
                                               double *__restrict__ Vinv, double *__restrict__ gp,
                                               double *__restrict__ jn2p,
                                               double *__restrict__ partial, unsigned long long *__restrict__ pivmm,
                                               const int32_t *__restrict__ sg_chunk,
                                               const int32_t *__restrict__ sg_tile_chunk0,
                                               const uint8_t *__restrict__ sg_lc, const double *__restrict__ sg_uv,
                                               const double *__restrict__ sg_w, unsigned *__restrict__ det_timeouts,
                                               SigLds &sy, double *sh) {
    constexpr bool IO = NCX > 6;
    constexpr bool io_simple = IO && IOS != 0;
    constexpr int NBLK = RB * (RB + 1) / 2, PROWS = RB * 16, LDK = SIG_LDK, NW = sig_waves(RB, IO), NT = 64 * NW;
    constexpr int CAMW = IO ? SIG_CAMW_IO : SIG_CAMW;
    extern __shared__ double smem[];
    double *stile = smem;                            // packed lower triangle of the tile's block of S (negated sum)
    constexpr int STILE = sig_stile(IO);
    double *vt = stile + STILE;                      // [128] -(W V^-1 g) by tile row
    double *camw = vt + 128;                         // [21][SIG_CAMW]
    double *wave_base = camw + sig_cams(IO) * CAMW;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // DBAT_HIP_ABLATE & 32 (measurement build): phase clocks of wave 0 (100 MHz ticks), summed over the tiles into g_tile2_prof
    const bool prof = DBAT_ABLATE(d, 32) && t == 0;
    long long tp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = prof ? wall_clock64() : 0;
    double *pan = wave_base + (size_t)wave * (PROWS * LDK);                // [PROWS][LDK]
    const int c0 = d.tile_cam_start[tile];
    const int ncam = d.tile_cam_start[tile + 1] - c0;
    const int io0 = IO ? d.tile_io_start[tile] : 0;
    const int nio = IO ? d.tile_io_start[tile + 1] - io0 : 0;      // IO rows of the tile (and of each of its chunks)
    const int nrows = 6 * ncam + nio;
    const int ch0 = sg_tile_chunk0[tile], ch1 = sg_tile_chunk0[tile + 1];
    // the first chunk of every wave is fixed (no ticket), and its descriptor is requested before anything else:
    // the load travels while the tile is cleared and the camera records are staged
    int ch = ch0 + wave;
    int nd = 0, nlc = 0;                             // lane l < 8: word l of the next descriptor; l < 16: its camera list
    if (ch < ch1) {
        if (lane < 8) nd = sg_chunk[8 * (int64_t)ch + lane];
        if (lane < 16) nlc = sg_lc[16 * (int64_t)ch + lane];
    }
    for (int i = t; i < STILE + 128; i += NT) stile[i] = 0.0;
    for (int i = t; i < NW * (PROWS * LDK); i += NT) wave_base[i] = 0.0;
    for (int i = t; i < ncam * CAMW; i += NT) {
        const int c = i / CAMW, f = i - c * CAMW;
        const CamRec &C = cams[d.tile_cams[c0 + c]];
        if constexpr (IO) camw[i] = reinterpret_cast<const double *>(&C)[f];
        else camw[i] = f < CAMW - 1 ? reinterpret_cast<const double *>(&C)[f] : (double)C.eo_est;
    }
    if (t < nrows) {
        const int gr = t < 6 * ncam ? 6 * d.tile_cams[c0 + t / 6] + t % 6 : 6 * d.nc + d.tile_iocols[io0 + t - 6 * ncam];
        sy.grow[t] = gr;
        if (d.deterministic) sy.urow[t] = d.det_u[gr];
    }
    if (t == 0 && d.deterministic) sy.uf = d.det_u[d.NS];
    if constexpr (IO) { for (int i = t; i < 16 * ncam; i += NT) sy.camio[i >> 4][i & 15] = d.tile_cam_io[(size_t)c0 * 16 + i]; }
    if (t == 0) { sy.next_chunk = ch0 + NW; sy.abort_ = 0; sy.flush_turn = 0; }
    for (int i = t; i < NW * 80; i += NT) { sy.tmap[i / 80][i % 80] = 0; sy.toff[i / 80][i % 80] = 0; }
    auto lap = [&](int i) { if (prof) { const long long now = wall_clock64(); tp[i] += now - tlast; tlast = now; } };
    __syncthreads();
    lap(0);
    double pmin = 1e300, pmax = 0.0, rr = 0.0;
    const int dix[3] = {0, 3, 5};
    // the descriptor of the wave's next chunk is fetched while the current one is being worked on
    auto grab = [&]() -> int {
        int c = 0;
        if (lane == 0) c = atomicAdd(&sy.next_chunk, 1);
        return __builtin_amdgcn_readfirstlane(c);
    };
    while (ch < ch1) {
        const int pt0 = __builtin_amdgcn_readlane(nd, 0), npts = __builtin_amdgcn_readlane(nd, 1);
        const int k = __builtin_amdgcn_readlane(nd, 2);
        const int gm = __builtin_amdgcn_readlane(nd, 4), gi0 = __builtin_amdgcn_readlane(nd, 5);
        const int uv0 = __builtin_amdgcn_readlane(nd, 6);
        if (lane < k) sy.lc[wave][lane] = nlc;
        {   // tile row of the chunk's rows 0 .. 6k-1 (row 6k is the right-hand side)
            const int kq = lane / 6;
            const int lcq = __builtin_amdgcn_ds_bpermute(4 * (kq < k ? kq : 0), nlc);
            const int tr1 = 6 * lcq + (lane - 6 * kq);
            if (lane < 6 * k) { sy.tmap[wave][lane] = (short)tr1; sy.toff[wave][lane] = tr1 * (tr1 + 1) / 2; }
            // (ds_bpermute returns 0 from lanes outside EXEC: both exchanges run with the whole wave)
            const int kq2 = (lane + 64) / 6;
            const int lcq2 = __builtin_amdgcn_ds_bpermute(4 * (kq2 < k ? kq2 : 0), nlc);
            const int tr2 = 6 * lcq2 + (lane + 64 - 6 * kq2);
            if (lane + 64 < 6 * k) { sy.tmap[wave][lane + 64] = (short)tr2; sy.toff[wave][lane + 64] = tr2 * (tr2 + 1) / 2; }
            if (lane < nio) {                        // IO rows follow the camera rows
                const int tr3 = 6 * ncam + lane;
                sy.tmap[wave][6 * k + lane] = (short)tr3; sy.toff[wave][6 * k + lane] = tr3 * (tr3 + 1) / 2;
            }
        }
        const int ch_cur = ch;
        ch = grab();
        if (ch < ch1) {
            if (lane < 8) nd = sg_chunk[8 * (int64_t)ch + lane];
            if (lane < 16) nlc = sg_lc[16 * (int64_t)ch + lane];
        }
        for (int i = lane; i < PROWS * LDK; i += 64) pan[i] = 0.0;     // rows / k-columns this chunk does not write
        __builtin_amdgcn_wave_barrier();
        lap(1);
        // ------------------------------------------------------------ pass 1: lane = object point
        double pR[6] = {0, 0, 0, 0, 0, 0}, pY[3] = {0, 0, 0}, pQ[3] = {0, 0, 0};     // of this lane's point, for pass 2
        unsigned pEst = 0;
        // Self-calibration, the rule (IOS): all cameras of the tile share one IO block, IO column q = tile
        // IO row q.  Then the IO rows of a point, Z_io = (sum_j E_io,j' B_j) R, are summed where the
        // point's observations are visited one after the other anyway -- in pass 1, in the lane's
        // registers -- instead of by 24 LDS atomics per observation (ten observations on the same
        // addresses) in pass 2.  Tiles that mix IO blocks run the IOS = 0 instantiation with the atomics.
        constexpr int NQ1 = NCX - 6;                       // IO columns of one camera
        constexpr int NQ = io_simple ? IOS * NQ1 : 1;       // IO rows of the tile: one or two IO blocks
        double pZio[NQ][3];
#pragma unroll
        for (int q = 0; q < NQ; ++q) pZio[q][0] = pZio[q][1] = pZio[q][2] = 0.0;
        {
            // A short chunk takes several lanes per point: with G = 32, 16 or 8 point slots (the smallest that
            // holds the chunk) the lanes l, l + G, l + 2G, ... share the point's cameras (slots h, h + 64/G, ...)
            // and add their sums up afterwards -- the loop over the cameras is a serial chain of evaluations,
            // so a chunk of 7 points (C1, C2) takes a quarter of the time of a chunk of 64.
            const int G = npts > 32 ? 64 : (npts > 16 ? 32 : (npts > 8 ? 16 : 8));
            const int pi = lane & (G - 1);
            const int jh = lane / G, jstep = 64 / G;
            const bool act = pi < npts;
            const int pt = DBAT_ABLATE(d, 1024) ? (pi & 31) : pt0 + (act ? pi : 0);      // (1024, measurement build: every load of pass 1 from the cache)
            const int64_t zp = d.NS + 3 * (int64_t)pt;
            double Q[3] = {z[zp], z[zp + 1], z[zp + 2]};
            const unsigned est = (d.z_est[zp] ? 1u : 0u) | (d.z_est[zp + 1] ? 2u : 0u) | (d.z_est[zp + 2] ? 4u : 0u);
            const double prw[3] = {d.z_prw[zp], d.z_prw[zp + 1], d.z_prw[zp + 2]};     // (used after the sums: requested with the point)
            const double prv[3] = {d.z_prv[zp], d.z_prv[zp + 1], d.z_prv[zp + 2]};
            double V[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
            const int64_t q0 = DBAT_ABLATE(d, 1024) ? (int64_t)(pi & 31) : uv0 + gi0 + (act ? pi : 0);
            const double2 *uvp = reinterpret_cast<const double2 *>(sg_uv), *wp = reinterpret_cast<const double2 *>(sg_w);
            // V += B'B, g += B'r of one observation (weighted, fixed coordinates masked)
            auto accumulate = [&](const double (&r)[2], const double (&B)[2][3]) {
                if (act) rr = fma2(rr, r[0], r[0], r[1], r[1]);
                V[0] = fma2(V[0], B[0][0], B[0][0], B[1][0], B[1][0]);
                V[1] = fma2(V[1], B[0][0], B[0][1], B[1][0], B[1][1]);
                V[2] = fma2(V[2], B[0][0], B[0][2], B[1][0], B[1][2]);
                V[3] = fma2(V[3], B[0][1], B[0][1], B[1][1], B[1][1]);
                V[4] = fma2(V[4], B[0][1], B[0][2], B[1][1], B[1][2]);
                V[5] = fma2(V[5], B[0][2], B[0][2], B[1][2], B[1][2]);
                g[0] = fma2(g[0], B[0][0], r[0], B[1][0], r[1]);
                g[1] = fma2(g[1], B[0][1], r[0], B[1][1], r[1]);
                g[2] = fma2(g[2], B[0][2], r[0], B[1][2], r[1]);
            };
            const int kk = DBAT_ABLATE(d, 128) ? 0 : (DBAT_ABLATE(d, 4) ? 1 : k);      // (128: no camera at all in pass 1)
            if constexpr (!IO) {
                // Fixed IO: TWO cameras per trip.  One evaluation is a dependent chain of ~120 f64 operations and the
                // SIMD holds two waves, so a single chain leaves the issue slots half empty (r03a_c3_summary.md: 40 %
                // of the wave cycles issue-stalled); two independent chains interleave.  The accumulators of pass 2
                // are not live here, so the registers are there.  A lane whose second camera does not exist repeats
                // its first with zero weights (lanes of a short chunk start at different cameras).
                // All the image coordinates of the lane's share are requested up front (a trip of two evaluations takes
                // a few hundred cycles, a load that misses the L2 a few thousand: fetching one trip ahead left every trip
                // waiting), so the chunk pays ONE memory latency here, together with the point and its prior above.
                constexpr int NT1 = (SIG_KMAXR + 1) / 2;                 // trips of two cameras
                double2 uvl[2 * NT1];
#pragma unroll
                for (int tq = 0; tq < 2 * NT1; ++tq) {
                    const int jj = jh + tq * jstep;
                    uvl[tq] = double2{0, 0};
                    if (jj < k) uvl[tq] = uvp[q0 + (int64_t)jj * (DBAT_ABLATE(d, 1024) ? 32 : gm)];
                }
#pragma unroll
                for (int tq = 0; tq < NT1; ++tq) {
                    const int j = jh + 2 * tq * jstep;
                    if (j < kk) {
                        const bool vb = j + jstep < kk;
                        const int j2 = vb ? j + jstep : j;
                        const CamRec &CA = *reinterpret_cast<const CamRec *>(camw + sy.lc[wave][j] * CAMW);
                        const CamRec &CB = *reinterpret_cast<const CamRec *>(camw + sy.lc[wave][j2] * CAMW);
                        const double2 ca = uvl[2 * tq], cb = vb ? uvl[2 * tq + 1] : uvl[2 * tq];
                        double wa0 = CA.w[0], wa1 = CA.w[1], wb0 = CB.w[0], wb1 = CB.w[1];
                        if constexpr (PW) {           // (observations with their own standard deviations: the rare case)
                            const double2 cwa = wp[q0 + (int64_t)j * gm], cwb = wp[q0 + (int64_t)j2 * gm];
                            wa0 = cwa.x; wa1 = cwa.y; wb0 = cwb.x; wb1 = cwb.y;
                        }
                        const double zb = vb ? 1.0 : 0.0;
                        wb0 *= zb; wb1 *= zb;
                        // (the coordinates a point does not estimate are masked once per point, after the sums)
                        double rA[2], BA[2][3], rB[2], BB[2][3];
                        sig_eval_rB(CA, Q, ca.x, ca.y, wa0, wa1, rA, BA);
                        sig_eval_rB(CB, Q, cb.x, cb.y, wb0, wb1, rB, BB);
                        accumulate(rA, BA);
                        accumulate(rB, BB);
                    }
                }
            } else {
            double2 uv_n = double2{0, 0}, w_n = double2{0, 0};
            if (jh < k) { uv_n = uvp[q0 + (int64_t)jh * gm]; if constexpr (PW) w_n = wp[q0 + (int64_t)jh * gm]; }
            for (int j = jh; j < kk; j += jstep) {
                const int lc = sy.lc[wave][j];
                const CamRec &C = *reinterpret_cast<const CamRec *>(camw + lc * CAMW);
                const int64_t q = q0 + (int64_t)j * gm;
                const double uu = uv_n.x, vv = uv_n.y;
                const double w0 = PW ? w_n.x : C.w[0], w1 = PW ? w_n.y : C.w[1];
                if (j + jstep < k) {                  // next slot's image coordinates, one slot ahead
                    uv_n = uvp[q + (int64_t)jstep * gm];
                    if constexpr (PW) w_n = wp[q + (int64_t)jstep * gm];
                }
                double r[2], A[2][6], B[2][3], Cf[2][MAXIO];
                obs_eval<MODEL, true, io_simple>(C, d.nK, d.nP, Q, uu, vv, r, A, B, Cf);    // A is dead code here
                r[0] *= w0; r[1] *= w1;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const double m = ((est >> c) & 1u) ? 1.0 : 0.0;
                    B[0][c] *= w0 * m; B[1][c] *= w1 * m;
                }
                if constexpr (io_simple) {
                    {                                // W_io += E_io' B, into the rows of the camera's IO block
                        double Eio[2][NCX];
                        io_columns<NCX>(C, Cf, w0, w1, Eio);
                        const int ncol = min(C.ncol, NCX);
                        auto into = [&](auto off) {
                            constexpr int OFF = decltype(off)::value;
#pragma unroll
                            for (int q = 0; q < NQ1; ++q)
                                if (6 + q < ncol) {
                                    const double e0 = Eio[0][6 + q], e1 = Eio[1][6 + q];
                                    pZio[OFF + q][0] = fma2(pZio[OFF + q][0], e0, B[0][0], e1, B[1][0]);
                                    pZio[OFF + q][1] = fma2(pZio[OFF + q][1], e0, B[0][1], e1, B[1][1]);
                                    pZio[OFF + q][2] = fma2(pZio[OFF + q][2], e0, B[0][2], e1, B[1][2]);
                                }
                        };
                        if (IOS == 2 && sy.camio[lc][0] >= NQ1) into(std::integral_constant<int, (IOS == 2 ? NQ1 : 0)>{});   // (uniform: the camera is)
                        else into(std::integral_constant<int, 0>{});
                    }
                }
                accumulate(r, B);
            }
            }
            for (int m = G; m < 64; m <<= 1) {        // the other lanes of the point (uniform loop: butterfly over the slices)
#pragma unroll
                for (int c = 0; c < 6; ++c) V[c] += lane_get(V[c], lane ^ m);
#pragma unroll
                for (int c = 0; c < 3; ++c) g[c] += lane_get(g[c], lane ^ m);
                if constexpr (io_simple) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
#pragma unroll
                        for (int c = 0; c < 3; ++c) pZio[q][c] += lane_get(pZio[q][c], lane ^ m);
                }
            }
            const bool writer = act && lane == pi;    // one lane per point writes to HBM
            if constexpr (!IO) {
                if (est != 7u) {                      // fixed coordinates: B(:, c) = 0
                    const double m0 = (est & 1u) ? 1.0 : 0.0, m1 = (est & 2u) ? 1.0 : 0.0, m2 = (est & 4u) ? 1.0 : 0.0;
                    V[0] *= m0; V[1] *= m0 * m1; V[2] *= m0 * m2; V[3] *= m1; V[4] *= m1 * m2; V[5] *= m2;
                    g[0] *= m0; g[1] *= m1; g[2] *= m2;
                }
            }
            if (act) {
                double jn[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const double pw = prw[c];
                    if (pw > 0) { V[dix[c]] += pw; g[c] += pw * (Q[c] - prv[c]); }
                    jn[c] = V[dix[c]];
                    if (writer) jn2p[3 * (int64_t)pt + c] = jn[c];
                    if ((est >> c) & 1u) V[dix[c]] += lambda; else V[dix[c]] = 1.0;
                }
                double inv[6], Rpb[6];
                point_block_factor(V, Rpb, inv);          // V^-1 = R R', R lower triangular, straight from V
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
                if (writer) {
#pragma unroll
                    for (int c = 0; c < 6; ++c) Vinv[6 * (int64_t)pt + c] = Rpb[c];     // the factor: V^-1 = R R'
#pragma unroll
                    for (int c = 0; c < 3; ++c) gp[3 * (int64_t)pt + c] = g[c];
                }
                const double r00 = Rpb[0], r10 = Rpb[1], r20 = Rpb[2], r11 = Rpb[3], r21 = Rpb[4], r22 = Rpb[5];
                pR[0] = r00; pR[1] = r10; pR[2] = r20; pR[3] = r11; pR[4] = r21; pR[5] = r22;
                if constexpr (!IO || io_simple) {
                    // pass 2 multiplies B R with B unmasked: row c of R is zeroed for a coordinate that is not estimated
                    // (column c of B then never contributes)
                    if (est != 7u) {
                        if (!(est & 1u)) pR[0] = 0.0;
                        if (!(est & 2u)) { pR[1] = 0.0; pR[3] = 0.0; }
                        if (!(est & 4u)) { pR[2] = 0.0; pR[4] = 0.0; pR[5] = 0.0; }
                    }
                }
                pY[0] = r00 * g[0] + r10 * g[1] + r20 * g[2];                  // y = R' g
                pY[1] = r11 * g[1] + r21 * g[2];
                pY[2] = r22 * g[2];
                pQ[0] = Q[0]; pQ[1] = Q[1]; pQ[2] = Q[2];
                pEst = est;
                if constexpr (io_simple) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {       // Z_io = W_io R
                        const double w0_ = pZio[q][0], w1_ = pZio[q][1], w2_ = pZio[q][2];
                        pZio[q][0] = w0_ * r00 + w1_ * r10 + w2_ * r20;
                        pZio[q][1] = w1_ * r11 + w2_ * r21;
                        pZio[q][2] = w2_ * r22;
                    }
                }
            } else if constexpr (io_simple) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) pZio[q][0] = pZio[q][1] = pZio[q][2] = 0.0;      // no point in this lane
            }
        }
        lap(2);
        // trace(J'J) only (the first linearisation of levenberg_marquardt.m:76-95 serves lambda0 = c trace / n and
        // nothing else): the squared column norms of the points are out, the Schur complement is not wanted
        if (d.trace_only || DBAT_ABLATE(d, 64)) continue;      // (64, measurement build: no pass 2 at all)
        // ------------------------------------------------------------ pass 2: lane = observation
        mfma_d4 acc[NBLK];
#pragma unroll
        for (int s = 0; s < NBLK; ++s) acc[s] = mfma_d4{0, 0, 0, 0};
        const int ppr = min(SIG_PPR, 64 / k);        // points per round
        const int ir = lane / k, j = lane - ir * k;
        const bool lane_on = ir < ppr;
        const int ry = 6 * k + nio;                  // row of y; rows 6k .. ry-1 are the IO rows
        const int rbk = (ry + 1 + 15) >> 4;          // row blocks this chunk needs
        const int lcj = lane_on ? sy.lc[wave][j] : 0;
        const CamRec &C = *reinterpret_cast<const CamRec *>(camw + lcj * CAMW);
        unsigned eo_est;
        if constexpr (IO) eo_est = C.eo_est; else eo_est = (unsigned)camw[lcj * CAMW + CAMW - 1];
        const double2 *uvp2 = reinterpret_cast<const double2 *>(sg_uv), *wp2 = reinterpret_cast<const double2 *>(sg_w);
        const int64_t qj = uv0 + (int64_t)j * gm + gi0;
        double2 uv2 = double2{0, 0}, w2 = double2{0, 0};
        if (lane_on && ir < npts) { uv2 = uvp2[qj + ir]; if constexpr (PW) w2 = wp2[qj + ir]; }
        for (int p0 = 0; p0 < npts; p0 += ppr) {
            const int i = p0 + ir;
            const bool on = lane_on && i < npts;
            const double2 uv_c = uv2, w_c = w2;
            if (lane_on && i + ppr < npts) { uv2 = uvp2[qj + i + ppr]; if constexpr (PW) w2 = wp2[qj + i + ppr]; }   // next round's
            double Zr[6][3];
#pragma unroll
            for (int a = 0; a < 6; ++a) Zr[a][0] = Zr[a][1] = Zr[a][2] = 0.0;
            double y3[3] = {0, 0, 0};
            if constexpr (IO) {
                if constexpr (!io_simple) {          // the IO rows are sums over a point's observations: start from zero
                    for (int q = lane; q < nio * LDK; q += 64) pan[6 * k * LDK + q] = 0.0;
                } else {
                    // Z_io of the round's points from the lanes that hold them (pass-1 layout: lane = point;
                    // lanes without a point hold zeros and clear the previous round's columns)
                    const int nhold = npts > 32 ? 64 : (npts > 16 ? 32 : (npts > 8 ? 16 : 8));     // point slots of the pass-1 layout
                    const int pil = lane & (nhold - 1);
                    const int irw = pil - p0;
                    if (lane == pil && irw >= 0 && irw < ppr) {
#pragma unroll
                        for (int q = 0; q < NQ; ++q)
                            if (q < nio) {
                                double *pio = pan + (6 * k + q) * LDK + 3 * irw;
                                pio[0] = pZio[q][0]; pio[1] = pZio[q][1]; pio[2] = pZio[q][2];
                            }
                    }
                    if (p0 + ppr > nhold && lane < p0 + ppr - nhold) {   // columns of points nhold, nhold+1, ...: no lane holds them
                        const int irz = nhold - p0 + lane;
#pragma unroll
                        for (int q = 0; q < NQ; ++q)
                            if (q < nio) {
                                double *pio = pan + (6 * k + q) * LDK + 3 * irz;
                                pio[0] = 0.0; pio[1] = 0.0; pio[2] = 0.0;
                            }
                    }
                }
            }
            // R | y | Q | est of this lane's point, from the lane that holds it
            const int src = on ? i : lane;
            double gR[6], gY[3], gQ[3];
            if (!DBAT_ABLATE(d, 256)) {              // (256, measurement build: no gathers)
#pragma unroll
            for (int c = 0; c < 6; ++c) gR[c] = lane_get(pR[c], src);
#pragma unroll
            for (int c = 0; c < 3; ++c) { gY[c] = lane_get(pY[c], src); gQ[c] = lane_get(pQ[c], src); }
            } else {
#pragma unroll
                for (int c = 0; c < 6; ++c) gR[c] = pR[c];
#pragma unroll
                for (int c = 0; c < 3; ++c) { gY[c] = pY[c]; gQ[c] = pQ[c]; }
            }
            unsigned est = 7u;
            if constexpr (IO && !io_simple) est = (unsigned)__builtin_amdgcn_ds_bpermute(4 * src, (int)pEst);
            // fixed IO, and self-calibrating tiles whose IO rows were summed in pass 1: the camera rows of Z need the
            // blocks A and B only -- they do not depend on (u, v) -- by the lean evaluation
            if constexpr (!IO || io_simple) {
                if (on && !DBAT_ABLATE(d, 2)) {
                    const double Q[3] = {gQ[0], gQ[1], gQ[2]};
                    const double w0 = PW ? w_c.x : C.w[0], w1 = PW ? w_c.y : C.w[1];
                    sig_eval_Z(C, Q, w0, w1, gR, Zr);
                    if ((eo_est & 63u) != 63u) {
#pragma unroll
                        for (int a = 0; a < 6; ++a)
                            if (!((eo_est >> a) & 1u)) { Zr[a][0] = 0.0; Zr[a][1] = 0.0; Zr[a][2] = 0.0; }
                    }
                    y3[0] = gY[0]; y3[1] = gY[1]; y3[2] = gY[2];
                }
            } else
            if (on && !DBAT_ABLATE(d, 2)) {
                const double Q[3] = {gQ[0], gQ[1], gQ[2]};
                const double uu = uv_c.x, vv = uv_c.y;
                const double w0 = PW ? w_c.x : C.w[0], w1 = PW ? w_c.y : C.w[1];
                double r[2], A[2][6], B[2][3], Cf[2][MAXIO];
                obs_eval<MODEL, true, (IO && !io_simple)>(C, d.nK, d.nP, Q, uu, vv, r, A, B, Cf);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const double m = ((est >> c) & 1u) ? 1.0 : 0.0;
                    B[0][c] *= w0 * m; B[1][c] *= w1 * m;
                }
                const double r00 = gR[0], r10 = gR[1], r20 = gR[2], r11 = gR[3], r21 = gR[4], r22 = gR[5];
                if constexpr (IO && !io_simple) {    // IO columns of this camera -> the chunk's IO rows (LDS atomics)
                    const int ncol = min(C.ncol, NCX);
                    double Eio[2][NCX];
                    io_columns<NCX>(C, Cf, w0, w1, Eio);
#pragma unroll
                    for (int q = 0; q < NCX - 6; ++q)
                        if (6 + q < ncol) {
                            const double e0 = Eio[0][6 + q], e1 = Eio[1][6 + q];
                            const double wa0 = e0 * B[0][0] + e1 * B[1][0];
                            const double wa1 = e0 * B[0][1] + e1 * B[1][1];
                            const double wa2 = e0 * B[0][2] + e1 * B[1][2];
                            double *pio = pan + (6 * k + sy.camio[lcj][q]) * LDK + 3 * ir;
                            atomic_add_f64(pio, wa0 * r00 + wa1 * r10 + wa2 * r20);
                            atomic_add_f64(pio + 1, wa1 * r11 + wa2 * r21);
                            atomic_add_f64(pio + 2, wa2 * r22);
                        }
                }
                // Z = W R = E' (B R): the 2 x 3 block B R once per observation (12 operations), then one 2-term
                // product per element of Z (36) -- instead of W = E'B (36) followed by W R (36)
                const double br00 = B[0][0] * r00 + B[0][1] * r10 + B[0][2] * r20, br01 = B[0][1] * r11 + B[0][2] * r21, br02 = B[0][2] * r22;
                const double br10 = B[1][0] * r00 + B[1][1] * r10 + B[1][2] * r20, br11 = B[1][1] * r11 + B[1][2] * r21, br12 = B[1][2] * r22;
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    const double m = ((eo_est >> a) & 1u) ? 1.0 : 0.0;
                    const double e0 = A[0][a] * w0 * m, e1 = A[1][a] * w1 * m;
                    Zr[a][0] = e0 * br00 + e1 * br10;
                    Zr[a][1] = e0 * br01 + e1 * br11;
                    Zr[a][2] = e0 * br02 + e1 * br12;
                }
                y3[0] = gY[0]; y3[1] = gY[1]; y3[2] = gY[2];
            }
            if (lane_on && !DBAT_ABLATE(d, 512)) {   // lanes of missing points overwrite the previous round with zeros  (512, measurement build: no panel writes)
                double *pr = pan + (6 * j) * LDK + 3 * ir;
#pragma unroll
                for (int a = 0; a < 6; ++a) { pr[a * LDK] = Zr[a][0]; pr[a * LDK + 1] = Zr[a][1]; pr[a * LDK + 2] = Zr[a][2]; }
                if (j == 0) { double *py = pan + ry * LDK + 3 * ir; py[0] = y3[0]; py[1] = y3[1]; py[2] = y3[2]; }
            }
            lds_fence();
            __builtin_amdgcn_wave_barrier();
            lap(3);
            // 18 k-columns = 4 1/2 k-steps: all operands of the round are read first, then the products
            // run back to back; the last step's lanes 32..63 would read k-columns 18, 19 (the next row)
            const int ksteps = DBAT_ABLATE(d, 1) ? 0 : (3 * min(ppr, npts - p0) + 3) >> 2;
            const double *zr = pan + (lane & 15) * LDK + (lane >> 4);
            // operands of k-step ks+1 are read while the products of k-step ks run
            auto load_ops = [&](int ks, double (&o)[RB]) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
                    o[rb] = (rb < rbk && ks < ksteps && !(ks == 4 && lane >= 32)) ? zr[rb * 16 * LDK + 4 * ks] : 0.0;
            };
            double opa[RB], opb[RB];
            load_ops(0, opa);
#pragma unroll
            for (int ks = 0; ks < 5; ++ks) {
                double (&cur)[RB] = (ks & 1) ? opb : opa;
                double (&nxt)[RB] = (ks & 1) ? opa : opb;
                if (ks + 1 < 5) load_ops(ks + 1, nxt);
                if (ks < ksteps) {
#pragma unroll
                    for (int r1 = 0; r1 < RB; ++r1)
                        if (r1 < rbk) {
#pragma unroll
                            for (int r2 = 0; r2 <= r1; ++r2)
                                acc[r1 * (r1 + 1) / 2 + r2] = __builtin_amdgcn_mfma_f64_16x16x4f64(
                                    cur[r1], cur[r2], acc[r1 * (r1 + 1) / 2 + r2], 0, 0, 0);
                        }
                }
            }
            lds_fence();
            __builtin_amdgcn_wave_barrier();
            lap(4);
        }
        // ------------------------------------------------------------ chunk -> tile (LDS atomics)
        // one predicated ds_add_f64 per accumulator element; vt follows stile, so the right-hand-side
        // row (6k) only changes the index and the sign
        // deterministic mode: the chunks add to the tile in their order (they are taken in that order; all the tile's waves
        // are resident, so a wave only ever waits for waves that are running) -- the tile's sums are the same bits then,
        // and they go onto the grid of S once, when the tile is flushed (one rounding per tile and element, not per chunk)
        if (d.deterministic) {
            volatile int *turn = &sy.flush_turn;
            int spins = 0;
            for (; *turn != ch_cur - ch0 && spins < (1 << 24); ++spins) __builtin_amdgcn_s_sleep(1);
            // the cap is there for the case that the argument above fails: the chunks would then add out of order and the
            // run would lose its determinism without a word -- the host turns this count into an error (Core::sync)
            if (spins >= (1 << 24) && lane == 0) atomicAdd(det_timeouts, 1u);
        }
        {
            const int r6k = ry;                      // the row of y closes the chunk's rows
            int tcm[RB];
#pragma unroll
            for (int r2 = 0; r2 < RB; ++r2) tcm[r2] = sy.tmap[wave][min(16 * r2 + (lane & 15), 79)];
            // Rows beyond the row of y were zero in the panel, so the accumulators are exactly 0.0 there and their row
            // map entries are those of an earlier chunk (or the zeros of the set-up): the blocks below the diagonal add
            // every element, without a test -- a zero lands on some valid element of the tile.  The diagonal blocks drop
            // their upper triangle and the (y, y) element.
#pragma unroll
            for (int r1 = 0; r1 < RB; ++r1) {
                if (r1 >= rbk) break;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int lr = 16 * r1 + (lane >> 4) + 4 * e;
                    const bool yrow = lr == r6k;
                    double *row = stile + (yrow ? STILE : sy.toff[wave][min(lr, 79)]);
                    if (!DBAT_ABLATE(d, 8)) {
#pragma unroll
                        for (int r2 = 0; r2 < r1; ++r2) {
                            const double v = acc[r1 * (r1 + 1) / 2 + r2][e];
                            atomic_add_f64(row + tcm[r2], yrow ? -v : v);
                        }
                        const double v = acc[r1 * (r1 + 1) / 2 + r1][e];
                        const int lcol = 16 * r1 + (lane & 15);
                        if (lcol <= lr && lcol < r6k && v != 0.0) atomic_add_f64(row + tcm[r1], yrow ? -v : v);
                    }
                }
            }
        }
        if (d.deterministic) {
            lds_fence();
            if (lane == 0) { volatile int *turn = &sy.flush_turn; *turn = ch_cur - ch0 + 1; }
        }
        __builtin_amdgcn_wave_barrier();
        lap(5);
    }
    lap(6);
    __syncthreads();
    lap(6);
    // ---------------------------------------------------------------- tile -> HBM
    // one column per wave and trip: consecutive lanes, consecutive rows (neighbouring addresses in S).  All the
    // LDS reads of the wave's columns are issued first (the accumulators are dead: the registers are there), then the
    // atomics -- one LDS latency per tile instead of one per column
    {
        constexpr int NCOLW = (128 + NW - 1) / NW;   // columns per wave at most
        double v0[NCOLW], v1[NCOLW];
#pragma unroll
        for (int q = 0; q < NCOLW; ++q) {
            const int tc = wave + q * NW, tr = tc + lane;
            v0[q] = (tc < nrows && tr < nrows) ? stile[tr * (tr + 1) / 2 + tc] : 0.0;
            v1[q] = (tc < nrows && tr + 64 < nrows) ? stile[(tr + 64) * (tr + 65) / 2 + tc] : 0.0;
            // deterministic mode: onto the grid of the element's place in S (kernels.hpp DevProblem::deterministic) -- the
            // additions of the tiles are exact then, in whatever order they arrive
            if (d.deterministic && tc < nrows) {
                const double uc = sy.urow[tc];
                if (tr < nrows) v0[q] = det_round(v0[q], sy.urow[tr], uc);
                if (tr + 64 < nrows) v1[q] = det_round(v1[q], sy.urow[tr + 64], uc);
            }
        }
        if (!DBAT_ABLATE(d, 16)) {
#pragma unroll
            for (int q = 0; q < NCOLW; ++q) {
                const int tc = wave + q * NW, tr = tc + lane;
                if (tc < nrows) {
                    const int64_t gcol = sy.grow[tc] * d.ldS;
                    if (v0[q] != 0.0) atomic_add_f64(S + gcol + sy.grow[tr], -v0[q]);
                    if (v1[q] != 0.0) atomic_add_f64(S + gcol + sy.grow[tr + 64], -v1[q]);
                }
            }
        }
    }
    for (int i = t; i < nrows; i += NT) {
        const double v = d.deterministic ? det_round(vt[i], sy.urow[i], sy.uf) : vt[i];
        if (v != 0.0) atomic_add_f64(g_red + sy.grow[i], v);
    }
    double accr[1] = {rr};
    block_sum<1>(accr, sh);
    if (t == 0) partial[part_slot] = accr[0];
    pmin = pmin < 1e300 ? sqrt(pmin) : pmin; pmax = sqrt(pmax);
    for (int off = 32; off > 0; off >>= 1) {
        pmin = fmin(pmin, __shfl_down(pmin, off, 64));
        pmax = fmax(pmax, __shfl_down(pmax, off, 64));
    }
    if ((t & 63) == 0 && pmax > 0.0) {
        atomicMin(pivmm, (unsigned long long)__double_as_longlong(pmin));
        atomicMax(pivmm + 1, (unsigned long long)__double_as_longlong(pmax));
    }
    lap(7);
    if (prof)
        for (int i = 0; i < 8; ++i) if (tp[i]) atomicAdd(&g_tile2_prof[i], (unsigned long long)tp[i]);
}

// PERSISTENT: the launch has one workgroup per CU (the tile's LDS allows no more), and every workgroup takes tiles
// from a ticket counter, longest first, until none is left -- a tile is 150 us of work, and a workgroup that is
// dispatched for one tile only pays the dispatch and the release of 150 KB of LDS every time (round 4: about 14 us
// per tile between the end of one workgroup and the first instruction of the next).  Every workgroup draws exactly one
// ticket beyond the last tile; the one that draws the very last ticket of the launch puts the counter back to zero.
template <int MODEL, int RB, int NCX, bool PW>
__global__ __launch_bounds__(64 * sig_waves(RB, (NCX > 6))) void k_build_sig(DevProblem d, const double *__restrict__ z,
                                                   const CamRec *__restrict__ cams, double lambda, int scale,
                                                   double *__restrict__ S, double *__restrict__ g_red,
                                                   double *__restrict__ Vinv, double *__restrict__ gp,
                                                   double *__restrict__ jn2p,
                                                   double *__restrict__ partial, unsigned long long *__restrict__ pivmm,
                                                   const int32_t *__restrict__ sg_chunk,
                                                   const int32_t *__restrict__ sg_tile_chunk0,
                                                   const uint8_t *__restrict__ sg_lc, const double *__restrict__ sg_uv,
                                                   const double *__restrict__ sg_w, unsigned *__restrict__ tile_ctr) {
    __shared__ unsigned s_ticket;
    __shared__ SigLds sy;                            // (one copy for the three instantiations of build_sig_tile)
    __shared__ double sh[16];
    for (;;) {
        if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add(tile_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned ti = s_ticket;
        if (ti >= (unsigned)d.ntiles) {
            if (threadIdx.x == 0 && ti == (unsigned)d.ntiles + gridDim.x - 1)
                __hip_atomic_store(tile_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        const int tile = d.tile_order[ti];
        int ios = 0;
        if constexpr (NCX > 6) ios = d.tile_io_simple ? d.tile_io_simple[tile] : 0;       // IO blocks of the tile, if it qualifies
#ifdef DBAT_SIG_ONLY_IOS
        ios = DBAT_SIG_ONLY_IOS;     // (register experiments: one instantiation per kernel)
#endif
        if (NCX > 6 && ios == 1)
            build_sig_tile<MODEL, RB, NCX, (NCX > 6 ? 1 : 0), PW>(d, tile, (int)ti, z, cams, lambda, scale, S, g_red, Vinv, gp, jn2p, partial, pivmm, sg_chunk, sg_tile_chunk0, sg_lc, sg_uv, sg_w, tile_ctr + 2, sy, sh);
        else if (NCX > 6 && ios == 2)
            build_sig_tile<MODEL, RB, NCX, (NCX > 6 ? 2 : 0), PW>(d, tile, (int)ti, z, cams, lambda, scale, S, g_red, Vinv, gp, jn2p, partial, pivmm, sg_chunk, sg_tile_chunk0, sg_lc, sg_uv, sg_w, tile_ctr + 2, sy, sh);
        else
            build_sig_tile<MODEL, RB, NCX, 0, PW>(d, tile, (int)ti, z, cams, lambda, scale, S, g_red, Vinv, gp, jn2p, partial, pivmm, sg_chunk, sg_tile_chunk0, sg_lc, sg_uv, sg_w, tile_ctr + 2, sy, sh);
        __syncthreads();                             // the tile's LDS is free again (and s_ticket may be redrawn)
    }
}

// k_backsub_sig: back-substitution dp = -V^-1 (g_p + W' dc) and ||J p||^2 over the image rows for the
// points of the signature chunks (K7, K8).  One wave per chunk, lane = object point; the chunk's
// cameras are the same for every lane, so their records and their steps dc are wave-uniform
// (scalar loads), (u,v) comes coalesced from the slot-major copy, and the sums over a point's
// observations stay in the lane's registers: no LDS, no atomics, no per-lane camera gathers.
// One sweep over the k cameras (see the formula below).
template <int MODEL, int NCX>
__global__ __launch_bounds__(256) void k_backsub_sig(DevProblem d, const double *__restrict__ z,
                                                     const CamRec *__restrict__ cams,
                                                     const double *__restrict__ Vinv, const double *__restrict__ gp,
                                                     double *__restrict__ dz, double *__restrict__ partial /* [grid][2] */,
                                                     const int32_t *__restrict__ sg_chunk, int nchunks,
                                                     const int32_t *__restrict__ sg_gcam,
                                                     const double *__restrict__ sg_uv, const double *__restrict__ sg_w) {
    constexpr int CW = (int)((sizeof(CamRec) + 7) / 8);
    __shared__ double sh[8];
    __shared__ double crec[4][SIG_KMAXR][CW];        // the chunk's camera records, fetched once by the whole wave
    __shared__ double dcs[4][SIG_KMAXR][NCX];        // ... and their steps dc, column by column
    const int t = threadIdx.x, lane = t & 63;
    const int wv = t >> 6;
    const int ch = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (t >> 6));
    double acc[2] = {0, 0};
    if (ch < nchunks) {
        const int32_t *cd = sg_chunk + 8 * (int64_t)ch;
        const int pt0 = cd[0], npts = cd[1], k = cd[2], gm = cd[4], gi0 = cd[5], uv0 = cd[6];
        const int mycam = lane < k ? sg_gcam[16 * (int64_t)ch + lane] : 0;
        for (int base = 0; base < k * CW; base += 64) {      // whole-wave exchange: ds_bpermute reads 0 from lanes outside EXEC
            const int idx = base + lane, c = min(idx / CW, k - 1), f = idx - c * CW;
            const int cam = __builtin_amdgcn_ds_bpermute(4 * c, mycam);
            if (idx < k * CW) crec[wv][c][f] = reinterpret_cast<const double *>(cams + cam)[f];
        }
        __builtin_amdgcn_wave_barrier();
        for (int idx = lane; idx < k * NCX; idx += 64) {
            const int c = idx / NCX, a = idx - c * NCX;
            const CamRec &C = *reinterpret_cast<const CamRec *>(crec[wv][c]);
            const bool on = a < (NCX > 6 ? min(C.ncol, NCX) : 6) && (NCX > 6 || ((C.eo_est >> a) & 1u));
            dcs[wv][c][a] = on ? dz[C.col[a]] : 0.0;
        }
        __builtin_amdgcn_wave_barrier();
        // A short chunk takes several lanes per point, as pass 1 of k_build_sig does: with G = 32, 16 or 8 point slots (the
        // smallest that holds the chunk) the lanes l, l + G, ... share the point's cameras and add their sums up afterwards
        // (round 6: a chunk of C2 / C4 has seven points -- 57 of a wave's 64 lanes did nothing while seven ran ten
        // evaluations one after the other).
        const int G = npts > 32 ? 64 : (npts > 16 ? 32 : (npts > 8 ? 16 : 8));
        const int pi = lane & (G - 1), jh = lane / G, jstep = 64 / G;
        const bool act = pi < npts;
        const int pt = pt0 + (act ? pi : 0);
        const int64_t zp = d.NS + 3 * (int64_t)pt;
        const double Q[3] = {z[zp], z[zp + 1], z[zp + 2]};
        const unsigned est = (d.z_est[zp] ? 1u : 0u) | (d.z_est[zp + 1] ? 2u : 0u) | (d.z_est[zp + 2] ? 4u : 0u);
        const double2 *uvp = reinterpret_cast<const double2 *>(sg_uv), *wp = reinterpret_cast<const double2 *>(sg_w);
        const int64_t q0 = uv0 + gi0 + (act ? pi : 0);
        // one sweep over the cameras: with t_j = E_j dc_j (2-vector of observation j),
        //   dp = -V^-1 (g_p + sum_j B_j' t_j)
        //   sum_j |t_j + B_j dp|^2 = sum |t_j|^2 + 2 dp' (sum B_j' t_j) + dp' (sum B_j' B_j) dp
        // so |J p|^2 of the point's rows needs no second evaluation of its observations
        double sB[3] = {0, 0, 0}, V0[6] = {0, 0, 0, 0, 0, 0}, tt2 = 0;
        // (two copies of the sweep, with the weights of the observations or of the camera: a select between a
        // global and an LDS address inside the loop becomes a flat load)
        auto sweep = [&](auto per_obs_w) {
        for (int j = jh; j < k; j += jstep) {
            const CamRec &C = *reinterpret_cast<const CamRec *>(crec[wv][j]);
            double2 uv = {0, 0};
            if constexpr (NCX > 6) uv = uvp[q0 + (int64_t)j * gm];        // (fixed IO: E and B do not depend on the image point)
            double w0, w1;
            if constexpr (decltype(per_obs_w)::value) { const double2 ww = wp[q0 + (int64_t)j * gm]; w0 = ww.x; w1 = ww.y; }
            else { w0 = C.w[0]; w1 = C.w[1]; }
            double B[2][3], t0 = 0, t1 = 0;
            if constexpr (NCX > 6) {
                // self-calibration: t = E dc straight from the pieces of the model (no 2 x 14 block E, no IO
                // derivative block in registers).  The plan routes a problem here only if every camera has
                // the usual eight IO columns (Plan::all_std8); anything else takes k_backsub
                obs_step_dot<MODEL, true>(C, Q, uv.x, uv.y, w0, w1, est, dcs[wv][j], t0, t1, B);
            } else {
                sig_step_dot6(C, Q, w0, w1, dcs[wv][j], t0, t1, B);
            }
            // (two accumulating FMAs per sum: the kernel is bound by the vector instructions it issues)
#define DBAT_ACC2(acc, a, b, c, e) acc = __builtin_fma(a, b, __builtin_fma(c, e, acc))
            DBAT_ACC2(tt2, t0, t0, t1, t1);
            DBAT_ACC2(sB[0], B[0][0], t0, B[1][0], t1);
            DBAT_ACC2(sB[1], B[0][1], t0, B[1][1], t1);
            DBAT_ACC2(sB[2], B[0][2], t0, B[1][2], t1);
            DBAT_ACC2(V0[0], B[0][0], B[0][0], B[1][0], B[1][0]);
            DBAT_ACC2(V0[1], B[0][0], B[0][1], B[1][0], B[1][1]);
            DBAT_ACC2(V0[2], B[0][0], B[0][2], B[1][0], B[1][2]);
            DBAT_ACC2(V0[3], B[0][1], B[0][1], B[1][1], B[1][1]);
            DBAT_ACC2(V0[4], B[0][1], B[0][2], B[1][1], B[1][2]);
            DBAT_ACC2(V0[5], B[0][2], B[0][2], B[1][2], B[1][2]);
#undef DBAT_ACC2
        }
        };
        if (sg_w) sweep(std::true_type{}); else sweep(std::false_type{});
        for (int m = G; m < 64; m <<= 1) {           // the other lanes of the point (uniform loop: butterfly over the slices)
#pragma unroll
            for (int c = 0; c < 3; ++c) sB[c] += lane_get(sB[c], lane ^ m);
#pragma unroll
            for (int c = 0; c < 6; ++c) V0[c] += lane_get(V0[c], lane ^ m);
            tt2 += lane_get(tt2, lane ^ m);
        }
        {
            const double s0 = gp[3 * (int64_t)pt] + sB[0], s1 = gp[3 * (int64_t)pt + 1] + sB[1], s2 = gp[3 * (int64_t)pt + 2] + sB[2];
            double p0, p1, p2;
            point_block_solve_neg(Vinv + 6 * (int64_t)pt, s0, s1, s2, p0, p1, p2);
            const double d0 = (est & 1u) ? p0 : 0.0, d1 = (est & 2u) ? p1 : 0.0, d2 = (est & 4u) ? p2 : 0.0;
            if (act && lane == pi) {                 // one lane per point writes and counts
                dz[zp] = d0; dz[zp + 1] = d1; dz[zp + 2] = d2;
                acc[0] = tt2 + 2.0 * (d0 * sB[0] + d1 * sB[1] + d2 * sB[2])
                       + d0 * (V0[0] * d0 + V0[1] * d1 + V0[2] * d2) + d1 * (V0[1] * d0 + V0[3] * d1 + V0[4] * d2)
                       + d2 * (V0[2] * d0 + V0[4] * d1 + V0[5] * d2);
            }
        }
    }
    block_sum<2>(acc, sh);
    if (t == 0) { partial[2 * blockIdx.x] = acc[0]; partial[2 * blockIdx.x + 1] = acc[1]; }
}

}  // namespace dbat
