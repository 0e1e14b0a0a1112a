// Blocked dense Cholesky of the reduced camera system (K6), FP64, gfx950.
//
// Solves S q = b for the symmetric positive definite reduced system
// (order NS = 6*nc + nIOu) that replaces MATLAB's `\` on the full normal
// matrix (gauss_newton_armijo.m:172, levenberg_marquardt.m:119,
// levenberg_marquardt_powell.m:277) after the object points have been
// eliminated.
//
// Storage: lower triangle of S in a column-major array with leading dimension
// lda >= NS+1 and NS+1 columns; row NS of the array holds the right-hand side
// b' (one extra row below the matrix).  A blocked factorisation S = L L' then
// leaves y' = (L^-1 b)' in that row -- the forward substitution rides along
// with the panel solves -- and a blocked backward substitution L' q = y
// finishes the solve.
//
// Left-looking over block columns of OB = 256, inner panels of NB = 64:
//   outer J :  A[J:, J:J+OB] -= L[J:, 0:J] L[J:J+OB, 0:J]'     one rocBLAS dgemm, k = J
//   inner i :  k_potf2      diagonal 64 x 64 block, held in REGISTERS by one
//                           workgroup (lane = row, wave = column class);
//                           produces L11 and L11^-1
//              k_trsm64     X = A21 L11^-T = A21 (L11^-1)'  as a 64x64x64 product on
//                           the f64 matrix cores (v_mfma_f64_16x16x4_f64)
//              k_update64   rest of the block column -= X X_sub'   (same MFMA tile)
//   backward:  k_backsolve  per 64-panel, last to first, q_j = (L11^-1)' y_j
#pragma once
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <stdint.h>

#include <algorithm>
#include <vector>

namespace dbat {

constexpr int CHOL_NB = 64;     // inner panel width
constexpr int CHOL_OB = 256;    // outer block column width

typedef double chol_d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Cholesky of one 64 x 64 diagonal block, A11 = L L', and Linv = L^-1.
// 320 threads.  Waves 0-3 factor: lane tx = row, wave ty holds columns
// k = ty + 4q (q = 0..15) of its row in registers; every scaled pivot column
// goes to LDS (one barrier per step, the step loop fully unrolled so all
// register indices are static; 1/sqrt by v_rsq_f64 + two Newton steps, no
// division on the critical path).  Wave 4 trails one step behind and applies
// the same eliminations to an identity: lane c holds column c of L^-1.
// info gets the 1-based index of the first non-positive pivot (LAPACK potrf
// convention).
__global__ __launch_bounds__(320) void k_potf2(double *__restrict__ A, int64_t lda, int n, int j0,
                                               double *__restrict__ Linv, int *__restrict__ info) {
    constexpr int NB = 64;
    __shared__ double Lc[NB * NB];                      // Lc[j*NB + i] = L(i, j)
    __shared__ double idv[NB];                          // 1 / L(j, j)
    const int t = threadIdx.x, tx = t & 63, ty = t >> 6;
    const int nb = min(NB, n - j0);
    double *A0 = A + (int64_t)j0 * lda + j0;
    if (ty < 4) {
        double a[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int k = ty + 4 * q;
            a[q] = (tx < nb && k < nb && tx >= k) ? A0[(int64_t)k * lda + tx] : (tx == k ? 1.0 : 0.0);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (ty == (j & 3)) {                        // the wave that owns column j
                const double v = a[j >> 2];
                const double ajj = readlane_f64(v, j);
                if (!(ajj > 0.0) && tx == 0 && j < nb && info && *info == 0) *info = j0 + j + 1;
                double id = __builtin_amdgcn_rsq(ajj);
                id = id * (1.5 - 0.5 * ajj * id * id);
                id = id * (1.5 - 0.5 * ajj * id * id);
                const double l = tx > j ? v * id : (tx == j ? ajj * id : 0.0);
                a[j >> 2] = l;
                Lc[j * NB + tx] = l;
                if (tx == j) idv[j] = id;
            }
            __syncthreads();
            const double lrow = Lc[j * NB + tx];        // L(tx, j)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int k = ty + 4 * q;               // wave-uniform
                if (k > j) a[q] -= lrow * Lc[j * NB + k];   // entries above the diagonal are junk, never read
            }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int k = ty + 4 * q;
            if (tx < nb && k < nb && tx >= k) A0[(int64_t)k * lda + tx] = a[q];
        }
    } else {
        double m[NB];                                   // column tx of L^-1, rows in registers
#pragma unroll
        for (int i = 0; i < NB; ++i) m[i] = i == tx ? 1.0 : 0.0;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            __syncthreads();                            // column j and idv[j] are in LDS
            const double mj = m[j] * idv[j];
            m[j] = mj;
#pragma unroll
            for (int i = j + 1; i < NB; ++i) m[i] -= Lc[j * NB + i] * mj;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) Linv[tx * NB + i] = (i < nb && tx < nb) ? m[i] : 0.0;   // Linv(i, c=tx)
    }
}

// One 64 x 64 x 64 product on the f64 matrix cores, transposed orientation so
// that the result is written along the column-major columns:
//   D(c, r) = sum_m  P(c, m) * Q(r, m),   c, r, m in [0, 64)
// Pm/Qm are LDS images [m][c] / [m][r] with row stride LD.  Wave w computes the
// rows c in [16w, 16w+16) for all 64 r: acc[rt] covers r in [16rt, 16rt+16).
// Result element: acc[rt][e] = D(c = 16w + (lane>>4) + 4e, r = 16rt + (lane&15)).
template <int LD>
__device__ __forceinline__ void mfma_tile64(const double *Pm, const double *Qm, int w, int lane,
                                            chol_d4 acc[4]) {
    // the operands of k-step kk+1 are on their way from LDS while k-step kk multiplies (one wave per
    // SIMD: nothing else hides the LDS latency; bench/potf_micro.hip: 2.81 -> 2.38 us per product)
    const double *pa = Pm + (lane >> 4) * LD + 16 * w + (lane & 15), *pb = Qm + (lane >> 4) * LD + (lane & 15);
    double an = pa[0], bn[4] = {pb[0], pb[16], pb[32], pb[48]};
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
        const double a = an, b0 = bn[0], b1 = bn[1], b2 = bn[2], b3 = bn[3];
        if (kk + 1 < 16) {
            an = pa[4 * (kk + 1) * LD];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) bn[rt] = pb[4 * (kk + 1) * LD + 16 * rt];
        }
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b2, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b3, acc[3], 0, 0, 0);
    }
}

// X = A21 * (L11^-1)' for 64 rows per workgroup (the right-hand-side row
// included):  X(r, c) = sum_m A21(r, m) Linv(c, m).
// Row tiles: the rows below a panel that can hold non-zeros are the BAND rows
// [r_begin, band_end) -- the envelope of the reduced system, see CholEnvelope --
// followed by the dense TAIL rows [tail_begin, nrows_total) (IO unknowns and
// the right-hand-side row).  Tile t < nband is a band tile.
__device__ __forceinline__ void row_tile(int t, int nband, int r_begin, int band_end, int tail_begin,
                                         int nrows_total, int64_t &r0, int &rlimit) {
    if (t < nband) { r0 = (int64_t)r_begin + 64 * (int64_t)t; rlimit = band_end; }
    else { r0 = (int64_t)tail_begin + 64 * (int64_t)(t - nband); rlimit = nrows_total; }
}

__global__ __launch_bounds__(256) void k_trsm64(double *__restrict__ A, int64_t lda, int nrows_total, int j0,
                                                const double *__restrict__ Linv, int nband, int band_end,
                                                int tail_begin) {
    constexpr int NB = 64, LD = 65;
    __shared__ double Pm[NB * LD];                      // Pm[m][c] = Linv(c, m)
    __shared__ double Qm[NB * LD];                      // Qm[m][r] = A21(r, m)
    const int t = threadIdx.x, tx = t & 63, ty = t >> 6;
    int64_t r0; int rlimit;
    row_tile(blockIdx.x, nband, j0 + NB, band_end, tail_begin, nrows_total, r0, rlimit);
    nrows_total = rlimit;
    const bool ok = r0 + tx < nrows_total;
    double *rowp = A + (int64_t)j0 * lda + r0 + tx;
#pragma unroll 4
    for (int mm = ty; mm < NB; mm += 4) {
        Pm[mm * LD + tx] = Linv[mm * NB + tx];
        Qm[mm * LD + tx] = ok ? rowp[(int64_t)mm * lda] : 0.0;
    }
    __syncthreads();
    chol_d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    mfma_tile64<LD>(Pm, Qm, ty, tx, acc);
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        const int64_t r = r0 + 16 * rt + (tx & 15);
        if (r < nrows_total)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = 16 * ty + (tx >> 4) + 4 * e;
                A[(int64_t)(j0 + c) * lda + r] = acc[rt][e];
            }
    }
}

// Inner update of the current block column:  for the 64-column block cb (columns
// [cj, cj+64)) and 64 rows per workgroup:
//   A(r, cj + c) -= sum_m X(r, m) X(cj + c, m),   X = A[:, j0:j0+64]
// gridDim.y indexes the column blocks to the right of the panel inside the
// outer block; only rows r >= cj are touched.
__global__ __launch_bounds__(256) void k_update64(double *__restrict__ A, int64_t lda, int nrows_total, int j0,
                                                  int band_end, int tail_begin) {
    constexpr int NB = 64, LD = 65;
    __shared__ double Pm[NB * LD];                      // Pm[m][c] = X(cj + c, m)
    __shared__ double Qm[NB * LD];                      // Qm[m][r] = X(r0 + r, m)
    const int t = threadIdx.x, tx = t & 63, ty = t >> 6;
    const int cb = blockIdx.y;
    const int64_t cj = (int64_t)j0 + NB * (cb + 1);
    const int ntot = nrows_total;
    // rows of this column block: band [cj, band_end) then tail [max(tail_begin,cj), ntot)
    const int bend = band_end > cj ? band_end : (int)cj;
    const int nband = (bend - (int)cj + 63) / 64;
    const int tb = tail_begin > cj ? tail_begin : (int)cj;
    int64_t r0; int rlimit;
    row_tile(blockIdx.x, nband, (int)cj, bend, tb > bend ? tb : bend, ntot, r0, rlimit);
    nrows_total = rlimit;
    if (r0 >= nrows_total) return;
    const bool ok = r0 + tx < nrows_total;
    const bool okc = cj + tx < ntot;
    const double *Xc = A + (int64_t)j0 * lda + cj + tx;
    const double *Xr = A + (int64_t)j0 * lda + r0 + tx;
#pragma unroll 4
    for (int mm = ty; mm < NB; mm += 4) {
        Pm[mm * LD + tx] = okc ? Xc[(int64_t)mm * lda] : 0.0;
        Qm[mm * LD + tx] = ok ? Xr[(int64_t)mm * lda] : 0.0;
    }
    __syncthreads();
    chol_d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    mfma_tile64<LD>(Pm, Qm, ty, tx, acc);
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        const int64_t r = r0 + 16 * rt + (tx & 15);
        if (r < nrows_total)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int64_t c = cj + 16 * ty + (tx >> 4) + 4 * e;
                if (r >= c) A[c * lda + r] -= acc[rt][e];
            }
    }
}

// Left-looking update of the dense tail rows (IO unknowns, right-hand side) for
// one outer block column:  C(tail+r, J+c) -= sum_{k<J} L(tail+r, k) L(J+c, k).
// A skinny product (<= 64 rows per z-slice, <= 256 columns, k = J): workgroup
// (x = 64-column tile, y = 128-wide k chunk, z = 64-row tile) runs two
// 64x64x64 MFMA tiles and adds its piece with f64 atomics.
__global__ __launch_bounds__(256) void k_tail_gemm(double *__restrict__ A, int64_t lda, int J, int ob,
                                                   int tail_begin, int nrows_total) {
    constexpr int NB = 64, LD = 65;
    __shared__ double Pm[NB * LD];                      // Pm[m][c] = L(J + 64*ct + c, k0 + m)
    __shared__ double Qm[NB * LD];                      // Qm[m][r] = L(tail_begin + 64*rt + r, k0 + m)
    const int t = threadIdx.x, tx = t & 63, ty = t >> 6;
    const int ct = blockIdx.x, rt = blockIdx.z;
    const int64_t crow = (int64_t)J + 64 * ct + tx;     // row of L that belongs to column J+64ct+tx of C
    const int64_t rrow = (int64_t)tail_begin + 64 * rt + tx;
    const bool okc = 64 * ct + tx < ob, okr = rrow < nrows_total;
    chol_d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int k0 = blockIdx.y * 128; k0 < min(J, (int)(blockIdx.y + 1) * 128); k0 += 64) {
        __syncthreads();
#pragma unroll 4
        for (int mm = ty; mm < NB; mm += 4) {
            const bool okk = k0 + mm < J;
            Pm[mm * LD + tx] = (okc && okk) ? A[(int64_t)(k0 + mm) * lda + crow] : 0.0;
            Qm[mm * LD + tx] = (okr && okk) ? A[(int64_t)(k0 + mm) * lda + rrow] : 0.0;
        }
        __syncthreads();
        mfma_tile64<LD>(Pm, Qm, ty, tx, acc);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t r = (int64_t)tail_begin + 64 * rt + 16 * q + (tx & 15);
        if (r < nrows_total)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = 64 * ct + 16 * ty + (tx >> 4) + 4 * e;
                if (c < ob) unsafeAtomicAdd(A + (int64_t)(J + c) * lda + r, -acc[q][e]);
            }
    }
}

// Backward substitution L' q = y, one launch per 64-panel from the last to the
// first.  y (n entries, work vector) holds the right-hand side already reduced
// by the panels after j0; the solved panel goes to qout.  Every workgroup
// first computes q_j = (L11^-1)' y_j redundantly (64 dot products), then
// removes the panel's contribution from its slice of the earlier entries:
//   y[0:j0] -= L[j0:j0+nb, 0:j0]' q_j.
__global__ __launch_bounds__(256) void k_backsolve(const double *__restrict__ A, int64_t lda, int n, int j0,
                                                   const double *__restrict__ Linv,
                                                   double *__restrict__ y, double *__restrict__ qout, int c_begin) {
    constexpr int NB = 64;
    __shared__ double q[NB];
    __shared__ double part[4][NB];
    const int t = threadIdx.x, tx = t & 63, ty = t >> 6;
    const int nb = min(NB, n - j0);
    // q(c) = sum_{k>=c} Linv(k, c) y(k): wave ty sums k = ty, ty+4, ... ; lane = c
    {
        double s = 0;
#pragma unroll 4
        for (int k = ty; k < NB; k += 4) {
            const double yk = k < nb ? y[j0 + k] : 0.0;
            s += Linv[tx * NB + k] * yk;               // Linv(k, c=tx) stored at [c*NB + k]
        }
        part[ty][tx] = s;
    }
    __syncthreads();
    if (t < NB) {
        const double s = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
        q[t] = s;
        if (blockIdx.x == 0 && t < nb) qout[j0 + t] = s;
    }
    __syncthreads();
    const int c = c_begin + blockIdx.x * 256 + t;       // columns left of the panel's envelope hold zeros
    if (c < j0) {
        const double *col = A + (int64_t)c * lda + j0;
        double v[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) v[k] = k < nb ? col[k] : 0.0;   // all loads in flight together
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
        for (int k = 0; k < NB; k += 4) { s0 += v[k] * q[k]; s1 += v[k + 1] * q[k + 1]; s2 += v[k + 2] * q[k + 2]; s3 += v[k + 3] * q[k + 3]; }
        y[c] -= (s0 + s1) + (s2 + s3);
    }
}

// Envelope (profile) of the reduced system.  Row r of S (and of its Cholesky
// factor: no fill outside the envelope) has its first non-zero in column
// rowfirst[r]; rows >= tail0 are dense (IO unknowns, right-hand side).  For
// bundle adjustment rowfirst comes from the camera co-visibility graph: camera
// c only couples to cameras that share an object point with it.  With image
// sequences / flight strips in acquisition order the band is narrow and the
// block-column updates shrink from (n-J) x J to band x band products; a fully
// connected network degenerates to the dense case (band_end = tail0).
struct CholEnvelope {
    int n = 0, tail0 = 0;
    std::vector<int> rowfirst;      // [n]
    std::vector<int> band_end;      // per 64-column panel jb: rows [.., band_end) reach into columns <= 64*jb+63
    std::vector<int> panel_first;   // per 64-row panel: first non-zero column of any of its rows
    void build(int n_, int tail0_, const std::vector<int> &first) {
        n = n_; tail0 = tail0_ < n_ ? tail0_ : n_; rowfirst = first;
        const int nblk = (n + CHOL_NB - 1) / CHOL_NB;
        std::vector<int> reach(nblk, 0);
        for (int r = 0; r < tail0; ++r) { const int b = rowfirst[r] / CHOL_NB; if (r + 1 > reach[b]) reach[b] = r + 1; }
        band_end.assign(nblk, 0);
        int m = 0;
        for (int b = 0; b < nblk; ++b) { if (reach[b] > m) m = reach[b]; band_end[b] = m; }
        panel_first.assign(nblk, 0);
        for (int b = 0; b < nblk; ++b) {
            int f = n;
            for (int r = b * CHOL_NB; r < (b + 1) * CHOL_NB && r < n; ++r) f = std::min(f, r < tail0 ? rowfirst[r] : 0);
            panel_first[b] = f;
        }
    }
    void build_dense(int n_) { std::vector<int> f(n_, 0); build(n_, n_, f); }
};

struct BlockChol {
    // linv_work: ceil(n/NB) * NB*NB doubles (inverse factors of the diagonal blocks)
    static size_t linv_doubles(int n) { return (size_t)((n + CHOL_NB - 1) / CHOL_NB) * CHOL_NB * CHOL_NB; }

    // Factor the lower triangle of the n x n matrix in A (lda >= n+1, n+1
    // columns allocated) and solve A q = b where b' sits in row n of A.
    // q -> q_out (n entries).  info_dev: LAPACK-style failure index (0 = ok).
    static void solve(rocblas_handle blas, hipStream_t stream, double *A, int64_t lda, int n,
                      double *q_out, double *y_work, double *linv_work, int *info_dev,
                      const CholEnvelope &env) {
        constexpr int NB = CHOL_NB, OB = CHOL_OB;
        (void)hipMemsetAsync(info_dev, 0, sizeof(int), stream);
        const double one = 1.0, mone = -1.0;
        const int ntot = n + 1;                         // matrix rows + the right-hand-side row
        const int nblk = (n + NB - 1) / NB;
        for (int J = 0; J < n; J += OB) {
            const int ob = n - J < OB ? n - J : OB;
            if (J > 0) {
                // left-looking update of the block column with everything factored so far,
                // restricted to the envelope: band rows x the columns they can reach, then
                // the dense tail rows over all columns
                const int jb_last = std::min(nblk - 1, (J + ob - 1) / NB);
                const int tail_begin = std::max(env.tail0, J);
                const int bend = std::min(std::max(env.band_end[jb_last], J), tail_begin);
                int kmin = J;
                for (int r = J; r < bend; ++r) kmin = std::min(kmin, env.rowfirst[r]);
                kmin = (kmin / NB) * NB;
                const int ntail = ntot - tail_begin;
                if (bend == tail_begin && kmin == 0 && bend > J) {
                    // dense case: band and tail are one contiguous row range over all columns
                    rocblas_dgemm(blas, rocblas_operation_none, rocblas_operation_transpose, ntot - J, ob, J, &mone,
                                  A + J, (rocblas_int)lda, A + J, (rocblas_int)lda, &one,
                                  A + (int64_t)J * lda + J, (rocblas_int)lda);
                } else {
                    if (bend > J && kmin < J)
                        rocblas_dgemm(blas, rocblas_operation_none, rocblas_operation_transpose, bend - J, ob, J - kmin,
                                      &mone, A + (int64_t)kmin * lda + J, (rocblas_int)lda,
                                      A + (int64_t)kmin * lda + J, (rocblas_int)lda, &one,
                                      A + (int64_t)J * lda + J, (rocblas_int)lda);
                    if (ntail > 256) {
                        rocblas_dgemm(blas, rocblas_operation_none, rocblas_operation_transpose, ntail, ob, J,
                                      &mone, A + tail_begin, (rocblas_int)lda, A + J, (rocblas_int)lda, &one,
                                      A + (int64_t)J * lda + tail_begin, (rocblas_int)lda);
                    } else if (ntail > 0) {
                        hipLaunchKernelGGL(k_tail_gemm, dim3((ob + 63) / 64, (J + 127) / 128, (ntail + 63) / 64),
                                           dim3(256), 0, stream, A, lda, J, ob, tail_begin, ntot);
                    }
                }
            }
            for (int j0 = J; j0 < J + ob; j0 += NB) {
                double *Linv = linv_work + (size_t)(j0 / NB) * NB * NB;
                hipLaunchKernelGGL(k_potf2, dim3(1), dim3(320), 0, stream, A, lda, n, j0, Linv, info_dev);
                const int nb = n - j0 < NB ? n - j0 : NB;
                const int below = ntot - (j0 + nb);
                if (below <= 0) break;
                if (nb == NB) {
                    const int r_begin = j0 + NB;
                    const int tail_begin = std::max(env.tail0, r_begin);
                    const int bend = std::min(std::max(env.band_end[j0 / NB], r_begin), tail_begin);
                    const int nband = (bend - r_begin + 63) / 64;
                    const int ntail = (ntot - tail_begin + 63) / 64;
                    hipLaunchKernelGGL(k_trsm64, dim3(nband + ntail), dim3(256), 0, stream, A, lda, ntot, j0, Linv,
                                       nband, bend, tail_begin);
                    const int ncb = (J + ob - (j0 + NB) + NB - 1) / NB;   // column blocks left in this outer block
                    if (ncb > 0)
                        hipLaunchKernelGGL(k_update64, dim3(nband + ntail, ncb), dim3(256), 0, stream, A, lda, ntot, j0,
                                           bend, tail_begin);
                } else {
                    // ragged last panel (nb < 64): it is the last one, only the rhs row is below
                    rocblas_dtrsm(blas, rocblas_side_right, rocblas_fill_lower, rocblas_operation_transpose,
                                  rocblas_diagonal_non_unit, below, nb, &one, A + (int64_t)j0 * lda + j0, (rocblas_int)lda,
                                  A + (int64_t)j0 * lda + j0 + nb, (rocblas_int)lda);
                }
            }
        }
        // y' is row n of A: gather it, then the blocked backward substitution
        (void)hipMemcpy2DAsync(y_work, sizeof(double), A + n, lda * sizeof(double), sizeof(double), n,
                               hipMemcpyDeviceToDevice, stream);
        const int last = ((n - 1) / NB) * NB;
        for (int j0 = last; j0 >= 0; j0 -= NB) {
            const int c_begin = std::min(j0, (env.panel_first[j0 / NB] / 256) * 256);
            const int grid = j0 > c_begin ? (j0 - c_begin + 255) / 256 : 1;
            hipLaunchKernelGGL(k_backsolve, dim3(grid), dim3(256), 0, stream, A, lda, n, j0,
                               linv_work + (size_t)(j0 / NB) * NB * NB, y_work, q_out, c_begin);
        }
    }
};

}  // namespace dbat
