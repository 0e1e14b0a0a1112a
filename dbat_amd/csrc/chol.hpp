// Shared pieces of the Cholesky of the reduced camera system (K6), FP64, gfx950: the 64 x 64 x 64 tile
// product on the f64 matrix cores and the envelope of the reduced system (camera co-visibility band).
// The factorisation itself is the persistent dataflow kernel of chol_df.hpp; it replaces MATLAB's `\`
// on the full normal matrix (gauss_newton_armijo.m:172, levenberg_marquardt.m:119,
// levenberg_marquardt_powell.m:277) after the object points have been eliminated.
//
// Storage of the reduced system: lower triangle of S in a column-major array with leading dimension
// lda >= NS+1 and NS+1 columns; row NS of the array holds the right-hand side b' (one extra row below the
// matrix), so that the forward substitution rides along with the factorisation.
// (Rounds 1-2 also kept a multi-launch blocked factorisation on rocBLAS here -- BlockChol, four launches per
// 64-column panel -- for A/B runs; it was 4x slower than the dataflow kernel and is gone.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <vector>

namespace dbat {

constexpr int CHOL_NB = 64;     // inner panel width

typedef double chol_d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
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


}  // namespace dbat
