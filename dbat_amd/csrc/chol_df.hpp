// Dataflow Cholesky of the reduced camera system (K6): ONE persistent kernel.
//
// The blocked factorisation of chol.hpp is a chain of ~n/64 dependent panels
// with three or four kernel launches each; for the narrow envelopes of bundle
// adjustment (camera co-visibility band of ~13 tiles) every one of those
// kernels is latency-, not throughput-bound.  Here the same arithmetic runs as
// a task graph inside one launch:
//
//   task (i,k), i >= k, one 64 x 64 tile of the envelope, left-looking:
//       T  = A(i,k) - sum_{j<k} L(i,j) L(k,j)'              f64 MFMA, accumulators in registers, each term
//                                                           as soon as its two tiles exist; the tiles of the
//                                                           next term are in flight while one multiplies
//       i == k :  L(k,k) = chol(T), Linv_k = L(k,k)^-1      (df_potf2: 16-column panels in registers,
//                                                           trailing updates on the matrix cores)
//                 L(p,k) = T(p,k) Linv_k'                   for the FIRST tile below the diagonal (the next
//                                                           link of the dependent chain), whose own task
//                                                           only sums and publishes T (DfJob modes 1, 2)
//       i  > k :  L(i,k) = T Linv_k'                        f64 MFMA, the other tiles of the column
//   The sum runs over the columns j whose tiles exist in both tile rows (the
//   pattern is closed under fill).  Sums of more than 96 terms (the dense IO
//   rows, the right-hand side) are cut into helper tasks that leave partial sums.
//
// Two layouts (DfView): IN PLACE -- S where it lies, natural order, pattern =
// envelope of the camera co-visibility band; PERMUTED -- the cameras in
// nested-dissection order, the factor in compact 64 x 64 tiles (every task
// fetches its tile of P S P' from S itself), pattern from a symbolic tile
// factorisation.  With the permutation the independent parts of the camera
// network factor concurrently and only the separators form a dependent chain;
// in natural (acquisition) order all n/64 panels do.
//
// Workgroups take tasks from an atomic counter.  The list is in a topological
// order -- every dependency of a task has a smaller number and is therefore
// already owned by a running workgroup: no deadlock for any grid size -- chosen
// among several such orders by a simulation of the kernel as a list schedule
// (finish_setup).  Completion is published per tile by storing the solve's epoch
// into the tile's flag after the tile's own stores have completed; tiles and
// flags move with agent-scope (sc1) accesses because the L2 of the eight XCDs
// are not coherent among themselves.  A spin cap turns any scheduling accident
// into an error code instead of a hung GPU.
//
// The right-hand side is tile row nT (one valid row, row n of the array): the
// forward substitution rides along exactly as in chol.hpp; the backward
// substitution follows as one task per panel.
#pragma once
#include "env.hpp"
#include <cstdio>
#include <functional>
#include <queue>

#include "chol.hpp"
#include "nd.hpp"

namespace dbat {

constexpr int DF_LD = 65;
constexpr int DF_SPIN_CAP = 1 << 22;

struct DfTask { int i, k; };
// One entry of the task list.  np >= 0: the task that owns tile (i,k): the products j in [jlo, k),
// then the partial sums of its np helpers (slots part .. part+np-1), then the factorisation / the
// L^-1 product.  np < 0: a helper: the products j in [jlo, jhi) of tile (i,k) into partial slot
// `part` -- the sums of the dense rows (IO unknowns, right-hand side: one product per tile column
// of the whole system) are cut into pieces that run side by side.
// mode: 0 plain; 1: tile (i,k) is the FIRST tile below the diagonal of its column (the next link of
// the dependent chain): its task only sums, publishes T = A - sum in the tile and is done -- the
// diagonal task of column k (mode 2, p = that tile row) multiplies it by L^-T right after its own
// factorisation, so the chain does not pass through a second workgroup (store, flag, load of L^-1).
// mode 4 (multi-GPU, phase A): tile (i,k) of a TOP separator column: this rank's share of its Schur complement,
// T_r = A_r(i,k) - sum over the columns j of the rank's OWN domain, is left in the tile and summed over the
// ranks before the top separators are factored (phase B, every rank).  | 8: no identity on the padding rows
// (they are contributed by rank 0 only).
struct DfJob { int i, k, jlo, jhi, part, np, mode, p; };

// Tiles that one workgroup writes and others read inside the same launch move
// with agent-scope relaxed atomics (sc1 loads/stores that bypass the per-XCD
// L2): no cache-wide write-back/invalidate is needed around the flags.
__device__ __forceinline__ double ld_coh(const double *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_coh(double *p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// 16-byte agent-scope loads (global_load_dwordx4 sc1): 8-byte sc1 accesses run at 0.54 ... 0.70 of
// their rate.  The compiler does not count inline-asm loads in its s_waitcnt bookkeeping: every
// group of loads is closed by df_wait16 on the loaded registers before their first use.
typedef double df_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ df_d2 ld_coh16(const double *p) {
    df_d2 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// A FINISHED tile of the compact storage never changes again inside the launch, nobody but its
// owner touched its (tile-private, 32 KB aligned) cache lines before it was finished, and the L2
// are invalidated between launches: after its flag has been seen a tile may be read THROUGH the L2
// of the reader's XCD (sc0: past the CU's vector cache only).  Every tile is read by dozens of
// tasks; with sc1 loads each of those reads went to memory (C4: 7.8 GB per factorisation).
// Not for the in-place layout, where the right-hand-side row shares lines with the last tile row.
__device__ __forceinline__ df_d2 ld_l2_16(const double *p) {
    df_d2 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ double ld_tile(const double *p, bool l2) {
    return l2 ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
              : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void df_wait16(df_d2 &a, df_d2 &b) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b) : : "memory");
}
// 64 x 64 tile (column-major, leading dimension ld, ld even, 16-byte aligned) -> Pm[c*LD + r]:
// thread (tx, ty) takes the row pair 2*(tx&31) of the columns 2*(ty + 4q) + (tx>>5), q < 8
template <int LD>
__device__ __forceinline__ void df_load_tile16(const double *T, int64_t ld, double *Pm, int tx, int ty, bool l2) {
    const double *src = T + 2 * (tx & 31);
    df_d2 v[8];
    if (l2) {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = ld_l2_16(src + (int64_t)(2 * (ty + 4 * q) + (tx >> 5)) * ld);
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = ld_coh16(src + (int64_t)(2 * (ty + 4 * q) + (tx >> 5)) * ld);
    }
#pragma unroll
    for (int q = 0; q < 8; q += 2) df_wait16(v[q], v[q + 1]);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        double *dst = Pm + (2 * (ty + 4 * q) + (tx >> 5)) * LD + 2 * (tx & 31);
        dst[0] = v[q].x; dst[1] = v[q].y;
    }
}
// The same in two halves for the pipelined sum: the loads of the next product are in flight while
// the matrix cores work on the current one.
__device__ __forceinline__ void df_tile16_issue(const double *T, int64_t ld, df_d2 (&v)[8], int tx, int ty, bool l2) {
    const double *src = T + 2 * (tx & 31);
    if (l2) {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = ld_l2_16(src + (int64_t)(2 * (ty + 4 * q) + (tx >> 5)) * ld);
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = ld_coh16(src + (int64_t)(2 * (ty + 4 * q) + (tx >> 5)) * ld);
    }
}
template <int LD>
__device__ __forceinline__ void df_tile16_commit(df_d2 (&v)[8], double *Pm, int tx, int ty) {
#pragma unroll
    for (int q = 0; q < 8; q += 2) df_wait16(v[q], v[q + 1]);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        double *dst = Pm + (2 * (ty + 4 * q) + (tx >> 5)) * LD + 2 * (tx & 31);
        dst[0] = v[q].x; dst[1] = v[q].y;
    }
}

// Thread 0 of the workgroup waits until *flag == epoch.  Returns false on abort.
__device__ __forceinline__ bool df_spin(const int *flag, int epoch, int *abort_flag) {
    int spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
        __builtin_amdgcn_s_sleep(1);
        ++spins;
        if ((spins & 255) == 0) {
            if (spins > DF_SPIN_CAP) __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
        }
    }
    return true;
}

// Cholesky of one 64 x 64 block and its inverse on 256 threads, blocked by 16
// columns, entirely in LDS/registers.  Tm[c*DF_TLD + r], r in [0,128), holds the
// AUGMENTED block [T; I]: the column operations that turn T into L turn the
// identity rows into L^-T, i.e. row 64+c ends up as column c of L^-1.
// Per 16-column panel p:
//   A  every wave holds the 16 diagonal rows of the panel in lanes 0-15 and 16
//      of the 64 other live rows (rows below the diagonal block + the identity
//      rows that are non-zero so far) in lanes 16-31, one row per lane, the 16
//      panel columns in registers.  16 elimination steps: pivot and multipliers
//      travel by v_readlane (SGPR broadcast), no LDS and no barrier inside.
//   B  trailing update of the columns right of the panel, 16 x 16 tiles on the
//      f64 matrix cores (4 MFMAs each).
// Entry: T (lower triangle, nb valid rows/columns) in Tm rows 0..63; exit: L
// in the same place, L^-T in rows 64..127.  info: LAPACK index of the first
// non-positive pivot.
constexpr int DF_TLD = 130;     // 64 * 130 doubles = the two 64 x 65 operand tiles

// a += (m of lane K of the same 16-lane row) * b: DPP row_newbcast on the f64 operation itself -- one
// instruction per update where a v_readlane broadcast needs three (bench/potf_micro.hip: 2593 instead of
// 3092 ticks per 16-column panel, same bits).  FIRST: m and b were just written by instructions the
// assembler cannot see from inside the asm statement (DPP read-after-write wait states).
template <int K, bool FIRST>
__device__ __forceinline__ void df_fmac_bcast(double &a, double m, double b) {
    if constexpr (FIRST)
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(m), "v"(b), "n"(K));
    else
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(m), "v"(b), "n"(K));
}
template <int K, bool FIRST = true>
struct DfUpdFrom {
    static __device__ __forceinline__ void run(double (&a)[16], double m, double nly) {
        if constexpr (K < 16) { df_fmac_bcast<K, FIRST>(a[K], m, nly); DfUpdFrom<K + 1, false>::run(a, m, nly); }
    }
};
__device__ __forceinline__ double df_swap16(double v) {          // the value of lane ^ 16 (ds_swizzle, no LDS memory)
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_ds_swizzle((int)(b & 0xffffffffll), 0x401F);
    const int hi = __builtin_amdgcn_ds_swizzle((int)(b >> 32), 0x401F);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// v with the odd 16-lane rows replaced by the even ones (row 1 <- row 0, row 3 <- row 2): the diagonal rows' column
// mirrored into the lanes of the other rows.  ds_bpermute_b32 with the source lane (lane & ~16): the exchange runs in
// the LDS pipe and costs the vector pipe no issue slot -- the elimination is bound by what the vector pipe issues
// (v_permlane16_swap_b32: two swaps, three moves and their wait states per column; bench/potf2_micro.hip).
__device__ __forceinline__ double df_mirror16(double v, int src4 /* 4 * (lane & ~16) */) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_ds_bpermute(src4, (int)(b & 0xffffffffll));
    const int hi = __builtin_amdgcn_ds_bpermute(src4, (int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// Column J of the 16-column panel elimination (see df_potf2), and the columns after it.
template <int J>
struct DfElimCol {
    // m: the diagonal rows' column J in BOTH 16-lane rows of every half of the wave (df_mirror16 of a[J]) -- the
    // multipliers of the other rows (lanes 16-31) are the diagonal rows' (lanes 0-15) entries.  The mirror of the NEXT
    // column is requested as soon as that column is final (right after its update by this one), so that its trip through
    // the LDS pipe runs under the remaining updates.
    static __device__ __forceinline__ void run(double (&a)[16], double &piv, unsigned &badmask, double m, int src4) {
        badmask |= !(piv > 0.0) ? 1u << J : 0u;
        const double araw = a[J];
        constexpr int J1 = J + 1 < 16 ? J + 1 : 0;
        // 1/sqrt(d) = y0 (1 + e/2 + 3 e^2/8) and 1/d = y0^2 (1 + e + e^2), e = 1 - d y0^2 (v_rsq_f64 is good to 2^-23: third
        // order).  Every f64 operation costs its 6.5 ticks of issue whether it depends on the one before or not
        // (bench/potf_micro.hip), so a column costs what it issues: the next pivot is the updated a[J+1] of lane J+1
        // itself, read back after the update (one DPP operation + two v_readlane), not a scalar copy of the same
        // arithmetic (four v_readlane + two operations): 161 -> 134 ticks per column.
        const double y0 = __builtin_amdgcn_rsq(piv);
        const double t = y0 * y0;
        const double e = __builtin_fma(-piv, t, 1.0);
        const double y = __builtin_fma(y0, __builtin_fma(e, 0.375, 0.5) * e, y0);
        const double y2 = __builtin_fma(t, __builtin_fma(e, e, e), t);
        const double nly = -(araw * y2);
        double mn = 0.0;
        if constexpr (J < 15) {
            df_fmac_bcast<J1, true>(a[J1], m, nly);
            mn = df_mirror16(a[J1], src4);
            piv = readlane_f64(a[J1], J1);
            __builtin_amdgcn_sched_barrier(0);       // (... and not after the updates, where the scheduler would put them)
        }
        a[J] = araw * y;
        DfUpdFrom<J + 2, false>::run(a, m, nly);
        if constexpr (J < 15) DfElimCol<J + 1>::run(a, piv, badmask, mn, src4);
    }
};

// Trailing update of df_potf2 after panel p.  Tiles, numbered cb-major: for cb in p+1..3: T row
// blocks rb = cb..3 (rowbase 16rb), then the identity blocks ib = 0..p (rowbase 64+16ib; ib == p is
// touched for the first time: starts from zero).  9 / 7 / 4 tiles for p = 0 / 1 / 2; wave w takes
// tiles w, w+4, w+8.  Straight-line code: all operands of the wave's NS tiles are fetched before
// the first product (a slot beyond the last tile repeats the last tile and is not stored).
template <int NS>
__device__ __forceinline__ void df_trail(double *Tm, int p, int w, int lane) {
    constexpr int LD = DF_TLD;
    const int ntile = p == 0 ? 9 : (p == 1 ? 7 : 4);
    chol_d4 c4[NS];
    double av[NS][4], bv[NS][4];
    int cbs[NS], rbs[NS];
    bool valid[NS];
#pragma unroll
    for (int s3 = 0; s3 < NS; ++s3) {
        int n = w + 4 * s3;
        valid[s3] = n < ntile;
        n = valid[s3] ? n : ntile - 1;
        const int step = n < 4 ? 0 : (n < 7 ? 1 : 2);
        const int cb = p + 1 + step;
        n -= step == 0 ? 0 : (step == 1 ? 4 : 7);
        const int nT_ = 4 - cb;
        const int rowbase = n < nT_ ? 16 * (cb + n) : 64 + 16 * (n - nT_);
        const bool fresh = n == nT_ + p;
        cbs[s3] = cb; rbs[s3] = rowbase;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const double v = Tm[(16 * cb + (lane >> 4) + 4 * e) * LD + (fresh ? 16 * cb : rowbase) + (lane & 15)];
            c4[s3][e] = fresh ? 0.0 : v;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = 16 * p + 4 * q + (lane >> 4);
            av[s3][q] = -Tm[m * LD + 16 * cb + (lane & 15)];           // A[cc][m] = -P(16cb + cc, m)
            bv[s3][q] = Tm[m * LD + rowbase + (lane & 15)];            // B[m][rr] =  P(rowbase + rr, m)
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int s3 = 0; s3 < NS; ++s3) c4[s3] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s3][q], bv[s3][q], c4[s3], 0, 0, 0);
#pragma unroll
    for (int s3 = 0; s3 < NS; ++s3)
        if (valid[s3]) {
#pragma unroll
            for (int e = 0; e < 4; ++e) Tm[(16 * cbs[s3] + (lane >> 4) + 4 * e) * LD + rbs[s3] + (lane & 15)] = c4[s3][e];
        }
}

// What the chain role (df_chain_role) has a factorisation do on the side -- work whose latency then hides under the
// panels instead of standing between two links of the chain:
//   pub_flag        the tile stores this wave issued before the call are awaited under the first panel's elimination and
//                   the tile's flag is raised after that panel's barrier (a write-through takes 2.5 us);
//   look1, look2    flags thread 0 reads while panel 2 / panel 3 are eliminated (-> look_out[2] | look_out[0..1]);
//   tsrc            if look1 was up under panel 2, every thread requests its 16 operand values of the link product
//                   (T(p,k), element kk at tsrc + 4 kk tld) before panel 3: ordinary agent-scope loads the compiler
//                   counts, in flight across the panel (tb_loaded).
struct DfHook {
    int *pub_flag = nullptr;
    int epoch = 0;
    const int *look1 = nullptr, *look2 = nullptr;
    int *look_out = nullptr;
    const double *tsrc = nullptr;
    int64_t tld = 0;
    bool tb_loaded = false;
    double tb[16];
};
struct DfNoHook {};

// bench/potf2_micro.hip -DDBAT_POTF2_STAGES: shader ticks of thread 0 between the stations of a panel, summed over the
// panels into tr[16 + station] (tr then points into LDS)
#ifdef DBAT_POTF2_STAGES
#define DF_STG(n) do { if (t == 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const long long now_ = (long long)__builtin_readcyclecounter(); \
    atomicAdd((unsigned long long *)(tr + 16 + (n)), (unsigned long long)(now_ - stg_last_)); stg_last_ = now_; } } while (0)
#else
#define DF_STG(n) do { } while (0)
#endif
template <bool PUB = false, class HOOK = DfNoHook>
__device__ __forceinline__ void df_potf2(double *Tm, int nb, int j0, int *info, long long *tr, HOOK *H = nullptr) {
    constexpr int LD = DF_TLD;
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
#ifdef DBAT_POTF2_STAGES
    long long stg_last_ = (long long)__builtin_readcyclecounter();
#endif
    // The identity rows are never initialised in LDS: block ib is the identity until panel ib
    // (synthesised in registers there), zero right of it until the trailing update of panel ib
    // (accumulators start from zero there), and nobody reads it left of its diagonal block.
    __syncthreads();                                    // the caller's T is in place
#pragma unroll 1
    for (int p = 0; p < 4; ++p) {
        // ---- A: panel columns [16p, 16p+16)
        {
            const int nlowT = 48 - 16 * p;                  // T rows below the diagonal block
            int row, ident = -1;                            // ident: column of the 1 of a fresh identity row
            if (lane < 16) row = 16 * p + lane;
            else {
                const int s = 16 * w + (lane - 16);
                row = s < nlowT ? 16 * (p + 1) + s : 64 + (s - nlowT);
                if (lane < 32 && s - nlowT >= 16 * p) ident = s - nlowT - 16 * p;
            }
            const bool act = lane < 32;
            int lk1 = 0, lk2 = 0;
            if constexpr (PUB) {
                if (p == 3 && H->tsrc && H->look_out[2] == H->epoch) {
#pragma unroll
                    for (int kk = 0; kk < 16; ++kk) H->tb[kk] = ld_coh(H->tsrc + (int64_t)(4 * kk) * H->tld);
                    H->tb_loaded = true;
                }
                if (p >= 2 && t == 0) {
                    if (H->look1 && !H->tb_loaded) lk1 = __hip_atomic_load(H->look1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (H->look2 && p == 3) lk2 = __hip_atomic_load(H->look2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            DF_STG(0);
            double a[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) a[q] = (act && ident < 0) ? Tm[(16 * p + q) * LD + row] : (ident == q ? 1.0 : 0.0);
            DF_STG(1);
#ifndef DBAT_POTF2_STAGES
            if (tr && t == 0 && p == 1) tr[5] = wall_clock64();
#endif
            unsigned badmask = 0;                           // bit j: pivot j of the panel is not positive
            // The wave is alone on its SIMD and every operation costs its issue slot: a column costs what it
            // issues.  The multipliers c_k = a_j(lane k) are taken BEFORE the column is scaled (DPP row broadcasts
            // inside the f64 operation, see df_fmac_bcast), every update is a_k -= c_k (a_j / d), and the next
            // pivot is the updated a_{j+1} of lane j+1 (DfElimCol).
            // (Padding columns of a ragged block are identity columns: their pivots are 1.)
            double piv = readlane_f64(a[0], 0);
            // (all 64 lanes run it although only the lower 32 hold rows: with the upper half out of EXEC the elimination
            // measured 1.44 instead of 1.03 us per panel -- no pass is skipped, and the branch costs)
            const int src4 = 4 * (lane & ~16);
            DfElimCol<0>::run(a, piv, badmask, df_mirror16(a[0], src4), src4);
            DF_STG(2);
#ifndef DBAT_POTF2_STAGES
            if (tr && t == 0 && p == 1) tr[13] = wall_clock64();
#endif
            if (lane >= 16 ? act : w == 0) {
#pragma unroll
                for (int q = 0; q < 16; ++q) Tm[(16 * p + q) * LD + row] = a[q];
            }
            DF_STG(3);
            if (badmask != 0 && t == 0 && *info == 0) *info = j0 + 16 * p + __builtin_ctz(badmask) + 1;
            if constexpr (PUB) {
                if (p == 2 && t == 0) H->look_out[2] = lk1;
                if (p == 3 && t == 0) { H->look_out[0] = H->tb_loaded ? H->epoch : lk1; H->look_out[1] = lk2; }
            }
        }
        if constexpr (PUB) { if (p == 0 && H->pub_flag) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __syncthreads();
        if constexpr (PUB) {
            if (p == 0 && H->pub_flag && t == 0) __hip_atomic_store(H->pub_flag, H->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        DF_STG(4);
#ifndef DBAT_POTF2_STAGES
        if (tr && t == 0) tr[6 + 2 * p] = wall_clock64();
#endif
        if (p == 3) break;
        // ---- B: columns cb > p:  Tm[16cb + cc][rowbase + rr] -= sum_m P(rowbase + rr, m) P(16cb + cc, m)
        if (p == 0) df_trail<3>(Tm, 0, w, lane);
        else if (p == 1) df_trail<2>(Tm, 1, w, lane);
        else df_trail<1>(Tm, 2, w, lane);
        DF_STG(5);
        __syncthreads();
        DF_STG(6);
#ifndef DBAT_POTF2_STAGES
        if (tr && t == 0) tr[7 + 2 * p] = wall_clock64();
#endif
    }
}

// Every lane of the wave polls the same flag (one transaction).  false on abort.
__device__ __forceinline__ bool df_spin_wave(const int *flag, int epoch, int *abort_flag) {
    int spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
        __builtin_amdgcn_s_sleep(1);
        ++spins;
        if ((spins & 255) == 0) {
            if (spins > DF_SPIN_CAP) __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
        }
    }
    return true;
}

constexpr unsigned long long DF_SENTINEL = 0xFFFFFFFFFFFFFFFFull;   // q entries not yet solved (a NaN no arithmetic produces)

// Where the tiles live.  The kernel addresses tile (i,k) -- rows [64i, 64i+64)
// x columns [64k, 64k+64) of the (possibly permuted) reduced system, i == nT:
// the right-hand-side row -- as  base + toff[i*nT + k] + c*ld + r.
//   in place : base = S, toff = 64k*ldS + 64i (i == nT: + n), ld = ldS
//   tiles    : base = compact tile storage, toff = 4096 * tile number, ld = 64
// rowbits[i] is the set of columns j <= i whose tile (i,j) is structurally
// non-zero after fill (W 64-bit words per row); a task only multiplies tiles
// that exist in both of its rows.
struct DfView {
    double *base;
    const int64_t *toff;        // [(nT+1)*nT], -1 = structurally zero
    int64_t ld;
    const uint64_t *rowbits;    // [(nT+1)*W]
    int W;
    const int *iperm;           // permuted index -> natural index (nullptr: identity)
    const double *S;            // != nullptr (compact tiles): the tasks fetch their tile of P S P' from S themselves
    int64_t ldS;                //   (natural order, lower triangle, right-hand side in row n_nat) instead of a gather launch
    int n_nat;
    int no_l2;                  // 1: tiles with agent-scope loads everywhere (0 would read finished tiles through the L2: no gain measured)
};

// Backward substitution task of panel j:  q_j = Linv_j' (y_j - sum_{i>j} L(i,j)' q_i).
// Lane = row of the tile L(i,j), 16 of its columns per thread in registers; the
// products with q_i accumulate lane-wise over all tiles of the block column and
// are summed across lanes once at the end.  Tiles are fetched before their q_i
// is polled, so when the last q (panel j+1) arrives only 16 FMAs, the reduction
// and the 64 x 64 product with Linv_j' remain.  q entries (in the order of the
// factorised system) double as their own flags (DF_SENTINEL until solved); the
// solution also goes to q_nat in natural order.
__device__ __forceinline__ bool df_backward(double *smem, const DfView &V, int n, int nT, int j,
                                            const int *__restrict__ bk_ptr, const int *__restrict__ bk_idx,
                                            const int *flags, int epoch, int *abort_flag, const double *linv_all,
                                            double *q_out, double *q_nat, const double *qscale, double *dz_out) {
    constexpr int NB = 64, LD = 65;
    const int t = threadIdx.x, tx = t & 63, ty = t >> 6;
    const int nc = min(NB, n - NB * j);
    const int64_t col0 = (int64_t)NB * j;
    double *red = smem;                                 // [c*LD + lane]
    double *part = smem + NB * LD;                      // [4][64]
    double *sv = part + 4 * NB;                         // [64]
    if (!df_spin_wave(flags + (int64_t)j * nT + j, epoch, abort_flag)) return false;
    if (!df_spin_wave(flags + (int64_t)nT * nT + j, epoch, abort_flag)) return false;
    double lv[16];                                      // Linv(k = ty + 4q, c = tx)
    const bool l2 = V.iperm != nullptr && !(V.no_l2 & 1);     // compact tiles: finished tiles through the L2 (see ld_l2_16)
#pragma unroll
    for (int q = 0; q < 16; ++q) lv[q] = ld_tile(linv_all + (size_t)j * NB * NB + tx * NB + ty + 4 * q, l2);
    const double yv = tx < nc ? ld_coh(V.base + V.toff[(int64_t)nT * nT + j] + (int64_t)tx * V.ld) : 0.0;   // y_j(tx)
    double acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.0;
    for (int e = bk_ptr[j]; e < bk_ptr[j + 1]; ++e) {
        const int i = bk_idx[e];
        const int nr = min(NB, n - NB * i);
        if (!df_spin_wave(flags + (int64_t)i * nT + j, epoch, abort_flag)) return false;
        double v[16];
        const double *Lt = V.base + V.toff[(int64_t)i * nT + j] + tx;
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = (tx < nr && ty + 4 * q < nc) ? ld_tile(Lt + (int64_t)(ty + 4 * q) * V.ld, l2) : 0.0;
        double qv = 0.0;
        int spins = 0;
        for (;;) {
            qv = tx < nr ? ld_coh(q_out + (int64_t)NB * i + tx) : 0.0;
            if (__all((unsigned long long)__double_as_longlong(qv) != DF_SENTINEL)) break;
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 255) == 0) {
                if (spins > DF_SPIN_CAP) __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
            }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] += v[q] * qv;
    }
    // sum over lanes: red[c][lane], c = ty + 4q
#pragma unroll
    for (int q = 0; q < 16; ++q) red[(ty + 4 * q) * LD + tx] = acc[q];
    __syncthreads();
    {
        double s = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += red[tx * LD + 16 * ty + r];
        part[ty * NB + tx] = s;
    }
    __syncthreads();
    if (ty == 0) sv[tx] = yv - ((part[tx] + part[NB + tx]) + (part[2 * NB + tx] + part[3 * NB + tx]));
    __syncthreads();
    {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) s += lv[q] * sv[ty + 4 * q];
        red[ty * NB + tx] = s;
    }
    __syncthreads();
    if (ty == 0 && tx < nc) {
        double qj = (red[tx] + red[NB + tx]) + (red[2 * NB + tx] + red[3 * NB + tx]);
        if (qj != qj) qj = __longlong_as_double(0x7ff8000000000000ll);   // any NaN -> the canonical one, never DF_SENTINEL
        st_coh(q_out + col0 + tx, qj);
        if (q_nat != q_out || dz_out) {
            const int zn = V.iperm ? V.iperm[col0 + tx] : (int)(col0 + tx);
            if (zn >= 0 && q_nat != q_out) q_nat[zn] = qj;          // < 0: padding row of the permuted system
            if (zn >= 0 && dz_out) dz_out[zn] = qscale[zn] * qj;    // the step of the unscaled system (D q)
        }
    }
    __syncthreads();                                    // smem is reused by the next task
    return true;
}

// ---- The dependent chain in its own role.
// Consecutive diagonal tiles of a separator are a chain: column k+1 cannot start before the link L(k+1,k) exists.
// Run as ordinary tasks a link costs 19 us (round 4: store -> write-through -> flag -> poll -> load on two parallel
// paths), of which df_potf2 is 8.7.  Here the first workgroups that arrive take the CHAIN ROLE: they factor diagonal
// tiles only, and one that has factored column k keeps L(k+1,k) in LDS and goes on with column k+1 --
//     T(k+1,k+1) = T'(k+1,k+1) - L(k+1,k) L(k+1,k)'
// with T' = the tile minus all EARLIER products, summed in advance by a worker task (DfJob mode 16; it lands in the
// otherwise unused diagonal tile of the compact storage).  Nothing the chain stores is awaited where the chain would
// notice: L^-1(k) goes out right after the factorisation and is acknowledged under the link product, L(k+1,k) is
// acknowledged under the first panel of the next factorisation (df_potf2<true>), and the loads of T(k+1,k) and
// T'(k+1,k+1) are issued BEFORE those stores (the memory counter of a wave retires in order).
// No deadlock: columns are taken by ticket in task order.  A workgroup CONTINUES with column k+1 only if that column's
// T' is there already (two bounded looks): everything else the column needs then depends on finished columns and on
// worker tasks only.  A ticket for a column that continues a chain is passed over while the column before it may still
// go on (giveup[k-1] != epoch); a workgroup that gives up raises giveup[k] and, if the ticket of k+1 has been passed
// over by then (counter > pos[k+1]), takes k+1 itself as if by that ticket (claim[] arbitrates).  Whoever takes a
// column without its link in LDS fetches the link from memory.  A chain workgroup that finds no column left becomes a
// worker.
struct DfChain {
    const int *cols;     // the tile columns in ticket (= task) order
    const int *par;      // [nT] tile row of the first tile below the diagonal of column k (the link); -1: none
    const int *bits;     // [nT] 1: T'(k,k) leaves out the product with column k-1 (k continues a chain)  2: k+1 continues k
    int *claim;          // [nT] == epoch: taken in this solve
    int *dflags;         // [nT] == epoch: T'(k,k) is in the diagonal tile
    int *giveup;         // [nT] == epoch: the workgroup that factored column k does not go on with k+1 (whoever holds k+1's ticket takes it)
    const int *pos;      // [nT] position of column k in cols
    int ncols, nwg, ctr_slot, role_slot;
    int trace_potf2;     // measurement build: clocks inside df_potf2 as well (they cost about a microsecond per factorisation)
};

// T'(k,k) in the accumulator layout of mfma_tile64: element (c = 16 ty + (tx >> 4) + 4 e, r = 16 rt + (tx & 15)).
// (Ordinary agent-scope loads throughout the chain role: the compiler counts them, so they may stay in flight across
// whatever follows.  Registers written by inline-asm loads are only safe while the compiler neither merges nor moves
// them before the wait -- it did, once, and a link was multiplied with whatever the registers held.)
__device__ __forceinline__ void df_issue_acc_tile(const double *T, int64_t ld, double (&tp)[16], int tx, int ty) {
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int e = 0; e < 4; ++e) tp[4 * rt + e] = ld_coh(T + (int64_t)(16 * ty + (tx >> 4) + 4 * e) * ld + 16 * rt + (tx & 15));
}

// returns false on abort (the kernel returns), true when no column is left
__device__ __forceinline__ bool df_chain_role(double *smem, const DfView &V, int nT, const DfChain &C, int *flags, int *tflags,
                                              int *ctl, int epoch, double *linv_all, int *info, double *ldiag,
                                              int *s_word, long long *trace) {
    constexpr int NB = 64, XLD = DF_LD;
    double *Tm = smem, *Xm = smem + NB * DF_TLD;        // the augmented block of df_potf2 | L(k,k-1) as the operand [m][r] of the next update
    const int t = threadIdx.x, tx = t & 63, ty = t >> 6;
    int *abort_flag = ctl + 1, *counter = ctl + C.ctr_slot;
    int *s_task = s_word, *s_ok = s_word + 1, *look = s_word + 2;      // look[3]: flags seen under the factorisation's last panels
    int k = -1;
    int *pub = nullptr;                                 // flag of the link tile stored last, raised once its stores are out
    double tp[16];
    // per-column clocks (measurement build): 0 taken / continued, 1 T' and the link are there, 2 block assembled, 3 factored,
    // 4 flags seen, 5 operands requested, 6 link done, 7 end; 8: 1 = by ticket, 0 = continued; 9: 1 = T' of the next column
    // was there at the first look, 2 = at the second, 0 = not / no next column; 10: 1 = T(p,k) requested under the last panel
#define DF_CLK(slot) do { if (trace && t == 0) trace[(int64_t)k * 16 + (slot)] = wall_clock64(); } while (0)
#define DF_VAL(slot, v) do { if (trace && t == 0) trace[(int64_t)k * 16 + (slot)] = (v); } while (0)
    bool fresh = true;                                  // column k was taken without its link in LDS
    for (;;) {
        if (k < 0) {
            for (;;) {                                  // the next column nobody has taken
                if (t == 0) {
                    const int tk = atomicAdd(counter, 1);
                    int kk = tk < C.ncols ? C.cols[tk] : -1;
                    if (kk >= 0 && (C.bits[kk] & 1) && atomicAdd(C.giveup + kk - 1, 0) != epoch) kk = -2;     // its chain may still go on
                    if (kk >= 0 && atomicExch(C.claim + kk, epoch) == epoch) kk = -2;
                    *s_task = kk;
                }
                __syncthreads();
                k = *s_task;
                __syncthreads();
                if (k != -2) break;
            }
            if (k < 0) return true;
            fresh = true;
        }
        if (fresh) {
            fresh = false;
            DF_CLK(0); DF_VAL(8, 1);
            const bool cont = (C.bits[k] & 1) != 0;
            if (t == 0) {
                bool ok = df_spin(C.dflags + k, epoch, abort_flag);
                if (ok && cont) ok = df_spin(flags + (int64_t)k * nT + (k - 1), epoch, abort_flag);
                *s_ok = ok;
            }
            __syncthreads();
            if (!*s_ok) { if (t == 0) *info = -1; return false; }
            df_issue_acc_tile(V.base + V.toff[(int64_t)k * nT + k], V.ld, tp, tx, ty);
            if (cont) df_load_tile16<XLD>(V.base + V.toff[(int64_t)k * nT + (k - 1)], V.ld, Xm, tx, ty, false);
            __syncthreads();
        }
        DF_CLK(1);
        // ---- column k: T = T' - L(k,k-1) L(k,k-1)' -> the augmented block
        {
            chol_d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
            if (C.bits[k] & 1) mfma_tile64<XLD>(Xm, Xm, ty, tx, acc);
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = 16 * ty + (tx >> 4) + 4 * e, r = 16 * rt + (tx & 15);
                    Tm[c * DF_TLD + r] = r >= c ? tp[4 * rt + e] - acc[rt][e] : 0.0;
                }
        }
        DF_CLK(2);
        const int p = C.par[k];                         // (>= 0: the right-hand-side row lies below every column)
        const bool nxt = (C.bits[k] & 2) != 0;
        double *Tpk = V.base + V.toff[(int64_t)p * nT + k];
        DfHook H;
        H.pub_flag = pub; H.epoch = epoch;
        H.look1 = tflags + k; H.look2 = nxt ? C.dflags + k + 1 : nullptr; H.look_out = look;
        // the operand of the link product straight into registers: thread (tx, ty) multiplies T(r = 16 ty + (tx & 15), m = 4 kk + (tx >> 4))
        H.tsrc = Tpk + (int64_t)(tx >> 4) * V.ld + 16 * ty + (tx & 15); H.tld = V.ld;
        df_potf2<true>(Tm, NB, NB * k, info, trace && C.trace_potf2 ? trace + (int64_t)(nT + k) * 16 : nullptr, &H);
        DF_CLK(3);
        pub = nullptr;
        const double piv = ty == 0 ? Tm[tx * DF_TLD + tx] : 0.0;
        // what the looks under the last panels saw is in LDS for everybody: no barrier on the usual path
        bool have_t = H.tb_loaded || look[0] == epoch;
        bool pre = nxt && look[1] == epoch;
        if (!have_t || (nxt && !pre)) {                 // ... otherwise thread 0 looks again (T(p,k): until it is there)
            if (t == 0) {
                const bool ok = have_t || df_spin(tflags + k, epoch, abort_flag);
                int pr = pre ? 1 : 0;
                if (ok && nxt && !pre && __hip_atomic_load(C.dflags + k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch) pr = 1;
                *s_ok = ok ? 1 + pr : 0;
            }
            __syncthreads();
            const int st = *s_ok;
            if (!st) { if (t == 0) *info = -1; return false; }
            pre = st == 2;
        }
        // (nobody else can take column k+1 before giveup[k] is raised: a plain store marks it)
        if (pre && t == 0) __hip_atomic_store(C.claim + k + 1, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        DF_CLK(4); DF_VAL(9, pre ? 1 : 0); DF_VAL(10, H.tb_loaded ? 1 : 0);
        if (!H.tb_loaded) {
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) H.tb[kk] = ld_coh(H.tsrc + (int64_t)(4 * kk) * V.ld);
        }
        // L^-1 -> Linv[c*64 + i] = Linv(i, c) = Tm[i][64 + c], behind the operand loads (a wave's memory counter retires in order)
        double *Linv = linv_all + (size_t)k * NB * NB;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int c = ty + 4 * q;
            st_coh(Linv + c * NB + tx, (c >> 4) <= (tx >> 4) ? Tm[tx * DF_TLD + 64 + c] : 0.0);
        }
        // T' of the next column BEHIND the stores: it is wanted after the link product, when the stores are out anyway
        if (pre) df_issue_acc_tile(V.base + V.toff[(int64_t)(k + 1) * nT + k + 1], V.ld, tp, tx, ty);
        DF_CLK(5);
        // the link: X(r, c) = sum_{m <= c} T(r, m) Linv(c, m)  (see the diagonal task of k_chol_df)
        const int nrp = p == nT ? 1 : NB;
        {
            chol_d4 x[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
            const double *pl = Tm + (tx & 15) * DF_TLD + 64 + (tx >> 4);
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const double b = H.tb[kk];
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
                    if (kk < 4 * (cb + 1)) {
                        const double a = pl[16 * cb * DF_TLD + 4 * kk];
                        x[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, x[cb], 0, 0, 0);
                    }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // L^-1 is out (issued before the product)
            // ... and its flag goes up with the last wave that gets here (an LDS count, no barrier): the workers' way from
            // L^-1(k) to the sums of column k+1 is as long as the chain's own way to the end of the next factorisation
            if (tx == 0) {
                if (atomicAdd(s_word + 5, 1) == 3) {
                    s_word[5] = 0;
                    __hip_atomic_store(flags + (int64_t)k * nT + k, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = 16 * cb + (tx >> 4) + 4 * e, r = 16 * ty + (tx & 15);
                    if (r < nrp) st_coh(Tpk + (int64_t)c * V.ld + r, x[cb][e]);
                    Xm[c * XLD + r] = x[cb][e];
                }
        }
        if (ldiag && ty == 0) { const int zn = V.iperm[NB * k + tx]; if (zn >= 0) ldiag[zn] = piv; }
        if (t == 0 && nxt && !pre) {                    // a second, bounded look (the loads then queue behind the stores)
            int got = 0;
            for (int s = 0; s < 6 && !got; ++s) {
                if (__hip_atomic_load(C.dflags + k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch)
                    got = atomicExch(C.claim + k + 1, epoch) != epoch ? 1 : -1;
                else __builtin_amdgcn_s_sleep(8);
            }
            *s_task = got > 0;
        }
        __syncthreads();
        DF_CLK(6);
        if (nxt && !pre && *s_task) {
            DF_VAL(9, 2);
            df_issue_acc_tile(V.base + V.toff[(int64_t)(k + 1) * nT + k + 1], V.ld, tp, tx, ty);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                            // every wave's stores of L(k+1,k) are out
            if (t == 0) __hip_atomic_store(flags + (int64_t)p * nT + k, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            DF_CLK(7);
            k = k + 1;
            DF_CLK(0); DF_VAL(8, 0);
            continue;
        }
        if (pre) {                                      // on with column k+1; L(k+1,k)'s flag follows inside its factorisation
            pub = flags + (int64_t)p * nT + k;
            DF_CLK(7);
            k = k + 1;
            DF_CLK(0); DF_VAL(8, 0);
            continue;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) {
            __hip_atomic_store(flags + (int64_t)p * nT + k, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int take = 0;
            if (nxt) {                                  // the next column: to whoever holds its ticket -- or, if that ticket was passed over, to this workgroup
                // Store-buffering pair: a ticket holder adds to the counter and THEN reads giveup[k] (its second
                // atomic depends on the first's result); this side raises giveup[k] and THEN reads the counter.  Both may
                // not see the old value, or column k+1 is never taken: the exchange must have been performed at the L2
                // before the counter is read -- its result is consumed here, so the wave waits for it (a non-returning
                // exchange could still be in flight while the counter, in another L2 channel, is read).
                const int was = atomicExch(C.giveup + k, epoch);
                asm volatile("" ::"v"(was) : "memory");
                if (atomicAdd(counter, 0) > C.pos[k + 1] && atomicExch(C.claim + k + 1, epoch) != epoch) take = 1;
            }
            *s_task = take;
        }
        DF_CLK(7);
        __syncthreads();
        const int take = *s_task;
        __syncthreads();
        if (take) { k = k + 1; fresh = true; }
        else k = -1;
    }
#undef DF_CLK
#undef DF_VAL
}

// CHAIN (compact tiles): the diagonal tiles are factored by the workgroups of the chain role (df_chain_role); the task
// list holds no diagonal task, the sums of the diagonal tiles are DfJob mode 16.  !CHAIN: everything is a task (the
// in-place layout, DBAT_HIP_DF_CHAIN=0).
template <bool CHAIN>
__global__ __launch_bounds__(256) void k_chol_df(DfView V, int n, int nT, const DfJob *__restrict__ tasks,
                                                 int ntasks, int *__restrict__ flags, int *__restrict__ ctl, int epoch,
                                                 double *__restrict__ linv_all, int *__restrict__ info,
                                                 long long *__restrict__ trace, const int *__restrict__ bk_ptr,
                                                 const int *__restrict__ bk_idx, double *__restrict__ q_out,
                                                 double *__restrict__ q_nat, double *__restrict__ ldiag,
                                                 const double *__restrict__ qscale, double *__restrict__ dz_out,
                                                 double *__restrict__ parts, int nparts,
                                                 const int *__restrict__ bk_list, int nbk, int ctr_slot, DfChain C) {
    constexpr int NB = 64, LD = DF_LD;
    __shared__ double smem[(CHAIN ? 3 : 2) * NB * LD];  // Pm | Qm, or the augmented block of df_potf2 (chain role: and the link)
    double *Pm = smem, *Qm = smem + NB * LD;
    __shared__ int s_word[8];                           // task / ok, and three words of the chain role
    int &s_task = s_word[0], &s_ok = s_word[1];
    const int t = threadIdx.x, tx = t & 63, ty = t >> 6;
    int *counter = ctl + ctr_slot, *abort_flag = ctl + 1;     // (slot 0, or 2 for the second launch of a solve)
    const bool l2 = V.iperm != nullptr && !(V.no_l2 & 1);     // compact tiles: finished tiles through the L2 (see ld_l2_16)
    if constexpr (CHAIN) {                              // the first workgroups to arrive factor the diagonal tiles
        if (t == 0) { s_task = atomicAdd(ctl + C.role_slot, 1); s_word[5] = 0; }
        __syncthreads();
        const int arrival = s_task;
        __syncthreads();
        if (arrival < C.nwg &&
            !df_chain_role(smem, V, nT, C, flags, flags + (int64_t)(nT + 1) * nT + nparts, ctl, epoch, linv_all, info, ldiag, s_word,
                           trace ? trace + (int64_t)(ntasks + nT) * 16 : nullptr))
            return;
    }
    for (;;) {
        if (t == 0) s_task = atomicAdd(counter, 1);
        __syncthreads();
        const int task = s_task;
        if (task >= ntasks) {
            // backward substitution, the panels of bk_list (last to first; q_out == nullptr: factor only)
            if (q_out == nullptr || task >= ntasks + nbk) return;
            if (trace && t == 0) trace[task * 16 + 0] = wall_clock64();
            if (!df_backward(smem, V, n, nT, bk_list[task - ntasks], bk_ptr, bk_idx, flags, epoch, abort_flag,
                             linv_all, q_out, q_nat, qscale, dz_out)) {
                if (t == 0) *info = -1;
                return;
            }
            if (trace && t == 0) trace[task * 16 + 4] = wall_clock64();
            continue;
        }
        const DfJob jb = tasks[task];
        const int i = jb.i, k = jb.k;
        const bool helper = jb.np < 0;                  // a piece of a long sum (see DfJob)
        const int jend = jb.jhi;                        // products j in [jb.jlo, jend) (owner of a whole sum: jhi = k)
        int *pflags = flags + (int64_t)(nT + 1) * nT;   // flags of the helpers' partial sums
        int *tflags = pflags + nparts;                  // per column: T of its sum-only task (DfJob mode 1) is in the tile
        const int nr = i == nT ? 1 : min(NB, n - NB * i);
        const int64_t col0 = (int64_t)NB * k;
        const int nc = min(NB, n - NB * k);
        double *Tik = V.base + V.toff[(int64_t)i * nT + k];
        if (trace && t == 0) trace[task * 16 + 0] = wall_clock64();
        // the accumulators (layout: c = 16*ty + (tx>>4) + 4e, r = 16*rt + (tx&15)) start from MINUS the
        // original tile values, so that T = -(acc) at the end and nothing else stays in registers over the sum
        chol_d4 acc[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = 16 * ty + (tx >> 4) + 4 * e, r = 16 * rt + (tx & 15);
                double v = 0.0;
                if (!helper && r < nr && c < nc) {
                    if (V.S) {                          // the tile of P S P' straight from S
                        const int cn = V.iperm[NB * k + c];
                        if (i == nT) v = (r == 0 && cn >= 0) ? V.S[(int64_t)cn * V.ldS + V.n_nat] : 0.0;
                        else {
                            const int rn = V.iperm[NB * i + r];
                            if (cn >= 0 && rn >= 0) v = rn >= cn ? V.S[(int64_t)cn * V.ldS + rn] : V.S[(int64_t)rn * V.ldS + cn];
                            else v = (i == k && r == c && !(jb.mode & 8)) ? 1.0 : 0.0;      // padding rows: identity
                        }
                    } else v = Tik[(int64_t)c * V.ld + r];
                }
                acc[rt][e] = -v;
            }
        bool alive = true;
        // T -= L(i,j) L(k,j)' over the columns j < k present in both tile rows
        const uint64_t *bi = V.rowbits + (size_t)i * V.W, *bk = V.rowbits + (size_t)k * V.W;
        const int wend = (jend - 1) >> 6;               // last word with a bit below jend (jend = 0: no products)
        auto word = [&](int w) {
            uint64_t m = bi[w] & bk[w];
            if (w == (jb.jlo >> 6)) m &= ~0ull << (jb.jlo & 63);
            if (w == (jend >> 6)) m &= (1ull << (jend & 63)) - 1;
            return m;
        };
        int jw = jb.jlo >> 6;
        uint64_t jm = jend > jb.jlo ? word(jw) : 0ull;
        auto next_j = [&]() -> int {                    // next common column in [jlo, jend), -1 at the end
            for (;;) {
                if (jm) { const int j = 64 * jw + __builtin_ctzll(jm); jm &= jm - 1; return j; }
                if (jend <= jb.jlo || ++jw > wend) return -1;
                jm = word(jw);
            }
        };
        auto spin_both = [&](int j) {                   // thread 0: both tiles of product j exist (blocking)
            bool ok = df_spin(flags + (int64_t)k * nT + j, epoch, abort_flag);
            if (ok && i != k) ok = df_spin(flags + (int64_t)i * nT + j, epoch, abort_flag);
            return ok;
        };
        // whole tiles (in the compact storage the right-hand-side row is a whole tile too, zero below its row)
        if (nc == NB && (nr == NB || i == k || (V.iperm && i == nT)) && !(V.ld & 1)) {
            // Software pipeline: while the matrix cores work on product n, the tiles of product n+1 are
            // on their way into registers -- if its flags were up when thread 0 looked (one non-blocking
            // look, issued before the wait for the tiles of product n, so its latency hides there).
            // Otherwise the workgroup finishes product n first and then waits.  2.0 instead of 3.7 us per
            // product for a task whose inputs exist (the long sums of the dense IO / right-hand-side rows).
            df_d2 rp[8], rq[8];
#ifdef DBAT_HIP_PROFILING
            const bool abl_f = (V.no_l2 & 6) == 6;                               // 3: ... and no look at the next product's flags either
            const bool abl_q = (V.no_l2 & 6) != 0, abl_p = (V.no_l2 & 4) != 0;   // DBAT_HIP_DF_ABLATE: operand tiles not fetched (timing only)
            if (abl_q) for (int q = 0; q < 8; ++q) { rq[q].x = 1e-3 * tx; rq[q].y = 1e-3 * ty; }
            if (abl_p) for (int q = 0; q < 8; ++q) { rp[q].x = 1e-3 * tx; rp[q].y = 1e-3 * ty; }
#else
            constexpr bool abl_q = false, abl_p = false, abl_f = false;
#endif
            // jc: the product whose tiles are on their way; jn: the one after it.  Thread 0 looks at jn's flags ONE
            // PRODUCT AHEAD -- the look is issued before the MFMA pass of the product before jc and read after jc's tiles
            // have arrived, so its latency (an agent-scope load: 1.5 us) hides under that pass.  Round 4 looked at the top
            // of jc's own round and paid the latency in every product (C4: 10 % of the factorisation).  A look that found
            // the flags down is repeated at the top of the round (fresh), as before.
            int jc = next_j(), jn = -1;
            int f1 = 0, f2 = 0;
            auto look = [&](int j) {
                f1 = f2 = epoch;
                if (t == 0 && j >= 0 && !abl_f) {
                    f1 = __hip_atomic_load(flags + (int64_t)k * nT + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (i != k) f2 = __hip_atomic_load(flags + (int64_t)i * nT + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            };
            // The tile offsets are a table in memory (V.toff) and the compiler reads it with VECTOR loads followed by
            // s_waitcnt vmcnt(0) -- placed between the requests of L(k,j) and L(i,j) that wait also drained the first tile
            // before the second was even requested, in every product (found in the ISA in round 5: 5.2 us per product
            // where the MFMA pass takes 2.4).  The offsets of a product are read ONE ROUND BEFORE its tiles are requested,
            // where nothing is in flight behind them.
            int64_t ok_n = 0, oi_n = 0;              // offsets of jn's tiles
            auto offsets = [&](int j, int64_t &ok_, int64_t &oi_) {
                ok_ = j >= 0 ? V.toff[(int64_t)k * nT + j] : 0;
                oi_ = j >= 0 && i != k ? V.toff[(int64_t)i * nT + j] : 0;
            };
            auto issue = [&](int64_t ok_, int64_t oi_) {
                if (!abl_p) df_tile16_issue(V.base + ok_, V.ld, rp, tx, ty, l2);
                if (i != k && !abl_q) df_tile16_issue(V.base + oi_, V.ld, rq, tx, ty, l2);
            };
            if (jc >= 0) {
                int64_t ok_c, oi_c;
                offsets(jc, ok_c, oi_c);
                jn = next_j();
                offsets(jn, ok_n, oi_n);
                if (t == 0) s_ok = spin_both(jc);
                __syncthreads();
                if (!s_ok) alive = false;
                else {
                    issue(ok_c, oi_c);
                    look(jn);
                }
            }
            while (alive && jc >= 0) {
                // (the row words next_j reads and the offsets are vector loads: taken here, where nothing is in flight
                // behind them)
                const int jnn = jn >= 0 ? next_j() : -1;
                int64_t ok_nn, oi_nn;
                offsets(jnn, ok_nn, oi_nn);
                if (t == 0 && jn >= 0 && (f1 != epoch || f2 != epoch)) look(jn);      // down a product ago: look again
                __syncthreads();                        // the previous MFMA pass has read Pm/Qm
                df_tile16_commit<LD>(rp, Pm, tx, ty);
                if (i != k) df_tile16_commit<LD>(rq, Qm, tx, ty);
                if (t == 0) s_ok = jn >= 0 && f1 == epoch && f2 == epoch;
                __syncthreads();
                const bool early = s_ok;
                if (early) { issue(ok_n, oi_n); look(jnn); }     // (the look's answer is read after the next round's tiles have arrived)
                mfma_tile64<LD>(Pm, i == k ? Pm : Qm, ty, tx, acc);
                if (jn >= 0 && !early) {
                    __syncthreads();                    // s_ok has been read by everybody
                    if (t == 0) s_ok = spin_both(jn);
                    __syncthreads();
                    if (!s_ok) { alive = false; break; }
                    issue(ok_n, oi_n);
                    look(jnn);
                }
                jc = jn; jn = jnn; ok_n = ok_nn; oi_n = oi_nn;
            }
        } else {
            for (int j = next_j(); j >= 0; j = next_j()) {
                if (t == 0) s_ok = spin_both(j);
                __syncthreads();                        // also: the previous MFMA pass has read Pm/Qm
                if (!s_ok) { alive = false; break; }
                const double *Lk = V.base + V.toff[(int64_t)k * nT + j] + tx;     // L(64k + tx, 64j + m)
                const double *Li = V.base + V.toff[(int64_t)i * nT + j] + tx;     // L(row0 + tx, 64j + m)
                const bool okc = tx < nc, okr = tx < nr && i != k;
                double vk[16], vi[16];                  // all loads in flight before the first use
#pragma unroll
                for (int q = 0; q < 16; ++q) vk[q] = okc ? ld_tile(Lk + (int64_t)(ty + 4 * q) * V.ld, l2) : 0.0;
#pragma unroll
                for (int q = 0; q < 16; ++q) vi[q] = okr ? ld_tile(Li + (int64_t)(ty + 4 * q) * V.ld, l2) : 0.0;
#pragma unroll
                for (int q = 0; q < 16; ++q) { Pm[(ty + 4 * q) * LD + tx] = vk[q]; Qm[(ty + 4 * q) * LD + tx] = vi[q]; }
                __syncthreads();
                mfma_tile64<LD>(Pm, i == k ? Pm : Qm, ty, tx, acc);
            }
        }
        if (!alive) { if (t == 0) *info = -1; return; }
        __syncthreads();
        if (helper) {                                   // partial sum -> its slot (tile layout), flag, next task
            double *slot = parts + (size_t)jb.part * 4096;
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = 16 * ty + (tx >> 4) + 4 * e, r = 16 * rt + (tx & 15);
                    st_coh(slot + c * 64 + r, acc[rt][e]);
                }
            if (trace && t == 0) trace[task * 16 + 1] = wall_clock64();
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            if (t == 0) __hip_atomic_store(pflags + jb.part, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (trace && t == 0) trace[task * 16 + 4] = wall_clock64();
            continue;
        }
        for (int h = 0; h < jb.np; ++h) {               // the helpers' pieces of this tile's sum
            if (t == 0) s_ok = df_spin(pflags + jb.part + h, epoch, abort_flag);
            __syncthreads();
            if (!s_ok) { if (t == 0) *info = -1; return; }
            const double *slot = parts + (size_t)(jb.part + h) * 4096;
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = 16 * ty + (tx >> 4) + 4 * e, r = 16 * rt + (tx & 15);
                    acc[rt][e] += ld_coh(slot + c * 64 + r);
                }
            __syncthreads();                            // s_ok is rewritten in the next round
        }
        if (trace && t == 0) trace[task * 16 + 1] = wall_clock64();
        if (jb.mode & 4) {                              // this rank's share of a top separator tile: T_r -> the tile, nothing else
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = 16 * ty + (tx >> 4) + 4 * e, r = 16 * rt + (tx & 15);
                    if (r < nr && c < nc) Tik[(int64_t)c * V.ld + r] = -acc[rt][e];
                }
            __syncthreads();
            continue;
        }
        if (jb.mode == 1 || jb.mode == 16) {            // sum only: T -> the tile; the diagonal task of the column finishes it
                                                        // (16: T' of the diagonal tile itself, for the chain role)
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = 16 * ty + (tx >> 4) + 4 * e, r = 16 * rt + (tx & 15);
                    if (r < nr && c < nc) st_coh(Tik + (int64_t)c * V.ld + r, -acc[rt][e]);
                }
            if (trace && t == 0) trace[task * 16 + 3] = wall_clock64();
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            if (t == 0) __hip_atomic_store(jb.mode == 1 ? tflags + k : C.dflags + k, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (trace && t == 0) trace[task * 16 + 4] = wall_clock64();
            continue;
        }
        double *Linv = linv_all + (size_t)k * NB * NB;
        if (!CHAIN && i == k) {
            // T(r, c) = -acc -> augmented block, ragged part = identity, upper triangle = 0
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = 16 * ty + (tx >> 4) + 4 * e, r = 16 * rt + (tx & 15);
                    const double v = -acc[rt][e];
                    smem[c * DF_TLD + r] = (r < nc && c < nc) ? (r >= c ? v : 0.0) : (r == c ? 1.0 : 0.0);
                }
            if (trace && t == 0) trace[task * 16 + 2] = wall_clock64();
            if (trace && t == 0) trace[task * 16 + 14] = (long long)__builtin_readcyclecounter();
            df_potf2(smem, nc, (int)col0, info, trace ? trace + task * 16 : nullptr);
            if (trace && t == 0) trace[task * 16 + 15] = (long long)__builtin_readcyclecounter();
            if (ldiag && ty == 0 && tx < nc) {          // the pivots, by natural index (k_prior_jv takes their extremes)
                const int zn = V.iperm ? V.iperm[col0 + tx] : (int)(col0 + tx);
                if (zn >= 0) ldiag[zn] = smem[tx * DF_TLD + tx];
            }
            if (jb.mode == 2) {
                __syncthreads();                        // the pivots have been read: the L rows may go
                // The next link of the chain, here instead of in a second workgroup: L(p,k) = T(p,k) L^-T with
                // T from the column's sum-only task (published long ago as a rule) and L^-T where df_potf2
                // left it.  The L rows of the augmented block are free in the compact layout (the diagonal
                // tile is not stored): T goes there, smem[m*DF_TLD + r] = T(r, m).
                const int p = jb.p;
                const int nrp = p == nT ? 1 : min(NB, n - NB * p);
                double *Tpk = V.base + V.toff[(int64_t)p * nT + k];
                // (no prefetch across df_potf2: registers written by in-flight inline-asm loads must not be
                // moved by the compiler, and under df_potf2's register pressure they are)
                if (t == 0) s_ok = df_spin(tflags + k, epoch, abort_flag);
                __syncthreads();
                if (!s_ok) { if (t == 0) *info = -1; return; }
                df_load_tile16<DF_TLD>(Tpk, V.ld, smem, tx, ty, false);
                __syncthreads();
                // X(r, c) = sum_{m <= c} T(r, m) Linv(c, m), Linv(c, m) = smem[c*DF_TLD + 64 + m].  Wave w takes
                // the rows r in [16w, 16w+16) and all four column blocks cb, k-steps m < 16(cb+1) only
                // (L^-1 is lower triangular): 40 instead of 64 products per wave, evenly.
                chol_d4 x[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
                const double *pq = smem + (tx >> 4) * DF_TLD + 16 * ty + (tx & 15);          // T(r = 16w + lane%16, m = 4kk + lane/16)
                const double *pl = smem + (tx & 15) * DF_TLD + 64 + (tx >> 4);               // Linv(c = 16cb + lane%16, m = 4kk + lane/16)
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) {
                    const double b = pq[4 * kk * DF_TLD];
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb)
                        if (kk < 4 * (cb + 1)) {
                            const double a = pl[16 * cb * DF_TLD + 4 * kk];
                            x[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, x[cb], 0, 0, 0);
                        }
                }
                // x[cb][e] = X(c = 16cb + (lane>>4) + 4e, r = 16w + (lane&15))
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int c = 16 * cb + (tx >> 4) + 4 * e, r = 16 * ty + (tx & 15);
                        if (r < nrp && c < nc) st_coh(Tpk + (int64_t)c * V.ld + r, x[cb][e]);
                    }
                __builtin_amdgcn_s_waitcnt(0);
                __syncthreads();                        // L(p,k) is out: the chain goes on while L^-1 is stored
                if (t == 0) __hip_atomic_store(flags + (int64_t)p * nT + k, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // L -> tile (lower triangle), L^-1 -> Linv[c*64 + i] = Linv(i, c) = smem[i][64 + c]
#pragma unroll 4
            for (int c = ty; c < NB; c += 4) {
                // L(k,k) itself is only wanted where the factor stays in S (in place: posterior covariance);
                // in the compact-tile layout nothing reads the diagonal tile again (the solves use L^-1)
                if (!V.iperm && tx < nc && c < nc && tx >= c) st_coh(Tik + (int64_t)c * V.ld + tx, smem[c * DF_TLD + tx]);
                st_coh(Linv + c * NB + tx, (tx < nc && c < nc && (c >> 4) <= (tx >> 4)) ? smem[tx * DF_TLD + 64 + c] : 0.0);   // L^-1 is lower triangular
            }
        } else {
            // T(r, c) = -acc  ->  Qm[c][r]
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = 16 * ty + (tx >> 4) + 4 * e, r = 16 * rt + (tx & 15);
                    Qm[c * LD + r] = -acc[rt][e];
                }
            if (t == 0) s_ok = df_spin(flags + (int64_t)k * nT + k, epoch, abort_flag);
            __syncthreads();
            if (!s_ok) { if (t == 0) *info = -1; return; }
            if (trace && t == 0) trace[task * 16 + 2] = wall_clock64();
            df_load_tile16<LD>(Linv, NB, Pm, tx, ty, l2);
            __syncthreads();
            {
                chol_d4 x[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
                mfma_tile64<LD>(Pm, Qm, ty, tx, x);
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int c = 16 * ty + (tx >> 4) + 4 * e, r = 16 * rt + (tx & 15);
                        if (r < nr && c < nc) st_coh(Tik + (int64_t)c * V.ld + r, x[rt][e]);
                    }
            }
        }
        if (trace && t == 0) trace[task * 16 + 3] = wall_clock64();
        __builtin_amdgcn_s_waitcnt(0);                      // this wave's tile stores have reached the coherence point
        __syncthreads();
        if (t == 0) {
            __hip_atomic_store(flags + (int64_t)i * nT + k, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (trace && t == 0) trace[task * 16 + 4] = wall_clock64();
    }
}

// What k_chol_df expects to find reset when the tasks fetch their tiles themselves: error code,
// task counter / abort flag, "not solved yet" in q.
__global__ void k_df_reset(int *__restrict__ info, int *__restrict__ ctl, unsigned long long *__restrict__ qflag, int nq) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { *info = 0; for (int q = 0; q < 8; ++q) ctl[q] = 0; }
    if (i < nq) qflag[i] = DF_SENTINEL;
}

// ---- Selected inversion: the entries of Z = inv(P S P') ON THE PATTERN OF THE FACTOR, from the compact tile factor
// (Takahashi's recurrence by tiles -- the reference's own tool for this is code/test/sparseinv/sparseinv.c; bundle_cov.m:63-117
// inverts blocks of a permuted Cholesky factor).  With column k of the factor, L_kk and L_Jk (J: the tile rows below the
// diagonal), and M_Jk = L_Jk L_kk^-1:
//     Z_Jk = - Z_JJ M_Jk,        Z_kk = L_kk^-T L_kk^-1 - M_Jk' Z_Jk,
// from the last column to the first.  Z_JJ lies inside the pattern (it is closed under fill) and belongs to later
// columns.  Columns whose J are all done form a LEVEL; a level is two launches (off-diagonal tiles, then the diagonal
// ones), and the M tiles -- which overwrite L, the factor is formed anew by every solve -- one launch up front.
// Cost: sum_k |J_k|^2 tile products, the count of the factorisation itself; memory: one more set of compact tiles
// (C4: 0.35 GB) where the dense inverse took 7.2 GB and 9 TF (rocsolver_dpotri).
// loaders: a 64 x 64 tile stored column-major (ld 64) into an LDS operand image
template <int LD>
__device__ __forceinline__ void si_load(const double *T, double *Pm, int tx, int ty, double sign = 1.0) {     // Pm[c*LD + r] = T(r, c)
#pragma unroll
    for (int q = 0; q < 16; ++q) { const int c = ty + 4 * q; Pm[c * LD + tx] = sign * T[c * 64 + tx]; }
}
template <int LD>
__device__ __forceinline__ void si_load_T(const double *T, double *Pm, int tx, int ty, double sign = 1.0) {   // Pm[r*LD + c] = T(r, c)
#pragma unroll
    for (int q = 0; q < 16; ++q) { const int c = ty + 4 * q; Pm[tx * LD + c] = sign * T[c * 64 + tx]; }
}
struct SiTask { int i, k; };
// M(i,k) = L(i,k) L_kk^-1, in place
__global__ __launch_bounds__(256) void k_selinv_m(double *__restrict__ tiles, const int64_t *__restrict__ toff, int nT,
                                                  const double *__restrict__ linv_all, const SiTask *__restrict__ tasks) {
    constexpr int LD = DF_LD;
    __shared__ double smem[2 * 64 * LD];
    double *Pm = smem, *Qm = smem + 64 * LD;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const SiTask tk = tasks[blockIdx.x];
    double *T = tiles + toff[(int64_t)tk.i * nT + tk.k];
    si_load_T<LD>(linv_all + (size_t)tk.k * 4096, Pm, tx, ty);        // P(c, m) = Linv(m, c)
    si_load<LD>(T, Qm, tx, ty);                                        // Q(r, m) = L(r, m)
    __syncthreads();
    chol_d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    mfma_tile64<LD>(Pm, Qm, ty, tx, acc);                              // D(c, r) = sum_m Linv(m, c) L(r, m) = M(r, c)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int e = 0; e < 4; ++e) T[(16 * ty + (tx >> 4) + 4 * e) * 64 + 16 * rt + (tx & 15)] = acc[rt][e];
}
// Z(i,k) = - sum_{j in J(k)} Z(i,j) M(j,k),  i in J(k)
__global__ __launch_bounds__(256) void k_selinv_off(double *__restrict__ Z, const double *__restrict__ M, const int64_t *__restrict__ toff,
                                                    int nT, const int *__restrict__ bk_ptr, const int *__restrict__ bk_idx,
                                                    const SiTask *__restrict__ tasks) {
    constexpr int LD = DF_LD;
    __shared__ double smem[2 * 64 * LD];
    double *Pm = smem, *Qm = smem + 64 * LD;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const SiTask tk = tasks[blockIdx.x];
    const int i = tk.i, k = tk.k;
    chol_d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int e = bk_ptr[k]; e < bk_ptr[k + 1]; ++e) {
        const int j = bk_idx[e];
        __syncthreads();
        si_load_T<LD>(M + toff[(int64_t)j * nT + k], Pm, tx, ty);                       // P(c, m) = M(j,k)(m, c)
        if (i >= j) si_load<LD>(Z + toff[(int64_t)i * nT + j], Qm, tx, ty);             // Q(r, m) = Z(i,j)(r, m)
        else si_load_T<LD>(Z + toff[(int64_t)j * nT + i], Qm, tx, ty);                  //         = Z(j,i)(m, r)
        __syncthreads();
        mfma_tile64<LD>(Pm, Qm, ty, tx, acc);
    }
    double *T = Z + toff[(int64_t)i * nT + k];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int e = 0; e < 4; ++e) T[(16 * ty + (tx >> 4) + 4 * e) * 64 + 16 * rt + (tx & 15)] = -acc[rt][e];
}
// Z(k,k) = L_kk^-T L_kk^-1 - sum_{j in J(k)} M(j,k)' Z(j,k)   (both triangles are stored)
__global__ __launch_bounds__(256) void k_selinv_diag(double *__restrict__ Z, const double *__restrict__ M, const int64_t *__restrict__ toff,
                                                     int nT, const int *__restrict__ bk_ptr, const int *__restrict__ bk_idx,
                                                     const double *__restrict__ linv_all, const int *__restrict__ cols) {
    constexpr int LD = DF_LD;
    __shared__ double smem[2 * 64 * LD];
    double *Pm = smem, *Qm = smem + 64 * LD;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int k = cols[blockIdx.x];
    chol_d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    si_load_T<LD>(linv_all + (size_t)k * 4096, Pm, tx, ty);            // P(c, m) = Linv(m, c)
    __syncthreads();
    mfma_tile64<LD>(Pm, Pm, ty, tx, acc);                              // D(c, r) = sum_m Linv(m, c) Linv(m, r)
    for (int e = bk_ptr[k]; e < bk_ptr[k + 1]; ++e) {
        const int j = bk_idx[e];
        __syncthreads();
        si_load_T<LD>(Z + toff[(int64_t)j * nT + k], Pm, tx, ty);                       // P(c, m) = Z(j,k)(m, c)
        si_load_T<LD>(M + toff[(int64_t)j * nT + k], Qm, tx, ty, -1.0);                 // Q(r, m) = -M(j,k)(m, r)
        __syncthreads();
        mfma_tile64<LD>(Pm, Qm, ty, tx, acc);
    }
    double *T = Z + toff[(int64_t)k * nT + k];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int e = 0; e < 4; ++e) T[(16 * ty + (tx >> 4) + 4 * e) * 64 + 16 * rt + (tx & 15)] = acc[rt][e];
}

// Host side: schedule (tile pattern, task list, tables) and launch.
//   setup_inplace   factor S where it lies, natural order, envelope pattern
//   setup_permuted  nested-dissection order of the cameras + symbolic tile
//                   factorisation; the numeric phase gathers P S P' into compact
//                   tiles first.  Independent sub-blocks of the camera network
//                   then factor concurrently and only the separators form a chain,
//                   instead of one chain over all n/64 panels.
struct DataflowChol {
    int n = 0, n_nat = 0, nT = 0, W = 0, ntasks = 0, ntiles = 0, epoch = 0, grid = 512;   // n: order of the factorised system (padded)
    bool permuted = false;
    // The caller's last kernel before solve() has done what k_df_reset does (k_finish takes it along: reset_args);
    // consumed by the next solve().
    bool reset_done = false;
    bool reset_args(int *&ctl, unsigned long long *&q, int &nq) const {
        if (!permuted || !d_ctl || !d_qperm) return false;
        ctl = d_ctl; q = reinterpret_cast<unsigned long long *>(d_qperm); nq = nT * CHOL_NB;
        return true;
    }
    long long n_products = 0;                           // 64x64x64 tile products of the update phase (schedule statistic)
    int *d_flags = nullptr, *d_ctl = nullptr, *d_bk_ptr = nullptr, *d_bk_idx = nullptr, *d_iperm = nullptr;
    int64_t *d_toff = nullptr;
    uint64_t *d_rowbits = nullptr;
    DfJob *d_tasks = nullptr, *d_tasksB = nullptr;      // task list (one rank: everything; several: this rank's domain) / the top separators
    int ntasksB = 0, nbk = 0;
    int *d_bk_list = nullptr;                           // panels of the backward substitution in task order
    uint64_t *d_rowbits_top = nullptr;                  // rowbits restricted to the top columns (second launch)
    bool two_phase = false;                             // several ranks: domain + shares, all-reduce of the top tiles, top
    int64_t top_tile0 = 0, n_top_tiles = 0;             // the tiles of the top columns are the last n_top_tiles of d_tiles
    std::vector<double> sim_critical_us, sim_schedule_us;   // per task list: critical path / simulated list schedule (model, us)
    std::vector<DfJob> h_tasksB;
    DfTask *d_tile_ij = nullptr;
    int nparts = 0;                                     // partial-sum slots of the helper tasks
    bool env_l2 = false;                                // finished tiles through the L2: measured in round 2, no gain (kept off)
    int env_df_ablate = 0;                              // measurement build (DBAT_HIP_DF_ABLATE): operand fetches of the tile products off
    // the chain role (df_chain_role): the diagonal tiles of the compact layout, per task list (A: all / this rank's domain, B: top)
    bool use_chain = false;
    int nchain = 0, nchainB = 0, chain_wg = 0, chain_wgB = 0;
    int *d_chain_cols = nullptr, *d_chain_colsB = nullptr, *d_chain_par = nullptr, *d_chain_bits = nullptr, *d_chain_bitsB = nullptr, *d_chain_pos = nullptr;
    double *d_parts = nullptr;
    double *d_tiles = nullptr, *d_qperm = nullptr;
    // selected inversion (selected_inverse): Z on the factor's pattern, the schedule by levels
    double *d_ztiles = nullptr;
    SiTask *d_si_m = nullptr, *d_si_off = nullptr;
    int *d_si_diag = nullptr, *d_perm = nullptr;
    int n_si_m = 0;
    std::vector<int> si_off_ptr, si_diag_ptr;           // per level: ranges into d_si_off / d_si_diag
    std::vector<uint64_t> h_rowbits;
    long long *d_trace = nullptr;                       // optional per-task timestamps (DBAT_HIP_DF_TRACE=file)
    std::vector<DfJob> h_tasks;
    std::vector<int> perm;                              // natural -> permuted (empty: identity)
    // doubles of the linv_work argument of solve(): one 64 x 64 inverse per tile row of the
    // factorised (padded) system
    size_t linv_doubles() const { return (size_t)std::max(nT, 1) * CHOL_NB * CHOL_NB; }

    void release() {
        void *ps[] = {d_flags, d_ctl, d_bk_ptr, d_bk_idx, d_iperm, d_toff, d_rowbits, d_tasks, d_tile_ij, d_tiles, d_qperm, d_parts,
                      d_tasksB, d_bk_list, d_rowbits_top, d_chain_cols, d_chain_colsB, d_chain_par, d_chain_bits, d_chain_bitsB, d_chain_pos,
                      d_ztiles, d_si_m, d_si_off, d_si_diag, d_perm};
        for (void *p : ps) if (p) (void)hipFree(p);
        d_flags = d_ctl = d_bk_ptr = d_bk_idx = d_iperm = nullptr; d_toff = nullptr; d_rowbits = nullptr;
        d_tasks = nullptr; d_tile_ij = nullptr; d_tiles = d_qperm = d_parts = nullptr;
        d_tasksB = nullptr; d_bk_list = nullptr; d_rowbits_top = nullptr;
        d_chain_cols = d_chain_colsB = d_chain_par = d_chain_bits = d_chain_bitsB = d_chain_pos = nullptr;
        d_ztiles = nullptr; d_si_m = d_si_off = nullptr; d_si_diag = d_perm = nullptr; n_si_m = 0;
        si_off_ptr.clear(); si_diag_ptr.clear();
    }
    // per-task timestamps of the last solve (100 MHz ticks): task, i, k, t[0..15], jlo, jhi, mode per line
    void dump_trace(hipStream_t stream, const char *path) const {
        if (!d_trace) return;
        std::vector<long long> h((size_t)(ntasks + 3 * nT) * 16);
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(h.data(), d_trace, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        FILE *f = fopen(path, "w");
        if (!f) return;
        for (int t = 0; t < ntasks + nT; ++t) {
            // owner tasks: i, k; helpers of long sums: -(i + 2), k; backward substitution: -1, panel
            const int i = t < ntasks ? (h_tasks[t].np < 0 ? -(h_tasks[t].i + 2) : h_tasks[t].i) : -1, k = t < ntasks ? h_tasks[t].k : nT - 1 - (t - ntasks);
            fprintf(f, "%d,%d,%d", t, i, k);
            for (int q = 0; q < 16; ++q) fprintf(f, ",%lld", h[(size_t)t * 16 + q]);
            fprintf(f, ",%d,%d,%d\n", t < ntasks ? h_tasks[t].jlo : 0, t < ntasks ? (h_tasks[t].np < 0 ? h_tasks[t].jhi : h_tasks[t].k) : 0,
                    t < ntasks ? h_tasks[t].mode : 0);   // its products: j in [jlo, jhi); mode (DfJob)
        }
        for (int k = 0; k < nT && use_chain; ++k) {      // the chain role's columns (df_chain_role): i = -3
            fprintf(f, "%d,-3,%d", ntasks + nT + k, k);
            for (int q = 0; q < 16; ++q) fprintf(f, ",%lld", h[(size_t)(ntasks + nT + k) * 16 + q]);
            fprintf(f, ",0,0,0\n");
        }
        for (int k = 0; k < nT && use_chain; ++k) {      // ... and the clocks inside their factorisation (df_potf2): i = -4
            fprintf(f, "%d,-4,%d", ntasks + 2 * nT + k, k);
            for (int q = 0; q < 16; ++q) fprintf(f, ",%lld", h[(size_t)(ntasks + 2 * nT + k) * 16 + q]);
            fprintf(f, ",0,0,0\n");
        }
        fclose(f);
    }
    template <class T>
    static bool up(T *&dst, const std::vector<T> &v) {
        if (hipMalloc((void **)&dst, std::max<size_t>(v.size(), 1) * sizeof(T)) != hipSuccess) return false;
        if (!v.empty()) (void)hipMemcpy(dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
        return true;
    }
    // common part: nz[i] = bitset of columns k <= i with a structurally non-zero tile (i, k), i = 0..nT
    // (row nT = right-hand side, all columns), fill included
    // One task list (see DfJob).  PHASE 0: the whole factorisation (one rank).  Several ranks (col_owner: the
    // rank whose domain a tile column belongs to, -1 = top separator / IO): PHASE 1 = the columns of this rank's
    // domain and its share of every top tile (mode 4); PHASE 2 = the top columns, products over top columns only.
    struct JobList { std::vector<DfJob> jobs; std::vector<int> dptr, dep, own, sumjob, presum, bits, par; };
    void build_jobs(int phase, const std::vector<uint64_t> &rowbits, const std::vector<int> &col_owner, int rank, JobList &L) {
        auto has = [&](int i, int k) { return (rowbits[(size_t)i * W + (k >> 6)] >> (k & 63)) & 1ull; };
        const int split_min = std::max(env_int("DBAT_HIP_DF_SPLIT", 96), 2);
        const int chunk = std::max(env_int("DBAT_HIP_DF_CHUNK", 32), 1);
        const bool merge = permuted;                            // compact tiles only (see k_chol_df)
        auto top = [&](int k) { return phase != 0 && col_owner[k] < 0; };
        auto mine = [&](int k) { return phase == 0 || col_owner[k] == rank; };
        int dom_lo = nT, dom_hi = 0;                            // this rank's domain: a contiguous range of tile columns
        if (phase == 1) for (int k = 0; k < nT; ++k) if (col_owner[k] == rank) { dom_lo = std::min(dom_lo, k); dom_hi = k + 1; }
        if (dom_hi <= dom_lo) dom_lo = dom_hi = 0;
        L.jobs.clear(); L.dptr.assign(1, 0); L.dep.clear();
        L.own.assign((size_t)(nT + 1) * nT, -1); L.sumjob.assign(nT, -1);
        L.presum.assign(nT, -1); L.bits.assign(nT, 0); L.par.assign(nT, -1);
        const bool chain = use_chain && merge;
        auto inphase = [&](int c) { return c < nT && (phase == 0 || (phase == 1 ? col_owner[c] == rank : col_owner[c] < 0)); };
        std::vector<int> js;
        for (int k = 0; k < nT; ++k) {
            // emit the job(s) of tile (i,k) with the products js; returns the owner job
            auto emit = [&](int i, int mode, int par, int jhi_owner) {
                n_products += (long long)js.size();
                const int nj = (int)js.size();
                const int nch = nj > split_min ? (nj + chunk - 1) / chunk : 1;
                auto push_products = [&](int q0, int q1) {
                    for (int q = q0; q < q1; ++q) { L.dep.push_back(L.own[(size_t)k * nT + js[q]]); L.dep.push_back(i != k ? L.own[(size_t)i * nT + js[q]] : -1); }
                    L.dptr.push_back((int)L.dep.size());
                };
                for (int c = 0; c + 1 < nch; ++c) {             // helpers
                    L.jobs.push_back(DfJob{i, k, js[chunk * c], js[chunk * (c + 1)], nparts + c, -1, 0, -1});
                    push_products(chunk * c, chunk * (c + 1));
                }
                const int me = (int)L.jobs.size();
                L.jobs.push_back(DfJob{i, k, nch > 1 ? js[chunk * (nch - 1)] : (nj ? js[0] : jhi_owner), jhi_owner, nparts, nch - 1, mode, i == k ? par : -1});   // (no products: an empty range)
                push_products(chunk * (nch - 1), nj);
                nparts += nch - 1;
                return me;
            };
            if (phase == 1 && top(k)) {                         // this rank's share of the top tiles of column k
                for (int i = k; i <= nT; ++i) {
                    if (!has(i, k)) continue;
                    js.clear();
                    for (int j = dom_lo; j < std::min(dom_hi, k); ++j) if (has(i, j) && has(k, j)) js.push_back(j);
                    emit(i, rank == 0 ? 4 : 12, -1, std::min(dom_hi, k));
                }
                continue;
            }
            if (phase == 1 ? !mine(k) : (phase == 2 && !top(k))) continue;
            auto products = [&](int i) {
                js.clear();
                for (int j = 0; j < k; ++j) if (has(i, j) && has(k, j) && (phase != 2 || top(j))) js.push_back(j);
            };
            int par = -1;
            if (merge) for (int i = k + 1; i <= nT && par < 0; ++i) if (has(i, k)) par = i;
            // the sum-only task first: the diagonal task depends on it (job numbers stay topological)
            if (par >= 0) { products(par); L.sumjob[k] = emit(par, 1, -1, k); }
            products(k);
            L.par[k] = par;
            if (chain) {
                // the chain role factors the tile: its sum is a task of its own (mode 16) -- without the product with
                // column k-1 where this column continues the chain of k-1 (the link then stays in the workgroup's LDS)
                const bool cont = k > 0 && (L.bits[k - 1] & 2) != 0;
                if (cont) js.pop_back();                        // (has(k, k-1): the link itself)
                L.presum[k] = emit(k, 16, -1, cont ? k - 1 : k);
                js.clear();
                L.bits[k] = (cont ? 1 : 0) | ((par == k + 1 && k + 1 < nT && inphase(k + 1)) ? 2 : 0);
            }
            L.own[(size_t)k * nT + k] = emit(k, par >= 0 ? 2 : 0, par, k);
            if (par >= 0) L.own[(size_t)par * nT + k] = L.own[(size_t)k * nT + k];      // L(par,k) is published by the diagonal task
            for (int i = k + 1; i <= nT; ++i)
                if (has(i, k) && i != par) { products(i); L.own[(size_t)i * nT + k] = emit(i, 0, -1, k); }
        }
    }
    // Task order = the order in which workgroups take them.  Any topological order is
    // deadlock free (a dependency always has a smaller number, so it is owned by a running
    // workgroup); column-major order, however, hands out ALL tiles of the first blocks of
    // the dissection before the first tile of the others, and the few hundred resident
    // workgroups then sit in the dependent chains of a few leaves while the other leaves
    // have not started.  Candidate orders are tried on a model of the kernel (measured
    // costs, us: bench/chol_trace.py, bench/chol_path.py) and the best one is taken.
    void order_jobs(JobList &L, const char *what) {
        std::vector<DfJob> &jobs = L.jobs;
        const std::vector<int> &dptr = L.dptr, &dep = L.dep, &own = L.own, &sumjob = L.sumjob;
        const int nt = (int)jobs.size();
        if (nt == 0) return;
        auto is_helper = [&](int t) { return jobs[t].np < 0; };
        auto is_share = [&](int t) { return (jobs[t].mode & 4) != 0; };
        auto is_diag = [&](int t) { return jobs[t].np >= 0 && jobs[t].i == jobs[t].k && !is_share(t) && jobs[t].mode != 16; };
        auto nprod_of = [&](int t) { return (dptr[t + 1] - dptr[t]) / 2; };
        // every dependency of job t: product tiles, its helpers (the np jobs right before it), the diagonal tile
        auto for_deps = [&](int t, const std::function<void(int)> &f) {
            for (int q = dptr[t]; q < dptr[t + 1]; ++q) if (dep[q] >= 0) f(dep[q]);
            if (!is_helper(t)) {
                for (int h = 1; h <= jobs[t].np; ++h) f(t - h);
                if (jobs[t].mode == 0 && jobs[t].i != jobs[t].k) f(own[(size_t)jobs[t].k * nT + jobs[t].k]);
                if (jobs[t].mode == 2) f(sumjob[jobs[t].k]);
                if (is_diag(t) && L.presum[jobs[t].k] >= 0) {   // chain role: the tile's sum, and the column it continues
                    f(L.presum[jobs[t].k]);
                    if (L.bits[jobs[t].k] & 1) f(own[(size_t)(jobs[t].k - 1) * nT + jobs[t].k - 1]);
                }
            }
        };
        const double c_prod = 2.5, c_diag = 13.5, c_off = 4.5, c_hop = 2.0, c_add = 1.5, c_help = 2.0;
        auto tail_of = [&](int t) { return is_helper(t) || is_share(t) || jobs[t].mode == 1 || jobs[t].mode == 16 ? c_help : (is_diag(t) ? c_diag + (jobs[t].mode == 2 ? 2.5 : 0.0) : c_off); };
        // (1) earliest start: every input the moment it exists, unlimited workgroups
        std::vector<double> est(nt, 0.0), fin(nt, 0.0);
        for (int t = 0; t < nt; ++t) {
            double e = 0.0;
            for_deps(t, [&](int d) { e = std::max(e, fin[d]); });
            est[t] = e;
            fin[t] = e + c_hop + (nprod_of(t) ? c_prod : 0.0) + tail_of(t);
        }
        // (2) bottom level: the longest way from a task to the end of the factorisation
        std::vector<double> blev(nt, 0.0);
        for (int t = nt - 1; t >= 0; --t) {
            blev[t] += c_hop + c_prod + tail_of(t);
            for_deps(t, [&](int d) { blev[d] = std::max(blev[d], blev[t]); });
        }
        // The kernel as a list schedule: `workers` resident workgroups take the tasks in order; a
        // workgroup consumes its inputs in its fixed order, every product after both tiles exist.
        const int workers = std::max(1, std::min(grid, 256));      // 400 registers per lane: one workgroup per CU
        auto simulate = [&](const std::vector<int> &ord) {
            std::vector<double> done(nt, 0.0);
            std::priority_queue<double, std::vector<double>, std::greater<double>> freeat;
            for (int w = 0; w < workers; ++w) freeat.push(0.0);
            double last = 0.0;
            for (int t : ord) {
                double tc = freeat.top(); freeat.pop();
                for (int q = dptr[t]; q + 1 < dptr[t + 1]; q += 2) {
                    double av = done[dep[q]];
                    if (dep[q + 1] >= 0) av = std::max(av, done[dep[q + 1]]);
                    tc = std::max(tc, av + c_hop) + c_prod;
                }
                if (!is_helper(t)) {
                    for (int h = jobs[t].np; h >= 1; --h) tc = std::max(tc, done[t - h] + c_hop) + c_add;
                    if (jobs[t].mode == 0 && jobs[t].i != jobs[t].k) tc = std::max(tc, done[own[(size_t)jobs[t].k * nT + jobs[t].k]] + c_hop);
                    if (jobs[t].mode == 2) tc = std::max(tc, done[sumjob[jobs[t].k]] + c_hop - c_diag);    // T is only wanted after the factorisation
                    if (is_diag(t) && L.presum[jobs[t].k] >= 0) {
                        tc = std::max(tc, done[L.presum[jobs[t].k]] + c_hop);
                        if (L.bits[jobs[t].k] & 1) tc = std::max(tc, done[own[(size_t)(jobs[t].k - 1) * nT + jobs[t].k - 1]]);
                    }
                }
                tc += tail_of(t);
                done[t] = tc; last = std::max(last, tc);
                freeat.push(tc);
            }
            return last;
        };
        // Candidates, all topological (Kahn's algorithm: a task is eligible once its dependencies are
        // placed; among the eligible ones the smallest key goes first).  Keys: earliest start minus
        // beta x bottom level -- beta = 0: in the order the inputs appear; large beta: critical path
        // first -- and earliest start minus the task's own serial work (long sums get their
        // workgroup early).  The model ranks them; the best simulated one is taken.
        std::vector<int> indeg0(nt, 0), sptr(nt + 1, 0), succ;
        for (int t = 0; t < nt; ++t) for_deps(t, [&](int d) { ++indeg0[t]; ++sptr[d + 1]; });
        for (int t = 0; t < nt; ++t) sptr[t + 1] += sptr[t];
        succ.resize(sptr[nt]);
        {
            std::vector<int> fill(sptr.begin(), sptr.end() - 1);
            for (int t = 0; t < nt; ++t) for_deps(t, [&](int d) { succ[fill[d]++] = t; });
        }
        auto kahn = [&](const std::vector<double> &key) {
            std::vector<int> ord(nt), indeg(indeg0);
            typedef std::pair<double, int> KI;
            std::priority_queue<KI, std::vector<KI>, std::greater<KI>> pq;
            for (int t = 0; t < nt; ++t) if (!indeg[t]) pq.push(KI(key[t], t));
            size_t n_out = 0;
            while (!pq.empty()) {
                const int t = pq.top().second; pq.pop();
                ord[n_out++] = t;
                for (int q = sptr[t]; q < sptr[t + 1]; ++q) if (--indeg[succ[q]] == 0) pq.push(KI(key[succ[q]], succ[q]));
            }
            return ord;
        };
        const double betas[] = {0.0, 0.25, 0.5, 1.0, 2.0, 4.0, 1e6};
        const int ncand = (int)(sizeof(betas) / sizeof(betas[0])) + 1;
        std::vector<int> best_ord;
        double tbest = 1e300;
        int best = -1;
        std::vector<double> key(nt), tcs(ncand);
        const int forced = env_int("DBAT_HIP_DF_ORDER", -1);
        for (int c = 0; c < ncand; ++c) {
            if (c + 1 < ncand) for (int t = 0; t < nt; ++t) key[t] = est[t] - betas[c] * blev[t];
            else for (int t = 0; t < nt; ++t) key[t] = est[t] - c_prod * nprod_of(t);
            std::vector<int> ord = kahn(key);
            tcs[c] = simulate(ord);
            if (forced == c || (forced < 0 && tcs[c] < tbest)) { tbest = tcs[c]; best = c; best_ord.swap(ord); }
        }
        double cp = 0.0;
        for (int t = 0; t < nt; ++t) cp = std::max(cp, fin[t]);
        sim_critical_us.push_back(cp); sim_schedule_us.push_back(tbest);
        if (env_on("DBAT_HIP_PLAN_STATS")) {
            fprintf(stderr, "[chol %s] %d tasks, critical path %.0f us; simulated with %d workgroups, key = earliest start - beta x bottom level:",
                    what, nt, cp, workers);
            for (int c = 0; c + 1 < ncand; ++c) fprintf(stderr, " beta %g: %.0f us,", betas[c], tcs[c]);
            fprintf(stderr, " earliest start - own work: %.0f us -> candidate %d\n", tcs[ncand - 1], best);
        }
        // a final task addresses its helpers by slot, not by position: any topological order will do
        std::vector<DfJob> sorted(nt);
        for (int t = 0; t < nt; ++t) sorted[t] = jobs[best_ord[t]];
        jobs.swap(sorted);
    }
    // common part: nz[i] = bitset of columns k <= i with a structurally non-zero tile (i, k), i = 0..nT
    // (row nT = right-hand side, all columns), fill included.  col_owner (empty: one rank): see build_jobs.
    bool finish_setup(const std::vector<uint64_t> &rowbits, const std::vector<int64_t> &toff,
                      const std::vector<int> &col_owner = std::vector<int>(), int rank = 0) {
        auto has = [&](int i, int k) { return (rowbits[(size_t)i * W + (k >> 6)] >> (k & 63)) & 1ull; };
        if (const char *g = env_get("DBAT_HIP_DF_GRID")) grid = std::min(std::max(atoi(g), 1), 4096);
        nparts = 0; n_products = 0;
        two_phase = !col_owner.empty();
        h_rowbits = rowbits;
        // The chain role pays where the dependent chain of the separators decides the time (C1 ... C3: 0.61 -> 0.59 ms at C3);
        // where the tile products do (C4: 116 000 of them, 3.07 -> 3.29 ms) its workgroups and the extra sum tasks cost more
        // than the shorter links return.  Both times from the pattern: products x 2.6 us over the workgroups against the
        // longest chain x 17 us (measured costs, MI355X).  DBAT_HIP_DF_CHAIN=0 / 1 forces it off / on.
        use_chain = permuted && env_int("DBAT_HIP_DF_CHAIN", 1) != 0;
        if (use_chain && !two_phase && !env_get("DBAT_HIP_DF_CHAIN")) {
            long long nprod = 0;
            int longest = 0, cur = 0;
            for (int k = 0; k < nT; ++k) {
                int par = -1;
                for (int i = k; i <= nT; ++i) {
                    if (!has(i, k)) continue;
                    if (i > k && par < 0) par = i;
                    for (int w = 0; w < W; ++w) {
                        uint64_t m = rowbits[(size_t)i * W + w] & rowbits[(size_t)k * W + w];
                        if (w == (k >> 6)) m &= (1ull << (k & 63)) - 1;
                        else if (w > (k >> 6)) m = 0;
                        nprod += __builtin_popcountll(m);
                    }
                }
                cur = (k > 0 && cur > 0) ? cur : 1;
                longest = std::max(longest, cur);
                cur = par == k + 1 ? cur + 1 : 0;
            }
            const double t_products = (double)nprod * 2.6 / std::min(grid, 256), t_chain = longest * 17.0;
            use_chain = t_products < 0.6 * t_chain;
            if (env_on("DBAT_HIP_PLAN_STATS"))
                fprintf(stderr, "[chol] %lld tile products (%.0f us over the workgroups), longest chain %d links (%.0f us): chain role %s\n",
                        nprod, t_products, longest, t_chain, use_chain ? "on" : "off");
        }
        // the diagonal tasks leave the ordered list for the chain role: their order is the order of the tickets
        auto split_chain = [&](JobList &L, std::vector<DfJob> &tasks, std::vector<int> &cols) {
            tasks.clear(); cols.clear();
            for (const DfJob &j : L.jobs) {
                if (use_chain && j.np >= 0 && j.i == j.k && (j.mode == 0 || j.mode == 2)) cols.push_back(j.k);
                else tasks.push_back(j);
            }
        };
        auto chain_width = [&](const std::vector<int> &cols, const std::vector<int> &bits) {
            int runs = 0;
            for (int k : cols) runs += !(bits[k] & 1);
            return std::max(1, std::min(runs, env_int("DBAT_HIP_DF_CHAIN_WG", 32)));
        };
        std::vector<int> colsA, colsB;
        sim_critical_us.clear(); sim_schedule_us.clear();
        JobList L;
        std::vector<int> bkl;                                   // panels of the backward substitution, in task order
        std::vector<uint64_t> rowbits_top;
        if (!two_phase) {
            build_jobs(0, rowbits, col_owner, rank, L);
            order_jobs(L, "all");
            split_chain(L, h_tasks, colsA); ntasks = (int)h_tasks.size();
            for (int j = nT - 1; j >= 0; --j) bkl.push_back(j);
        } else {
            build_jobs(1, rowbits, col_owner, rank, L);
            order_jobs(L, "domain");
            split_chain(L, h_tasks, colsA); ntasks = (int)h_tasks.size();
            JobList B;
            build_jobs(2, rowbits, col_owner, rank, B);
            order_jobs(B, "top");
            split_chain(B, h_tasksB, colsB); ntasksB = (int)h_tasksB.size();
            if (!up(d_tasksB, h_tasksB)) return false;
            if (use_chain) {
                for (int k = 0; k < nT; ++k) if (B.par[k] >= 0) L.par[k] = B.par[k];      // (one table: a column is in one of the two lists)
                nchainB = (int)colsB.size(); chain_wgB = chain_width(colsB, B.bits);
                if (!up(d_chain_colsB, colsB) || !up(d_chain_bitsB, B.bits)) return false;
            }
            // the top columns first (every rank), then this rank's domain: a dependency always has a smaller number
            for (int j = nT - 1; j >= 0; --j) if (col_owner[j] < 0) bkl.push_back(j);
            for (int j = nT - 1; j >= 0; --j) if (col_owner[j] == rank) bkl.push_back(j);
            rowbits_top = rowbits;                              // the products of the second launch: top columns only
            for (int i = 0; i <= nT; ++i)
                for (int k = 0; k < nT; ++k)
                    if (col_owner[k] >= 0) rowbits_top[(size_t)i * W + (k >> 6)] &= ~(1ull << (k & 63));
            if (!up(d_rowbits_top, rowbits_top)) return false;
        }
        nbk = (int)bkl.size();
        if (use_chain) {
            nchain = (int)colsA.size(); chain_wg = chain_width(colsA, L.bits);
            std::vector<int> pos(nT, 0);                          // (one table: a column is in one of the two lists)
            for (size_t q = 0; q < colsA.size(); ++q) pos[colsA[q]] = (int)q;
            for (size_t q = 0; q < colsB.size(); ++q) pos[colsB[q]] = (int)q;
            if (!up(d_chain_cols, colsA) || !up(d_chain_bits, L.bits) || !up(d_chain_par, L.par) || !up(d_chain_pos, pos)) return false;
            if (env_on("DBAT_HIP_PLAN_STATS")) {
                int runs = 0, longest = 0, cur = 0;
                for (int k = 0; k < nT; ++k) { if (L.bits[k] & 1) ++cur; else { cur = 1; } longest = std::max(longest, cur); }
                for (int k : colsA) runs += !(L.bits[k] & 1);
                fprintf(stderr, "[chol] chain role: %d diagonal tiles in %d chains (longest %d), %d workgroups%s\n", nchain, runs, longest, chain_wg,
                        two_phase ? " (domain)" : "");
            }
        }
        if (env_on("DBAT_HIP_DF_TRACE") && !two_phase && hipMalloc(&d_trace, (size_t)(ntasks + 3 * nT) * 16 * sizeof(long long)) != hipSuccess) d_trace = nullptr;
        std::vector<int> bptr(nT + 1, 0), bidx;
        for (int j = 0; j < nT; ++j) {
            for (int i = nT - 1; i > j; --i)
                if (has(i, j)) bidx.push_back(i);
            bptr[j + 1] = (int)bidx.size();
        }
        if (!up(d_tasks, h_tasks) || !up(d_bk_ptr, bptr) || !up(d_bk_idx, bidx) || !up(d_rowbits, rowbits) || !up(d_toff, toff) || !up(d_bk_list, bkl))
            return false;
        if (hipMalloc(&d_flags, ((size_t)(nT + 1) * nT + (size_t)nparts + 4 * (size_t)nT) * sizeof(int)) != hipSuccess) return false;   // tiles, helper slots, T of the sum-only tasks, T' of the diagonal tiles, claims
        if (nparts > 0 && hipMalloc(&d_parts, (size_t)nparts * 4096 * sizeof(double)) != hipSuccess) return false;
        if (hipMalloc(&d_ctl, 8 * sizeof(int)) != hipSuccess) return false;
        (void)hipMemset(d_flags, 0, ((size_t)(nT + 1) * nT + (size_t)nparts + 4 * (size_t)nT) * sizeof(int));
        (void)hipMemset(d_ctl, 0, 8 * sizeof(int));
        env_l2 = env_int("DBAT_HIP_DF_L2", 0) != 0;
        env_df_ablate = 2 * (env_int("DBAT_HIP_DF_ABLATE", 0) & 3);     // measurement build: 1 = L(i,j) not fetched, 2 = neither tile
        epoch = 0;
        return true;
    }
    bool setup_inplace(const CholEnvelope &env, int64_t lda) {
        release();
        permuted = false; perm.clear();
        n = n_nat = env.n; nT = (n + CHOL_NB - 1) / CHOL_NB; W = (nT + 64) / 64;
        std::vector<uint64_t> rb((size_t)(nT + 1) * W, 0);
        std::vector<int64_t> toff((size_t)(nT + 1) * nT, -1);
        for (int i = 0; i <= nT; ++i) {
            const int kf = i < nT ? env.panel_first[i] / CHOL_NB : 0;
            for (int k = kf; k <= std::min(i, nT - 1); ++k) {
                rb[(size_t)i * W + (k >> 6)] |= 1ull << (k & 63);
                toff[(size_t)i * nT + k] = (int64_t)CHOL_NB * k * lda + (i == nT ? (int64_t)n : (int64_t)CHOL_NB * i);
            }
        }
        return finish_setup(rb, toff);
    }
    // adj: symmetric co-visibility bitsets of the nc cameras (adj_words 64-bit words per camera);
    // nd: the nested-dissection order of the cameras (nd.hpp; with nd.nparts > 1 its blocks carry the rank
    // that owns them); unknowns: 6 per camera, then nio dense IO unknowns.
    bool setup_permuted(int nc, int nio, const uint64_t *adj, int adj_words, const NdTree &nd, int rank) {
        release();
        permuted = true;
        n_nat = 6 * nc + nio;
        const std::vector<int> &order = nd.order, &block_end = nd.block_end;
        // A block starts on a tile boundary (padding rows = identity) where the tile that straddled the boundary
        // would chain two INDEPENDENT blocks together, or two owners (several ranks).  A separator directly follows
        // the last block of the part it was cut from and depends on it anyway: its rows go on in that block's last
        // tile -- one link less in the dependent chain per level of the dissection (DBAT_HIP_ND_PAD_ALL=1: as before).
        std::vector<int> rowpos(nc);                    // first row of every camera in the factorised order
        int off = 0;
        {
            const bool pad_all = env_on("DBAT_HIP_ND_PAD_ALL");
            size_t b = 0; int q0 = 0;
            for (int q = 0; q < nc; ++q) {
                if (q == q0) {
                    // ... but only where that saves a tile: the separator's rows and the ragged tail before them fit
                    // into fewer tiles together than apart (otherwise the separator merely loses its alignment)
                    const int sep_rows = 6 * (block_end[b] - q0), used = off % CHOL_NB;
                    const bool saves = used > 0 && (used + sep_rows + CHOL_NB - 1) / CHOL_NB < 1 + (sep_rows + CHOL_NB - 1) / CHOL_NB;
                    // ... or the separator is short (two tiles at most): its alignment is worth less than the rows the
                    // padding would add to the chain (measured: C1 0.196 -> 0.176 ms, C2 0.692 -> 0.681; longer ones: C2 loses)
                    const int join_small = env_int("DBAT_HIP_ND_JOIN_SMALL", 2 * CHOL_NB);
                    const bool joins = !pad_all && (saves || (used > 0 && sep_rows <= join_small)) && b > 0 && b < nd.block_sep.size() && nd.block_sep[b] &&
                                       nd.block_owner[b] == nd.block_owner[b - 1];
                    if (!joins) off = (off + CHOL_NB - 1) / CHOL_NB * CHOL_NB;
                }
                rowpos[order[q]] = off; off += 6;
                while (b < block_end.size() && block_end[b] <= q + 1) { q0 = q + 1; ++b; }
            }
        }
        const int io0 = nio > 0 ? (off + CHOL_NB - 1) / CHOL_NB * CHOL_NB : off;
        n = (io0 + nio + CHOL_NB - 1) / CHOL_NB * CHOL_NB;
        nT = n / CHOL_NB; W = (nT + 64) / 64;
        perm.assign(n_nat, 0);
        std::vector<int> iperm((size_t)n, -1);
        for (int c = 0; c < nc; ++c)
            for (int a = 0; a < 6; ++a) perm[6 * c + a] = rowpos[c] + a;
        for (int u = 0; u < nio; ++u) perm[6 * nc + u] = io0 + u;
        for (int z = 0; z < n_nat; ++z) iperm[perm[z]] = z;
        auto pos6 = [&](int c) { return rowpos[c]; };
        // ---- tile pattern of P S P' (lower), IO rows and the right-hand side dense, then symbolic fill
        std::vector<uint64_t> rb((size_t)(nT + 1) * W, 0);
        auto setbit = [&](int i, int k) { rb[(size_t)i * W + (k >> 6)] |= 1ull << (k & 63); };
        for (int a = 0; a < nc; ++a) {
            const int ta0 = pos6(a) / CHOL_NB, ta1 = (pos6(a) + 5) / CHOL_NB;
            for (int w = 0; w < adj_words; ++w) {
                uint64_t m = adj[(size_t)a * adj_words + w];
                while (m) {
                    const int b = 64 * w + __builtin_ctzll(m);
                    m &= m - 1;
                    if (b >= nc) break;
                    const int tb0 = pos6(b) / CHOL_NB, tb1 = (pos6(b) + 5) / CHOL_NB;
                    for (int ti = ta0; ti <= ta1; ++ti)
                        for (int tj = tb0; tj <= tb1; ++tj) setbit(std::max(ti, tj), std::min(ti, tj));
                }
            }
            for (int ti = ta0; ti <= ta1; ++ti) for (int tj = ta0; tj <= ti; ++tj) setbit(ti, tj);
        }
        for (int i = 0; i < nT; ++i) setbit(i, i);
        for (int i = io0 / CHOL_NB; i < nT && nio > 0; ++i) for (int k = 0; k <= i; ++k) setbit(i, k);
        for (int k = 0; k < nT; ++k) setbit(nT, k);
        {   // fill: eliminating column k couples all rows below it
            std::vector<int> rows;
            for (int k = 0; k < nT; ++k) {
                rows.clear();
                for (int i = k + 1; i <= nT; ++i) if ((rb[(size_t)i * W + (k >> 6)] >> (k & 63)) & 1ull) rows.push_back(i);
                for (size_t a = 0; a < rows.size(); ++a)
                    for (size_t b = 0; b <= a; ++b)
                        if (rows[b] < nT) setbit(rows[a], rows[b]);
            }
        }
        // ---- several ranks: the rank that owns every tile column (a block starts on a tile boundary, so a tile
        // column lies in one block); -1: top separator, IO unknowns
        std::vector<int> col_owner;
        if (nd.nparts > 1) {
            col_owner.assign(nT, -1);
            size_t b = 0;
            for (int q = 0; q < nc; ++q) {
                while (b < block_end.size() && block_end[b] <= q) ++b;
                const int ow = b < nd.block_owner.size() ? nd.block_owner[b] : -1;
                for (int t = rowpos[order[q]] / CHOL_NB; t <= (rowpos[order[q]] + 5) / CHOL_NB; ++t) col_owner[t] = ow;
            }
        }
        // ---- compact tile storage, column-major over the pattern; the tiles of the top columns last and
        // contiguous (what the ranks sum between the two launches)
        std::vector<int64_t> toff((size_t)(nT + 1) * nT, -1);
        std::vector<DfTask> tile_ij;
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1) top_tile0 = (int64_t)tile_ij.size();
            for (int k = 0; k < nT; ++k) {
                if ((!col_owner.empty() && col_owner[k] < 0) != (pass == 1)) continue;
                for (int i = k; i <= nT; ++i)
                    if ((rb[(size_t)i * W + (k >> 6)] >> (k & 63)) & 1ull) {
                        toff[(size_t)i * nT + k] = (int64_t)tile_ij.size() * 4096;
                        tile_ij.push_back(DfTask{i, k});
                    }
            }
        }
        ntiles = (int)tile_ij.size();
        n_top_tiles = ntiles - top_tile0;
        if (!up(d_tile_ij, tile_ij) || !up(d_iperm, iperm)) return false;
        if (hipMalloc(&d_tiles, (size_t)ntiles * 4096 * sizeof(double)) != hipSuccess) return false;
        (void)hipMemset(d_tiles, 0, (size_t)ntiles * 4096 * sizeof(double));     // right-hand-side tiles: only their first row is ever written
        if (hipMalloc(&d_qperm, (size_t)nT * CHOL_NB * sizeof(double)) != hipSuccess) return false;
        if (env_on("DBAT_HIP_PLAN_STATS"))
            fprintf(stderr, "[chol] order %d (%d with block padding, %zu blocks), %d tile rows, %d tiles (%.1f MB), dense lower triangle would be %d tiles\n",
                    n_nat, n, block_end.size(), nT, ntiles, ntiles * 32768.0 / 1e6, nT * (nT + 1) / 2 + nT);
        if (env_on("DBAT_HIP_PLAN_STATS")) {
            fprintf(stderr, "[chol] cameras per block, last (root separator) first:");
            for (size_t b = block_end.size(); b-- > 0 && block_end.size() - b <= 24;)
                fprintf(stderr, " %d", block_end[b] - (b ? block_end[b - 1] : 0));
            fprintf(stderr, "\n");
        }
        if (env_on("DBAT_HIP_PLAN_STATS") && !col_owner.empty()) {
            int ntop = 0, nmine = 0;
            for (int k = 0; k < nT; ++k) { ntop += col_owner[k] < 0; nmine += col_owner[k] == rank; }
            fprintf(stderr, "[chol] rank %d of %d: %d tile columns in its domain, %d top columns, %lld top tiles (%.1f MB summed over the ranks per factorisation)\n",
                    rank, nd.nparts, nmine, ntop, (long long)n_top_tiles, n_top_tiles * 32768.0 / 1e6);
        }
        return finish_setup(rb, toff, col_owner, rank);
    }
    // Z = inv(P S P') on the pattern of the factor, after solve() in the compact (permuted) layout on the same stream.
    // Overwrites the factor's off-diagonal tiles with M = L L_kk^-1.  One rank only.  false: out of device memory.
    bool selinv_supported() const { return permuted && !two_phase; }
    bool selected_inverse(hipStream_t stream, const double *linv_work) {
        if (!d_ztiles) {                                // the schedule, once
            auto has = [&](int i, int k) { return (h_rowbits[(size_t)i * W + (k >> 6)] >> (k & 63)) & 1ull; };
            std::vector<SiTask> mt, off;
            std::vector<int> level(nT, 0), diag;
            int nlev = 0;
            for (int k = nT - 1; k >= 0; --k) {
                int lv = 0;
                for (int i = k + 1; i < nT; ++i) if (has(i, k)) { lv = std::max(lv, level[i] + 1); mt.push_back(SiTask{i, k}); }
                level[k] = lv; nlev = std::max(nlev, lv + 1);
            }
            si_off_ptr.assign(nlev + 1, 0); si_diag_ptr.assign(nlev + 1, 0);
            for (int lv = 0; lv < nlev; ++lv) {
                for (int k = nT - 1; k >= 0; --k) {
                    if (level[k] != lv) continue;
                    diag.push_back(k);
                    for (int i = k + 1; i < nT; ++i) if (has(i, k)) off.push_back(SiTask{i, k});
                }
                si_off_ptr[lv + 1] = (int)off.size(); si_diag_ptr[lv + 1] = (int)diag.size();
            }
            n_si_m = (int)mt.size();
            if (!up(d_si_m, mt) || !up(d_si_off, off) || !up(d_si_diag, diag) || !up(d_perm, perm)) return false;
            if (hipMalloc(&d_ztiles, (size_t)ntiles * 4096 * sizeof(double)) != hipSuccess) { d_ztiles = nullptr; return false; }
        }
        if (n_si_m > 0)
            hipLaunchKernelGGL(k_selinv_m, dim3(n_si_m), dim3(256), 0, stream, d_tiles, d_toff, nT, linv_work, d_si_m);
        for (size_t lv = 0; lv + 1 < si_off_ptr.size(); ++lv) {
            const int no = si_off_ptr[lv + 1] - si_off_ptr[lv], nd = si_diag_ptr[lv + 1] - si_diag_ptr[lv];
            if (no > 0)
                hipLaunchKernelGGL(k_selinv_off, dim3(no), dim3(256), 0, stream, d_ztiles, (const double *)d_tiles, d_toff, nT, d_bk_ptr,
                                   d_bk_idx, d_si_off + si_off_ptr[lv]);
            if (nd > 0)
                hipLaunchKernelGGL(k_selinv_diag, dim3(nd), dim3(256), 0, stream, d_ztiles, (const double *)d_tiles, d_toff, nT, d_bk_ptr,
                                   d_bk_idx, linv_work, d_si_diag + si_diag_ptr[lv]);
        }
        return true;
    }
    int selinv_levels() const { return (int)si_off_ptr.size() - 1; }
    // Factor and solve S q = b (b' in row n of S, lower triangle of S, leading dimension lda).
    // q -> q_out (natural order).  In-place mode leaves L in S; permuted mode leaves S untouched.
    // ldiag (optional, n entries): the pivots diag(L) by natural index.  linv_work as
    // BlockChol::linv_doubles.  info_dev: > 0 first non-positive pivot (index in the factorised
    // order), -1 dataflow abort (spin cap).
    void solve(hipStream_t stream, double *A, int64_t lda, double *q_out, double *linv_work, int *info_dev,
               double *ldiag = nullptr, const double *qscale = nullptr, double *dz_out = nullptr) {
        if (d_trace) (void)hipMemsetAsync(d_trace, 0, (size_t)(ntasks + 3 * nT) * 16 * sizeof(long long), stream);
        ++epoch;
        DfView V;
        V.toff = d_toff; V.rowbits = d_rowbits; V.W = W; V.no_l2 = (env_l2 ? 0 : 1) | env_df_ablate;     // measured: no gain from the L2 path
        double *qflag;
        V.S = nullptr; V.ldS = lda; V.n_nat = n_nat;
        if (permuted) {
            // every task fetches its own tile of P S P' from S (the rows of the right-hand-side tiles below
            // the first stay zero from the set-up): a small reset instead of the gather launch
            if (!reset_done)
                hipLaunchKernelGGL(k_df_reset, dim3((nT * CHOL_NB + 255) / 256), dim3(256), 0, stream, info_dev, d_ctl,
                                   reinterpret_cast<unsigned long long *>(d_qperm), nT * CHOL_NB);
            reset_done = false;
            V.base = d_tiles; V.ld = 64; V.iperm = d_iperm; qflag = d_qperm; V.S = A;
        } else {
            (void)hipMemsetAsync(info_dev, 0, sizeof(int), stream);
            (void)hipMemsetAsync(d_ctl, 0, 8 * sizeof(int), stream);
            V.base = A; V.ld = lda; V.iperm = nullptr; qflag = q_out;
            (void)hipMemsetAsync(q_out, 0xFF, (size_t)n * sizeof(double), stream);
        }
        if (permuted && use_chain)
            hipLaunchKernelGGL(k_chol_df<true>, dim3(chain_grid(ntasks + nT, chain_wg)), dim3(256), 0, stream, V, n, nT, d_tasks, ntasks,
                               d_flags, d_ctl, epoch, linv_work, info_dev, d_trace, d_bk_ptr, d_bk_idx, qflag, q_out, ldiag,
                               qscale, dz_out, d_parts, nparts, d_bk_list, nbk, 0, chain_arg(0));
        else
            hipLaunchKernelGGL(k_chol_df<false>, dim3(std::min(grid, ntasks + nT)), dim3(256), 0, stream, V, n, nT, d_tasks, ntasks,
                               d_flags, d_ctl, epoch, linv_work, info_dev, d_trace, d_bk_ptr, d_bk_idx, qflag, q_out, ldiag,
                               qscale, dz_out, d_parts, nparts, d_bk_list, nbk, 0, DfChain{});
    }
    // chain role: the argument of a launch (which = 0: the whole / the domain's list, 1: the top separators), and its grid --
    // at least one worker beside the chain workgroups
    DfChain chain_arg(int which) const {
        DfChain C;
        int *dfl = d_flags + (size_t)(nT + 1) * nT + nparts + nT;
        C.cols = which ? d_chain_colsB : d_chain_cols; C.ncols = which ? nchainB : nchain;
        C.par = d_chain_par; C.bits = which ? d_chain_bitsB : d_chain_bits;
        C.dflags = dfl; C.claim = dfl + nT; C.giveup = dfl + 2 * nT; C.pos = d_chain_pos;
        C.nwg = which ? chain_wgB : chain_wg;
        C.ctr_slot = which ? 5 : 3; C.role_slot = which ? 6 : 4;
        C.trace_potf2 = env_int("DBAT_HIP_DF_TRACE_POTF2", 0);
        return C;
    }
    int chain_grid(int work, int wg) const { return std::max(std::min(grid, work + wg), wg + 1); }
    // ---- several ranks (two_phase), permuted layout only.  solve_domain: this rank factors the columns of its
    // own domain from ITS S (complete there: nd.hpp) and leaves its share of every top tile -- A_r minus the
    // domain's updates -- in the tile; the caller sums the top tiles over the ranks (top_tiles(), one
    // all-reduce); solve_top: every rank factors the top separators from the sums and runs the backward
    // substitution for the top and for its own domain.  The steps of the other domains' cameras are not
    // computed here (their dz entries are left alone): no observation of this rank refers to them.
    double *top_tiles() const { return d_tiles + top_tile0 * 4096; }
    int64_t top_tiles_count() const { return n_top_tiles * 4096; }
    void solve_domain(hipStream_t stream, double *A, int64_t lda, double *linv_work, int *info_dev, double *ldiag) {
        ++epoch;
        DfView V;
        V.toff = d_toff; V.rowbits = d_rowbits; V.W = W; V.no_l2 = (env_l2 ? 0 : 1) | env_df_ablate;
        V.ldS = lda; V.n_nat = n_nat; V.base = d_tiles; V.ld = 64; V.iperm = d_iperm; V.S = A;
        hipLaunchKernelGGL(k_df_reset, dim3((nT * CHOL_NB + 255) / 256), dim3(256), 0, stream, info_dev, d_ctl,
                           reinterpret_cast<unsigned long long *>(d_qperm), nT * CHOL_NB);
        if (ntasks > 0 && use_chain)
            hipLaunchKernelGGL(k_chol_df<true>, dim3(chain_grid(ntasks, chain_wg)), dim3(256), 0, stream, V, n, nT, d_tasks, ntasks,
                               d_flags, d_ctl, epoch, linv_work, info_dev, (long long *)nullptr, d_bk_ptr, d_bk_idx, (double *)nullptr,
                               (double *)nullptr, ldiag, (const double *)nullptr, (double *)nullptr, d_parts, nparts, d_bk_list, 0, 0, chain_arg(0));
        else if (ntasks > 0)
            hipLaunchKernelGGL(k_chol_df<false>, dim3(std::min(grid, ntasks)), dim3(256), 0, stream, V, n, nT, d_tasks, ntasks,
                               d_flags, d_ctl, epoch, linv_work, info_dev, (long long *)nullptr, d_bk_ptr, d_bk_idx, (double *)nullptr,
                               (double *)nullptr, ldiag, (const double *)nullptr, (double *)nullptr, d_parts, nparts, d_bk_list, 0, 0, DfChain{});
    }
    void solve_top(hipStream_t stream, int64_t lda, double *q_out, double *linv_work, int *info_dev, double *ldiag,
                   const double *qscale, double *dz_out) {
        DfView V;
        V.toff = d_toff; V.rowbits = d_rowbits_top; V.W = W; V.no_l2 = 1;
        V.ldS = lda; V.n_nat = n_nat; V.base = d_tiles; V.ld = 64; V.iperm = d_iperm; V.S = nullptr;   // the tiles hold the summed shares
        if (use_chain)
            hipLaunchKernelGGL(k_chol_df<true>, dim3(chain_grid(ntasksB + nbk, chain_wgB)), dim3(256), 0, stream, V, n, nT, d_tasksB, ntasksB,
                               d_flags, d_ctl, epoch, linv_work, info_dev, (long long *)nullptr, d_bk_ptr, d_bk_idx, d_qperm, q_out, ldiag,
                               qscale, dz_out, d_parts, nparts, d_bk_list, nbk, 2, chain_arg(1));
        else
            hipLaunchKernelGGL(k_chol_df<false>, dim3(std::min(grid, ntasksB + nbk)), dim3(256), 0, stream, V, n, nT, d_tasksB, ntasksB,
                               d_flags, d_ctl, epoch, linv_work, info_dev, (long long *)nullptr, d_bk_ptr, d_bk_idx, d_qperm, q_out, ldiag,
                               qscale, dz_out, d_parts, nparts, d_bk_list, nbk, 2, DfChain{});
    }
};

}  // namespace dbat
