// Dataflow Cholesky of the reduced camera system (K6): ONE persistent kernel.
//
// The blocked factorisation of chol.hpp is a chain of ~n/64 dependent panels
// with three or four kernel launches each; for the narrow envelopes of bundle
// adjustment (camera co-visibility band of ~13 tiles) every one of those
// kernels is latency-, not throughput-bound.  Here the same arithmetic runs as
// a task graph inside one launch:
//
//   task (i,k), i >= k, one 64 x 64 tile of the envelope, left-looking:
//       T  = A(i,k) - sum_{j=jlo}^{k-1} L(i,j) L(k,j)'      f64 MFMA, accumulators in registers,
//                                                           each term as soon as its two tiles exist
//       i == k :  L(k,k) = chol(T), Linv_k = L(k,k)^-1      (df_potf2: 16-column panels in registers,
//                                                           trailing updates on the matrix cores)
//       i  > k :  L(i,k) = T Linv_k'                        f64 MFMA
//   jlo = max(kfirst[i], kfirst[k]) -- the envelope is closed under fill.
//
// Workgroups take tasks from an atomic counter in column-major order, so every
// dependency of a task has a smaller number and is already owned by a running
// workgroup: no deadlock for any grid size.  Completion is published per tile
// by storing the solve's epoch into the tile's flag after the tile's own
// stores have completed; tiles and flags move with agent-scope (sc1) accesses
// because the L2 of the eight XCDs are not coherent among themselves.  A spin cap turns any
// scheduling accident into an error code instead of a hung GPU.
//
// The right-hand side is tile row nT (one valid row, row n of the array): the
// forward substitution rides along exactly as in chol.hpp.
#pragma once
#include "chol.hpp"

namespace dbat {

constexpr int DF_LD = 65;
constexpr int DF_SPIN_CAP = 1 << 22;

struct DfTask { int i, k; };

// Tiles that one workgroup writes and others read inside the same launch move
// with agent-scope relaxed atomics (sc1 loads/stores that bypass the per-XCD
// L2): no cache-wide write-back/invalidate is needed around the flags.
__device__ __forceinline__ double ld_coh(const double *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_coh(double *p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

__device__ __forceinline__ void df_potf2(double *Tm, int nb, int j0, int *info, long long *tr) {
    constexpr int LD = DF_TLD;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    // identity rows
    for (int idx = t; idx < 64 * 64; idx += 256) { const int c = idx >> 6, r = idx & 63; Tm[c * LD + 64 + r] = r == c ? 1.0 : 0.0; }
    __syncthreads();
#pragma unroll 1
    for (int p = 0; p < 4; ++p) {
        // ---- A: panel columns [16p, 16p+16)
        {
            const int nlowT = 48 - 16 * p;                  // T rows below the diagonal block
            int row;
            if (lane < 16) row = 16 * p + lane;
            else { const int s = 16 * w + (lane - 16); row = s < nlowT ? 16 * (p + 1) + s : 64 + (s - nlowT); }
            const bool act = lane < 32;
            double a[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) a[q] = act ? Tm[(16 * p + q) * LD + row] : 0.0;
            int bad = 0;                                    // first non-positive pivot of the panel, 1-based
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const double piv = readlane_f64(a[j], j);
                bad = (bad == 0 && !(piv > 0.0) && 16 * p + j < nb) ? 16 * p + j + 1 : bad;
                double id = __builtin_amdgcn_rsq(piv);
                id = id * (1.5 - 0.5 * piv * id * id);
                id = id * (1.5 - 0.5 * piv * id * id);
                const double l = a[j] * id;
                a[j] = l;
#pragma unroll
                for (int k = j + 1; k < 16; ++k) a[k] -= l * readlane_f64(l, k);
            }
            if (lane >= 16 ? act : w == 0) {
#pragma unroll
                for (int q = 0; q < 16; ++q) Tm[(16 * p + q) * LD + row] = a[q];
            }
            if (bad != 0 && t == 0 && *info == 0) *info = j0 + bad;
        }
        __syncthreads();
        if (tr && t == 0) tr[6 + 2 * p] = wall_clock64();
        if (p == 3) break;
        // ---- B: columns cb > p:  Tm[16cb + cc][rowbase + rr] -= sum_m P(rowbase + rr, m) P(16cb + cc, m)
        {
            // tiles: for cb in p+1..3: T row blocks rb = cb..3 (rowbase 16rb), identity blocks ib = 0..p (rowbase 64+16ib)
            int cnt = 0;
#pragma unroll 1
            for (int cb = p + 1; cb < 4; ++cb) {
                const int nT_ = 4 - cb, nI = p + 1;
#pragma unroll 1
                for (int q = 0; q < nT_ + nI; ++q, ++cnt) {
                    if ((cnt & 3) != w) continue;
                    const int rowbase = q < nT_ ? 16 * (cb + q) : 64 + 16 * (q - nT_);
                    chol_d4 c4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) c4[e] = Tm[(16 * cb + (lane >> 4) + 4 * e) * LD + rowbase + (lane & 15)];
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const int m = 16 * p + 4 * s + (lane >> 4);
                        const double av = -Tm[m * LD + 16 * cb + (lane & 15)];      // A[cc][m] = -P(16cb + cc, m)
                        const double bv = Tm[m * LD + rowbase + (lane & 15)];       // B[m][rr] =  P(rowbase + rr, m)
                        c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c4, 0, 0, 0);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) Tm[(16 * cb + (lane >> 4) + 4 * e) * LD + rowbase + (lane & 15)] = c4[e];
                }
            }
        }
        __syncthreads();
        if (tr && t == 0) tr[7 + 2 * p] = wall_clock64();
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

// Backward substitution task of panel j:  q_j = Linv_j' (y_j - sum_{i>j} L(i,j)' q_i).
// Lane = row of the tile L(i,j), 16 of its columns per thread in registers; the
// products with q_i accumulate lane-wise over all tiles of the block column and
// are summed across lanes once at the end.  Tiles are fetched before their q_i
// is polled, so when the last q (panel j+1) arrives only 16 FMAs, the reduction
// and the 64 x 64 product with Linv_j' remain.  q entries double as their own
// flags (DF_SENTINEL until solved).
__device__ __forceinline__ bool df_backward(double *smem, const double *A, int64_t lda, int n, int nT, int j,
                                            const int *__restrict__ bk_ptr, const int *__restrict__ bk_idx,
                                            const int *flags, int epoch, int *abort_flag, const double *linv_all,
                                            double *q_out) {
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
#pragma unroll
    for (int q = 0; q < 16; ++q) lv[q] = ld_coh(linv_all + (size_t)j * NB * NB + tx * NB + ty + 4 * q);
    const double yv = tx < nc ? ld_coh(A + (col0 + tx) * lda + n) : 0.0;   // y_j(tx) (every wave)
    double acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.0;
    for (int e = bk_ptr[j]; e < bk_ptr[j + 1]; ++e) {
        const int i = bk_idx[e];
        const int nr = min(NB, n - NB * i);
        if (!df_spin_wave(flags + (int64_t)i * nT + j, epoch, abort_flag)) return false;
        double v[16];
        const double *Lt = A + col0 * lda + (int64_t)NB * i + tx;
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = (tx < nr && ty + 4 * q < nc) ? ld_coh(Lt + (int64_t)(ty + 4 * q) * lda) : 0.0;
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
    if (ty == 0 && tx < nc) st_coh(q_out + col0 + tx, (red[tx] + red[NB + tx]) + (red[2 * NB + tx] + red[3 * NB + tx]));
    __syncthreads();                                    // smem is reused by the next task
    return true;
}

__global__ __launch_bounds__(256) void k_chol_df(double *__restrict__ A, int64_t lda, int n, int nT,
                                                 const int *__restrict__ kfirst, const DfTask *__restrict__ tasks,
                                                 int ntasks, int *__restrict__ flags, int *__restrict__ ctl, int epoch,
                                                 double *__restrict__ linv_all, int *__restrict__ info,
                                                 long long *__restrict__ trace, const int *__restrict__ bk_ptr,
                                                 const int *__restrict__ bk_idx, double *__restrict__ q_out) {
    constexpr int NB = 64, LD = DF_LD;
    __shared__ double smem[2 * NB * LD];                // Pm | Qm, or the augmented block of df_potf2
    double *Pm = smem, *Qm = smem + NB * LD;
    __shared__ int s_task, s_ok;
    const int t = threadIdx.x, tx = t & 63, ty = t >> 6;
    int *counter = ctl, *abort_flag = ctl + 1;
    for (;;) {
        if (t == 0) s_task = atomicAdd(counter, 1);
        __syncthreads();
        const int task = s_task;
        if (task >= ntasks) {
            // backward substitution, panels last to first (q_out == nullptr: factor only)
            if (q_out == nullptr || task >= ntasks + nT) return;
            if (trace && t == 0) trace[task * 16 + 0] = wall_clock64();
            if (!df_backward(smem, A, lda, n, nT, nT - 1 - (task - ntasks), bk_ptr, bk_idx, flags, epoch, abort_flag,
                             linv_all, q_out)) {
                if (t == 0) *info = -1;
                return;
            }
            if (trace && t == 0) trace[task * 16 + 4] = wall_clock64();
            continue;
        }
        const int i = tasks[task].i, k = tasks[task].k;
        const int64_t row0 = i == nT ? (int64_t)n : (int64_t)NB * i;
        const int nr = i == nT ? 1 : min(NB, n - NB * i);
        const int64_t col0 = (int64_t)NB * k;
        const int nc = min(NB, n - NB * k);
        const int jlo = max(kfirst[i], kfirst[k]);
        if (trace && t == 0) trace[task * 16 + 0] = wall_clock64();
        // original tile values in the accumulator layout: (c = 16*ty + (tx>>4) + 4e, r = 16*rt + (tx&15))
        chol_d4 orig[4];
        {
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = 16 * ty + (tx >> 4) + 4 * e, r = 16 * rt + (tx & 15);
                    orig[rt][e] = (r < nr && c < nc) ? A[(col0 + c) * lda + row0 + r] : 0.0;
                }
        }
        chol_d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        bool alive = true;
        for (int j = jlo; j < k; ++j) {
            if (t == 0) {
                bool ok = df_spin(flags + (int64_t)k * nT + j, epoch, abort_flag);
                if (ok && i != k) ok = df_spin(flags + (int64_t)i * nT + j, epoch, abort_flag);
                s_ok = ok;
            }
            __syncthreads();                                // also: the previous MFMA pass has read Pm/Qm
            if (!s_ok) { alive = false; break; }
            {
                const double *Lk = A + (int64_t)NB * j * lda + col0 + tx;     // L(64k + tx, 64j + m)
                const double *Li = A + (int64_t)NB * j * lda + row0 + tx;     // L(row0 + tx, 64j + m)
                const bool okc = tx < nc, okr = tx < nr && i != k;
                double vk[16], vi[16];                      // all loads in flight before the first use
#pragma unroll
                for (int q = 0; q < 16; ++q) vk[q] = okc ? ld_coh(Lk + (int64_t)(ty + 4 * q) * lda) : 0.0;
#pragma unroll
                for (int q = 0; q < 16; ++q) vi[q] = okr ? ld_coh(Li + (int64_t)(ty + 4 * q) * lda) : 0.0;
#pragma unroll
                for (int q = 0; q < 16; ++q) { Pm[(ty + 4 * q) * LD + tx] = vk[q]; Qm[(ty + 4 * q) * LD + tx] = vi[q]; }
            }
            __syncthreads();
            mfma_tile64<LD>(Pm, i == k ? Pm : Qm, ty, tx, acc);
        }
        if (!alive) { if (t == 0) *info = -1; return; }
        __syncthreads();
        if (trace && t == 0) trace[task * 16 + 1] = wall_clock64();
        double *Linv = linv_all + (size_t)k * NB * NB;
        if (i == k) {
            // T(r, c) = orig - acc -> augmented block, ragged part = identity, upper triangle = 0
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = 16 * ty + (tx >> 4) + 4 * e, r = 16 * rt + (tx & 15);
                    const double v = orig[rt][e] - acc[rt][e];
                    smem[c * DF_TLD + r] = (r < nc && c < nc) ? (r >= c ? v : 0.0) : (r == c ? 1.0 : 0.0);
                }
            if (trace && t == 0) trace[task * 16 + 2] = wall_clock64();
            df_potf2(smem, nc, (int)col0, info, trace ? trace + task * 16 : nullptr);
            // L -> A (lower triangle), L^-1 -> Linv[c*64 + i] = Linv(i, c) = smem[i][64 + c]
            double *A0 = A + col0 * lda + col0;
#pragma unroll 4
            for (int c = ty; c < NB; c += 4) {
                if (tx < nc && c < nc && tx >= c) st_coh(A0 + (int64_t)c * lda + tx, smem[c * DF_TLD + tx]);
                st_coh(Linv + c * NB + tx, (tx < nc && c < nc) ? smem[tx * DF_TLD + 64 + c] : 0.0);
            }
        } else {
            // T(r, c) = orig - acc  ->  Qm[c][r]
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = 16 * ty + (tx >> 4) + 4 * e, r = 16 * rt + (tx & 15);
                    Qm[c * LD + r] = orig[rt][e] - acc[rt][e];
                }
            if (t == 0) s_ok = df_spin(flags + (int64_t)k * nT + k, epoch, abort_flag);
            __syncthreads();
            if (!s_ok) { if (t == 0) *info = -1; return; }
            if (trace && t == 0) trace[task * 16 + 2] = wall_clock64();
            {
                double vk[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) vk[q] = ld_coh(Linv + (ty + 4 * q) * NB + tx);
#pragma unroll
                for (int q = 0; q < 16; ++q) Pm[(ty + 4 * q) * LD + tx] = vk[q];
            }
            __syncthreads();
            {
                chol_d4 x[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
                mfma_tile64<LD>(Pm, Qm, ty, tx, x);
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int c = 16 * ty + (tx >> 4) + 4 * e, r = 16 * rt + (tx & 15);
                        if (r < nr && c < nc) st_coh(A + (col0 + c) * lda + row0 + r, x[rt][e]);
                    }
            }
        }
        if (trace && t == 0) trace[task * 16 + 3] = wall_clock64();
        __builtin_amdgcn_s_waitcnt(0);                      // this wave's tile stores have reached the coherence point
        __syncthreads();
        if (t == 0) __hip_atomic_store(flags + (int64_t)i * nT + k, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (trace && t == 0) trace[task * 16 + 4] = wall_clock64();
    }
}

struct DataflowChol {
    int n = 0, nT = 0, ntasks = 0, epoch = 0, grid = 512;
    int *d_kfirst = nullptr, *d_flags = nullptr, *d_ctl = nullptr, *d_bk_ptr = nullptr, *d_bk_idx = nullptr;
    DfTask *d_tasks = nullptr;
    long long *d_trace = nullptr;                       // optional per-task timestamps (chol_test)
    std::vector<int> panel_first;                       // for the backward pass (as CholEnvelope)

    void release() {
        if (d_kfirst) (void)hipFree(d_kfirst);
        if (d_flags) (void)hipFree(d_flags);
        if (d_ctl) (void)hipFree(d_ctl);
        if (d_tasks) (void)hipFree(d_tasks);
        if (d_bk_ptr) (void)hipFree(d_bk_ptr);
        if (d_bk_idx) (void)hipFree(d_bk_idx);
        d_kfirst = d_flags = d_ctl = d_bk_ptr = d_bk_idx = nullptr; d_tasks = nullptr;
    }
    bool setup(const CholEnvelope &env) {
        release();
        n = env.n; nT = (n + CHOL_NB - 1) / CHOL_NB; epoch = 0;
        panel_first = env.panel_first;
        std::vector<int> kf(nT + 1, 0);
        for (int i = 0; i < nT; ++i) kf[i] = env.panel_first[i] / CHOL_NB;
        std::vector<DfTask> tasks;
        for (int k = 0; k < nT; ++k)
            for (int i = k; i <= nT; ++i)
                if (kf[i] <= k) tasks.push_back(DfTask{i, k});
        ntasks = (int)tasks.size();
        // backward pass: per panel j the tile rows i > j (matrix rows only) that reach column j, last first
        std::vector<int> bptr(nT + 1, 0), bidx;
        for (int j = 0; j < nT; ++j) {
            for (int i = nT - 1; i > j; --i)
                if (kf[i] <= j) bidx.push_back(i);
            bptr[j + 1] = (int)bidx.size();
        }
        if (bidx.empty()) bidx.push_back(0);
        if (hipMalloc(&d_bk_ptr, bptr.size() * sizeof(int)) != hipSuccess) return false;
        if (hipMalloc(&d_bk_idx, bidx.size() * sizeof(int)) != hipSuccess) return false;
        (void)hipMemcpy(d_bk_ptr, bptr.data(), bptr.size() * sizeof(int), hipMemcpyHostToDevice);
        (void)hipMemcpy(d_bk_idx, bidx.data(), bidx.size() * sizeof(int), hipMemcpyHostToDevice);
        if (hipMalloc(&d_kfirst, (nT + 1) * sizeof(int)) != hipSuccess) return false;
        if (hipMalloc(&d_flags, (size_t)(nT + 1) * nT * sizeof(int)) != hipSuccess) return false;
        if (hipMalloc(&d_ctl, 2 * sizeof(int)) != hipSuccess) return false;
        if (hipMalloc(&d_tasks, tasks.size() * sizeof(DfTask)) != hipSuccess) return false;
        (void)hipMemcpy(d_kfirst, kf.data(), (nT + 1) * sizeof(int), hipMemcpyHostToDevice);
        (void)hipMemcpy(d_tasks, tasks.data(), tasks.size() * sizeof(DfTask), hipMemcpyHostToDevice);
        (void)hipMemset(d_flags, 0, (size_t)(nT + 1) * nT * sizeof(int));
        if (const char *g = getenv("DBAT_HIP_DF_GRID")) grid = atoi(g);
        return true;
    }
    // Factor the lower triangle of the n x n matrix in A (lda >= n+1, n+1 columns)
    // and solve A q = b, b' in row n of A; q -> q_out.  linv_work as
    // BlockChol::linv_doubles.  info_dev: > 0 first non-positive pivot (LAPACK
    // potrf convention), -1 dataflow abort (spin cap).  One kernel launch.
    void solve(hipStream_t stream, double *A, int64_t lda, double *q_out, double *linv_work, int *info_dev) {
        (void)hipMemsetAsync(info_dev, 0, sizeof(int), stream);
        (void)hipMemsetAsync(d_ctl, 0, 2 * sizeof(int), stream);
        (void)hipMemsetAsync(q_out, 0xFF, (size_t)n * sizeof(double), stream);      // DF_SENTINEL
        ++epoch;
        hipLaunchKernelGGL(k_chol_df, dim3(std::min(grid, ntasks + nT)), dim3(256), 0, stream, A, lda, n, nT, d_kfirst,
                           d_tasks, ntasks, d_flags, d_ctl, epoch, linv_work, info_dev, d_trace, d_bk_ptr, d_bk_idx,
                           q_out);
    }
};

}  // namespace dbat
