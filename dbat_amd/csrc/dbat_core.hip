// Device engine + host damping loops + C ABI of the MI355X bundle core.
// See include/dbat_hip.h for the boundary and DESIGN.md for the data layout.
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <thread>
#include <atomic>
#include <vector>

#include "../../include/dbat_hip.h"
#include "chol_df.hpp"
#include "kernels.hpp"
#include "sig.hpp"
#include "heavy.hpp"
#include "plan.hpp"
#include "resect.hpp"

namespace dbat {

static thread_local std::string g_err;

struct DeviceError { std::string msg; };
struct UsageError { std::string msg; };

#define HIPCHK(expr)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            throw DeviceError{std::string(#expr) + ": " + hipGetErrorString(e_)};          \
    } while (0)

// every kernel launch is checked: a launch rejected for its grid, LDS size or resources would
// otherwise be skipped silently and the loops would carry on with stale buffers
#define LAUNCHK(...)                                                                       \
    do {                                                                                   \
        hipLaunchKernelGGL(__VA_ARGS__);                                                   \
        HIPCHK(hipGetLastError());                                                         \
    } while (0)

// RAII: make the handle's device current for the duration of an ABI call
struct DeviceGuard {
    int prev = -1, dev = -1;
    explicit DeviceGuard(int device) : dev(device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) HIPCHK(hipSetDevice(dev));
    }
    ~DeviceGuard() { if (prev >= 0 && prev != dev) (void)hipSetDevice(prev); }
};

#define NCCLCHK(expr)                                                                      \
    do {                                                                                   \
        ncclResult_t r_ = (expr);                                                          \
        if (r_ != ncclSuccess)                                                             \
            throw DeviceError{std::string(#expr) + ": " + ncclGetErrorString(r_)};         \
    } while (0)

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    void alloc(size_t count) {
        n = count;
        if (count) HIPCHK(hipMalloc((void **)&p, count * sizeof(T)));
    }
    template <class A>
    void upload(const std::vector<T, A> &v) {
        alloc(v.size());
        if (!v.empty()) HIPCHK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    }
    ~DevBuf() { if (p) (void)hipFree(p); }
};

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield");
#endif
}

#define DISPATCH_MODEL(KERNEL, ...)                                                          \
    switch (P.model) {                                                                       \
        case 2: KERNEL(2, __VA_ARGS__); break;                                               \
        case 3: KERNEL(3, __VA_ARGS__); break;                                               \
        case 4: KERNEL(4, __VA_ARGS__); break;                                               \
        default: KERNEL(5, __VA_ARGS__); break;                                              \
    }

struct Core {
    Plan P;
    DevProblem d{};
    hipStream_t stream = nullptr;
    int device = 0;
    int n_cu = 256;                  // compute units: launch size of the persistent tile kernel
    rocblas_handle blas = nullptr;
    // pinned host mailbox for the scalar read-backs of the damping loops (a copy to pageable memory is
    // staged and costs a blit kernel of ~18 us each, seven per LM step): [0..31] scalars, [32..35]
    // build sums, [40..47] pivots / info, [48..51] the pivot reset pattern
    double *hpin = nullptr;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t kev[13] = {};     // per-kernel brackets, recorded only while timing is on ([8], [9]: the all-reduce of the top tiles; [10 .. 12]: the kernels of the heavy / giant points)
    bool timing = false;
    // static problem data
    DevBuf<int32_t> cam_ncol, cam_col, cam_iorow, io_src, o_cam, o_pt;
    DevBuf<uint32_t> cam_eo_est, o_seg;
    DevBuf<double> io_fixed, px, cam_w, z_prw, z_prv, o_uv, o_w;
    // deterministic mode (dbat_hip_set_deterministic): ticket counters and the turn of every tile / chunk at them
    bool deterministic = false;
    DevBuf<double> det_cam_part, det_io_part, det_rr, det_u;      // kernels.hpp DevProblem::deterministic
    DevBuf<int32_t> det_cam_chunks;
    DevBuf<double> o_rhs;               // fixed IO: the corrected image coordinates (k_uv_to_rhs), what d.o_uv points at
    bool uv_pre = false;
    DevBuf<uint8_t> z_est, z_mine, o_lc, o_pidx;
    DevBuf<int32_t> tile_batch, tile_cam_start, tile_cams, tile_io_start, tile_iocols;
    DevBuf<uint8_t> tile_io_simple;
    DevBuf<uint8_t> tile_cam_io;
    DevBuf<int32_t> sg_chunk, sg_tile_chunk0, sg_gcam;   // signature groups (sig.hpp)
    int64_t sg_nchunks = 0;
    DevBuf<uint8_t> sg_lc;
    DevBuf<double> sg_uv, sg_w;
    bool use_sig = false;
    int sig_rb = 4;
    int tile_ncx = 6;
    int64_t ntiles = 0;
    size_t lds_tile2 = 0, lds_tile3 = 0;
    bool use_tile3 = false;
    int tile2_pc = TILE2_PC;
    bool use_tile2 = true;
    DevBuf<int64_t> o_row, batch_start, x2z, giant_start, cm_chunk_start;
    DevBuf<int32_t> cm_pt, cm_chunk_cam, tile_order;
    DevBuf<double> cm_uv, cm_w;
    int64_t n_cm_chunks = 0, n_cm_chunks_all = 0;
    DevBuf<double> giant_W;
    int64_t ngiant = 0;
    // heavy / giant points on the matrix cores (heavy.hpp; Plan::hv_*)
    bool use_heavy = false;
    HeavyDev hv{};
    DevBuf<int32_t> hv_obs_dst, hv_pt_io0, hv_io_dst, hv_io_pt, hv_pt_y, hv_grp_nb, hv_grp_row, hv_task, hv_ops;
    DevBuf<uint8_t> hv_obs_ld, hv_obs_ioloc, hv_io_ld;
    DevBuf<double> hv_Z;
    int giant_threads = 256;            // DBAT_HIP_GIANT_THREADS (64/128/256): tests force several chunks per point
    // state
    DevBuf<CamRec> cams, cams_f;                     // camera records at the linearisation point / at the last objective evaluation
    bool cams_at_lin = false;                        // cams holds the records of zlin
    // experiment switches, read once at set-up (never in the per-iteration path)

    const char *env_df_trace = nullptr;
    DevBuf<unsigned> gctr;                           // tickets of the in-kernel grid sums (zero between launches)
    DevBuf<double> gpart, rpart;                     // their per-block partial sums; sink of k_residual's (unused) sums
    DevBuf<double> z, zt, dz, zlin, vtmp, vtmp2, xbuf;  // NZ each (xbuf: n)
    DevBuf<double> red;      // [S | g_red | g_c | diagU | scal(8)]
    // Vinv: six doubles per object point -- since round 3 the FACTOR R of the point block's inverse (V^-1 = R R',
    // kernels.hpp point_block_factor), not the inverse; gp: B'r per point; jn2p: squared column norms of the point columns
    DevBuf<double> jn2c, dscale, rhs, Vinv, gp, jn2p, partial, scal;
    DevBuf<int> info;
    DevBuf<double> linv, ldiag;
    DataflowChol dfchol;                // persistent task-graph Cholesky, cameras in nested-dissection order (chol_df.hpp)
    DataflowChol dfchol_ip;             // the same in place, natural order (posterior covariance: L must end up in S)
    bool use_perm = true;
    bool mg_subtree = false;            // several ranks, domain sharding (nd.hpp): local factorisation + all-reduce of the top tiles
    DevBuf<uint8_t> piv_have;           // ... pivots of the reduced system that this rank computes
    bool chol_in_place = false;         // next factorisation must leave L in S
    int64_t ldS = 0;
    CholEnvelope env;
    DevBuf<unsigned long long> pivmm;   // [0..1] point pivots min/max, [2..3] reduced-system pivots
    double *S = nullptr, *g_red = nullptr, *g_c = nullptr, *diagU = nullptr, *red_scal = nullptr;
    int64_t red_count = 0;
    // multi-GPU: envelope of S + the vectors, contiguous (what the all-reduce carries)
    DevBuf<double> pk;
    DevBuf<int> col_bend;
    DevBuf<int64_t> col_off;
    int64_t pk_s_count = 0;
    int env_tail0 = 0;
    int64_t nb = 0, nobs = 0;
    int grid_obs = 1, grid_z = 1, grid_zs = 1;       // grid_zs: blocks of 1024 threads of the kernels that end in a grid sum
    size_t lds_build = 0, lds_back = 0;
    // multi-GPU: the RCCL communicator of this handle's rank (dbat_hip_comm_init), or the
    // caller's all-reduce callback (gloo / host tests)
    ncclComm_t nccl = nullptr;
    dbat_hip_allreduce_fn allreduce = nullptr;
    void *allreduce_user = nullptr;
    DevBuf<double> zgather;             // [NZ] owned entries of a z-vector, summed over the ranks
    bool multi() const { return nccl != nullptr || allreduce != nullptr; }
    // linearisation state
    double f_lin = 0, trace_jtj = 0, lambda_lin = 0;
    int scale_lin = 0;
    bool have_lin = false;
    bool s_valid = false;      // S holds an unfactorised reduced system
    bool s_dense_dirty = true; // S may hold non-zeros outside its envelope (fresh allocation, dense inverse)
    // counters
    int n_res_evals = 0, n_lin = 0, n_solves = 0, n_trace_only = 0;

    ~Core() {
        for (auto &e : ev) if (e) (void)hipEventDestroy(e);
        for (auto &e : kev) if (e) (void)hipEventDestroy(e);
        for (auto &e : st_ev) if (e) (void)hipEventDestroy(e);
        dfchol.release(); dfchol_ip.release();
        if (hpin) (void)hipHostFree(hpin);
        if (nccl) (void)ncclCommDestroy(nccl);
        if (blas) rocblas_destroy_handle(blas);
        if (stream) (void)hipStreamDestroy(stream);
    }

    void init(const dbat_hip_problem &pb) {
        const bool init_clock = env_int("DBAT_HIP_PLAN_STATS", 0) >= 2;
        auto t_init = std::chrono::steady_clock::now();
        auto lapi = [&](const char *what) {
            if (!init_clock) return;
            const auto now = std::chrono::steady_clock::now();
            fprintf(stderr, "[create clock] %-60s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_init).count());
            t_init = now;
        };
        device = pb.device;
        HIPCHK(hipSetDevice(pb.device));
        {
            hipDeviceProp_t prop;
            HIPCHK(hipGetDeviceProperties(&prop, pb.device));
            n_cu = std::max(1, prop.multiProcessorCount);
        }
        HIPCHK(hipStreamCreate(&stream));
        // coherent (fine-grained) on purpose: the kernels' stores must become visible to the spinning host
        // thread without a stream synchronisation, whatever HIP_HOST_COHERENT says
        HIPCHK(hipHostMalloc((void **)&hpin, 64 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
        memset(hpin, 0, 64 * sizeof(double));        // ticket slot 63 must not match the first wait by accident
        {
            const double big = 1e300;
            memcpy(&hpin[48], &big, 8); hpin[49] = 0.0; memcpy(&hpin[50], &big, 8); hpin[51] = 0.0;   // bit patterns: min = 1e300, max = +0
        }
        lapi("device, stream, pinned mailbox (the first call of a process initialises the HIP runtime: ~0.1 s)");
        for (auto &e : ev) HIPCHK(hipEventCreate(&e));
        for (auto &e : kev) HIPCHK(hipEventCreate(&e));
        lapi("events");
        nb = (int64_t)P.batch_start.size() - 1;
        nobs = (int64_t)P.o_cam.size();
        cam_ncol.upload(P.cam_ncol); cam_col.upload(P.cam_col); cam_iorow.upload(P.cam_iorow);
        cam_eo_est.upload(P.cam_eo_est); io_src.upload(P.io_src); io_fixed.upload(P.io_fixed);
        px.upload(P.px); cam_w.upload(P.cam_w);
        z_est.upload(P.z_est); z_mine.upload(P.z_mine); z_prw.upload(P.z_prw); z_prv.upload(P.z_prv);
        o_cam.upload(P.o_cam); o_pt.upload(P.o_pt); o_uv.upload(P.o_uv); o_w.upload(P.o_w);
        o_seg.upload(P.o_seg); o_row.upload(P.o_row); batch_start.upload(P.batch_start);
        x2z.upload(P.x2z);
        o_lc.upload(P.o_lc); o_pidx.upload(P.o_pidx); tile_batch.upload(P.tile_batch); tile_cam_start.upload(P.tile_cam_start); tile_cams.upload(P.tile_cams);
        tile_io_start.upload(P.tile_io_start); tile_iocols.upload(P.tile_iocols); tile_cam_io.upload(P.tile_cam_io);
        tile_ncx = P.shared_eo ? MAXCOL : (P.ncolmax <= 6 ? 6 : (P.ncolmax <= 14 ? 14 : (P.ncolmax <= 15 ? 15 : MAXCOL)));
        ntiles = (P.CMAX && P.nb_tiled > 0) ? (int64_t)P.tile_batch.size() - 1 : 0;
        d.nc = P.nc; d.np = P.np; d.nIOrows = P.nIOrows; d.nK = P.nK; d.nP = P.nP; d.nIOu = P.nIOu;
        d.ncolmax = P.ncolmax; d.BT = P.BT; d.NS = P.NS; d.NZ = P.NZ; d.nobs = nobs; d.nb = nb;
        d.cam_ncol = cam_ncol.p; d.cam_col = cam_col.p; d.cam_iorow = cam_iorow.p; d.cam_eo_est = cam_eo_est.p;
        d.io_src = io_src.p; d.io_fixed = io_fixed.p; d.px = px.p; d.cam_w = cam_w.p;
        d.z_est = z_est.p; d.z_mine = z_mine.p; d.z_prw = z_prw.p; d.z_prv = z_prv.p;
        d.o_cam = o_cam.p; d.o_pt = o_pt.p; d.o_uv = o_uv.p; d.o_w = P.uniform_w ? nullptr : o_w.p;
        d.o_seg = o_seg.p; d.o_row = o_row.p; d.batch_start = batch_start.p;
        d.CMAX = P.CMAX; d.ablate = env_int("DBAT_HIP_ABLATE", 0); d.trace_only = 0;
        d.any_prior = 0;
        for (double w : P.z_prw) if (w > 0) { d.any_prior = 1; break; }
        d.deterministic = 0; d.det_cam_part = d.det_io_part = d.det_rr = d.det_u = nullptr; d.det_cam_chunks = nullptr;
        env_df_trace = env_get("DBAT_HIP_DF_TRACE");
        d.ntiles = (int)ntiles; d.o_lc = o_lc.p; d.o_pidx = o_pidx.p;
        tile_order.upload(P.tile_order); d.tile_order = tile_order.p;
        cm_pt.upload(P.cm_pt); cm_uv.upload(P.cm_uv); cm_w.upload(P.cm_w);
        cm_chunk_cam.upload(P.cm_chunk_cam); cm_chunk_start.upload(P.cm_chunk_start);
        n_cm_chunks = P.n_cm_chunks_tiled; n_cm_chunks_all = (int64_t)P.cm_chunk_cam.size();
        ngiant = P.giant_start.empty() ? 0 : (int64_t)P.giant_start.size() - 1;
        if (ngiant > 0) {
            giant_start.upload(P.giant_start);
            giant_W.alloc((size_t)(P.giant_start.back() - P.giant_start.front()) * P.ncolmax * 3);
        }
        d.ngiant = (int)ngiant; d.giant_start = giant_start.p; d.giant_W = giant_W.p;
        use_heavy = P.hv_ok;
        if (use_heavy) {
            hv_obs_dst.upload(P.hv_obs_dst); hv_obs_ld.upload(P.hv_obs_ld); hv_obs_ioloc.upload(P.hv_obs_ioloc);
            hv_pt_io0.upload(P.hv_pt_io0); hv_io_dst.upload(P.hv_io_dst); hv_io_ld.upload(P.hv_io_ld); hv_io_pt.upload(P.hv_io_pt);
            hv_pt_y.upload(P.hv_pt_y); hv_grp_nb.upload(P.hv_grp_nb); hv_grp_row.upload(P.hv_grp_row);
            hv_task.upload(P.hv_task); hv_ops.upload(P.hv_ops);
            // rows of a slot that its point does not touch are never written: zero once, for good
            hv_Z.alloc((size_t)P.hv_z_doubles);
            HIPCHK(hipMemset(hv_Z.p, 0, (size_t)P.hv_z_doubles * sizeof(double)));
            hv.obs_dst = hv_obs_dst.p; hv.obs_ld = hv_obs_ld.p; hv.obs_ioloc = hv_obs_ioloc.p; hv.pt_io0 = hv_pt_io0.p;
            hv.io_dst = hv_io_dst.p; hv.io_ld = hv_io_ld.p; hv.io_pt = hv_io_pt.p; hv.pt_y = hv_pt_y.p;
            hv.grp_nb = hv_grp_nb.p; hv.grp_row = hv_grp_row.p; hv.task = hv_task.p; hv.ops = hv_ops.p;
            hv.obs0 = P.hv_obs0; hv.pt0 = P.hv_pt0; hv.ntasks = P.hv_ntasks; hv.max_batch_slots = P.hv_max_batch_slots;
        }
        if (const char *e = env_get("DBAT_HIP_GIANT_THREADS")) giant_threads = atoi(e);     // (64 | 128 | 256: env_validate has refused anything else)
        d.tile_batch = tile_batch.p; d.tile_cam_start = tile_cam_start.p; d.tile_cams = tile_cams.p;
        d.tile_io_start = tile_io_start.p; d.tile_iocols = tile_iocols.p; d.tile_cam_io = tile_cam_io.p;
        tile_io_simple.upload(P.tile_io_simple); d.tile_io_simple = P.tile_io_simple.empty() ? nullptr : tile_io_simple.p;
        use_sig = P.sg_ok && ntiles > 0 && tile_ncx <= 14;
        if (use_sig) {
            sg_chunk.upload(P.sg_chunk); sg_tile_chunk0.upload(P.sg_tile_chunk0); sg_lc.upload(P.sg_lc); sg_gcam.upload(P.sg_gcam);
            sg_nchunks = (int64_t)P.sg_chunk.size() / 8;
            sg_uv.upload(P.sg_uv);
            if (!P.uniform_w) sg_w.upload(P.sg_w);
            sig_rb = P.sg_rows_max <= 64 ? 4 : 5;
        }
        lapi("allocations and uploads of the plan");
        cams.alloc(P.nc); cams_f.alloc(P.nc);
        z.alloc(P.NZ); zt.alloc(P.NZ); dz.alloc(P.NZ); zlin.alloc(P.NZ); vtmp.alloc(P.NZ); vtmp2.alloc(P.NZ);
        xbuf.alloc(std::max<int64_t>(P.n, 1));
        ldS = ((P.NS + 1 + 7) / 8) * 8;
        d.ldS = ldS;
        const int64_t s_count = ldS * (P.NS + 1);
        red_count = s_count + 3 * P.NS + 8;
        red.alloc(red_count);
        S = red.p; g_red = S + s_count; g_c = g_red + P.NS; diagU = g_c + P.NS; red_scal = diagU + P.NS;
        ldiag.alloc(P.NS);
        {   // envelope of the reduced system from the camera co-visibility graph; IO rows are dense
            std::vector<int> first((size_t)P.NS, 0);
            for (int c = 0; c < P.nc; ++c)
                for (int k = 0; k < 6; ++k) first[(size_t)6 * c + k] = 6 * P.cam_first[c];
            if (P.shared_eo) env.build_dense((int)P.NS);   // shared EO: columns couple beyond the co-visibility band
            else env.build((int)P.NS, 6 * P.nc, first);
            {   // packed layout of the envelope for the all-reduce
                const int NSi = (int)P.NS;
                std::vector<int> cb((size_t)NSi);
                std::vector<int64_t> co((size_t)NSi + 1, 0);
                env_tail0 = env.tail0;
                for (int c = 0; c < NSi; ++c) {
                    const int be = c >= env.tail0 ? c : std::min(std::max(env.band_end[c / CHOL_NB], c + 1), env.tail0);
                    cb[c] = be;
                    co[c + 1] = co[c] + (be - c) + (NSi - std::max(env.tail0, be));
                }
                pk_s_count = co[NSi];
                col_bend.upload(cb); col_off.upload(co);
            }
            // nested-dissection order + compact tiles; in place (natural order, envelope pattern) where the factor
            // must end up in S (posterior covariance) and for shared EO blocks (no camera-wise dissection)
            use_perm = !P.shared_eo;
            if (!dfchol_ip.setup_inplace(env, ldS)) throw DeviceError{"out of device memory (Cholesky schedule)"};
            if (use_perm) {
                if (!dfchol.setup_permuted(P.nc, P.nIOu, P.cam_adj.data(), P.cam_adj_words, P.nd, P.rank))
                    throw DeviceError{"out of device memory (Cholesky schedule)"};
            }
            mg_subtree = P.mg_subtree && use_perm && dfchol.two_phase;
            if (mg_subtree) {     // pivots this rank computes: its own domain's and the (replicated) top separators' / IO
                std::vector<uint8_t> have((size_t)P.NS, 0);
                for (int c = 0; c < P.nc; ++c)
                    if (P.nd.cam_owner[c] < 0 || P.nd.cam_owner[c] == P.rank) for (int k = 0; k < 6; ++k) have[(size_t)6 * c + k] = 1;
                for (int64_t z = 6 * (int64_t)P.nc; z < P.NS; ++z) have[z] = 1;
                piv_have.upload(have);
            }
            linv.alloc(std::max(dfchol_ip.linv_doubles(), use_perm ? dfchol.linv_doubles() : (size_t)0));
        }
        lapi("envelope, schedules of the factorisation");
        jn2c.alloc(P.NS); dscale.alloc(P.NS); rhs.alloc(P.NS);
        Vinv.alloc((size_t)6 * P.np); gp.alloc((size_t)3 * P.np); jn2p.alloc((size_t)3 * P.np);
        HIPCHK(hipMemset(Vinv.p, 0, (size_t)6 * P.np * 8));
        HIPCHK(hipMemset(gp.p, 0, (size_t)3 * P.np * 8));
        HIPCHK(hipMemset(jn2p.p, 0, (size_t)3 * P.np * 8));
        grid_obs = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(nobs, 256), env_grid_obs()));
        grid_z = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(P.NZ, 256), 2048));
        grid_zs = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(P.NZ, 2048), 256));
        scal.alloc((size_t)16 + 4 * (size_t)P.nranks);
        HIPCHK(hipMemset(scal.p, 0, ((size_t)16 + 4 * (size_t)P.nranks) * sizeof(double)));   // unused slots travel through the all-reduce
        info.alloc(1);
        pivmm.alloc(4);
        gctr.alloc(16); HIPCHK(hipMemset(gctr.p, 0, 16 * sizeof(unsigned)));
        rpart.alloc((size_t)grid_obs);
        gpart.alloc((size_t)8 * std::max<int64_t>(std::max<int64_t>(grid_z, cdiv(P.NS, 256)), 2048));
        lds_build = ((size_t)P.BT * P.ncolmax * 3 + (size_t)P.BT * 18) * sizeof(double);
        lds_back = (size_t)P.BT * 6 * sizeof(double);
        // wave-specialised tile kernel: 256-observation batches, 16-point chunks, 2 panels
        tile2_pc = TILE2_PC;
        lds_tile2 = ((size_t)TILE2_NBUF * 3 * tile2_pc * TILE_LD + (size_t)256 * 9 + (size_t)128 * 15 + TILE_LD) * sizeof(double);
        use_tile2 = P.BT == 256 && P.ncolmax <= 15;      // (the plan does not tile anything else)
        partial.alloc((size_t)4 * std::max<int64_t>(std::max<int64_t>(std::max<int64_t>(nb + ntiles + ngiant + (int64_t)P.sg_chunk.size() / 8, n_cm_chunks_all), 2048), 1));
        set_lds_limits();
        HIPCHK(hipMemcpy(z.p, P.z0.data(), P.NZ * 8, hipMemcpyHostToDevice));
        HIPCHK(hipMemset(dz.p, 0, P.NZ * 8));
        HIPCHK(hipMemset(zt.p, 0, P.NZ * 8));          // (entries of other ranks' domains are never written)
        precompute_image_side();
        HIPCHK(hipDeviceSynchronize());
        lapi("work arrays, image side");
    }

    // Fixed interior orientation: the image side of every observation once per handle (kernels.hpp k_uv_to_rhs).
    // The point-major copy keeps the measured coordinates beside it (Jacobian export, forward intersection); the
    // camera-major and the slot-major copies are converted in place.
    void precompute_image_side() {
        uv_pre = !P.with_io;
        d.o_uv_raw = o_uv.p; d.uv_pre = uv_pre ? 1 : 0;
        if (!uv_pre || nobs == 0) return;
        prep_cams(z.p, cams_f.p);                    // interior orientation: the fixed values
        if (!o_rhs.p) o_rhs.alloc((size_t)2 * nobs);
#define L_RHS(M, dummy)                                                                                              \
        LAUNCHK((k_uv_to_rhs<M>), dim3((unsigned)cdiv(nobs, 256)), dim3(256), 0, stream, d.nK, d.nP, cams_f.p, nobs, o_cam.p, o_uv.p, o_rhs.p); \
        if (n_cm_chunks_all > 0) LAUNCHK((k_uv_to_rhs_cm<M>), dim3((unsigned)n_cm_chunks_all), dim3(256), 0, stream, d.nK, d.nP, cams_f.p, cm_chunk_cam.p, cm_chunk_start.p, cm_uv.p); \
        if (use_sig && sg_nchunks > 0) LAUNCHK((k_uv_to_rhs_sig<M>), dim3((unsigned)sg_nchunks), dim3(64), 0, stream, d.nK, d.nP, cams_f.p, sg_chunk.p, sg_gcam.p, sg_uv.p)
        DISPATCH_MODEL(L_RHS, 0)
#undef L_RHS
        d.o_uv = o_rhs.p;
    }

    // dbat_hip_set_values: new parameter values / prior observations for the same structure.  The host plan has them
    // already (plan_set_values); here the device copies, and everything a previous solve left behind is forgotten.
    void set_values(bool io_changed) {
        HIPCHK(hipMemcpyAsync(z.p, P.z0.data(), P.NZ * 8, hipMemcpyHostToDevice, stream));
        HIPCHK(hipMemcpyAsync(io_fixed.p, P.io_fixed.data(), P.io_fixed.size() * 8, hipMemcpyHostToDevice, stream));
        if (d.any_prior) {
            HIPCHK(hipMemcpyAsync(z_prw.p, P.z_prw.data(), P.NZ * 8, hipMemcpyHostToDevice, stream));
            HIPCHK(hipMemcpyAsync(z_prv.p, P.z_prv.data(), P.NZ * 8, hipMemcpyHostToDevice, stream));
        }
        HIPCHK(hipStreamSynchronize(stream));
        if (uv_pre && io_changed && nobs > 0) {
            // fixed interior orientation with other values: the corrected image coordinates again, from the measured ones
            // (the camera-major and slot-major copies were converted in place)
            HIPCHK(hipMemcpy(cm_uv.p, P.cm_uv.data(), P.cm_uv.size() * 8, hipMemcpyHostToDevice));
            if (use_sig && sg_nchunks > 0) HIPCHK(hipMemcpy(sg_uv.p, P.sg_uv.data(), P.sg_uv.size() * 8, hipMemcpyHostToDevice));
            d.o_uv = o_uv.p;
            precompute_image_side();
            HIPCHK(hipStreamSynchronize(stream));
        }
        cams_at_lin = false; have_lin = false; s_valid = false; pend_build = false; lin_pending = false;
        lambda_lin = 0; scale_lin = 0; f_lin = 0; trace_jtj = 0; near_singular = false; chol_in_place = false; replicate_next = false;
        n_res_evals = n_lin = n_solves = n_trace_only = 0;
    }

    // grid of the grid-stride observation kernels: one resident round (k_residual: 6 waves/SIMD
    // of 4-wave workgroups on 256 CUs)
    static int env_grid_obs() { return std::min(std::max(1, env_int("DBAT_HIP_GRID_OBS", 1536)), 1 << 20); }
    // kernels that use more than 64 KB of dynamic LDS must opt in
    void set_lds_limits() {
#define SET_LDS(K, BYTES) HIPCHK(hipFuncSetAttribute((const void *)(K), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BYTES)))
        if (use_tile2) {
            if (P.with_io) {
                SET_LDS((k_build_tile2<2, 14, TILE2_PC, TILE2_NBUF>), lds_tile2); SET_LDS((k_build_tile2<3, 14, TILE2_PC, TILE2_NBUF>), lds_tile2);
                SET_LDS((k_build_tile2<4, 14, TILE2_PC, TILE2_NBUF>), lds_tile2); SET_LDS((k_build_tile2<5, 14, TILE2_PC, TILE2_NBUF>), lds_tile2);
                SET_LDS((k_build_tile2<2, 15, TILE2_PC, TILE2_NBUF>), lds_tile2); SET_LDS((k_build_tile2<3, 15, TILE2_PC, TILE2_NBUF>), lds_tile2);
                SET_LDS((k_build_tile2<4, 15, TILE2_PC, TILE2_NBUF>), lds_tile2); SET_LDS((k_build_tile2<5, 15, TILE2_PC, TILE2_NBUF>), lds_tile2);
            }
        }
        lds_tile3 = ((size_t)TILE3_NBUF * 3 * TILE3_PC * TILE_LD + (size_t)256 * 9 + (size_t)256 * 9 + TILE_LD) * sizeof(double);
        // fixed IO: the variant with two producer groups (it holds at most 64 batch offsets per tile, the plan's
        // cap is 48); self-calibration: k_build_tile2
        use_tile3 = use_tile2 && !P.with_io;
        if (use_sig) {
#define SET_SIG(M, PW) SET_LDS((k_build_sig<M, 4, 6, PW>), sig_lds_bytes(4, false)); SET_LDS((k_build_sig<M, 5, 6, PW>), sig_lds_bytes(5, false)); \
                       SET_LDS((k_build_sig<M, 4, 14, PW>), sig_lds_bytes(4, true)); SET_LDS((k_build_sig<M, 5, 14, PW>), sig_lds_bytes(5, true));
            if (P.uniform_w) { SET_SIG(2, false); SET_SIG(3, false); SET_SIG(4, false); SET_SIG(5, false); }
            else { SET_SIG(2, true); SET_SIG(3, true); SET_SIG(4, true); SET_SIG(5, true); }
#undef SET_SIG
        }
        if (use_tile3) {
            SET_LDS((k_build_tile3<2, TILE3_PC, TILE3_NBUF>), lds_tile3); SET_LDS((k_build_tile3<3, TILE3_PC, TILE3_NBUF>), lds_tile3);
            SET_LDS((k_build_tile3<4, TILE3_PC, TILE3_NBUF>), lds_tile3); SET_LDS((k_build_tile3<5, TILE3_PC, TILE3_NBUF>), lds_tile3);
        }
        {
            const size_t lds_cov = ((size_t)P.BT * P.ncolmax * 3 + (size_t)P.BT * 6) * sizeof(double);
            SET_LDS((k_cov_points<2, false>), lds_cov); SET_LDS((k_cov_points<3, false>), lds_cov);
            SET_LDS((k_cov_points<4, false>), lds_cov); SET_LDS((k_cov_points<5, false>), lds_cov);
            SET_LDS((k_cov_points<2, true>), lds_cov); SET_LDS((k_cov_points<3, true>), lds_cov);
            SET_LDS((k_cov_points<4, true>), lds_cov); SET_LDS((k_cov_points<5, true>), lds_cov);
        }
        if (use_heavy) {
#define SET_HVZ(M) SET_LDS((k_heavy_z<M, 6>), heavy_z_lds_bytes(6, 0, true)); SET_LDS((k_heavy_z<M, 14>), heavy_z_lds_bytes(14, P.hv_max_batch_slots, true)); SET_LDS((k_heavy_z<M, 15>), heavy_z_lds_bytes(15, P.hv_max_batch_slots, true))
            SET_HVZ(2); SET_HVZ(3); SET_HVZ(4); SET_HVZ(5);
#undef SET_HVZ
        }
        SET_LDS((k_build<2, false>), lds_build); SET_LDS((k_build<3, false>), lds_build);
        SET_LDS((k_build<4, false>), lds_build); SET_LDS((k_build<5, false>), lds_build);
        SET_LDS((k_build<2, true>), lds_build); SET_LDS((k_build<3, true>), lds_build);
        SET_LDS((k_build<4, true>), lds_build); SET_LDS((k_build<5, true>), lds_build);
#undef SET_LDS
    }

    // v_mfma_f64_16x16x4_f64 instructions (2048 flops each) that one launch of the tile kernel executes: the symmetric
    // products it really runs, for the roofline entry beside the algorithmic (full product) count
    int64_t tile_kernel_mfma() const {
        if (!(ntiles > 0 && P.nb_tiled > 0)) return 0;
        int64_t n = 0;
        if (use_sig) {
            for (int64_t t = 0; t < ntiles; ++t) {
                const int nio = tile_ncx > 6 ? P.tile_io_start[t + 1] - P.tile_io_start[t] : 0;
                for (int64_t q = P.sg_tile_chunk0[t]; q < P.sg_tile_chunk0[t + 1]; ++q) {
                    const int npts = P.sg_chunk[8 * q + 1], k = P.sg_chunk[8 * q + 2];
                    const int ppr = std::min(SIG_PPR, 64 / std::max(k, 1)), rbk = (6 * k + nio + 1 + 15) >> 4;
                    for (int p0 = 0; p0 < npts; p0 += ppr) n += (int64_t)((3 * std::min(ppr, npts - p0) + 3) >> 2) * (rbk * (rbk + 1) / 2);
                }
            }
            return n;
        }
        // dense 128-row tiles: 36 lower-triangle blocks per k-step, chunks of PC points of every batch
        const int pc = (use_tile3 && tile_ncx == 6) ? TILE3_PC : tile2_pc;
        for (int64_t b = 0; b < P.nb_tiled; ++b) {
            const int64_t o1 = P.batch_start[b + 1];
            const int npts = o1 > P.batch_start[b] ? (int)P.o_pidx[o1 - 1] + 1 : 0;
            for (int p0 = 0; p0 < npts; p0 += pc) n += (int64_t)((3 * std::min(pc, npts - p0) + 3) >> 2) * 36;
        }
        return n;
    }

    // ---- helpers
    bool lin_pending = false;                        // k_finish's scalars are in the mailbox (or on their way), not read yet
    bool det_timeout_pending = false;                // ... and the timeout count of the deterministic signature kernel
    unsigned long long mb_seq = 0;                   // ticket of the last kernel that reports through mailbox slot 63
    bool mb_armed = false;                           // ... and such a kernel is the last one enqueued
    void sync() {
        // The kernel that ends a phase writes this wait's ticket into the mailbox after its results: spin on
        // it (a few microseconds after the kernel's last store) instead of sleeping in
        // hipStreamSynchronize; after 20 ms without the ticket fall back to the driver's wait (a wait that
        // long does not notice the driver's wake-up latency; the bound must cover a whole enqueued LM step --
        // 1.9 ms at C3, 15 ms at C4 -- because the host runs ahead of the stream: with 2 ms the C3 step
        // measured 2.32 instead of 1.88 ms).  A ticket hit skips the driver's wait, so
        // asynchronous device errors surface at the next real synchronisation: every ABI call that returns
        // results ends in one (z_to_x / read_scal).
        bool done = false;
        if (mb_armed) {
            volatile unsigned long long *slot = reinterpret_cast<volatile unsigned long long *>(hpin) + 63;
            const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(20);
            for (;;) {
                for (int spin = 0; spin < 512 && !done; ++spin) { done = *slot == mb_seq; if (!done) cpu_relax(); }
                if (done || std::chrono::steady_clock::now() >= t_end) break;
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            mb_armed = false;
        }
        if (!done) HIPCHK(hipStreamSynchronize(stream));
        if (det_timeout_pending) {
            det_timeout_pending = false;
            unsigned n_to = 0;
            memcpy(&n_to, hpin + 60, sizeof(unsigned));
            if (n_to) {
                HIPCHK(hipMemsetAsync(gctr.p + 7, 0, sizeof(unsigned), stream));
                throw DeviceError{"deterministic mode: " + std::to_string(n_to) + " chunk(s) of the signature kernel gave up waiting for their turn at "
                                  "the tile (spin cap): the sums of this linearisation are not order-fixed"};
            }
        }
        if (lin_pending) {      // trace(J'J): camera part from jn2c (estimated), point part from red_scal[1]
            f_lin = 0.5 * hpin[32];
            trace_jtj = hpin[34] + hpin[33];
            lin_pending = false;
        }
    }
    void mark(int i) { if (timing) HIPCHK(hipEventRecord(kev[i], stream)); }
    // ---- stage timers of a solve (dbat_hip_result.stage_s): an event where a stage is enqueued; the stream
    // time between consecutive events goes to the stage of the first.  Events come from a pool that grows
    // with the longest loop seen; nothing is read back before the loop has ended.
    std::vector<hipEvent_t> st_ev;
    std::vector<int> st_id;
    bool st_on = false;
    void stage(int id) {
        if (!st_on) return;
        if (st_id.size() == st_ev.size()) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); st_ev.push_back(e); }
        HIPCHK(hipEventRecord(st_ev[st_id.size()], stream));
        st_id.push_back(id);
    }
    void stages_begin() { st_id.clear(); st_on = true; stage(4); }
    void stages_end(double *out5) {
        stage(4);
        st_on = false;
        HIPCHK(hipStreamSynchronize(stream));
        for (int i = 0; i < 5; ++i) out5[i] = 0.0;
        for (size_t i = 0; i + 1 < st_id.size(); ++i) {
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, st_ev[i], st_ev[i + 1]));
            out5[st_id[i]] += 1e-3 * ms;
        }
    }
    // sum of [S | g_red | g_c | diagU | scalars] over the ranks: the envelope of S is packed
    // next to the vectors, one all-reduce, unpacked again
    void allreduce_system() {
        if (!multi()) return;
        const int64_t nvec = 3 * P.NS + 8;
        if (!pk.p) pk.alloc((size_t)(pk_s_count + nvec));
        LAUNCHK(k_pack_envelope, dim3((unsigned)P.NS), dim3(256), 0, stream, S, ldS, (int)P.NS, env_tail0,
                           col_bend.p, col_off.p, pk.p, 1);
        HIPCHK(hipMemcpyAsync(pk.p + pk_s_count, g_red, nvec * sizeof(double), hipMemcpyDeviceToDevice, stream));
        do_allreduce(pk.p, pk_s_count + nvec);
        LAUNCHK(k_pack_envelope, dim3((unsigned)P.NS), dim3(256), 0, stream, S, ldS, (int)P.NS, env_tail0,
                           col_bend.p, col_off.p, pk.p, 0);
        HIPCHK(hipMemcpyAsync(g_red, pk.p + pk_s_count, nvec * sizeof(double), hipMemcpyDeviceToDevice, stream));
    }
    void do_allreduce(double *buf, int64_t count) {
        if (nccl) {          // RCCL over xGMI, in place, ordered on the handle's stream
            NCCLCHK(ncclAllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, nccl, stream));
        } else if (allreduce) {     // installed only by multi-rank drivers (a one-rank group exercises the same path)
            if (allreduce(allreduce_user, buf, count, (void *)stream) != 0) throw DeviceError{"all-reduce callback failed"};
        }
    }
    // A z-vector whose entries are valid on their owning rank only (object points on their
    // shard, EO/IO on rank 0) -> the full vector on every rank.  Collective.
    const double *gathered(const double *z_dev) {
        if (!multi()) return z_dev;
        if (!zgather.p) zgather.alloc((size_t)P.NZ);
        LAUNCHK(k_mask_owned, dim3((unsigned)cdiv(P.NZ, 256)), dim3(256), 0, stream, P.NZ, z_mine.p, z_dev, zgather.p);
        do_allreduce(zgather.p, P.NZ);
        return zgather.p;
    }
    void read_scal(double *host, int n) {
        if (n <= 32) {
            HIPCHK(hipMemcpyAsync(hpin, scal.p, n * sizeof(double), hipMemcpyDeviceToHost, stream));
            sync();
            memcpy(host, hpin, n * sizeof(double));
            return;
        }
        HIPCHK(hipMemcpyAsync(host, scal.p, n * sizeof(double), hipMemcpyDeviceToHost, stream));
        sync();
    }
    void prep_cams(const double *zz, CamRec *into = nullptr) {
        if (!into) { into = cams.p; cams_at_lin = false; }      // build() sets it again once zlin == zz
        LAUNCHK(k_cam_prep, dim3((unsigned)cdiv(P.nc, 64)), dim3(64), 0, stream, d, zz, into);
    }
    void x_to_z(const double *x_host, double *z_dev) {
        // z keeps fixed entries; estimated entries overwritten from x
        HIPCHK(hipMemcpyAsync(z_dev, P.z0.data(), P.NZ * 8, hipMemcpyHostToDevice, stream));
        if (P.n) {
            HIPCHK(hipMemcpyAsync(xbuf.p, x_host, P.n * 8, hipMemcpyHostToDevice, stream));
            LAUNCHK(k_scatter_x, dim3((unsigned)cdiv(P.n, 256)), dim3(256), 0, stream, P.n, x2z.p, xbuf.p, z_dev);
        }
    }
    void z_to_x(const double *z_dev, double *x_host) {     // collective on a sharded handle
        if (!P.n) return;
        stage(4);
        z_dev = gathered(z_dev);
        LAUNCHK(k_gather_x, dim3((unsigned)cdiv(P.n, 256)), dim3(256), 0, stream, P.n, x2z.p, z_dev, xbuf.p);
        HIPCHK(hipMemcpyAsync(x_host, xbuf.p, P.n * 8, hipMemcpyDeviceToHost, stream));
        sync();
    }

    // ---- The parameter trace of a damping loop (bundle.m's T: x of every iteration), kept ON THE DEVICE while the loop
    // runs -- one gather kernel per column, no copy and no wait -- and brought down once at the end.  (Round 5 fetched
    // every column when it was made: a 24 MB copy into pageable memory and a synchronisation per iteration at C3, then a
    // second copy into the caller's array.)  Columns a loop never wrote come back as NaN.
    DevBuf<double> trace_dev;
    int trace_cap = 0, trace_n = 0;
    std::vector<uint8_t> trace_have;
    void trace_begin() { trace_n = 0; trace_have.clear(); }
    void trace_put(int k, const double *z_dev) {       // column k <- x(z_dev); collective on a sharded handle
        if (!P.n || k < 0) return;
        if (k >= trace_cap) {
            const int cap = std::max(8, std::max(2 * trace_cap, k + 1));
            DevBuf<double> nb;
            nb.alloc((size_t)P.n * cap);
            if (trace_cap > 0) HIPCHK(hipMemcpyAsync(nb.p, trace_dev.p, (size_t)P.n * trace_cap * 8, hipMemcpyDeviceToDevice, stream));
            HIPCHK(hipStreamSynchronize(stream));      // (the old buffer is freed with the swap)
            std::swap(trace_dev.p, nb.p);
            trace_cap = cap;
        }
        stage(4);
        z_dev = gathered(z_dev);
        LAUNCHK(k_gather_x, dim3((unsigned)cdiv(P.n, 256)), dim3(256), 0, stream, P.n, x2z.p, z_dev, trace_dev.p + (size_t)P.n * k);
        if ((int)trace_have.size() <= k) trace_have.resize((size_t)k + 1, 0);
        trace_have[k] = 1;
        trace_n = std::max(trace_n, k + 1);
    }
    void trace_truncate(int n) { trace_n = std::min(trace_n, std::max(n, 0)); }
    int trace_download(double *host, int cap_cols) {    // -> number of columns
        const int nt = std::min(trace_n, cap_cols);
        if (!P.n || nt <= 0) return std::max(nt, 0);
        HIPCHK(hipMemcpyAsync(host, trace_dev.p, (size_t)P.n * nt * 8, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        for (int k = 0; k < nt; ++k)
            if (!trace_have[k]) std::fill(host + (size_t)P.n * k, host + (size_t)P.n * (k + 1), NAN);
        if ((size_t)P.n * trace_cap * 8 > ((size_t)256 << 20)) {      // a kept handle does not sit on a large trace buffer
            DevBuf<double> old;
            std::swap(old.p, trace_dev.p);
            trace_cap = 0;
        }
        return nt;
    }

    // ---- K2: f = 0.5 r'r at zz (all ranks' sum).  Optionally store r.
    // f at x + alpha p (x, p as given; the point goes to out): the trial point and its camera records in one launch
    double eval_f_step(const double *x, double alpha, const double *pdir, double *out) {
        stage(3);
        if ((int64_t)cdiv(P.NZ, 256) < P.nc) { axpby(1.0, x, alpha, pdir, out); return eval_f(out, nullptr, nullptr); }
        LAUNCHK(k_axpby_cams, dim3((unsigned)cdiv(P.NZ, 256)), dim3(256), 0, stream, d, 1.0, x, alpha, pdir, out, cams_f.p);
        return eval_f(out, nullptr, nullptr, true);
    }
    double eval_f(const double *zz, double *r_w_out, double *r_unw_out, bool cams_ready = false) {
        stage(3);
        // own camera records: the ones of the linearisation point stay valid for the next solve
        if (!cams_ready) prep_cams(zz, cams_f.p);
        mark(6);
        // the objective value always comes from the camera-major kernel (one summation order for
        // every value the damping loops compare); the point-major one only when residuals are exported.
        // Both kernels finish their grid sums themselves; the second one adds the prior rows and hands
        // the total to the pinned mailbox (one rank) or to scal[0] for the all-reduce.
        // small projects without prior observations: the residual kernel's last block sums up and tells the host itself
        const bool res_tail = n_cm_chunks_all > 0 && n_cm_chunks_all <= 512 && !d.any_prior && !multi() && !r_w_out && !r_unw_out;
        if (n_cm_chunks_all > 0) {
#define L_RESCM(M, PRE) LAUNCHK((k_residual_cm<M, PRE>), dim3((unsigned)n_cm_chunks_all), dim3(256), 0, stream, d, zz, cams_f.p, cm_pt.p, cm_uv.p, P.uniform_w ? (const double *)nullptr : cm_w.p, cm_chunk_cam.p, cm_chunk_start.p, partial.p, \
                                res_tail ? gctr.p + 1 : (unsigned *)nullptr, scal.p, hpin, res_tail ? ++mb_seq : 0ull)
            if (uv_pre) { DISPATCH_MODEL(L_RESCM, true) } else { DISPATCH_MODEL(L_RESCM, false) }
#undef L_RESCM
        }
        mark(7);
        if (r_w_out || r_unw_out) {
#define L_RES(M, PRE) LAUNCHK((k_residual<M, PRE>), dim3(grid_obs), dim3(256), 0, stream, d, zz, cams_f.p, rpart.p, r_w_out, r_unw_out)
            if (uv_pre) { DISPATCH_MODEL(L_RES, true) } else { DISPATCH_MODEL(L_RES, false) }
#undef L_RES
        }
        if (!res_tail)
            LAUNCHK(k_prior_sq, dim3(grid_zs), dim3(1024), 0, stream, d, zz, gpart.p, gctr.p + 1, (const double *)partial.p, n_cm_chunks_all,
                    scal.p, multi() ? (double *)nullptr : hpin, ++mb_seq);
        double s;
        if (multi()) { do_allreduce(scal.p, 1); read_scal(&s, 1); }
        else { mb_armed = true; sync(); s = hpin[0]; }
        ++n_res_evals;
        return 0.5 * s;
    }

    // Deterministic mode: exact sums into the reduced system (kernels.hpp DevProblem::deterministic).  Covered: the
    // signature-group path with its camera-side kernels (C1 ... C4); the factorisation and the back-substitution are
    // order-fixed by construction.  Not covered yet: heavy / giant points and the tile kernels of irregular scenes
    // (per-point sums by LDS atomics), shared EO blocks, several ranks.
    // Every scene on one rank whose cameras have at most nine IO columns and own EO elements: the signature-group path keeps
    // its tile kernel (C1 ... C4: + 4 % at C3); scenes with irregular visibility (the reference's own projects) run the
    // column-list kernel k_build over ALL their batches instead of the wave-specialised tile kernels, whose per-point sums
    // are LDS atomics from several waves.
    int det_ncx() const { return P.ncolmax <= 6 ? 6 : (P.ncolmax <= 14 ? 14 : 15); }
    bool deterministic_supported() const { return P.nranks == 1 && !P.shared_eo && P.ncolmax <= 15 && n_cm_chunks_all > 0; }
    void set_deterministic(bool on) {
        if (on && !deterministic_supported())
            throw UsageError{"deterministic mode: one rank, no shared EO blocks (camera stations), at most nine estimated IO columns per camera"};
        if (on && !det_u.p) {
            const int nc = P.nc;
            det_cam_part.alloc((size_t)std::max<int64_t>(n_cm_chunks_all, 1) * DET_CP);
            det_io_part.alloc((size_t)nc * DET_IOP);
            det_rr.alloc((size_t)nc + 1 + DET_PRIOR_PARTS);
            det_u.alloc((size_t)P.NS + 1);
            // chunk ranges of every camera in the camera-major copy: its tiled chunks [2c, 2c+1), then -- offset 2 nc + 1 --
            // its untiled ones (both parts are sorted by camera)
            std::vector<int32_t> cr((size_t)2 * (2 * nc + 1), 0);
            for (int part = 0; part < 2; ++part) {
                const int64_t q0 = part == 0 ? 0 : n_cm_chunks, q1 = part == 0 ? n_cm_chunks : n_cm_chunks_all;
                int32_t *r = cr.data() + part * (2 * nc + 1);
                for (int c = 0; c < nc; ++c) { r[2 * c] = (int32_t)q1; r[2 * c + 1] = (int32_t)q1; }
                for (int64_t q = q0; q < q1; ++q) {
                    const int c = P.cm_chunk_cam[q];
                    if (r[2 * c] == (int32_t)q1) r[2 * c] = (int32_t)q;
                    r[2 * c + 1] = (int32_t)q + 1;
                }
                for (int c = 0; c < nc; ++c) if (r[2 * c] == (int32_t)q1) r[2 * c + 1] = (int32_t)q1;   // (a camera without chunks: empty range)
            }
            det_cam_chunks.upload(cr);
        }
        deterministic = on;
        d.deterministic = on ? 1 : 0;
        d.det_cam_part = det_cam_part.p; d.det_io_part = det_io_part.p; d.det_rr = det_rr.p; d.det_u = det_u.p;
        d.det_cam_chunks = det_cam_chunks.p;
    }
    // the camera side of a deterministic linearisation, after the camera-major kernel has left its chunk partials
    const double *det_z = nullptr;      // the point of the deterministic linearisation in progress
    void det_camera_side() {
        const int nio = (int)P.nIOu;
        const int ncx = det_ncx();
#define L_DCR(NCXV) LAUNCHK((k_det_cam_reduce<NCXV>), dim3((unsigned)P.nc), dim3(256), 0, stream, d, cams.p, S, g_c, g_red, diagU)
        if (ncx == 6) L_DCR(6); else if (ncx == 14) L_DCR(14); else L_DCR(15);
#undef L_DCR
        if (ncx > 6 && nio > 0) LAUNCHK((k_det_io_reduce<15>), dim3((unsigned)(nio * (nio + 1) / 2 + nio)), dim3(256), 0, stream, d, cams.p, nio, S, g_c, g_red, diagU);
        if (d.any_prior) LAUNCHK(k_det_prior_sq, dim3(DET_PRIOR_PARTS), dim3(256), 0, stream, d, det_z);
        LAUNCHK(k_det_rows, dim3((unsigned)cdiv(P.NS, 256)), dim3(256), 0, stream, d, (const double *)diagU);
        if (ncx == 6) LAUNCHK((k_det_round_cam<6>), dim3((unsigned)P.nc), dim3(256), 0, stream, d, cams.p, 0, S, g_red);
        else LAUNCHK((k_det_round_cam<15>), dim3((unsigned)P.nc + 1), dim3(256), 0, stream, d, cams.p, nio, S, g_red);
    }

    // ---- K1: linearise at zz with damping lambda; builds the reduced system.
    void build_enqueue(const double *zz, double lambda, int scale) {
        stage(0);
        const bool fused_first = !s_dense_dirty && P.NS >= P.nc;
        if (!fused_first) prep_cams(zz);
        else cams_at_lin = false;
        // only the envelope of S is ever written or read: zero that (and the vectors behind S); the whole
        // array once, and again after something filled it densely (the inverse of the posterior covariance)
        bool pivmm_set = false;
        if (s_dense_dirty) {
            HIPCHK(hipMemsetAsync(red.p, 0, red_count * sizeof(double), stream));
            s_dense_dirty = false;
        } else {
            // ... and the vectors behind S, and the pivot extremes {min, max} x {points, cameras}
            // ... and the camera records at zz, all in the first launch
            LAUNCHK(k_envelope_cams, dim3((unsigned)P.NS), dim3(256), 0, stream, d, zz, cams.p, S, ldS, (int)P.NS, env_tail0, col_bend.p, g_red, pivmm.p);
            pivmm_set = true;
        }
        if (!pivmm_set) HIPCHK(hipMemcpyAsync(pivmm.p, hpin + 48, 4 * sizeof(double), hipMemcpyHostToDevice, stream));
        // tiled batches through the MFMA kernel, the remaining ("heavy point") batches
        // -- or all of them when tiling is off -- through k_build
        int64_t npart = 0;
        // deterministic mode off the signature path: no tile kernel, k_build takes every batch
        const bool det_list = deterministic && !(use_sig && ntiles > 0 && P.nb_tiled > 0);
        const int64_t nb_tiled = det_list ? 0 : P.nb_tiled;
        if (deterministic) {
            // the camera side of EVERY observation (tiled or not) from the camera-major kernels, chunk partials summed in order
#define L_CAMN_ALL(M, NCXV) LAUNCHK((k_cam_normal<M, NCXV>), dim3((unsigned)n_cm_chunks_all), dim3(256), 0, stream, d, zz, cams.p, cm_pt.p, cm_uv.p, P.uniform_w ? (const double *)nullptr : cm_w.p, cm_chunk_cam.p, cm_chunk_start.p, S, g_c, g_red, diagU)
#define L_CAMN6_ALL(M, dummy) LAUNCHK((k_cam_normal6<M>), dim3((unsigned)n_cm_chunks_all), dim3(256), 0, stream, d, zz, cams.p, cm_pt.p, cm_uv.p, P.uniform_w ? (const double *)nullptr : cm_w.p, cm_chunk_cam.p, cm_chunk_start.p, S, g_c, g_red, diagU)
            const int ncx = det_ncx();
            if (ncx == 6) { DISPATCH_MODEL(L_CAMN6_ALL, 0) } else if (ncx == 14) { DISPATCH_MODEL(L_CAMN_ALL, 14) } else { DISPATCH_MODEL(L_CAMN_ALL, 15) }
#undef L_CAMN_ALL
#undef L_CAMN6_ALL
            det_z = zz;
            det_camera_side();
        }
        if (ntiles > 0 && nb_tiled > 0) {
            npart = ntiles;
#define L_TILE2(M, NCXV) LAUNCHK((k_build_tile2<M, NCXV, TILE2_PC, TILE2_NBUF>), dim3((unsigned)ntiles), dim3(512), lds_tile2, stream, d, zz, cams.p, lambda, scale, S, g_c, g_red, diagU, Vinv.p, gp.p, jn2p.p, partial.p, pivmm.p)
#define L_CAMN(M, NCXV) LAUNCHK((k_cam_normal<M, NCXV>), dim3((unsigned)n_cm_chunks), dim3(256), 0, stream, d, zz, cams.p, cm_pt.p, cm_uv.p, P.uniform_w ? (const double *)nullptr : cm_w.p, cm_chunk_cam.p, cm_chunk_start.p, S, g_c, g_red, diagU)
            if (!deterministic && use_tile2 && tile_ncx <= 15 && n_cm_chunks > 0) {
                // camera side of the tiled observations: J_c'J_c, J_c'r, squared column norms
#define L_CAMN6(M, dummy) LAUNCHK((k_cam_normal6<M>), dim3((unsigned)n_cm_chunks), dim3(256), 0, stream, d, zz, cams.p, cm_pt.p, cm_uv.p, P.uniform_w ? (const double *)nullptr : cm_w.p, cm_chunk_cam.p, cm_chunk_start.p, S, g_c, g_red, diagU)
                if (tile_ncx == 6) { DISPATCH_MODEL(L_CAMN6, 0) } else if (tile_ncx == 14) { DISPATCH_MODEL(L_CAMN, 14) } else { DISPATCH_MODEL(L_CAMN, 15) }
#undef L_CAMN6
            }
#undef L_CAMN
            mark(0);                                 // events around the tile kernel alone (bench roofline)
#define L_TILE3(M, DUMMY) LAUNCHK((k_build_tile3<M, TILE3_PC, TILE3_NBUF>), dim3((unsigned)ntiles), dim3(768), lds_tile3, stream, d, zz, cams.p, lambda, scale, S, g_red, Vinv.p, gp.p, jn2p.p, partial.p, pivmm.p)
#define L_SIGW(M, RBV, PW) LAUNCHK((k_build_sig<M, (RBV) % 8, (RBV) / 8, PW>), dim3((unsigned)std::min<int64_t>(ntiles, n_cu)), dim3(64 * sig_waves((RBV) % 8, (RBV) / 8 > 6)), sig_lds_bytes((RBV) % 8, (RBV) / 8 > 6), stream, d, zz, cams.p, lambda, scale, S, g_red, Vinv.p, gp.p, jn2p.p, partial.p, pivmm.p, sg_chunk.p, sg_tile_chunk0.p, sg_lc.p, sg_uv.p, PW ? sg_w.p : (const double *)nullptr, gctr.p + 5)
#define L_SIG(M, RBV) do { if (P.uniform_w) L_SIGW(M, RBV, false); else L_SIGW(M, RBV, true); } while (0)
            // (row blocks, camera-side columns) packed into one macro argument: RB + 8 * NCX
            if (use_sig && tile_ncx == 6 && sig_rb == 4) { DISPATCH_MODEL(L_SIG, 4 + 8 * 6) }
            else if (use_sig && tile_ncx == 6) { DISPATCH_MODEL(L_SIG, 5 + 8 * 6) }
            else if (use_sig && sig_rb == 4) { DISPATCH_MODEL(L_SIG, 4 + 8 * 14) }
            else if (use_sig) { DISPATCH_MODEL(L_SIG, 5 + 8 * 14) }
            else if (use_tile3 && tile_ncx == 6) { DISPATCH_MODEL(L_TILE3, 0) }
            else if (use_tile2 && tile_ncx == 14) { DISPATCH_MODEL(L_TILE2, 14) }
            else if (use_tile2 && tile_ncx == 15) { DISPATCH_MODEL(L_TILE2, 15) }
            else throw DeviceError{"internal: tiles without a tile kernel"};
            mark(1);
#undef L_TILE2
        }
        const bool no_tiles = !(ntiles > 0 && nb_tiled > 0);
        if (no_tiles) mark(0);                       // no tile kernel: the events bracket the kernels of the untiled points instead
        // heavy.hpp: the untiled points -- batches [P.nb_tiled, nb) and the giant points -- on the matrix cores; the
        // tiled batches that the deterministic mode keeps off the tile kernels stay with the column lists
        const int64_t nb_lists = use_heavy ? P.nb_tiled : nb;      // batches [nb_tiled, nb_lists) go through k_build
        if (nb_lists > nb_tiled) {
#define L_BUILD(M, IO) LAUNCHK((k_build<M, IO>), dim3((unsigned)(nb_lists - nb_tiled)), dim3(P.BT), lds_build, stream, d, zz, cams.p, lambda, scale, S, g_c, g_red, diagU, Vinv.p, gp.p, jn2p.p, partial.p + npart, pivmm.p, (int)nb_tiled)
            if (P.with_io) { DISPATCH_MODEL(L_BUILD, true) } else { DISPATCH_MODEL(L_BUILD, false) }
#undef L_BUILD
            npart += nb_lists - nb_tiled;
        }
        if (use_heavy) {
            mark(10);
            if (!deterministic && n_cm_chunks_all > n_cm_chunks) {
                // camera side of the untiled observations (deterministic mode: every chunk has been through it above)
                const unsigned nch = (unsigned)(n_cm_chunks_all - n_cm_chunks);
#define L_CAMNU(M, NCXV) LAUNCHK((k_cam_normal<M, NCXV>), dim3(nch), dim3(256), 0, stream, d, zz, cams.p, cm_pt.p, cm_uv.p, P.uniform_w ? (const double *)nullptr : cm_w.p, cm_chunk_cam.p + n_cm_chunks, cm_chunk_start.p + n_cm_chunks, S, g_c, g_red, diagU)
#define L_CAMN6U(M, dummy) LAUNCHK((k_cam_normal6<M>), dim3(nch), dim3(256), 0, stream, d, zz, cams.p, cm_pt.p, cm_uv.p, P.uniform_w ? (const double *)nullptr : cm_w.p, cm_chunk_cam.p + n_cm_chunks, cm_chunk_start.p + n_cm_chunks, S, g_c, g_red, diagU)
                if (tile_ncx == 6) { DISPATCH_MODEL(L_CAMN6U, 0) } else if (tile_ncx == 14) { DISPATCH_MODEL(L_CAMNU, 14) } else { DISPATCH_MODEL(L_CAMNU, 15) }
#undef L_CAMNU
#undef L_CAMN6U
            }
            if (nb > P.nb_tiled) {
#define L_HVZ(M, NCXV) LAUNCHK((k_heavy_z<M, NCXV>), dim3((unsigned)(nb - P.nb_tiled)), dim3(256), heavy_z_lds_bytes(NCXV, P.hv_max_batch_slots, deterministic), stream, d, hv, zz, cams.p, lambda, scale, hv_Z.p, Vinv.p, gp.p, jn2p.p, partial.p + npart, pivmm.p, (int)P.nb_tiled)
                if (tile_ncx == 6) { DISPATCH_MODEL(L_HVZ, 6) } else if (tile_ncx == 14) { DISPATCH_MODEL(L_HVZ, 14) } else { DISPATCH_MODEL(L_HVZ, 15) }
#undef L_HVZ
                npart += nb - P.nb_tiled;
            }
            if (ngiant > 0) {
#define L_HVG(M, NCXV) LAUNCHK((k_heavy_z_giant<M, NCXV>), dim3((unsigned)ngiant), dim3(giant_threads), 0, stream, d, hv, zz, cams.p, lambda, scale, hv_Z.p, Vinv.p, gp.p, jn2p.p, partial.p + npart, pivmm.p)
                if (tile_ncx == 6) { DISPATCH_MODEL(L_HVG, 6) } else if (tile_ncx == 14) { DISPATCH_MODEL(L_HVG, 14) } else { DISPATCH_MODEL(L_HVG, 15) }
#undef L_HVG
                npart += ngiant;
            }
            mark(11);
            LAUNCHK(k_heavy_syrk, dim3((unsigned)cdiv(hv.ntasks, 4)), dim3(256), 0, stream, d, hv, (const double *)hv_Z.p, S, g_red);
            mark(12);
        }
        if (no_tiles) mark(1);
        if (ngiant > 0 && !use_heavy) {               // points with more observations than a batch holds
#define L_GIANT(M, IO) LAUNCHK((k_build_giant<M, IO>), dim3((unsigned)ngiant), dim3(giant_threads), 0, stream, d, zz, cams.p, lambda, scale, S, g_c, g_red, diagU, Vinv.p, gp.p, jn2p.p, partial.p + npart, pivmm.p)
            if (P.with_io) { DISPATCH_MODEL(L_GIANT, true) } else { DISPATCH_MODEL(L_GIANT, false) }
#undef L_GIANT
            npart += ngiant;
        }
        if (deterministic && use_sig && !no_tiles) {
            // deterministic mode: chunks whose turn at the tile never came (sig.hpp, spin cap) -- read with the scalars
            HIPCHK(hipMemcpyAsync(hpin + 60, gctr.p + 7, sizeof(unsigned), hipMemcpyDeviceToHost, stream));
            det_timeout_pending = true;
        }
        if ((d.ablate & 32) && use_heavy && nb > P.nb_tiled) {      // phase profile of k_heavy_z (thread 0 of every batch)
            unsigned long long h[16];
            HIPCHK(hipStreamSynchronize(stream));
            HIPCHK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_tile2_prof), sizeof(h)));
            static const char *nm[6] = {"evaluate", "V, g sums", "point blocks", "EO rows of Z", "IO rows of Z", "tail"};
            fprintf(stderr, "[heavy_z prof, us per batch (thread 0) avg over %d batches]", (int)(nb - P.nb_tiled));
            for (int i = 0; i < 6; ++i) fprintf(stderr, " %s=%.2f", nm[i], h[8 + i] * 0.01 / (double)std::max<int64_t>(nb - P.nb_tiled, 1));
            fprintf(stderr, "\n");
            for (int i = 8; i < 16; ++i) h[i] = 0;
            HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_tile2_prof), h, sizeof(h)));
        }
        if ((d.ablate & 32) && use_sig) {            // phase profile of the signature kernel (wave 0 of every tile)
            unsigned long long h[16];
            HIPCHK(hipStreamSynchronize(stream));
            HIPCHK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_tile2_prof), sizeof(h)));
            static const char *nm[8] = {"tile init", "chunk head", "pass 1", "pass 2 eval+write", "pass 2 mfma", "chunk flush", "wait for the tile", "tile flush"};
            fprintf(stderr, "[sig prof, us per tile (wave 0) avg over %d tiles]", (int)ntiles);
            for (int i = 0; i < 8; ++i) fprintf(stderr, " %s=%.2f", nm[i], h[i] * 0.01 / (double)std::max<int64_t>(ntiles, 1));
            fprintf(stderr, "\n");
            memset(h, 0, sizeof(h));
            HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_tile2_prof), h, sizeof(h)));
        } else if ((d.ablate & 32) && use_tile2) {   // phase profile of the wave-specialised tile kernel
            unsigned long long h[16];
            HIPCHK(hipStreamSynchronize(stream));
            HIPCHK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_tile2_prof), sizeof(h)));
            static const char *nm[12] = {"stage(wait freed)", "P1", "pbarrier", "P2", "P3", "drain", "cons wait full", "cons zero+free", "tail/flush", "loop head", "cons product", "cons wait done"};
            fprintf(stderr, "[tile2 prof, us per tile avg over %d tiles]", (int)ntiles);
            for (int i = 0; i < 12; ++i) fprintf(stderr, " %s=%.2f", nm[i], h[i] * 0.01 / (double)std::max<int64_t>(ntiles, 1));
            fprintf(stderr, "\n");
            memset(h, 0, sizeof(h));
            HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_tile2_prof), h, sizeof(h)));
        }
        // red_scal[0] = the build kernels' residual sums + the prior rows' squares, red_scal[1] = owned
        // squared column norms of the point columns
        // (and keeps the copy of zz that the solve works from)
        LAUNCHK(k_build_tail, dim3(grid_zs), dim3(1024), 0, stream, d, zz, partial.p, npart, jn2p.p, gpart.p, gctr.p + 2, red_scal,
                zz != zlin.p ? zlin.p : (double *)nullptr);
    }
    // A linearisation at z that a damping loop has asked for but nobody has needed yet.  levenberg_marquardt.m
    // (:189-194) and levenberg_marquardt_powell.m linearise at every ACCEPTED point before they test for
    // termination; after the last accepted step that Jacobian is only returned (final.weighted.J), never
    // used for a step.  Here it is built when the next solve -- or a caller who wants the gradient, the
    // column norms or J v -- asks for it, and not at all otherwise: one build kernel chain less per solve
    // (C3: 6 -> 5 linearisations for 4 LM iterations).  The iterates are the same, bit for bit.
    bool pend_build = false;
    double pend_lambda = 0;
    int pend_scale = 0;
    void request_build(double lambda, int scale) {
        pend_build = true; pend_lambda = lambda; pend_scale = scale;
        lambda_lin = lambda; scale_lin = scale;
    }
    void ensure_build() {
        if (!pend_build) return;
        build(z.p, pend_lambda, pend_scale, true);
    }
    // the loop is over: the linearisation POINT is z (residuals, sigma0), whether or not its Jacobian was built
    void settle_lin_point() {
        if (!pend_build) return;
        copy(zlin.p, z.p);
        cams_at_lin = false; s_valid = false; have_lin = true;
    }
    // lazy: do not wait for the linearisation's scalars (f_lin, trace_jtj): the next sync() of any kind
    // picks them up.  The damping loops only need them after their first linearisation.
    // Several ranks.  Replicated mode: the envelope of S and the vectors are summed BEFORE k_finish; every rank
    // then holds, finishes and factors the whole system.  Domain mode (mg_subtree): S and the right-hand side
    // stay this rank's share -- complete for its own domain --, only the vectors [g_c | diagU | sums] are
    // summed here (the column scaling needs the complete column norms); k_finish adds the prior / damping /
    // fixed-element terms where this rank owns the column; the top separators are summed inside the
    // factorisation (factor_solve_enqueue).
    void allreduce_vectors() {
        if (!multi()) return;
        do_allreduce(g_c, 2 * P.NS + 8);             // [g_c | diagU | red_scal], contiguous behind g_red
    }
    // trace_only: the linearisation is wanted for trace(J'J) alone (Levenberg-Marquardt's lambda0): the signature
    // kernel stops after the per-point sums (squared column norms), the reduced system is NOT formed -- a third of
    // a build.  The handle has no linearisation afterwards (lambda_lin = NaN: the loop builds the damped system).
    // trace(J'J) at zz by one streaming pass (k_trace_cm): what Levenberg-Marquardt's first linearisation is for.
    // One rank only (several ranks sum it with the vectors of a build).
    void trace_only_pass(const double *zz) {
        stage(0);
        prep_cams(zz);
#define L_TRACE(M, NCXV) LAUNCHK((k_trace_cm<M, NCXV>), dim3((unsigned)n_cm_chunks_all), dim3(256), 0, stream, d, zz, cams.p, cm_pt.p, cm_uv.p, P.uniform_w ? (const double *)nullptr : cm_w.p, cm_chunk_cam.p, cm_chunk_start.p, partial.p)
        if (n_cm_chunks_all > 0) {
            if (tile_ncx == 6) { DISPATCH_MODEL(L_TRACE, 6) } else if (tile_ncx == 14) { DISPATCH_MODEL(L_TRACE, 14) }
            else if (tile_ncx == 15) { DISPATCH_MODEL(L_TRACE, 15) } else { DISPATCH_MODEL(L_TRACE, MAXCOL) }
        }
#undef L_TRACE
        LAUNCHK(k_trace_tail, dim3(grid_zs), dim3(1024), 0, stream, d, gpart.p, gctr.p + 6, (const double *)partial.p, n_cm_chunks_all, hpin, ++mb_seq);
        mb_armed = true; lin_pending = true;
        sync();
        pend_build = false;
        lambda_lin = NAN; have_lin = false; s_valid = false;
        ++n_trace_only;
    }
    void build(const double *zz, double lambda, int scale, bool lazy = false, bool trace_only = false) {
        if (trace_only && !multi() && P.nranks == 1) { trace_only_pass(zz); return; }
        pend_build = false;
        trace_only = trace_only && use_sig && ntiles > 0;
        d.trace_only = trace_only ? 1 : 0;             // an argument of this launch (DevProblem travels by value)
        try { build_enqueue(zz, lambda, scale); } catch (...) { d.trace_only = 0; throw; }
        d.trace_only = 0;
        if (mg_subtree) allreduce_vectors();
        else allreduce_system();
        finish_enqueue(zz, lambda, scale);           // also: trace(J'J) of the camera part and red_scal -> mailbox
        if (mg_subtree && replicate_next && multi()) allreduce_matrix();   // (posterior covariance: the whole system on every rank)
        cams_at_lin = true;                          // (zlin = zz: k_build_tail)
        lin_pending = true;
        if (!lazy) sync();
        lambda_lin = trace_only ? NAN : lambda;
        scale_lin = scale;
        have_lin = !trace_only;
        s_valid = !trace_only;
        if (!trace_only) ++n_lin; else ++n_trace_only;
    }
    bool replicate_next = false;     // domain mode: the next build sums the whole matrix after k_finish (in-place factorisation)
    // sum of the envelope of S (incl. the right-hand-side row) over the ranks, after k_finish
    void allreduce_matrix() {
        if (!pk.p) pk.alloc((size_t)(pk_s_count + 3 * P.NS + 8));
        LAUNCHK(k_pack_envelope, dim3((unsigned)P.NS), dim3(256), 0, stream, S, ldS, (int)P.NS, env_tail0, col_bend.p, col_off.p, pk.p, 1);
        do_allreduce(pk.p, pk_s_count);
        LAUNCHK(k_pack_envelope, dim3((unsigned)P.NS), dim3(256), 0, stream, S, ldS, (int)P.NS, env_tail0, col_bend.p, col_off.p, pk.p, 0);
    }
    void finish_enqueue(const double *zz, double lambda, int scale) {
        // one rank, compact tiles: the factorisation's reset rides along (DfChol::reset_args)
        int *df_ctl = nullptr; unsigned long long *df_q = nullptr; int df_nq = 0;
        const bool ride = use_perm && !chol_in_place && !(mg_subtree && multi()) && dfchol.reset_args(df_ctl, df_q, df_nq);
        LAUNCHK(k_finish, dim3((unsigned)cdiv(P.NS, 256)), dim3(256), 0, stream, d, zz, lambda, scale, S, g_c, g_red, diagU, jn2c.p, dscale.p,
                gpart.p, gctr.p + 3, (const double *)red_scal, scal.p, hpin, mg_subtree && multi() ? (const uint8_t *)z_mine.p : (const uint8_t *)nullptr,
                ride ? info.p : (int *)nullptr, ride ? df_ctl : (int *)nullptr, ride ? df_q : (unsigned long long *)nullptr, ride ? df_nq : 0);
        if (ride) dfchol.reset_done = true;
        if (scale)      // D S D on the envelope
            LAUNCHK(k_envelope_op, dim3((unsigned)P.NS), dim3(256), 0, stream, S, ldS, (int)P.NS, env_tail0, col_bend.p, (const double *)dscale.p);
    }

    // ---- K6: Cholesky of the reduced system; returns 0 or the failing pivot
    int factor_solve_enqueue() {
        // a schedule built for domain sharding publishes the top separators only after the sum over the ranks
        if (mg_subtree && !multi())
            throw UsageError{"this handle is one of several shards: attach a communicator (dbat_hip_comm_init) or an all-reduce "
                             "callback (dbat_hip_set_allreduce) before linearising or solving"};
        stage(1);
        mark(2);
        // Cholesky + both substitutions; q -> rhs.  One persistent dataflow kernel (chol_df.hpp)
        // (the dataflow kernel also writes the step of the unscaled system dc = D q)
        if (use_perm && !chol_in_place && mg_subtree && multi()) {
            // this rank's domain and its share of the top separators; the shares summed over the ranks (one
            // all-reduce of the top tiles); the top separators and the backward substitution on every rank
            dfchol.solve_domain(stream, S, ldS, linv.p, info.p, ldiag.p);
            mark(8);
            do_allreduce(dfchol.top_tiles(), dfchol.top_tiles_count());
            mark(9);
            dfchol.solve_top(stream, ldS, rhs.p, linv.p, info.p, ldiag.p, dscale.p, dz.p);
        } else if (use_perm && !chol_in_place) dfchol.solve(stream, S, ldS, rhs.p, linv.p, info.p, ldiag.p, dscale.p, dz.p);
        else dfchol_ip.solve(stream, S, ldS, rhs.p, linv.p, info.p, ldiag.p, dscale.p, dz.p);
        HIPCHK(hipGetLastError());                   // launches inside the factorisation helpers
        if (env_df_trace) (use_perm && !chol_in_place ? dfchol : dfchol_ip).dump_trace(stream, env_df_trace);
        mark(3);
        // the pivot extremes of the reduced system are taken in k_prior_jv (the tail of the solve)
        ++n_solves;
        return 0;
    }
    // ---- K7: back-substitution; sums {||Jp||^2, r'Jp, ||p||^2}
    void backsub_enqueue() {
        stage(2);
        mark(4);
        // the points of the signature chunks by k_backsub_sig (one wave per chunk), the other batches by
        // k_backsub; partial: [nb batches][2] (the tiled batches' slots stay zero), giants, sig workgroups
        const bool sig_bs = use_sig && P.sg_backsub_ok;
        const int64_t b_first = sig_bs ? P.nb_tiled : 0;
        const int64_t n_sig_wg = sig_bs ? cdiv(sg_nchunks, 4) : 0;
        // (the slots of the tiled batches are not written then, and not summed: see k_sum_partials below)
        if (n_sig_wg > 0) {
#define L_BACKS(M, NCXV) LAUNCHK((k_backsub_sig<M, NCXV>), dim3((unsigned)n_sig_wg), dim3(256), 0, stream, d, zlin.p, cams.p, Vinv.p, gp.p, dz.p, partial.p + 2 * (nb + ngiant), sg_chunk.p, (int)sg_nchunks, sg_gcam.p, sg_uv.p, P.uniform_w ? (const double *)nullptr : sg_w.p)
            if (tile_ncx == 6) { DISPATCH_MODEL(L_BACKS, 6) } else { DISPATCH_MODEL(L_BACKS, 14) }
#undef L_BACKS
        }
        if (nb > b_first) {
#define L_BACK(M, NCXV) LAUNCHK((k_backsub<M, NCXV>), dim3((unsigned)(nb - b_first)), dim3(P.BT), lds_back, stream, d, zlin.p, cams.p, Vinv.p, gp.p, dz.p, partial.p, (int)b_first)
            if (tile_ncx == 6) { DISPATCH_MODEL(L_BACK, 6) } else if (tile_ncx == 14) { DISPATCH_MODEL(L_BACK, 14) } else if (tile_ncx == 15) { DISPATCH_MODEL(L_BACK, 15) } else { DISPATCH_MODEL(L_BACK, MAXCOL) }
#undef L_BACK
        }
        if (ngiant > 0) {
#define L_BACKG(M, IO) LAUNCHK((k_backsub_giant<M, IO>), dim3((unsigned)ngiant), dim3(giant_threads), 0, stream, d, zlin.p, cams.p, Vinv.p, gp.p, dz.p, partial.p + 2 * nb)
            if (P.with_io) { DISPATCH_MODEL(L_BACKG, true) } else { DISPATCH_MODEL(L_BACKG, false) }
#undef L_BACKG
        }
        mark(5);
        // one launch: the sums of the back-substitution kernels, the prior rows' share, g'p, p'p, the pivot
        // extremes of the reduced system; on one rank straight into the pinned mailbox
        LAUNCHK(k_prior_jv, dim3(grid_zs), dim3(1024), 0, stream, d, zlin.p, dz.p, g_c, gp.p, gpart.p, gctr.p + 4,
                (const double *)(partial.p + 2 * b_first), nb - b_first + ngiant + n_sig_wg, (const double *)ldiag.p, pivmm.p,
                (const int *)info.p, scal.p, multi() ? (double *)nullptr : hpin, ++mb_seq,
                mg_subtree && multi() ? (const uint8_t *)piv_have.p : (const uint8_t *)nullptr);
    }
    // solve at the current linearisation: p in dz.  Returns true if the
    // factorisation failed outright (non-positive pivot / non-finite step);
    // near_singular mimics MATLAB's 'nearlySingularMatrix' warning through
    // CHOLMOD's estimate rcond = (min diag(L) / max diag(L))^2 < eps over the
    // pivots of the point blocks and of the reduced system.
    bool near_singular = false;
    const bool pivot_stats = env_on("DBAT_HIP_PIVOT_STATS");
    double rcond_est = 0.0;      // (min pivot / max pivot)^2 of the last solve; 0 after a failed factorisation
    int chol_info = 0;           // k_chol_df's info of the last solve: > 0 first non-positive pivot, -1 dataflow abort
    double piv_last[4] = {0, 0, 0, 0};   // {min, max} pivot of the point blocks, {min, max} of the reduced system
    bool solve(double &JpJp, double &rJp, double &pp) {
        ensure_build();
        if (!s_valid) build(zlin.p, lambda_lin, scale_lin);   // the factorisation overwrote S
        s_valid = false;
        factor_solve_enqueue();
        if (!cams_at_lin) { prep_cams(zlin.p); cams_at_lin = true; }     // something else used the camera records since build()
        backsub_enqueue();
        int hinfo = 0;
        unsigned long long hmm[4];
        // one all-reduce per solve: the 8 scalar sums (slot 7: a failed factorisation on any rank) and, behind
        // them, two {min,max} slots per rank with this rank's pivots of the point blocks and of the part of
        // the reduced system it factored (the extremes travel through the sum)
        const int nsl = multi() ? 4 * P.nranks : 0;
        std::vector<double> h((size_t)8 + nsl);
        if (nsl) {
            HIPCHK(hipMemcpyAsync(hpin + 40, pivmm.p, sizeof(hmm), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipMemcpyAsync(hpin + 44, info.p, sizeof(hinfo), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipMemsetAsync(scal.p + 8, 0, (size_t)nsl * sizeof(double), stream));
            HIPCHK(hipMemcpyAsync(scal.p + 8 + 4 * P.rank, pivmm.p, 4 * sizeof(double), hipMemcpyDeviceToDevice, stream));
            do_allreduce(scal.p, 8 + nsl);
            read_scal(h.data(), 8 + nsl);
        } else {             // k_prior_jv left everything in the mailbox
            mb_armed = true;
            sync();
            memcpy(h.data(), hpin, 8 * sizeof(double));
        }
        memcpy(hmm, hpin + 40, sizeof(hmm));
        memcpy(&hinfo, hpin + 44, sizeof(hinfo));
        JpJp = h[0] + h[4]; rJp = h[1] + h[5]; pp = h[6];
        double mm[4];
        memcpy(mm, hmm, sizeof(mm));
        double pmin = std::min(mm[0], mm[2]), pmax = std::max(mm[1], mm[3]);
        for (int r = 0; r < nsl / 2; ++r) { pmin = std::min(pmin, h[8 + 2 * r]); pmax = std::max(pmax, h[8 + 2 * r + 1]); }
        const bool failed = hinfo != 0 || (nsl && h[7] > 0) || !std::isfinite(pp) || !std::isfinite(JpJp);
        // a failed factorisation that ran in place may have left non-finite values in tile parts outside the
        // column envelope (0 x NaN), which the envelope-only clearing of the next build would not reach
        if (failed && !(use_perm && !chol_in_place)) s_dense_dirty = true;
        const double ratio = pmax > 0 ? pmin / pmax : 0.0;
        near_singular = failed || !(ratio * ratio >= 2.220446049250313e-16);
        rcond_est = failed ? 0.0 : ratio * ratio;
        chol_info = hinfo;
        for (int q = 0; q < 4; ++q) piv_last[q] = mm[q];
        if (pivot_stats)
            fprintf(stderr, "[solve] pivots of the point blocks %.3e ... %.3e, of the reduced system %.3e ... %.3e, info %d, rcond estimate %.3e\n",
                    mm[0], mm[1], mm[2], mm[3], hinfo, rcond_est);
        return failed;
    }
    // ---- posterior covariance blocks at z (bundle_cov.m): s0^2 * blocks of inv(J'J)
    std::vector<double> cop_tmp;
    void posterior_cov(double s0, double *hCEO, double *hCIO, double *hCOP, double *hSinv) {
        // inv(S): on one rank, from the compact nested-dissection factor, only the entries the blocks need (selected inversion,
        // chol_df.hpp) -- no dense inverse (C4: 7.2 GB and 9 TF of rocsolver_dpotri).  The dense inverse remains for several
        // ranks, for problems without the compact factor (shared EO blocks, DBAT_HIP_ND_OFF) and for a caller who asks for it.
        const bool selinv = use_perm && !multi() && !hSinv && dfchol.selinv_supported() && !env_on("DBAT_HIP_COV_DENSE");
        replicate_next = !selinv;                    // (domain sharding: the whole system on every rank for this one)
        try { build(z.p, 0.0, 0); } catch (...) { replicate_next = false; throw; }   // unscaled, undamped reduced system + V^-1 per point
        replicate_next = false;
        s_valid = false;
        SinvView SV;
        int hinfo = 0;
        if (selinv) {
            factor_solve_enqueue();                  // the compact factor (S itself is only read)
            HIPCHK(hipMemcpyAsync(&hinfo, info.p, sizeof(hinfo), hipMemcpyDeviceToHost, stream));
            sync();
            if (hinfo != 0) throw DeviceError{"posterior covariance: the reduced normal matrix is not positive definite"};
            if (!dfchol.selected_inverse(stream, linv.p)) throw DeviceError{"out of device memory (selected inverse)"};
            HIPCHK(hipGetLastError());
            SV.tiles = dfchol.d_ztiles; SV.toff = dfchol.d_toff; SV.perm = dfchol.d_perm; SV.nT = dfchol.nT;
        } else {
            chol_in_place = true;
            // the in-place factorisation stores whole 64 x 64 tiles, which straddle the column envelope that the
            // next build clears: from here on S must be cleared densely -- also when the factorisation fails
            s_dense_dirty = true;
            factor_solve_enqueue();                  // L in the lower triangle of S
            chol_in_place = false;
            HIPCHK(hipMemcpyAsync(&hinfo, info.p, sizeof(hinfo), hipMemcpyDeviceToHost, stream));
            sync();
            if (hinfo != 0) throw DeviceError{"posterior covariance: the reduced normal matrix is not positive definite"};
            if (!blas) {     // (created on first use: rocblas_create_handle costs 0.1 s, and only the dense inverse needs it)
                if (rocblas_create_handle(&blas) != rocblas_status_success) throw DeviceError{"rocblas_create_handle failed"};
                rocblas_set_stream(blas, stream);
            }
            if (rocsolver_dpotri(blas, rocblas_fill_lower, (rocblas_int)P.NS, S, (rocblas_int)ldS, info.p) != rocblas_status_success)
                throw DeviceError{"rocsolver_dpotri failed"};
            SV.dense = S; SV.ld = ldS;
        }
        have_lin = false;
        const double s02 = s0 * s0;
        prep_cams(z.p);
        DevBuf<double> dCEO, dCIO, dCOP;
        if (hCEO) dCEO.alloc((size_t)36 * P.nc);
        if (hCIO && P.nIOu > 0) dCIO.alloc((size_t)P.nIOu * P.nIOu);
        if (hCEO || (hCIO && P.nIOu > 0)) {
            const int64_t tot = 36 * (int64_t)P.nc + (int64_t)P.nIOu * P.nIOu;
            LAUNCHK(k_cov_cam, dim3((unsigned)cdiv(tot, 256)), dim3(256), 0, stream, d, SV, s02, dCEO.p, dCIO.p);
        }
        if (hCOP) {
            dCOP.alloc((size_t)9 * P.np);
            HIPCHK(hipMemsetAsync(dCOP.p, 0, (size_t)9 * P.np * sizeof(double), stream));
            const size_t lds_cov = ((size_t)P.BT * P.ncolmax * 3 + (size_t)P.BT * 6) * sizeof(double);
            if (nb > 0) {
#define L_COV(M, IO) LAUNCHK((k_cov_points<M, IO>), dim3((unsigned)nb), dim3(P.BT), lds_cov, stream, d, z.p, cams.p, Vinv.p, SV, s02, dCOP.p)
                if (P.with_io) { DISPATCH_MODEL(L_COV, true) } else { DISPATCH_MODEL(L_COV, false) }
#undef L_COV
            }
            if (ngiant > 0) {
#define L_COVG(M, IO) LAUNCHK((k_cov_giant<M, IO>), dim3((unsigned)ngiant), dim3(giant_threads), 0, stream, d, z.p, cams.p, Vinv.p, SV, s02, dCOP.p)
                if (P.with_io) { DISPATCH_MODEL(L_COVG, true) } else { DISPATCH_MODEL(L_COVG, false) }
#undef L_COVG
            }
            if (multi()) do_allreduce(dCOP.p, (int64_t)9 * P.np);   // blocks of the other shards' points
            cop_tmp.resize((size_t)9 * P.np);
            HIPCHK(hipMemcpyAsync(cop_tmp.data(), dCOP.p, (size_t)9 * P.np * sizeof(double), hipMemcpyDeviceToHost, stream));
        }
        if (hCEO) HIPCHK(hipMemcpyAsync(hCEO, dCEO.p, (size_t)36 * P.nc * sizeof(double), hipMemcpyDeviceToHost, stream));
        if (hCIO && P.nIOu > 0) HIPCHK(hipMemcpyAsync(hCIO, dCIO.p, (size_t)P.nIOu * P.nIOu * sizeof(double), hipMemcpyDeviceToHost, stream));
        if (hSinv)                                   // full inv(S) (lower triangle valid), NS x NS column-major, not scaled by s0^2
            HIPCHK(hipMemcpy2DAsync(hSinv, (size_t)P.NS * sizeof(double), S, (size_t)ldS * sizeof(double),
                                    (size_t)P.NS * sizeof(double), (size_t)P.NS, hipMemcpyDeviceToHost, stream));
        sync();
        if (hCOP)                                    // device blocks are in processing order
            for (int64_t p = 0; p < P.np; ++p)
                std::copy(cop_tmp.begin() + 9 * (int64_t)P.pt_rank[p], cop_tmp.begin() + 9 * (int64_t)P.pt_rank[p] + 9, hCOP + 9 * p);
    }
    // ||J v||^2 and r'Jv at the linearisation point, ||v||^2 over owned entries
    void jtimes(const double *v, double &JvJv, double &rJv, double &vv) {
        if (!cams_at_lin) { prep_cams(zlin.p); cams_at_lin = true; }
#define L_JT(M, NCXV) LAUNCHK((k_jtimes<M, NCXV>), dim3(grid_obs), dim3(256), 0, stream, d, zlin.p, cams.p, v, partial.p)
        if (tile_ncx == 6) { DISPATCH_MODEL(L_JT, 6) } else if (tile_ncx == 14) { DISPATCH_MODEL(L_JT, 14) } else if (tile_ncx == 15) { DISPATCH_MODEL(L_JT, 15) } else { DISPATCH_MODEL(L_JT, MAXCOL) }
#undef L_JT
        LAUNCHK(k_prior_jv, dim3(grid_zs), dim3(1024), 0, stream, d, zlin.p, v, g_c, gp.p, gpart.p, gctr.p + 4,
                (const double *)partial.p, (int64_t)grid_obs, (const double *)nullptr, pivmm.p, (const int *)info.p, scal.p, (double *)nullptr, 0ull);
        do_allreduce(scal.p, 8);
        double h[8];
        read_scal(h, 8);
        JvJv = h[0] + h[4]; rJv = h[1] + h[5]; vv = h[6];
    }
    double dot_owned(const double *a, const double *b) {
        LAUNCHK(k_dot, dim3(grid_z), dim3(256), 0, stream, P.NZ, z_mine.p, a, b, partial.p);
        LAUNCHK((k_sum_partials<1>), dim3(1), dim3(1024), 0, stream, partial.p, (int64_t)grid_z, scal.p, 0);
        do_allreduce(scal.p, 1);
        double s;
        read_scal(&s, 1);
        return s;
    }
    void axpby(double a, const double *x, double b, const double *y2, double *y) {
        LAUNCHK(k_axpby, dim3((unsigned)cdiv(P.NZ, 256)), dim3(256), 0, stream, P.NZ, a, x, b, y2, y);
    }
    void gradient(double *g) {
        LAUNCHK(k_gradient, dim3((unsigned)cdiv(P.NZ, 256)), dim3(256), 0, stream, d, g_c, gp.p, g);
    }
    void colnorm2(double *out) {
        LAUNCHK(k_jn2, dim3((unsigned)cdiv(P.NZ, 256)), dim3(256), 0, stream, d, jn2c.p, jn2p.p, out);
    }
    void copy(double *dst, const double *src) { HIPCHK(hipMemcpyAsync(dst, src, P.NZ * 8, hipMemcpyDeviceToDevice, stream)); }
    void accept_trial() { std::swap(z.p, zt.p); }    // z <- the trial point: the buffers change roles (nothing keeps their addresses)
};

// ============================================================================
// Host damping loops (F12).  Device vectors: z (current), zt (trial), dz (step).
// ============================================================================

struct LoopOut {
    std::vector<double> res, damp, aux;
    int code = 0, iters = 0;
    double f_final = 0;
};

static void export_residuals(Core &c, const double *zdev, double *r_unw, double *r_wgt, double *f);
static void jtimes_rows(Core &c, const double *v_dev, double *Jv);

// J*p as a host vector for a caller-supplied termFun: taken right after the solve, at the linearisation that produced p
// (levenberg_marquardt.m:217 tests the OLD Jp against the NEW r)
static void keep_Jp(Core &c, const dbat_hip_options &o, const double *p_dev, std::vector<double> &Jp) {
    if (!o.term_fun) return;
    Jp.resize((size_t)c.P.m);
    jtimes_rows(c, p_dev, Jp.data());
}
static bool term_fun(Core &c, const dbat_hip_options &o, double JpJp, double f, const std::vector<double> &Jp) {
    if (o.term_fun) {                                // termFun(Jp, r), r = the weighted residual at the current point
        std::vector<double> r((size_t)c.P.m);
        export_residuals(c, c.z.p, nullptr, r.data(), nullptr);
        return o.term_fun(o.term_user, Jp.data(), r.data(), c.P.m) != 0;
    }
    // bundle.m:186-192: relative  norm(Jp)<=tol*norm(r) ; absolute norm(r)<=tol
    const double nr = std::sqrt(2 * f);
    if (o.abs_term) return nr <= o.conv_tol;
    return std::sqrt(JpJp) <= o.conv_tol * nr;
}
// vetoFun(t) at the trial point c.zt
static bool vetoed(Core &c, const dbat_hip_options &o) {
    if (!o.veto_fun) return false;
    std::vector<double> x((size_t)c.P.n);
    c.z_to_x(c.zt.p, x.data());
    return o.veto_fun(o.veto_user, x.data(), c.P.n) != 0;
}

// the line the lsa solver prints with 'trace', handed to the caller while the loop runs (dbat_hip_trace_fn)
static inline void trace_line(const dbat_hip_options &o, int n, double res, double damp, int step = -1, double rho = NAN) {
    if (o.trace_fun) o.trace_fun(o.trace_user, o.damping, n, res, damp, step, rho);
}

static void push_trace(Core &c, const dbat_hip_options &o, LoopOut &) {
    if (!o.store_trace) return;
    c.trace_put(c.trace_n, c.z.p);
}

// Objective values: every f = 0.5*r'r the loops compare comes from the same
// kernel and reduction order (Core::eval_f), so the same point always gives
// the same bits -- as it does in the reference, where one resFun computes all
// of them.  (k_build also sums r'r, in a different order; that value is only
// reported through dbat_hip_linearize_solve.)

// lsa/gauss_newton_armijo.m:86-245, linesearch :249-290
static void loop_gna(Core &c, const dbat_hip_options &o, LoopOut &out) {
    int n = 0;
    std::vector<double> Jp_host;
    push_trace(c, o, out);
    double f = c.eval_f(c.z.p, nullptr, nullptr);
    while (true) {
        c.build(c.z.p, 0.0, 1, true);                             // :112-116, :166-170 (scalars: with the solve's sync)
        out.res.push_back(std::sqrt(2 * f));
        trace_line(o, n, out.res.back(), out.damp.empty() ? NAN : out.damp.back());   // :119-128
        if (n == 0 && !c.P.rank_ok) { out.code = -4; break; }     // :132-142
        double JpJp, rJp, pp;
        const bool failed = c.solve(JpJp, rJp, pp);               // :172-174
        if (failed || (o.singular_test && c.near_singular)) { out.code = -2; break; }  // :176-184
        keep_Jp(c, o, c.dz.p, Jp_host);
        if (term_fun(c, o, JpJp, f, Jp_host)) break;              // :191
        ++n;
        // linesearch
        const double f0 = f, fp0 = rJp;
        double alpha = 1.0;
        bool found = false;
        while (alpha >= o.alpha_min) {
            const double ft = c.eval_f_step(c.z.p, alpha, c.dz.p, c.zt.p);
            if (ft < f0 + o.mu * alpha * fp0 && !vetoed(c, o)) { found = true; f = ft; break; }     // :265-281
            alpha /= 2;
        }
        if (!found) alpha = 0.0;
        else c.accept_trial();
        out.damp.push_back(alpha);
        push_trace(c, o, out);
        if (alpha == 0.0) { out.code = -3; out.res.push_back(out.res.back()); break; }
        if (n > o.max_iter) { out.code = -1; out.res.push_back(std::sqrt(2 * f)); break; }
    }
    out.iters = n;
    out.f_final = f;
}

// lsa/gauss_markov.m:52-129 with the documented semantics
static void loop_gm(Core &c, const dbat_hip_options &o, LoopOut &out) {
    int n = 0;
    std::vector<double> Jp_host;
    push_trace(c, o, out);
    double f = 0;
    while (true) {
        f = c.eval_f(c.z.p, nullptr, nullptr);
        c.build(c.z.p, 0.0, 0, true);
        out.res.push_back(std::sqrt(2 * f));
        trace_line(o, n, out.res.back(), NAN);                    // :74-76
        double JpJp, rJp, pp;
        const bool failed = c.solve(JpJp, rJp, pp);               // :79
        if (failed || (o.singular_test && c.near_singular)) { out.code = -2; break; }
        keep_Jp(c, o, c.dz.p, Jp_host);
        if (term_fun(c, o, JpJp, f, Jp_host)) break;              // :94
        ++n;
        c.axpby(1.0, c.z.p, 1.0, c.dz.p, c.z.p);
        push_trace(c, o, out);
        if (n > o.max_iter) { out.code = -1; break; }
    }
    out.iters = n;
    out.f_final = f;
}

// lambda0 (and lambdaMin) given as fractions of trace(J'J)/n with lambda0 >= lambdaMin > 0: the undamped system of
// the first linearisation is never solved, only its trace is used
static bool lambda0_needs_trace(const dbat_hip_options &o) {
    const double l0 = std::fabs(o.lambda0), lm = std::fabs(o.lambda_min);
    return o.lambda0 < 0 && o.lambda_min < 0 && l0 >= lm && l0 > 0;
}

// lsa/levenberg_marquardt.m:52-250
static void loop_lm(Core &c, const dbat_hip_options &o, LoopOut &out) {
    int n = 0;
    double f = c.eval_f(c.z.p, nullptr, nullptr);
    // :76-82.  J at x0 is needed for trace(J'J) here; the first step solves (J'J + lambda0 I), built below
    c.build(c.z.p, 0.0, 0, false, lambda0_needs_trace(o));
    const double nx = (double)c.P.n;
    double lambda0 = o.lambda0, lambdaMin = o.lambda_min;
    if (lambda0 < 0) lambda0 = std::fabs(lambda0) * c.trace_jtj / nx;       // :88-90
    if (lambdaMin < 0) lambdaMin = std::fabs(lambdaMin) * c.trace_jtj / nx; // :93-95
    double lambda = lambda0;
    if (lambda < lambdaMin) lambda = 0;
    out.damp.push_back(lambda);
    double prevLambda = NAN;
    double JpJp = 0, rJp = 0, pp = 0;
    std::vector<double> Jp_host;
    while (true) {
        while (n <= o.max_iter) {
            if (c.lambda_lin != lambda) c.build(c.z.p, lambda, 0, true);      // (JTJ+lambda*I), :119
            const bool failed = c.solve(JpJp, rJp, pp);
            if (!failed) keep_Jp(c, o, c.dz.p, Jp_host);                      // :162 (Jp = J*p, before the trial point)
            out.res.push_back(std::sqrt(2 * f));
            if (n == 0 && !c.P.rank_ok) { out.code = -4; break; }             // :126-135
            if (failed) { out.code = -2; break; }   // the reference has no test here; MATLAB would continue on Inf/NaN
            out.damp.push_back(lambda);
            trace_line(o, n, out.res.back(), n == 0 ? NAN : lambda);              // :138-147
            if (o.store_trace) c.trace_put(n, c.z.p);
            ++n;
            const double fNew = c.eval_f_step(c.z.p, 1.0, c.dz.p, c.zt.p);                 // t = x+p
            if (fNew < f && !vetoed(c, o)) {                                  // :170-177
                c.accept_trial();
                lambda = lambda / 10;
                if (lambda < lambdaMin) lambda = 0;
                c.request_build(lambda, 0);                                   // :189-194 (built when the next solve needs it)
                f = fNew;
                break;
            } else {
                if (lambda == 0) lambda = lambdaMin; else lambda = lambda * 10;
            }
        }
        if (out.code != 0) break;
        if (prevLambda == 0 && term_fun(c, o, JpJp, f, Jp_host)) break;       // :217 (old Jp, new r)
        prevLambda = lambda;
        if (n > o.max_iter) { out.code = -1; break; }
    }
    if (o.store_trace) c.trace_put(n, c.z.p);
    out.res.push_back(std::sqrt(2 * f));                                      // :242
    out.iters = n;
    out.f_final = f;
    c.settle_lin_point();
}

// lsa/levenberg_marquardt_powell.m:60-230, dogleg :232-335
static void loop_lmp(Core &c, const dbat_hip_options &o, double delta0, LoopOut &out) {
    int n = 0;
    double delta = delta0;
    std::vector<double> rhos, steps;
    if (o.store_trace) c.trace_put(0, c.z.p);
    double f = c.eval_f(c.z.p, nullptr, nullptr);
    c.build(c.z.p, 0.0, 1);
    bool have_gn = false;
    std::vector<double> Jp_host;
    double gnJpJp = 0, gnrJp = 0, gnpp = 0;
    double *pGN = c.vtmp.p, *g = c.vtmp2.p;
    while (true) {
        out.res.push_back(std::sqrt(2 * f));
        if (n == 0 && !c.P.rank_ok) { out.code = -4; break; }
        // ---- dogleg(r,J,delta)
        if (!have_gn) {
            const bool failed = c.solve(gnJpJp, gnrJp, gnpp);     // :267-279 (scaled GN)
            if (failed) { out.code = -2; break; }   // no singular test in the reference's LMP
            c.copy(pGN, c.dz.p);
            have_gn = true;
        }
        int step;
        double JpJp, rJp;
        const double npGN = std::sqrt(gnpp);
        if (npGN <= delta) {                                       // :281-286
            step = 0; c.copy(c.dz.p, pGN); JpJp = gnJpJp; rJp = gnrJp;
        } else {
            c.gradient(g);                                         // g = J'r
            double gJJg, rJg, gg;
            c.jtimes(g, gJJg, rJg, gg);                            // :304-311
            const double lambdaStar = gg / gJJg;
            const double nCP = lambdaStar * std::sqrt(gg);
            if (nCP > delta) {                                     // :313-318
                step = 2;
                c.axpby(-delta / std::sqrt(gg), g, 0.0, g, c.dz.p);
            } else {                                               // :324-335
                step = 1;
                // CP = -lambdaStar*g ; A=|CP-pGN|^2, B=2 CP.(pGN-CP), C=|CP|^2-delta^2
                const double cp_pgn = -lambdaStar * c.dot_owned(g, pGN);
                const double cp2 = nCP * nCP;
                const double A = cp2 - 2 * cp_pgn + gnpp;
                const double B = 2 * (cp_pgn - cp2);
                const double Cc = cp2 - delta * delta;
                const double k = (-B + std::sqrt(B * B - 4 * A * Cc)) / (2 * A);
                // p = CP + k (pGN - CP) = (1-k)(-lambdaStar) g + k pGN
                c.axpby(-(1 - k) * lambdaStar, g, k, pGN, c.dz.p);
            }
            double vv;
            c.jtimes(c.dz.p, JpJp, rJp, vv);
        }
        out.damp.push_back(delta);
        steps.push_back(step);
        if (step == 0) {
            keep_Jp(c, o, c.dz.p, Jp_host);
            if (term_fun(c, o, gnJpJp, f, Jp_host)) break;         // :134-140
        }
        const double ft = c.eval_f_step(c.z.p, 1.0, c.dz.p, c.zt.p);
        const bool veto = vetoed(c, o);                            // :146-150
        const double predicted = -rJp - 0.5 * JpJp;                // :153
        const double actual = f - ft;
        const double rho = actual / predicted;
        rhos.push_back(rho);
        trace_line(o, n, out.res.back(), delta, step, rho);        // :160-164
        if (veto || rho <= o.rho_bad) {                            // :166-179
            delta = delta / 2;
            if (delta > npGN) delta = delta / std::exp2(std::ceil(std::log2(delta / npGN)));
        } else {
            c.accept_trial();
            c.request_build(0.0, 1);
            f = ft;
            have_gn = false;
            if (rho >= o.rho_good) delta = delta * 2;
        }
        if (o.store_trace) c.trace_put(n, c.z.p);
        ++n;
        if (n > o.max_iter) { out.code = -1; break; }
    }
    if (o.store_trace) {
        c.trace_put(n, c.z.p);
        c.trace_truncate(std::max(n, 1));                          // :229 trims to 1:n
    }
    out.aux = rhos;
    out.aux.resize((size_t)o.max_iter + 2, NAN);
    out.aux.insert(out.aux.end(), steps.begin(), steps.end());
    out.iters = n;
    out.f_final = f;
    c.settle_lin_point();
}

}  // namespace dbat

// ============================================================================
// C ABI
// ============================================================================
using namespace dbat;

struct dbat_hip_handle {
    std::unique_ptr<Core> core;
};

#define API_TRY try {
#define API_CATCH                                                                       \
    }                                                                                   \
    catch (const DeviceError &e) { g_err = e.msg; return DBAT_HIP_EDEVICE; }            \
    catch (const UsageError &e) { g_err = e.msg; return DBAT_HIP_EINVAL; }              \
    catch (const std::bad_alloc &) { g_err = "out of host memory"; return DBAT_HIP_ENOMEM; } \
    catch (const std::exception &e) { g_err = e.what(); return DBAT_HIP_EINVAL; }

extern "C" {

const char *dbat_hip_last_error(void) { return g_err.c_str(); }
int dbat_hip_abi_version(void) { return DBAT_HIP_ABI_VERSION; }

int dbat_hip_default_options(int32_t damping, dbat_hip_options *opt) {
    if (!opt || damping < 0 || damping > 3) { g_err = "bad damping"; return DBAT_HIP_EINVAL; }
    opt->term_fun = nullptr; opt->term_user = nullptr; opt->veto_fun = nullptr; opt->veto_user = nullptr;
    opt->trace_fun = nullptr; opt->trace_user = nullptr;
    opt->damping = damping; opt->max_iter = 20; opt->conv_tol = 1e-6; opt->abs_term = 0;
    opt->singular_test = 1; opt->store_trace = 1; opt->mu = 0.1; opt->alpha_min = 1e-9;
    opt->lambda0 = -1e-10; opt->lambda_min = -1e-10; opt->rho_bad = 0.25; opt->rho_good = 0.75;
    opt->delta0 = -1.0;
    return DBAT_HIP_OK;
}

int dbat_hip_plan(const dbat_hip_problem *prob, int64_t *n_params, int64_t *n_residuals,
                  int64_t *n_io, int64_t *n_eo, int64_t *n_op, int64_t *shard_pt_lo, int64_t *shard_pt_hi) {
    API_TRY
    if (!prob) { g_err = "null problem"; return DBAT_HIP_EINVAL; }
    Plan P;
    if (!build_plan(*prob, P, env_on("DBAT_HIP_PLAN_STATS"))) { g_err = P.err; return DBAT_HIP_EINVAL; }
    if (n_params) *n_params = P.n;
    if (n_residuals) *n_residuals = P.m;
    if (n_io) *n_io = P.nIO;
    if (n_eo) *n_eo = P.nEO;
    if (n_op) *n_op = P.nOP;
    if (shard_pt_lo) *shard_pt_lo = P.pt_lo;
    if (shard_pt_hi) *shard_pt_hi = P.pt_hi;
    return DBAT_HIP_OK;
    API_CATCH
}

int dbat_hip_plan_structural_rank_ok(const dbat_hip_problem *prob, int32_t *ok) {
    API_TRY
    if (!prob || !ok) { g_err = "null argument"; return DBAT_HIP_EINVAL; }
    Plan P;
    if (!build_plan(*prob, P, false)) { g_err = P.err; return DBAT_HIP_EINVAL; }
    *ok = P.rank_ok ? 1 : 0;
    return DBAT_HIP_OK;
    API_CATCH
}

int dbat_hip_plan_point_owner(const dbat_hip_problem *prob, int32_t *owner) {
    API_TRY
    if (!prob || !owner) { g_err = "null argument"; return DBAT_HIP_EINVAL; }
    const int nr = std::max(1, prob->shard_count);
    for (int r = 0; r < nr; ++r) {
        dbat_hip_problem pb = *prob;
        pb.shard_rank = r;
        Plan P;
        if (!build_plan(pb, P, false)) { g_err = P.err; return DBAT_HIP_EINVAL; }
        for (int64_t i = P.pt_lo; i < P.pt_hi; ++i) owner[P.porder[i]] = r;
    }
    return DBAT_HIP_OK;
    API_CATCH
}

int dbat_hip_plan_domain_map(const dbat_hip_problem *prob, int32_t *cam_owner, int32_t *subtree) {
    API_TRY
    if (!prob || !cam_owner) { g_err = "null argument"; return DBAT_HIP_EINVAL; }
    Plan P;
    if (!build_plan(*prob, P, false)) { g_err = P.err; return DBAT_HIP_EINVAL; }
    for (int c = 0; c < P.nc; ++c) cam_owner[c] = P.mg_subtree ? P.nd.cam_owner[c] : -1;
    if (subtree) *subtree = P.mg_subtree ? 1 : 0;
    return DBAT_HIP_OK;
    API_CATCH
}

int dbat_hip_plan_layout_stats(const dbat_hip_problem *prob, int64_t *st) {
    API_TRY
    if (!prob || !st) { g_err = "null argument"; return DBAT_HIP_EINVAL; }
    Plan P;
    if (!build_plan(*prob, P, true)) { g_err = P.err; return DBAT_HIP_EINVAL; }
    for (int i = 0; i < 16; ++i) st[i] = 0;
    st[0] = P.CMAX && P.nb_tiled > 0 ? (int64_t)P.tile_batch.size() - 1 : 0;
    st[1] = (int64_t)P.batch_start.size() - 1; st[2] = P.nb_tiled;
    st[3] = P.sg_ngroups; st[4] = P.sg_npoints; st[5] = (int64_t)P.sg_chunk.size() / 8;
    for (size_t q = 0; q < P.sg_chunk.size() / 8; ++q) {
        const int npts = P.sg_chunk[8 * q + 1], k = P.sg_chunk[8 * q + 2];
        st[6 + (npts > 32 ? 3 : (npts > 16 ? 2 : (npts > 8 ? 1 : 0)))]++;
        if (npts > std::min(6, 64 / std::max(k, 1))) st[10]++;      // SIG_PPR points per round of pass 2
    }
    st[11] = P.sg_kmax; st[12] = P.sg_rows_max; st[13] = P.sg_ok ? 1 : 0; st[14] = P.sg_backsub_ok ? 1 : 0; st[15] = P.hv_ok ? P.hv_ntasks : 0;
    return DBAT_HIP_OK;
    API_CATCH
}

int dbat_hip_plan_serialize(const dbat_hip_problem *prob, double *x0) {
    API_TRY
    if (!prob || !x0) { g_err = "null argument"; return DBAT_HIP_EINVAL; }
    Plan P;
    if (!build_plan(*prob, P, false)) { g_err = P.err; return DBAT_HIP_EINVAL; }
    for (int64_t i = 0; i < P.n; ++i) x0[i] = P.z0[P.x2z[i]];
    return DBAT_HIP_OK;
    API_CATCH
}

int dbat_hip_create(const dbat_hip_problem *prob, dbat_hip_handle **out) {
    API_TRY
    if (!prob || !out) { g_err = "null argument"; return DBAT_HIP_EINVAL; }
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        g_err = "no HIP device: the dbat_hip core has no CPU path";
        return DBAT_HIP_EDEVICE;
    }
    auto h = std::make_unique<dbat_hip_handle>();
    h->core = std::make_unique<Core>();
    const bool create_clock = env_int("DBAT_HIP_PLAN_STATS", 0) >= 2;
    const auto t_create0 = std::chrono::steady_clock::now();
    if (!build_plan(*prob, h->core->P, true)) {
        g_err = h->core->P.err;
        return g_err.find("not supported") != std::string::npos || g_err.find("more observations") != std::string::npos
                   ? DBAT_HIP_EUNSUPPORTED : DBAT_HIP_EINVAL;
    }
    {
        if (prob->device < 0 || prob->device >= ndev) { g_err = "bad device index"; return DBAT_HIP_EINVAL; }
        DeviceGuard dev_guard(prob->device);
        const auto t_create1 = std::chrono::steady_clock::now();
        h->core->init(*prob);
        if (create_clock)
            fprintf(stderr, "[create clock] host plan %.1f ms, device set-up (uploads, schedule of the factorisation, image side) %.1f ms\n",
                    std::chrono::duration<double, std::milli>(t_create1 - t_create0).count(),
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_create1).count());
    }
    *out = h.release();
    return DBAT_HIP_OK;
    API_CATCH
}

void dbat_hip_destroy(dbat_hip_handle *h) { delete h; }

int dbat_hip_structure_key(const dbat_hip_problem *prob, uint64_t *key) {
    API_TRY
    if (!prob || !key) { g_err = "null argument"; return DBAT_HIP_EINVAL; }
    if (prob->abi_version != DBAT_HIP_ABI_VERSION) { g_err = "ABI version mismatch"; return DBAT_HIP_EINVAL; }
    structure_key(*prob, key);
    return DBAT_HIP_OK;
    API_CATCH
}

int dbat_hip_handle_key(const dbat_hip_handle *h, uint64_t *key) {
    if (!h || !key) { g_err = "null argument"; return DBAT_HIP_EINVAL; }
    key[0] = h->core->P.key[0]; key[1] = h->core->P.key[1];
    return DBAT_HIP_OK;
}

int dbat_hip_set_values(dbat_hip_handle *h, const dbat_hip_problem *prob) {
    API_TRY
    if (!h || !prob) { g_err = "null argument"; return DBAT_HIP_EINVAL; }
    Core &c = *h->core;
    uint64_t key[2];
    if (prob->abi_version != DBAT_HIP_ABI_VERSION) { g_err = "ABI version mismatch"; return DBAT_HIP_EINVAL; }
    structure_key(*prob, key);
    if (key[0] != c.P.key[0] || key[1] != c.P.key[1]) {
        g_err = "dbat_hip_set_values: the problem's structure (sizes, visibility, image observations, masks, blocks, prior pattern, "
                "shard, device, DBAT_HIP_* environment) differs from the one this handle was created for: create a new handle";
        return DBAT_HIP_EINVAL;
    }
    DeviceGuard dev_guard(c.device);
    const size_t nio = (size_t)c.P.nIOrows * c.P.nc;
    const bool io_changed = memcmp(c.P.io_fixed.data(), prob->IO_val, nio * sizeof(double)) != 0;
    if (!plan_set_values(*prob, c.P)) { g_err = c.P.err; return DBAT_HIP_EINVAL; }
    c.set_values(io_changed);
    return DBAT_HIP_OK;
    API_CATCH
}

int64_t dbat_hip_num_params(const dbat_hip_handle *h) { return h ? h->core->P.n : -1; }
int64_t dbat_hip_num_residuals(const dbat_hip_handle *h) { return h ? h->core->P.m : -1; }

int dbat_hip_serialize(const dbat_hip_handle *h, double *x) {
    if (!h || !x) { g_err = "null argument"; return DBAT_HIP_EINVAL; }
    const Plan &P = h->core->P;
    for (int64_t i = 0; i < P.n; ++i) x[i] = P.z0[P.x2z[i]];
    return DBAT_HIP_OK;
}

int dbat_hip_deserialize(const dbat_hip_handle *h, const double *x, double *IO, double *EO, double *OP) {
    if (!h || !x) { g_err = "null argument"; return DBAT_HIP_EINVAL; }
    const Plan &P = h->core->P;
    std::vector<double> z(P.z0);
    for (int64_t i = 0; i < P.n; ++i) z[P.x2z[i]] = x[i];
    if (EO)                                             // shared elements fan out from their leading entry (deserialize.m:28-30)
        for (int c = 0; c < P.nc; ++c)
            for (int k = 0; k < 6; ++k) EO[6 * c + k] = z[P.cam_col[(size_t)c * MAXCOL + k]];
    if (OP)                                             // z holds the points in processing order
        for (int64_t p = 0; p < P.np; ++p)
            for (int k = 0; k < 3; ++k) OP[3 * p + k] = z[P.NS + 3 * (int64_t)P.pt_rank[p] + k];
    if (IO)
        for (size_t e = 0; e < P.io_src.size(); ++e)
            IO[e] = P.io_src[e] >= 0 ? z[6 * (int64_t)P.nc + P.io_src[e]] : P.io_fixed[e];
    return DBAT_HIP_OK;
}

int dbat_hip_structural_rank_ok(const dbat_hip_handle *h, int32_t *ok) {
    if (!h || !ok) { g_err = "null argument"; return DBAT_HIP_EINVAL; }
    *ok = h->core->P.rank_ok ? 1 : 0;
    return DBAT_HIP_OK;
}

namespace dbat {
static void export_residuals(Core &c, const double *zdev, double *r_unw, double *r_wgt, double *f) {
    // image rows via k_residual (reference row order), prior rows on the host
    DevBuf<double> tmp;
    double *dev_unw = nullptr;
    if (r_unw || r_wgt) { tmp.alloc(2 * std::max<int64_t>(c.P.no, 1)); dev_unw = tmp.p; HIPCHK(hipMemsetAsync(dev_unw, 0, 2 * c.P.no * 8, c.stream)); }
    const double ff = c.eval_f(zdev, nullptr, dev_unw);
    if (f) *f = ff;
    if (!(r_unw || r_wgt)) return;
    const Plan &P = c.P;
    // a sharded handle computes the rows of its own observations; the sum over the ranks
    // (zeros elsewhere) gives every rank the whole vector.  Weighted rows on the device
    // as well, so that the weights of the other shards' observations are not needed here.
    DevBuf<double> tmpw;
    if (r_wgt) {
        tmpw.alloc(2 * std::max<int64_t>(P.no, 1));
        HIPCHK(hipMemsetAsync(tmpw.p, 0, 2 * P.no * 8, c.stream));
        if (c.nobs > 0)
            LAUNCHK(k_weight_rows, dim3((unsigned)cdiv(c.nobs, 256)), dim3(256), 0, c.stream, c.d, dev_unw, tmpw.p);
    }
    if (c.multi()) {
        c.do_allreduce(dev_unw, 2 * P.no);
        if (r_wgt) c.do_allreduce(tmpw.p, 2 * P.no);
    }
    const double *zfull = c.gathered(zdev);
    std::vector<double> zh(P.NZ);
    if (r_unw) HIPCHK(hipMemcpyAsync(r_unw, dev_unw, 2 * P.no * 8, hipMemcpyDeviceToHost, c.stream));
    if (r_wgt) HIPCHK(hipMemcpyAsync(r_wgt, tmpw.p, 2 * P.no * 8, hipMemcpyDeviceToHost, c.stream));
    HIPCHK(hipMemcpyAsync(zh.data(), zfull, P.NZ * 8, hipMemcpyDeviceToHost, c.stream));
    c.sync();
    int64_t row = 2 * P.no;
    for (int64_t zi : P.prior_z) {
        const double e = zh[zi] - P.z_prv[zi];
        if (r_unw) r_unw[row] = e;
        if (r_wgt) r_wgt[row] = e * std::sqrt(P.z_prw[zi]);
        ++row;
    }
}
}  // namespace dbat

int dbat_hip_residual(dbat_hip_handle *h, const double *x, double *r_unweighted, double *f) {
    API_TRY
    if (!h || !x) { g_err = "null argument"; return DBAT_HIP_EINVAL; }
    Core &c = *h->core;
    DeviceGuard dev_guard(c.device);
    c.x_to_z(x, c.zt.p);
    export_residuals(c, c.zt.p, r_unweighted, nullptr, f);
    return DBAT_HIP_OK;
    API_CATCH
}

int dbat_hip_jacobian_blocks(dbat_hip_handle *h, const double *x, double *JEO, double *JOP, double *JIO) {
    API_TRY
    if (!h || !x) { g_err = "null argument"; return DBAT_HIP_EINVAL; }
    Core &c = *h->core;
    DeviceGuard dev_guard(c.device);
    const Plan &P = c.P;
    c.x_to_z(x, c.zt.p);
    c.prep_cams(c.zt.p);
    DevBuf<double> a, b, cc;
    const int64_t no = std::max<int64_t>(P.no, 1);
    if (JEO) { a.alloc(12 * no); HIPCHK(hipMemsetAsync(a.p, 0, 12 * no * 8, c.stream)); }
    if (JOP) { b.alloc(6 * no); HIPCHK(hipMemsetAsync(b.p, 0, 6 * no * 8, c.stream)); }
    if (JIO) { cc.alloc(2 * (int64_t)P.nIOrows * no); HIPCHK(hipMemsetAsync(cc.p, 0, 2 * (int64_t)P.nIOrows * no * 8, c.stream)); }
    if (c.nobs > 0) {
#define L_JB(M, dummy) LAUNCHK((k_jac_blocks<M>), dim3((unsigned)cdiv(c.nobs, 256)), dim3(256), 0, c.stream, c.d, c.zt.p, c.cams.p, a.p, b.p, cc.p)
        switch (P.model) { case 2: L_JB(2, 0); break; case 3: L_JB(3, 0); break; case 4: L_JB(4, 0); break; default: L_JB(5, 0); break; }
#undef L_JB
    }
    if (JEO) HIPCHK(hipMemcpyAsync(JEO, a.p, 12 * P.no * 8, hipMemcpyDeviceToHost, c.stream));
    if (JOP) HIPCHK(hipMemcpyAsync(JOP, b.p, 6 * P.no * 8, hipMemcpyDeviceToHost, c.stream));
    if (JIO) HIPCHK(hipMemcpyAsync(JIO, cc.p, 2 * (int64_t)P.nIOrows * P.no * 8, hipMemcpyDeviceToHost, c.stream));
    c.sync();
    return DBAT_HIP_OK;
    API_CATCH
}

int dbat_hip_jacobian_sample(dbat_hip_handle *h, const double *x, int64_t n, const int64_t *ip_col, double *res,
                             double *JEO, double *JOP, double *JIO) {
    API_TRY
    if (!h || !x || n < 0 || (n > 0 && (!ip_col || !res || !JEO || !JOP || !JIO))) { g_err = "null argument"; return DBAT_HIP_EINVAL; }
    Core &c = *h->core;
    const Plan &P = c.P;
    if (P.nranks > 1) { g_err = "dbat_hip_jacobian_sample: one-rank handles only"; return DBAT_HIP_EUNSUPPORTED; }
    if (n == 0) return DBAT_HIP_OK;
    // IP column -> position in processing order, for the requested columns only
    std::vector<int64_t> pos((size_t)n, -1);
    {
        std::vector<std::pair<int64_t, int64_t>> want((size_t)n);
        for (int64_t i = 0; i < n; ++i) {
            if (ip_col[i] < 0 || ip_col[i] >= P.no) { g_err = "IP column out of range"; return DBAT_HIP_EINVAL; }
            want[i] = {ip_col[i], i};
        }
        std::sort(want.begin(), want.end());
        const int64_t nobs = (int64_t)P.o_row.size();
        for (int64_t o = 0; o < nobs; ++o) {
            auto it = std::lower_bound(want.begin(), want.end(), std::make_pair(P.o_row[o], (int64_t)-1));
            for (; it != want.end() && it->first == P.o_row[o]; ++it) pos[it->second] = o;
        }
        // (a one-rank plan holds every IP column; the kernel indexes the observation arrays with these positions)
        for (int64_t i = 0; i < n; ++i)
            if (pos[i] < 0) { g_err = "dbat_hip_jacobian_sample: IP column " + std::to_string(ip_col[i]) + " is not in this handle's plan"; return DBAT_HIP_EINVAL; }
    }
    DeviceGuard dev_guard(c.device);
    c.x_to_z(x, c.zt.p);
    c.prep_cams(c.zt.p);
    DevBuf<int64_t> dpos;
    dpos.upload(pos);
    const int R = P.nIOrows;
    DevBuf<double> dr, a, b, cc;
    dr.alloc(2 * n); a.alloc(12 * n); b.alloc(6 * n); cc.alloc(2 * (int64_t)R * n);
#define L_JS(M, dummy) LAUNCHK((k_jac_sample<M>), dim3((unsigned)cdiv(n, 256)), dim3(256), 0, c.stream, c.d, c.zt.p, c.cams.p, n, dpos.p, dr.p, a.p, b.p, cc.p)
    switch (P.model) { case 2: L_JS(2, 0); break; case 3: L_JS(3, 0); break; case 4: L_JS(4, 0); break; default: L_JS(5, 0); break; }
#undef L_JS
    HIPCHK(hipMemcpyAsync(res, dr.p, 2 * n * 8, hipMemcpyDeviceToHost, c.stream));
    HIPCHK(hipMemcpyAsync(JEO, a.p, 12 * n * 8, hipMemcpyDeviceToHost, c.stream));
    HIPCHK(hipMemcpyAsync(JOP, b.p, 6 * n * 8, hipMemcpyDeviceToHost, c.stream));
    HIPCHK(hipMemcpyAsync(JIO, cc.p, 2 * (int64_t)R * n * 8, hipMemcpyDeviceToHost, c.stream));
    c.sync();
    return DBAT_HIP_OK;
    API_CATCH
}

int dbat_hip_jacobian_csc(dbat_hip_handle *h, const double *x, int32_t weighted, int64_t *nnz_out,
                          int64_t *colptr, int64_t *rowidx, double *val) {
    API_TRY
    if (!h || !x || !nnz_out) { g_err = "null argument"; return DBAT_HIP_EINVAL; }
    Core &c = *h->core;
    const Plan &P = c.P;
    if (P.nranks > 1) { g_err = "dbat_hip_jacobian_csc: one-rank handles only"; return DBAT_HIP_EUNSUPPORTED; }
    std::vector<int64_t> z2x((size_t)P.NZ, -1);
    for (int64_t i = 0; i < P.n; ++i) z2x[P.x2z[i]] = i;
    // columns of one observation: EO (6), IO (the camera's estimated IO columns), OP (3); -1 = no unknown
    const int R = P.nIOrows;
    auto cols_of = [&](int64_t o, int64_t *eo, int64_t *io, int *iorow, int &nio, int64_t *op) {
        const int cam = P.o_cam[o];
        for (int k = 0; k < 6; ++k) { const int32_t zc = P.cam_col[(size_t)cam * MAXCOL + k]; eo[k] = (P.cam_eo_est[cam] >> k) & 1u ? z2x[zc] : -1; }
        nio = P.cam_ncol[cam] - 6;
        for (int q = 0; q < nio; ++q) { io[q] = z2x[P.cam_col[(size_t)cam * MAXCOL + 6 + q]]; iorow[q] = P.cam_iorow[(size_t)cam * MAXIO + q]; }
        for (int k = 0; k < 3; ++k) op[k] = z2x[P.NS + 3 * (int64_t)P.o_pt[o] + k];
    };
    const int64_t nobs = (int64_t)P.o_cam.size();
    std::vector<int64_t> cnt((size_t)P.n + 1, 0);
    int64_t eo[6], io[MAXIO], op[3];
    int iorow[MAXIO], nio;
    for (int64_t o = 0; o < nobs; ++o) {
        cols_of(o, eo, io, iorow, nio, op);
        for (int k = 0; k < 6; ++k) if (eo[k] >= 0) cnt[eo[k] + 1] += 2;
        for (int q = 0; q < nio; ++q) if (io[q] >= 0) cnt[io[q] + 1] += 2;
        for (int k = 0; k < 3; ++k) if (op[k] >= 0) cnt[op[k] + 1] += 2;
    }
    for (int64_t zi : P.prior_z) cnt[z2x[zi] + 1] += 1;
    for (int64_t i = 0; i < P.n; ++i) cnt[i + 1] += cnt[i];
    *nnz_out = cnt[P.n];
    if (!colptr) return DBAT_HIP_OK;
    if (!rowidx || !val) { g_err = "null argument"; return DBAT_HIP_EINVAL; }
    DeviceGuard dev_guard(c.device);
    c.x_to_z(x, c.zt.p);
    c.prep_cams(c.zt.p);
    const int64_t no = std::max<int64_t>(P.no, 1);
    DevBuf<double> a, b, cc;
    a.alloc(12 * no); b.alloc(6 * no); cc.alloc(2 * (int64_t)R * no);
    if (c.nobs > 0) {
#define L_JB(M, dummy) LAUNCHK((k_jac_blocks<M>), dim3((unsigned)cdiv(c.nobs, 256)), dim3(256), 0, c.stream, c.d, c.zt.p, c.cams.p, a.p, b.p, cc.p)
        switch (P.model) { case 2: L_JB(2, 0); break; case 3: L_JB(3, 0); break; case 4: L_JB(4, 0); break; default: L_JB(5, 0); break; }
#undef L_JB
    }
    std::vector<double> JEO((size_t)12 * no), JOP((size_t)6 * no), JIO((size_t)2 * R * no);
    HIPCHK(hipMemcpyAsync(JEO.data(), a.p, JEO.size() * 8, hipMemcpyDeviceToHost, c.stream));
    HIPCHK(hipMemcpyAsync(JOP.data(), b.p, JOP.size() * 8, hipMemcpyDeviceToHost, c.stream));
    HIPCHK(hipMemcpyAsync(JIO.data(), cc.p, JIO.size() * 8, hipMemcpyDeviceToHost, c.stream));
    c.sync();
    for (int64_t i = 0; i <= P.n; ++i) colptr[i] = cnt[i];
    std::vector<int64_t> fill(cnt.begin(), cnt.end() - 1);
    // observations in REFERENCE row order (ascending IP column), so that the rows of a column ascend
    std::vector<int64_t> by_row((size_t)nobs);
    for (int64_t o = 0; o < nobs; ++o) by_row[P.o_row[o]] = o;
    for (int64_t k = 0; k < nobs; ++k) {
        const int64_t o = by_row[k];
        const int cam = P.o_cam[o];
        double w0 = 1.0, w1 = 1.0;
        if (weighted) { w0 = P.uniform_w ? P.cam_w[2 * cam] : P.o_w[2 * o]; w1 = P.uniform_w ? P.cam_w[2 * cam + 1] : P.o_w[2 * o + 1]; }
        cols_of(o, eo, io, iorow, nio, op);
        auto put = [&](int64_t col, double v0, double v1) {
            int64_t &f = fill[col];
            rowidx[f] = 2 * k; val[f] = v0 * w0; rowidx[f + 1] = 2 * k + 1; val[f + 1] = v1 * w1;
            f += 2;
        };
        for (int q = 0; q < 6; ++q) if (eo[q] >= 0) put(eo[q], JEO[12 * k + 2 * q], JEO[12 * k + 2 * q + 1]);
        for (int q = 0; q < nio; ++q) if (io[q] >= 0) put(io[q], JIO[2 * (int64_t)R * k + 2 * iorow[q]], JIO[2 * (int64_t)R * k + 2 * iorow[q] + 1]);
        for (int q = 0; q < 3; ++q) if (op[q] >= 0) put(op[q], JOP[6 * k + 2 * q], JOP[6 * k + 2 * q + 1]);
    }
    int64_t row = 2 * P.no;
    for (int64_t zi : P.prior_z) {                       // prior rows: selection rows of I (prior_obs.m:45-72)
        int64_t &f = fill[z2x[zi]];
        rowidx[f] = row++; val[f] = weighted ? std::sqrt(P.z_prw[zi]) : 1.0;
        ++f;
    }
    return DBAT_HIP_OK;
    API_CATCH
}

int dbat_hip_linearize_solve(dbat_hip_handle *h, const double *x, double lambda, int32_t scale_columns,
                             double *p, double *stats) {
    API_TRY
    if (!h || !x) { g_err = "null argument"; return DBAT_HIP_EINVAL; }
    Core &c = *h->core;
    DeviceGuard dev_guard(c.device);
    c.x_to_z(x, c.z.p);
    c.build(c.z.p, lambda, scale_columns);
    double JpJp, rJp, pp;
    c.solve(JpJp, rJp, pp);
    const bool singular = c.near_singular;
    if (p) c.z_to_x(c.dz.p, p);
    if (stats) {
        stats[0] = c.f_lin; stats[1] = JpJp; stats[2] = rJp; stats[3] = pp;
        stats[4] = c.trace_jtj; stats[5] = singular ? 1.0 : 0.0; stats[6] = c.rcond_est; stats[7] = (double)c.chol_info;
    }
    return DBAT_HIP_OK;
    API_CATCH
}

int dbat_hip_gradient(dbat_hip_handle *h, double *g) {
    API_TRY
    if (!h || !g || !h->core->have_lin) { g_err = "no linearisation"; return DBAT_HIP_EINVAL; }
    Core &c = *h->core;
    DeviceGuard dev_guard(c.device);
    c.ensure_build();                                 // (a linearisation the damping loop left to whoever needs it)
    c.gradient(c.vtmp.p);
    c.z_to_x(c.vtmp.p, g);
    return DBAT_HIP_OK;
    API_CATCH
}

int dbat_hip_colnorms(dbat_hip_handle *h, double *Jn) {
    API_TRY
    if (!h || !Jn || !h->core->have_lin) { g_err = "no linearisation"; return DBAT_HIP_EINVAL; }
    Core &c = *h->core;
    DeviceGuard dev_guard(c.device);
    c.ensure_build();                                 // (a linearisation the damping loop left to whoever needs it)
    c.colnorm2(c.vtmp.p);
    c.z_to_x(c.vtmp.p, Jn);
    for (int64_t i = 0; i < c.P.n; ++i) Jn[i] = std::sqrt(Jn[i]);
    return DBAT_HIP_OK;
    API_CATCH
}

int dbat_hip_jtimes_sqnorm(dbat_hip_handle *h, const double *v, double *sqnorm) {
    API_TRY
    if (!h || !v || !sqnorm || !h->core->have_lin) { g_err = "no linearisation"; return DBAT_HIP_EINVAL; }
    Core &c = *h->core;
    DeviceGuard dev_guard(c.device);
    c.ensure_build();                                 // (a linearisation the damping loop left to whoever needs it)
    HIPCHK(hipMemsetAsync(c.vtmp.p, 0, c.P.NZ * 8, c.stream));
    if (c.P.n) {
        HIPCHK(hipMemcpyAsync(c.xbuf.p, v, c.P.n * 8, hipMemcpyHostToDevice, c.stream));
        LAUNCHK(k_scatter_x, dim3((unsigned)cdiv(c.P.n, 256)), dim3(256), 0, c.stream, c.P.n, c.x2z.p, c.xbuf.p, c.vtmp.p);
    }
    double a, b, vv;
    c.jtimes(c.vtmp.p, a, b, vv);
    *sqnorm = a;
    return DBAT_HIP_OK;
    API_CATCH
}

namespace dbat {
// J v (v in z layout on the device) as host rows: image rows in the reference's order, then the prior rows -- at the
// linearisation the handle holds (zlin)
static void jtimes_rows(Core &c, const double *v_dev, double *Jv) {
    const Plan &P = c.P;
    if (!c.cams_at_lin) { c.prep_cams(c.zlin.p); c.cams_at_lin = true; }
    DevBuf<double> out;
    out.alloc(2 * std::max<int64_t>(P.no, 1));
    if (c.nobs > 0) {
#define L_JTV(M, NCXV) LAUNCHK((k_jtimes_vec<M, NCXV>), dim3((unsigned)cdiv(c.nobs, 256)), dim3(256), 0, c.stream, c.d, c.zlin.p, c.cams.p, v_dev, out.p)
        if (c.tile_ncx == 6) { DISPATCH_MODEL(L_JTV, 6) } else if (c.tile_ncx == 14) { DISPATCH_MODEL(L_JTV, 14) } else if (c.tile_ncx == 15) { DISPATCH_MODEL(L_JTV, 15) } else { DISPATCH_MODEL(L_JTV, MAXCOL) }
#undef L_JTV
    }
    std::vector<double> vz(P.NZ);
    HIPCHK(hipMemcpyAsync(Jv, out.p, 2 * P.no * 8, hipMemcpyDeviceToHost, c.stream));
    HIPCHK(hipMemcpyAsync(vz.data(), v_dev, P.NZ * 8, hipMemcpyDeviceToHost, c.stream));
    c.sync();
    int64_t row = 2 * P.no;                                  // prior rows: selection rows of I, weighted (prior_obs.m:45-72)
    for (int64_t zi : P.prior_z) Jv[row++] = vz[zi] * std::sqrt(P.z_prw[zi]);
}
}  // namespace dbat

int dbat_hip_jtimes(dbat_hip_handle *h, const double *v, double *Jv) {
    API_TRY
    if (!h || !v || !Jv || !h->core->have_lin) { g_err = "no linearisation"; return DBAT_HIP_EINVAL; }
    Core &c = *h->core;
    const Plan &P = c.P;
    if (P.nranks > 1) { g_err = "dbat_hip_jtimes: one-rank handles only"; return DBAT_HIP_EUNSUPPORTED; }
    DeviceGuard dev_guard(c.device);
    c.ensure_build();                                 // (a linearisation the damping loop left to whoever needs it)
    HIPCHK(hipMemsetAsync(c.vtmp.p, 0, P.NZ * 8, c.stream));
    if (P.n) {
        HIPCHK(hipMemcpyAsync(c.xbuf.p, v, P.n * 8, hipMemcpyHostToDevice, c.stream));
        LAUNCHK(k_scatter_x, dim3((unsigned)cdiv(P.n, 256)), dim3(256), 0, c.stream, P.n, c.x2z.p, c.xbuf.p, c.vtmp.p);
    }
    jtimes_rows(c, c.vtmp.p, Jv);
    return DBAT_HIP_OK;
    API_CATCH
}

int dbat_hip_solve(dbat_hip_handle *h, const dbat_hip_options *opt, double *x, dbat_hip_result *result,
                   double *res, double *damp, double *aux, double *trace) {
    API_TRY
    if (!h || !opt || !x || !result) { g_err = "null argument"; return DBAT_HIP_EINVAL; }
    if (opt->damping < 0 || opt->damping > 3) { g_err = "Unknown damping"; return DBAT_HIP_EINVAL; }
    if (opt->store_trace && !trace) { g_err = "store_trace without a trace buffer"; return DBAT_HIP_EINVAL; }
    if (opt->term_fun && h->core->P.nranks > 1) { g_err = "term_fun: one-rank handles only (J*p rows stay with their shard)"; return DBAT_HIP_EUNSUPPORTED; }
    Core &c = *h->core;
    DeviceGuard dev_guard(c.device);
    c.n_res_evals = c.n_lin = c.n_solves = c.n_trace_only = 0;
    c.trace_begin();
    c.x_to_z(x, c.z.p);
    c.lambda_lin = NAN;
    LoopOut out;
    const auto t0 = std::chrono::steady_clock::now();
    c.stages_begin();
    struct StageGuard { Core &c; ~StageGuard() { c.st_on = false; } } stage_guard{c};     // (a loop that throws must not leave the timers on)
    switch (opt->damping) {
        case DBAT_HIP_DAMP_GM: loop_gm(c, *opt, out); break;
        case DBAT_HIP_DAMP_GNA: loop_gna(c, *opt, out); break;
        case DBAT_HIP_DAMP_LM: loop_lm(c, *opt, out); break;
        default: {
            double delta0 = opt->delta0;
            if (!(delta0 > 0)) {                                   // bundle.m:325 delta0 = norm(x0)
                double s = 0;
                for (int64_t i = 0; i < c.P.n; ++i) s += x[i] * x[i];
                delta0 = std::sqrt(s);
            }
            loop_lmp(c, *opt, delta0, out);
        }
    }
    c.sync();
    result->time_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    c.stages_end(result->stage_s);
    result->code = out.code; result->iters = out.iters;
    c.z_to_x(c.z.p, x);
    const int cap_res = opt->max_iter + 3, cap_d = 2 * opt->max_iter + 4;
    result->n_res = (int)std::min<size_t>(out.res.size(), cap_res);
    if (res) for (int i = 0; i < result->n_res; ++i) res[i] = out.res[i];
    result->n_damp = (int)std::min<size_t>(out.damp.size(), cap_d);
    if (damp) for (int i = 0; i < result->n_damp; ++i) damp[i] = out.damp[i];
    if (aux) for (int i = 0; i < cap_d; ++i) aux[i] = i < (int)out.aux.size() ? out.aux[i] : NAN;
    result->n_trace = 0;
    if (opt->store_trace && trace) {
        const int cap_t = opt->max_iter + 2;
        result->n_trace = c.trace_download(trace, cap_t);
    }
    const double dof = (double)(c.P.m - c.P.n);
    result->sigma0 = std::sqrt(2 * out.f_final / dof);            // bundle.m:476-483
    result->n_residual_evals = c.n_res_evals; result->n_linearizations = c.n_lin; result->n_solves = c.n_solves;
    result->n_trace_only = c.n_trace_only;
    return DBAT_HIP_OK;
    API_CATCH
}

int dbat_hip_final_residuals(dbat_hip_handle *h, double *r_unweighted, double *r_weighted) {
    API_TRY
    if (!h || !h->core->have_lin) { g_err = "no linearisation"; return DBAT_HIP_EINVAL; }
    Core &c = *h->core;
    DeviceGuard dev_guard(c.device);
    export_residuals(c, c.zlin.p, r_unweighted, r_weighted, nullptr);
    return DBAT_HIP_OK;
    API_CATCH
}

int dbat_hip_set_deterministic(dbat_hip_handle *h, int32_t on) {
    API_TRY
    if (!h) { g_err = "null handle"; return DBAT_HIP_EINVAL; }
    Core &c = *h->core;
    if (on && !c.deterministic_supported()) {
        g_err = "deterministic mode covers the signature-group path on one rank (no heavy or giant points, no irregular-visibility tile kernels)";
        return DBAT_HIP_EUNSUPPORTED;
    }
    DeviceGuard dev_guard(c.device);
    c.set_deterministic(on != 0);
    return DBAT_HIP_OK;
    API_CATCH
}

int dbat_hip_set_allreduce(dbat_hip_handle *h, dbat_hip_allreduce_fn fn, void *user) {
    if (!h) { g_err = "null handle"; return DBAT_HIP_EINVAL; }
    h->core->allreduce = fn; h->core->allreduce_user = user;
    return DBAT_HIP_OK;
}

int dbat_hip_comm_unique_id(uint8_t *id) {
    API_TRY
    if (!id) { g_err = "null argument"; return DBAT_HIP_EINVAL; }
    static_assert(sizeof(ncclUniqueId) <= DBAT_HIP_UNIQUE_ID_BYTES, "unique id does not fit");
    ncclUniqueId uid;
    NCCLCHK(ncclGetUniqueId(&uid));
    memset(id, 0, DBAT_HIP_UNIQUE_ID_BYTES);
    memcpy(id, &uid, sizeof(uid));
    return DBAT_HIP_OK;
    API_CATCH
}

int dbat_hip_comm_init(dbat_hip_handle *h, const uint8_t *id) {
    API_TRY
    if (!h || !id) { g_err = "null argument"; return DBAT_HIP_EINVAL; }
    Core &c = *h->core;
    DeviceGuard dev_guard(c.device);
    if (c.nccl) { g_err = "the handle already has a communicator"; return DBAT_HIP_EINVAL; }
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    NCCLCHK(ncclCommInitRank(&c.nccl, c.P.nranks, uid, c.P.rank));
    return DBAT_HIP_OK;
    API_CATCH
}

int dbat_hip_comm_allreduce_host(dbat_hip_handle *h, double *buf, int64_t count, int32_t op) {
    API_TRY
    if (!h || !buf || count < 0 || op < 0 || op > 2) { g_err = "bad argument"; return DBAT_HIP_EINVAL; }
    Core &c = *h->core;
    DeviceGuard dev_guard(c.device);
    if (!c.nccl) return DBAT_HIP_OK;                  // one rank: the identity
    DevBuf<double> tmp;
    tmp.alloc((size_t)std::max<int64_t>(count, 1));
    HIPCHK(hipMemcpyAsync(tmp.p, buf, count * 8, hipMemcpyHostToDevice, c.stream));
    NCCLCHK(ncclAllReduce(tmp.p, tmp.p, (size_t)count, ncclDouble, op == 0 ? ncclSum : (op == 1 ? ncclMax : ncclMin), c.nccl, c.stream));
    HIPCHK(hipMemcpyAsync(buf, tmp.p, count * 8, hipMemcpyDeviceToHost, c.stream));
    c.sync();
    return DBAT_HIP_OK;
    API_CATCH
}

int dbat_hip_owned_mask(const dbat_hip_handle *h, uint8_t *mask) {
    if (!h || !mask) { g_err = "null argument"; return DBAT_HIP_EINVAL; }
    const Plan &P = h->core->P;
    for (int64_t i = 0; i < P.n; ++i) mask[i] = P.z_mine[P.x2z[i]];
    return DBAT_HIP_OK;
}

int dbat_hip_set_x(dbat_hip_handle *h, const double *x) {
    API_TRY
    if (!h || !x) { g_err = "null argument"; return DBAT_HIP_EINVAL; }
    Core &c = *h->core;
    DeviceGuard dev_guard(c.device);
    c.x_to_z(x, c.z.p);
    c.sync();
    return DBAT_HIP_OK;
    API_CATCH
}

int dbat_hip_bench_step(dbat_hip_handle *h, double lambda, int32_t scale_columns, double *ms) {
    API_TRY
    if (!h) { g_err = "null handle"; return DBAT_HIP_EINVAL; }
    Core &c = *h->core;
    DeviceGuard dev_guard(c.device);
    c.timing = true;
    HIPCHK(hipEventRecord(c.ev[0], c.stream));
    c.build_enqueue(c.z.p, lambda, scale_columns);
    if (c.mg_subtree) c.allreduce_vectors(); else c.allreduce_system();
    c.finish_enqueue(c.z.p, lambda, scale_columns);
    HIPCHK(hipEventRecord(c.ev[1], c.stream));
    c.factor_solve_enqueue();
    HIPCHK(hipEventRecord(c.ev[2], c.stream));
    c.backsub_enqueue();
    c.do_allreduce(c.scal.p, 8);
    HIPCHK(hipEventRecord(c.ev[3], c.stream));
    (void)c.eval_f_step(c.z.p, 1.0, c.dz.p, c.zt.p);              // trial-point residual, syncs
    HIPCHK(hipEventRecord(c.ev[4], c.stream));
    c.sync();
    c.timing = false;
    c.have_lin = true; c.lambda_lin = lambda; c.scale_lin = scale_columns; c.s_valid = false;
    if (ms) {
        for (int i = 0; i < 4; ++i) {
            float t = 0;
            HIPCHK(hipEventElapsedTime(&t, c.ev[i], c.ev[i + 1]));
            ms[i] = t;
        }
        for (int i = 0; i < 4; ++i) {      // k_build, potrf+potrs, k_backsub, k_residual
            float t = 0;
            HIPCHK(hipEventElapsedTime(&t, c.kev[2 * i], c.kev[2 * i + 1]));
            ms[4 + i] = t;
        }
        for (int i = 8; i < 16; ++i) ms[i] = 0.0;
        if (c.use_heavy) {                                   // heavy / giant points: camera side + Z rows, then the products
            float t = 0;
            HIPCHK(hipEventElapsedTime(&t, c.kev[10], c.kev[11])); ms[12] = t;
            HIPCHK(hipEventElapsedTime(&t, c.kev[11], c.kev[12])); ms[13] = t;
        }
        if (c.mg_subtree && c.multi() && c.use_perm) {      // domain sharding: the three parts of the factorisation
            float t = 0;
            HIPCHK(hipEventElapsedTime(&t, c.kev[2], c.kev[8])); ms[8] = t;       // own domain + shares of the top tiles
            HIPCHK(hipEventElapsedTime(&t, c.kev[8], c.kev[9])); ms[9] = t;       // all-reduce of the top tiles
            HIPCHK(hipEventElapsedTime(&t, c.kev[9], c.kev[3])); ms[10] = t;      // top separators + backward substitution
        }
    }
    return DBAT_HIP_OK;
    API_CATCH
}

int dbat_hip_posterior_cov(dbat_hip_handle *h, const double *x, double sigma0, double *CEO, double *CIO,
                           double *COP, double *Sinv) {
    API_TRY
    if (!h || !x) { g_err = "null argument"; return DBAT_HIP_EINVAL; }
    Core &c = *h->core;
    DeviceGuard dev_guard(c.device);
    c.x_to_z(x, c.z.p);
    c.posterior_cov(sigma0, CEO, CIO, COP, Sinv);
    return DBAT_HIP_OK;
    API_CATCH
}

int dbat_hip_forwintersect(dbat_hip_handle *h, const double *x, const uint8_t *skip, double *OP) {
    API_TRY
    if (!h || !x || !OP) { g_err = "null argument"; return DBAT_HIP_EINVAL; }
    Core &c = *h->core;
    DeviceGuard dev_guard(c.device);
    const Plan &P = c.P;
    c.x_to_z(x, c.zt.p);                              // IO / EO of the rays
    c.prep_cams(c.zt.p);
    DevBuf<double> dOP;
    dOP.alloc((size_t)3 * std::max(P.np, 1));
    HIPCHK(hipMemsetAsync(dOP.p, 0, (size_t)3 * P.np * sizeof(double), c.stream));
    if (c.nb > 0) LAUNCHK(k_forwintersect, dim3((unsigned)c.nb), dim3(P.BT), (size_t)P.BT * 9 * sizeof(double), c.stream, c.d, c.cams.p, dOP.p);
    if (c.ngiant > 0) LAUNCHK(k_forwintersect_giant, dim3((unsigned)c.ngiant), dim3(256), 0, c.stream, c.d, c.cams.p, dOP.p);
    if (c.multi()) c.do_allreduce(dOP.p, (int64_t)3 * P.np);             // every point is computed by its owner
    std::vector<double> tmp((size_t)3 * P.np);
    HIPCHK(hipMemcpyAsync(tmp.data(), dOP.p, tmp.size() * sizeof(double), hipMemcpyDeviceToHost, c.stream));
    c.sync();
    std::vector<uint8_t> seen((size_t)P.np, 0);       // points without any observation: NaN (pm_multiforwintersect.m:41)
    for (int32_t r : P.o_pt) seen[r] = 1;
    if (c.multi()) std::fill(seen.begin(), seen.end(), 1);               // (other shards' points arrive through the sum)
    const double nan = std::nan("");
    for (int64_t p = 0; p < P.np; ++p) {
        if (skip && skip[p]) continue;                // keeps the caller's value (forwintersect.m:32-36)
        const int64_t r = P.pt_rank[p];
        for (int k = 0; k < 3; ++k) OP[3 * p + k] = seen[r] ? tmp[3 * r + k] : nan;
    }
    return DBAT_HIP_OK;
    API_CATCH
}

int dbat_hip_resect(int32_t device, int32_t n_images, const int64_t *pt_start, const double *X, const double *xn,
                    const int64_t *tri_start, const int32_t *tri, double *P, double *rms) {
    API_TRY
    if (n_images < 0 || !pt_start || !tri_start || !P || !rms) { g_err = "null argument"; return DBAT_HIP_EINVAL; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_err = "no HIP device: the dbat_hip core has no CPU path"; return DBAT_HIP_EDEVICE; }
    if (device < 0 || device >= ndev) { g_err = "bad device index"; return DBAT_HIP_EINVAL; }
    if (n_images == 0) return DBAT_HIP_OK;
    if (pt_start[0] != 0 || tri_start[0] != 0) { g_err = "ranges must start at 0"; return DBAT_HIP_EINVAL; }
    for (int c = 0; c < n_images; ++c)
        if (pt_start[c + 1] < pt_start[c] || tri_start[c + 1] < tri_start[c]) { g_err = "ranges must ascend"; return DBAT_HIP_EINVAL; }
    const int64_t npt = pt_start[n_images], ntri = tri_start[n_images];
    if ((npt > 0 && (!X || !xn)) || (ntri > 0 && !tri)) { g_err = "null argument"; return DBAT_HIP_EINVAL; }
    for (int c = 0; c < n_images; ++c)
        for (int64_t t = 3 * tri_start[c]; t < 3 * tri_start[c + 1]; ++t)
            if (tri[t] < 0 || tri[t] >= pt_start[c + 1] - pt_start[c]) { g_err = "triangle index outside the camera's points"; return DBAT_HIP_EINVAL; }
    DeviceGuard dev_guard(device);
    DevBuf<int64_t> dps, dts;
    DevBuf<double> dX, dx, dP, dr;
    DevBuf<int32_t> dtri;
    dps.upload(std::vector<int64_t>(pt_start, pt_start + n_images + 1));
    dts.upload(std::vector<int64_t>(tri_start, tri_start + n_images + 1));
    dX.upload(std::vector<double>(X, X + 3 * npt)); dx.upload(std::vector<double>(xn, xn + 2 * npt));
    dtri.upload(std::vector<int32_t>(tri, tri + 3 * ntri));
    dP.alloc((size_t)12 * n_images); dr.alloc((size_t)n_images);
    LAUNCHK(k_resect, dim3((unsigned)n_images), dim3(64), 0, (hipStream_t)nullptr, n_images, dps.p, dX.p, dx.p, dts.p, dtri.p, 1, dP.p, dr.p);
    HIPCHK(hipMemcpy(P, dP.p, (size_t)12 * n_images * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(rms, dr.p, (size_t)n_images * sizeof(double), hipMemcpyDeviceToHost));
    return DBAT_HIP_OK;
    API_CATCH
}

int dbat_hip_chol_stats(const dbat_hip_handle *h, int64_t *st) {
    if (!h || !st) { g_err = "null argument"; return DBAT_HIP_EINVAL; }
    const Core &c = *h->core;
    const DataflowChol &f = c.use_perm ? c.dfchol : c.dfchol_ip;
    st[0] = f.n; st[1] = f.ntasks; st[2] = f.n_products; st[3] = f.nT; st[4] = c.use_perm ? 1 : 0; st[5] = 1;
    return DBAT_HIP_OK;
}

int dbat_hip_build_kernel_name(const dbat_hip_handle *h, char *buf, int32_t buf_len) {
    if (!h || !buf || buf_len < 2) { g_err = "null argument"; return DBAT_HIP_EINVAL; }
    const Core &c = *h->core;
    const char *nm = c.use_heavy ? "k_heavy_z + k_heavy_syrk" : "k_build";      // (nothing tiled: the kernels of the untiled points)
    if (c.ntiles > 0 && c.P.nb_tiled > 0) {
        if (c.use_sig) nm = "k_build_sig";
        else if (c.use_tile3 && c.tile_ncx == 6) nm = "k_build_tile3";
        else if (c.use_tile2 && c.tile_ncx <= 15) nm = "k_build_tile2";
        else nm = "k_build";
    }
    snprintf(buf, (size_t)buf_len, "%s", nm);
    return DBAT_HIP_OK;
}

int dbat_hip_info(const dbat_hip_handle *h, int64_t *info) {
    if (!h || !info) { g_err = "null argument"; return DBAT_HIP_EINVAL; }
    const Core &c = *h->core;
    info[0] = c.P.NS; info[1] = c.nb; info[2] = c.P.max_k; info[3] = c.nobs;
    info[4] = c.P.pt_hi - c.P.pt_lo; info[5] = c.P.BT; info[6] = c.P.ncolmax; info[7] = c.ntiles;
    info[8] = c.mg_subtree ? 1 : 0;
    info[9] = c.mg_subtree ? c.dfchol.top_tiles_count() : (c.P.nranks > 1 ? c.pk_s_count : 0);   // doubles summed in / before the factorisation
    info[10] = c.P.nranks > 1 ? (c.mg_subtree ? 2 * c.P.NS + 8 : 3 * c.P.NS + 8) : 0;          // ... and as vectors per linearisation
    info[11] = c.mg_subtree ? c.P.nd.n_top_cams : 0;
    info[12] = c.use_perm ? c.dfchol.nT : 0;
    info[13] = c.mg_subtree ? c.dfchol.ntasks : 0; info[14] = c.mg_subtree ? c.dfchol.ntasksB : 0;
    info[15] = c.tile_kernel_mfma();
    for (int i = 16; i < 24; ++i) info[i] = 0;
    if (c.use_heavy) {
        info[16] = c.P.hv_ntasks; info[17] = c.P.hv_mfma; info[18] = c.P.hv_ngroups; info[19] = c.P.hv_z_doubles * 8;
        info[20] = c.P.hv_npts; info[21] = c.nobs - c.P.hv_obs0; info[22] = c.P.hv_ks_per_task; info[23] = c.P.hv_alg_flops;
    }
    return DBAT_HIP_OK;
}


/* Digest of the host plan (debug / CPU tests only: the plan must be identical bit for bit whatever the number of
 * threads that built it).  out[k] = FNV-1a hash of the k-th field of Plan in the order of plan_digest_names(). */
#define DBAT_PLAN_FIELDS(X)                                                                                           \
    X(x2z) X(io_src) X(io_fixed) X(z_est) X(z_prw) X(z_prv) X(z_mine) X(z0) X(prior_z) X(porder) X(pt_rank) X(o_cam)    \
    X(o_pt) X(o_uv) X(o_w) X(o_seg) X(o_row) X(batch_start) X(o_lc) X(o_pidx) X(tile_batch) X(tile_cam_start)           \
    X(tile_order) X(tile_cams) X(cm_pt) X(cm_uv) X(cm_w) X(cm_chunk_cam) X(cm_chunk_start) X(giant_start)               \
    X(tile_io_start) X(tile_iocols) X(tile_cam_io) X(tile_io_simple) X(sg_chunk) X(sg_tile_chunk0) X(sg_lc) X(sg_gcam)  \
    X(sg_uv) X(sg_w) X(cam_w) X(cam_ncol) X(cam_col) X(cam_iorow) X(cam_eo_est) X(px) X(cam_first) X(cam_adj)
int dbat_hip_debug_plan_digest(const dbat_hip_problem *prob, uint64_t *out, int32_t n_out, char *names, int32_t names_len) {
    API_TRY
    if (!prob || !out) { g_err = "null argument"; return DBAT_HIP_EINVAL; }
    Plan P;
    if (!build_plan(*prob, P, true)) { g_err = P.err; return DBAT_HIP_EINVAL; }
    auto fnv = [](const void *p, size_t n) {
        uint64_t h = 1469598103934665603ull;
        const unsigned char *b = (const unsigned char *)p;
        for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
        return h ^ (uint64_t)n;
    };
    std::vector<uint64_t> hs;
    std::string nm;
#define X(F) hs.push_back(fnv(P.F.data(), P.F.size() * sizeof(P.F[0]))); nm += #F ",";
    DBAT_PLAN_FIELDS(X)
#undef X
    hs.push_back(fnv(P.nd.order.data(), P.nd.order.size() * sizeof(int))); nm += "nd.order,";
    hs.push_back(fnv(P.nd.block_end.data(), P.nd.block_end.size() * sizeof(int))); nm += "nd.block_end,";
    hs.push_back(fnv(P.nd.cam_owner.data(), P.nd.cam_owner.size() * sizeof(int))); nm += "nd.cam_owner,";
    const int64_t sc[] = {P.n, P.m, P.NS, P.NZ, P.nIOu, P.pt_lo, P.pt_hi, P.CMAX, P.nb_tiled, P.n_cm_chunks_tiled, P.sg_kmax, P.sg_rows_max,
                          P.sg_ngroups, P.sg_npoints, P.sg_ok, P.sg_backsub_ok, P.BT, P.ncolmax, P.with_io, P.uniform_w, P.all_std8, P.max_k,
                          P.shared_eo, P.rank_ok, P.order_dims, P.mg_subtree, P.n_tiles_io_simple, P.n_prior[0], P.n_prior[1], P.n_prior[2]};
    hs.push_back(fnv(sc, sizeof(sc))); nm += "scalars";
    if ((int)hs.size() > n_out) { g_err = "digest buffer too small"; return DBAT_HIP_EINVAL; }
    for (size_t i = 0; i < hs.size(); ++i) out[i] = hs[i];
    for (int i = (int)hs.size(); i < n_out; ++i) out[i] = 0;
    if (names && names_len > 0) snprintf(names, (size_t)names_len, "%s", nm.c_str());
    return (int)hs.size();
    API_CATCH
}

/* Host only (debug / CPU unit tests; never used by the product path): the index structures of the heavy / giant points
 * (Plan::hv_*, csrc/heavy.hpp) checked against their definition.  Every row of Z that k_heavy_z would write gets a
 * pseudo-random value (a hash of its point, its row of the reduced system and its k-column) at the place the plan gives
 * it; the tasks of k_heavy_syrk are then replayed on the host (same operand places, same flush rules) and compared with
 * the plain sum over the points of z_p z_p' over ALL their rows.  out[8]: { 1 if the plan takes the path, untiled points,
 * row groups, tasks, k-steps, largest absolute difference, largest absolute entry, number of compared entries }. */
int dbat_hip_debug_heavy_plan_selftest(const dbat_hip_problem *prob, double *out) {
    API_TRY
    if (!prob || !out) { g_err = "null argument"; return DBAT_HIP_EINVAL; }
    Plan P;
    if (!build_plan(*prob, P, true)) { g_err = P.err; return DBAT_HIP_EINVAL; }
    for (int i = 0; i < 8; ++i) out[i] = 0.0;
    if (!P.hv_ok) return DBAT_HIP_OK;
    const int64_t NS = P.NS;
    if (NS > 4000) { g_err = "selftest: reduced system too large for the dense host check"; return DBAT_HIP_EINVAL; }
    auto val = [](int64_t pt, int64_t row, int c) {
        uint64_t h = (uint64_t)pt * 0x9E3779B97F4A7C15ull ^ (uint64_t)(row + 1) * 0xC2B2AE3D27D4EB4Full ^ (uint64_t)(c + 1) * 0x165667B19E3779F9ull;
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        return (double)(int64_t)(h % 2001) / 1000.0 - 1.0;
    };
    std::vector<double> Zs((size_t)P.hv_z_doubles, 0.0);
    std::vector<double> ref((size_t)(NS + 1) * (NS + 1), 0.0), got((size_t)(NS + 1) * (NS + 1), 0.0);
    const int64_t ho0 = P.hv_obs0, ho1 = (int64_t)P.o_cam.size();
    // rows of every point: (row of the reduced system, values of the three k-columns)
    std::vector<int64_t> prow; std::vector<double> pval;
    int64_t o = ho0;
    while (o < ho1) {
        const int32_t pt = P.o_pt[o];
        const int32_t hp = pt - P.hv_pt0;
        prow.clear(); pval.clear();
        std::vector<int32_t> slot_row((size_t)(P.hv_pt_io0[hp + 1] - P.hv_pt_io0[hp]), -1);
        int64_t oe = o;
        for (; oe < ho1 && P.o_pt[oe] == pt; ++oe) {
            const int32_t c = P.o_cam[oe];
            const int32_t dst = P.hv_obs_dst[oe - ho0];
            const int ld = P.hv_obs_ld[oe - ho0];
            for (int a = 0; a < 6; ++a) {
                const int64_t row = P.cam_col[(size_t)c * MAXCOL + a];
                prow.push_back(row);
                for (int k = 0; k < 3; ++k) { const double v = val(pt, row, k); pval.push_back(v); Zs[(size_t)dst + (size_t)k * ld + a] = v; }
            }
            for (int q = 6; q < P.cam_ncol[c]; ++q) {
                const int sl = P.hv_obs_ioloc[(size_t)(oe - ho0) * Plan::HV_NIOC + (q - 6)];
                if (sl >= (int)slot_row.size()) { g_err = "selftest: IO slot out of range"; return DBAT_HIP_EINVAL; }
                const int32_t row = P.cam_col[(size_t)c * MAXCOL + q];
                if (slot_row[sl] >= 0 && slot_row[sl] != row) { g_err = "selftest: two IO columns in one slot"; return DBAT_HIP_EINVAL; }
                slot_row[sl] = row;
            }
        }
        for (size_t sl = 0; sl < slot_row.size(); ++sl) {
            if (slot_row[sl] < 0) { g_err = "selftest: IO slot without a column"; return DBAT_HIP_EINVAL; }
            const int gs = P.hv_pt_io0[hp] + (int)sl;
            if (P.hv_io_pt[gs] != hp) { g_err = "selftest: IO slot of another point"; return DBAT_HIP_EINVAL; }
            prow.push_back(slot_row[sl]);
            for (int k = 0; k < 3; ++k) { const double v = val(pt, slot_row[sl], k); pval.push_back(v); Zs[(size_t)P.hv_io_dst[gs] + (size_t)k * P.hv_io_ld[gs]] = v; }
        }
        prow.push_back(NS);
        for (int k = 0; k < 3; ++k) { const double v = val(pt, NS, k); pval.push_back(v); Zs[(size_t)P.hv_pt_y[2 * hp] + (size_t)k * P.hv_pt_y[2 * hp + 1]] = v; }
        for (size_t a = 0; a < prow.size(); ++a)
            for (size_t b = 0; b < prow.size(); ++b) {
                const int64_t ri = prow[a], rj = prow[b];
                if (ri < rj || (ri == NS && rj == NS)) continue;
                ref[(size_t)rj * (NS + 1) + ri] -= pval[3 * a] * pval[3 * b] + pval[3 * a + 1] * pval[3 * b + 1] + pval[3 * a + 2] * pval[3 * b + 2];
            }
        o = oe;
    }
    // the tasks, as k_heavy_syrk runs them
    int64_t nks_all = 0;
    for (int t = 0; t < P.hv_ntasks; ++t) {
        const int gi = P.hv_task[4 * t], gj = P.hv_task[4 * t + 1], ks0 = P.hv_task[4 * t + 2], nks = P.hv_task[4 * t + 3];
        const int nbi = P.hv_grp_nb[gi], nbj = P.hv_grp_nb[gj];
        nks_all += nks;
        std::vector<double> acc((size_t)48 * 48, 0.0);
        for (int ks = 0; ks < nks; ++ks)
            for (int kk = 0; kk < 4; ++kk) {
                const int32_t *op = P.hv_ops.data() + ((size_t)(ks0 + ks) * 4 + kk) * 2;
                if ((op[0] < 0) != (op[1] < 0)) { g_err = "selftest: half an operand"; return DBAT_HIP_EINVAL; }
                if (op[0] < 0) continue;
                for (int i = 0; i < 16 * nbi; ++i)
                    for (int j = 0; j < 16 * nbj; ++j) acc[(size_t)i * 48 + j] += Zs[(size_t)op[0] + i] * Zs[(size_t)op[1] + j];
            }
        for (int i = 0; i < 16 * nbi; ++i)
            for (int j = 0; j < 16 * nbj; ++j) {
                if (gi == gj && i < j) continue;
                const int32_t ri = P.hv_grp_row[(size_t)gi * 48 + i], cj = P.hv_grp_row[(size_t)gj * 48 + j];
                if (ri < 0 || cj < 0 || cj == NS) continue;
                if (ri < cj) { g_err = "selftest: rows of the groups are not ascending"; return DBAT_HIP_EINVAL; }
                got[(size_t)cj * (NS + 1) + ri] -= acc[(size_t)i * 48 + j];
            }
    }
    double dmax = 0, vmax = 0; int64_t ncmp = 0;
    for (size_t i = 0; i < ref.size(); ++i) {
        dmax = std::max(dmax, std::fabs(ref[i] - got[i])); vmax = std::max(vmax, std::fabs(ref[i]));
        ncmp += ref[i] != 0.0;
    }
    out[0] = 1; out[1] = P.hv_npts; out[2] = P.hv_ngroups; out[3] = P.hv_ntasks; out[4] = (double)nks_all;
    out[5] = dmax; out[6] = vmax; out[7] = (double)ncmp;
    return DBAT_HIP_OK;
    API_CATCH
}

/* Host evaluation of the per-observation model (debug / CPU unit tests of
 * model.hpp only; never used by the product path). */
int dbat_hip_debug_model_eval_host(int32_t model, int32_t nK, int32_t nP, const double *EO6, const double *IO,
                                   double px_size, const double *Q3, const double *uv, double *r2,
                                   double *A12, double *B6, double *C /* 2 x nIOrows */) {
    if (model < 2 || model > 5 || nK < 0 || nK > MAXK || nP < 0 || nP > MAXP) { g_err = "bad model"; return DBAT_HIP_EINVAL; }
    CamRec c{};
    c.c[0] = EO6[0]; c.c[1] = EO6[1]; c.c[2] = EO6[2];
    cam_rotation(EO6 + 3, c.Mt, c.sk, c.ck);
    c.f = IO[0]; c.pp[0] = IO[1]; c.pp[1] = IO[2]; c.b[0] = IO[3]; c.b[1] = IO[4];
    for (int k = 0; k < nK; ++k) c.K[k] = IO[5 + k];
    for (int k = 0; k < nP; ++k) c.P[k] = IO[5 + nK + k];
    c.sz = px_size;
    double r[2], A[2][6], B[2][3], Cf[2][MAXIO];
    switch (model) {
        case 2: obs_eval<2, true, true>(c, nK, nP, Q3, uv[0], uv[1], r, A, B, Cf); break;
        case 3: obs_eval<3, true, true>(c, nK, nP, Q3, uv[0], uv[1], r, A, B, Cf); break;
        case 4: obs_eval<4, true, true>(c, nK, nP, Q3, uv[0], uv[1], r, A, B, Cf); break;
        default: obs_eval<5, true, true>(c, nK, nP, Q3, uv[0], uv[1], r, A, B, Cf); break;
    }
    r2[0] = r[0]; r2[1] = r[1];
    for (int k = 0; k < 6; ++k) { A12[2 * k] = A[0][k]; A12[2 * k + 1] = A[1][k]; }
    for (int k = 0; k < 3; ++k) { B6[2 * k] = B[0][k]; B6[2 * k + 1] = B[1][k]; }
    for (int k = 0; k < 5 + nK + nP; ++k) { C[2 * k] = Cf[0][k]; C[2 * k + 1] = Cf[1][k]; }
    return DBAT_HIP_OK;
}

}  // extern "C"
