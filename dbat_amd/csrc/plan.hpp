// Host-side plan: index maps and the point-major batch layout.
//
// Restates (on the host, once per problem) the reference's
//   misc/buildserialindices.m:69-159,162-221   x order [IO;EO;OP], leading
//                                              elements of parameter blocks
//   misc/buildweightmatrix.m:13-43             sigma_mm = IP.std * pxSize
//   bundle.m:137-154                           prior.use &= est
// and builds the internal layout the kernels use:
//   z = [ EO (6*nc) | IOu (nIOu leading IO unknowns) | OP (3*np) ]
// The first NS = 6*nc+nIOu entries of z are the columns of the reduced
// (camera + IO) system; the OP part is eliminated by the Schur complement.
#pragma once
#include "env.hpp"
#include "par.hpp"
#include <atomic>
#include <chrono>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/dbat_hip.h"
#include "model.hpp"
#include "nd.hpp"

namespace dbat {

struct Plan {
    int nc = 0, np = 0, nIOrows = 0, nK = 0, nP = 0, model = 0;
    int64_t no = 0;
    int rank = 0, nranks = 1;
    // unknown vector
    int64_t n = 0, nIO = 0, nEO = 0, nOP = 0;      // sizes of the x sections
    int64_t NS = 0, NZ = 0;
    int nIOu = 0;
    std::vector<int64_t> x2z;                      // [n] z index of every x entry
    std::vector<int64_t> lead_io;                  // [nIOu] entry of IO.val (column-major) that leads every IO unknown
    uint64_t key[2] = {0, 0};                      // structure_key of the problem the plan was built from
    std::vector<int32_t> io_src;                   // [nIOrows*nc] IOu index or -1 (fixed)
    std::vector<double> io_fixed;                  // [nIOrows*nc] IO.val (used where io_src<0)
    std::vector<uint8_t> z_est;                    // [NZ]
    std::vector<double> z_prw, z_prv;              // [NZ] prior weight 1/sigma^2 (0 = none), prior value
    std::vector<uint8_t> z_mine;                   // [NZ] scalar sums over z counted by this rank
    std::vector<double> z0;                        // [NZ] initial values
    // prior residual rows in reference order (IO, EO, OP), buildserialindices.m:151-159
    std::vector<int64_t> prior_z;                  // z index of every prior row
    int64_t n_prior[3] = {0, 0, 0};
    int64_t m = 0;                                 // total residual rows
    // observations of this shard in processing (point-major) order
    int64_t pt_lo = 0, pt_hi = 0;                  // range in processing order
    std::vector<int32_t> porder;                   // processing order -> point
    std::vector<int32_t> pt_rank;                  // point -> position in the processing order; the object points
                                                   // sit in z in THAT order (neighbouring observations then
                                                   // gather neighbouring coordinates), o_pt holds ranks
    uvec<int32_t> o_cam, o_pt;
    uvec<double> o_uv;                      // 2 per obs
    uvec<double> o_w;                       // 2 per obs (1/sigma_mm) or empty if uniform
    uvec<uint32_t> o_seg;                   // (seg_start | seg_len<<16) within batch
    uvec<int64_t> o_row;                    // IP column (reference order)
    std::vector<int64_t> batch_start;              // [nb+1]
    // tiles: runs of batches whose observations touch at most CMAX cameras; the
    // Schur complement of a tile is accumulated in LDS and flushed once
    uvec<uint8_t> o_lc;                     // local camera index of every observation
    uvec<uint8_t> o_pidx;                   // ordinal of the observation's point inside its batch
    std::vector<int32_t> tile_batch;               // [ntiles+1] first batch of every tile
    std::vector<int32_t> tile_cam_start;           // [ntiles+1]
    std::vector<int32_t> tile_order;               // launch order: longest tiles first (shorter tail)
    std::vector<int32_t> tile_cams;                // global camera ids, ascending inside a tile
    int CMAX = 0;                                  // 0 = no tiling (global atomics)
    int64_t nb_tiled = 0;                          // batches [0,nb_tiled) belong to tiles; the rest hold heavy points
    // camera-major copy of the observations, tiled ones first (k_cam_normal: J_c'J_c, J_c'r per
    // camera over the tiled part; k_residual_cm over everything): chunks of one camera's
    // observations, at most CM_CHUNK each
    static constexpr int CM_CHUNK = 2048;
    uvec<int32_t> cm_pt;                    // object point
    uvec<double> cm_uv;                     // 2 per observation
    uvec<double> cm_w;                      // 2 per observation or empty if uniform
    std::vector<int32_t> cm_chunk_cam;             // camera of every chunk
    std::vector<int64_t> cm_chunk_start;           // [nchunks+1]
    int64_t n_cm_chunks_tiled = 0;                 // chunks of tiled observations come first
    // "giant" points (more observations than a batch holds) come last, outside the batches
    std::vector<int64_t> giant_start;              // [ngiant+1] first observation of every giant point
    // self-calibration: the estimated IO columns of a tile's cameras are extra rows
    // of the tile-local system (at most IOT of them), after the 6*ncam camera rows
    static constexpr int IOT = 16;
    std::vector<int32_t> tile_io_start;            // [ntiles+1]
    std::vector<int32_t> tile_iocols;              // IOu indices, ascending inside a tile
    std::vector<uint8_t> tile_cam_io;              // [#tile cams][16]: local IO row of the camera's j-th IO column
    std::vector<uint8_t> tile_io_simple;           // per tile: all its cameras share one IO block, IO column q = tile IO row q
    int32_t n_tiles_io_simple = 0;
    // signature groups (k_build_sig): consecutive points of a tile that are seen by exactly the
    // same cameras share their rows of the reduced system.  A chunk = at most 64 points of one
    // group; 8 ints per chunk {first point (processing order), #points, #cameras k, first
    // observation (point-major arrays), #points of the whole group, index of the chunk's first
    // point in the group, first observation of the group (slot-major copy), offset into sg_lc}.
    static constexpr int SG_KMAX = 13;             // 6*13 + 1 rows fit five 16-row blocks
#ifndef DBAT_SIG_IO_WAVES
#define DBAT_SIG_IO_WAVES 4
#endif
    static constexpr int SIG_IO_CMAX = DBAT_SIG_IO_WAVES == 8 ? 16 : 18;    // cameras per tile, self-calibration (sig.hpp SIG_IO_CAMS)
    static constexpr int SG_CHUNK = 64;
    std::vector<int32_t> sg_chunk;                 // [nchunks][8]
    std::vector<int32_t> sg_tile_chunk0;           // [ntiles+1]
    std::vector<uint8_t> sg_lc;                    // [nchunks][16] tile-local camera index of every slot
    std::vector<int32_t> sg_gcam;                  // [nchunks][16] the same cameras by their global index
    uvec<double> sg_uv;                     // slot-major copy of o_uv: group g, slot j, point i at 2*(obs0_g + j*m_g + i)
    uvec<double> sg_w;                      // the same for o_w (empty if uniform)
    int sg_kmax = 0;                               // largest k among the tiled points
    int sg_rows_max = 0;                           // most rows of a chunk: 6k + IO columns of its tile + 1
    int64_t sg_ngroups = 0, sg_npoints = 0;
    bool sg_ok = false;                            // the tiled points can go through k_build_sig
    bool sg_backsub_ok = false;                    // ... and through k_backsub_sig
    int BT = 256;
    int ncolmax = 6;
    bool with_io = false;
    bool uniform_w = true;
    std::vector<double> cam_w;                     // [2*nc] per-camera 1/sigma_mm when uniform
    std::vector<int32_t> cam_ncol, cam_col, cam_iorow;   // per camera column lists (MAXCOL / MAXIO strides)
    std::vector<uint32_t> cam_eo_est;
    bool all_std8 = false;                         // self-calibration: every camera estimates exactly cc px py K1 K2 K3 P1 P2 (nK = 3, nP = 2)
    std::vector<double> px;                        // [2*nc]
    std::vector<int32_t> cam_first;                // lowest camera index sharing an object point with each camera
    std::vector<uint64_t> cam_adj;                 // [nc][cam_adj_words] co-visibility graph (bit c2 of row c1), symmetric
    int cam_adj_words = 0;
    // nested dissection of the co-visibility graph (nd.hpp): elimination order of the reduced system and,
    // with several ranks, the domain of every rank (mg_subtree: object points follow their cameras' domain,
    // the domains factor locally and only the top separators' Schur complement is summed over the ranks)
    NdTree nd;
    bool mg_subtree = false;
    int max_k = 0;                                 // max observations of one point
    bool shared_eo = false;                        // EO.struct.block shares elements between images (camera stations):
                                                   // every shared element is ONE unknown in the slot of its leading
                                                   // entry (cam_col), the other entries' slots are inert
    bool rank_ok = true;                           // structural rank test
    // Heavy and giant points on the matrix cores (heavy.hpp).  The rows of the reduced system that these points touch
    // -- the EO rows of their cameras, the IO columns those cameras estimate, and one row for the right-hand side --
    // are cut into ROW GROUPS of at most 48 rows (eight cameras; three 16-row blocks of v_mfma_f64_16x16x4_f64).  A
    // point has a SLOT in every group it touches: 3 k-columns x (16 x row blocks) doubles of the scratch array Zs, which
    // k_heavy_z fills with the point's rows of Z = W R (V^-1 = R R').  The Schur complement of these points is then the
    // block-sparse product  S(Gi, Gj) -= sum over the points p in both groups  Z_p(Gi) Z_p(Gj)'  -- one list of slot
    // pairs per pair of groups, cut into tasks of at most hv_ks_per_task k-steps (k_heavy_syrk: one wave per task,
    // accumulators in registers, ONE flush per task).
    bool hv_ok = false;
    int64_t hv_obs0 = 0;                           // first observation (processing order) of the untiled points
    int32_t hv_pt0 = 0, hv_npts = 0;               // their range in the processing order
    int32_t hv_ngroups = 0, hv_ntasks = 0, hv_ks_per_task = 0;
    int64_t hv_z_doubles = 0;                      // size of Zs
    int64_t hv_mfma = 0;                           // v_mfma_f64_16x16x4_f64 instructions of one launch of k_heavy_syrk
    int64_t hv_alg_flops = 0;                      // sum over these points of 108 k + 216 k^2 (the full product Y W')
    int hv_max_io_slots = 0;                       // most IO columns of one point
    int hv_max_batch_slots = 0;                    // most IO slots of the points of one batch (k_heavy_z sums them in LDS)
    std::vector<int32_t> hv_grp_nb;                // [groups] 16-row blocks of the group (1 .. 3)
    std::vector<int32_t> hv_grp_row;               // [groups][48] row of the reduced system (-1: padding, NS: right-hand side)
    std::vector<int32_t> hv_obs_dst;               // [untiled observations] Zs index of (slot, k-column 0, first EO row of the camera)
    std::vector<uint8_t> hv_obs_ld;                // ... and the stride between the k-columns of that group (16 x row blocks)
    std::vector<uint8_t> hv_obs_ioloc;             // [untiled observations][HV_NIOC] slot of the camera's j-th IO column in the point's IO list
    std::vector<int32_t> hv_pt_io0;                // [points + 1] first IO slot of every point
    std::vector<int32_t> hv_io_dst;                // [IO slots] Zs index of (slot, k-column 0, row of the IO column)
    std::vector<uint8_t> hv_io_ld;
    std::vector<int32_t> hv_io_pt;                 // [IO slots] the point (index from hv_pt0)
    std::vector<int32_t> hv_pt_y;                  // [points][2] Zs index and stride of the point's right-hand-side row
    std::vector<int32_t> hv_task;                  // [tasks][4] group i, group j <= i, first k-step, k-steps
    std::vector<int32_t> hv_ops;                   // [k-steps][4][2] Zs index of the k-column lane group kk reads, in Gi and in Gj (-1: zero)
    static constexpr int HV_NIOC = 9;              // IO columns per camera at most (ncolmax <= 15)
    static constexpr int HV_MAXSLOTS = 48;         // IO columns per point at most
    int order_dims = 3;                            // dimensions of the point-ordering curve (2 = flat cloud)
    std::string err;
};

inline bool fail(Plan &P, const std::string &msg) { P.err = msg; return false; }

// serializeblock (buildserialindices.m:162-221) for one parameter array.
// Returns for every entry (column-major rows x cols) the index of its leading
// unknown inside the section (or -1 if fixed), and the list of leading entries.
inline void serialize_block(int rows, int cols, const int32_t *block, const uint8_t *est,
                            std::vector<int32_t> &dist, std::vector<int64_t> &lead_entries,
                            bool &simple) {
    dist.assign((size_t)rows * cols, -1);
    std::vector<uint8_t> leading((size_t)rows * cols, 0);
    simple = true;
    for (int i = 0; i < rows; ++i) {
        // first estimated occurrence of every non-zero block id in this row
        std::unordered_map<int32_t, int> seen;
        for (int j = 0; j < cols; ++j) {
            const size_t e = (size_t)j * rows + i;
            if (!est[e] || block[e] == 0) continue;
            if (seen.emplace(block[e], j).second) leading[e] = 1;
            else simple = false;
        }
    }
    // leading entries in column-major order define the x order (find(leading))
    lead_entries.clear();
    for (size_t e = 0; e < leading.size(); ++e)
        if (leading[e]) { dist[e] = (int32_t)lead_entries.size(); lead_entries.push_back((int64_t)e); }
    if (!simple) {
        for (int i = 0; i < rows; ++i) {
            std::unordered_map<int32_t, int32_t> lead_of;   // block id -> x index of its leading element
            for (int j = 0; j < cols; ++j) {
                const size_t e = (size_t)j * rows + i;
                if (leading[e]) lead_of[block[e]] = dist[e];
            }
            for (int j = 0; j < cols; ++j) {
                const size_t e = (size_t)j * rows + i;
                if (!est[e] || block[e] == 0 || leading[e]) continue;
                dist[e] = lead_of[block[e]];
            }
        }
    }
}

// Row groups, slots and pair tasks of the heavy / giant points (Plan::hv_*; kernels in heavy.hpp).  One thread: the
// lists are proportional to the observations of these points only, and every thread count must give the same plan.
inline void build_heavy_plan(Plan &P) {
    P.hv_ok = false; P.hv_ntasks = 0; P.hv_ngroups = 0; P.hv_z_doubles = 0; P.hv_mfma = 0;
    if (!P.CMAX || P.BT != 256 || P.shared_eo || P.ncolmax > 6 + Plan::HV_NIOC) return;
    if (env_int("DBAT_HIP_HEAVY", 1) == 0) return;
    const int64_t nb = (int64_t)P.batch_start.size() - 1;
    if (P.nb_tiled > nb) return;
    const int64_t ho0 = P.batch_start[P.nb_tiled], ho1 = (int64_t)P.o_cam.size();
    if (ho1 <= ho0) return;
    const int nc = P.nc;
    const int32_t hp0 = P.o_pt[ho0], hp1 = P.o_pt[ho1 - 1] + 1;
    const int32_t nhp = hp1 - hp0;
    // observation range of every point (a point's observations are contiguous, the points consecutive)
    std::vector<int64_t> pobs((size_t)nhp + 1, ho1);
    for (int64_t o = ho1 - 1; o >= ho0; --o) pobs[P.o_pt[o] - hp0] = o;
    for (int32_t i = nhp - 1; i >= 0; --i) if (pobs[i] > pobs[i + 1]) pobs[i] = pobs[i + 1];     // (a point without observations: empty range)
    // cameras and IO columns these points touch, ascending
    std::vector<int32_t> cam_idx((size_t)nc, -1), hcams;
    for (int64_t o = ho0; o < ho1; ++o) cam_idx[P.o_cam[o]] = 0;
    for (int c = 0; c < nc; ++c) if (cam_idx[c] == 0) { cam_idx[c] = (int32_t)hcams.size(); hcams.push_back(c); }
    std::vector<int32_t> io_idx((size_t)std::max(1, P.nIOu), -1), hio;
    for (int32_t c : hcams)
        for (int q = 6; q < P.cam_ncol[c]; ++q) io_idx[P.cam_col[(size_t)c * MAXCOL + q] - 6 * nc] = 0;
    for (int k = 0; k < P.nIOu; ++k) if (io_idx[k] == 0) { io_idx[k] = (int32_t)hio.size(); hio.push_back(k); }
    // groups: eight cameras each; then the IO columns and the row of the right-hand side, 48 rows each
    const int ngc = ((int)hcams.size() + 7) / 8;
    const int nior = (int)hio.size() + 1;            // IO rows + the right-hand side
    const int ngi = (nior + 47) / 48;
    const int ng = ngc + ngi;
    P.hv_ngroups = ng;
    P.hv_grp_nb.assign(ng, 0); P.hv_grp_row.assign((size_t)ng * 48, -1);
    for (size_t i = 0; i < hcams.size(); ++i)
        for (int a = 0; a < 6; ++a) P.hv_grp_row[(i / 8) * 48 + 6 * (i % 8) + a] = 6 * hcams[i] + a;
    for (int i = 0; i < nior; ++i)
        P.hv_grp_row[(size_t)(ngc + i / 48) * 48 + i % 48] = i < (int)hio.size() ? 6 * nc + hio[i] : (int32_t)P.NS;
    for (int g = 0; g < ng; ++g) {
        int rows = 0;
        for (int r = 0; r < 48; ++r) if (P.hv_grp_row[(size_t)g * 48 + r] >= 0) rows = r + 1;
        P.hv_grp_nb[g] = (rows + 15) / 16;
    }
    const int yg = ngc + (nior - 1) / 48, yrow = (nior - 1) % 48;
    // the groups of every point (ascending) with its slot in each; IO slots
    std::vector<int32_t> cnt((size_t)ng, 0);         // slots handed out per group
    std::vector<int64_t> pg0((size_t)nhp + 1, 0);    // prefix of the points' group lists
    std::vector<int32_t> pg_g, pg_slot;
    const int64_t nho = ho1 - ho0;
    P.hv_obs_dst.assign((size_t)nho, 0); P.hv_obs_ld.assign((size_t)nho, 16);
    P.hv_obs_ioloc.assign((size_t)nho * Plan::HV_NIOC, 255);
    P.hv_pt_io0.assign((size_t)nhp + 1, 0); P.hv_io_dst.clear(); P.hv_io_ld.clear(); P.hv_io_pt.clear();
    P.hv_pt_y.assign((size_t)2 * nhp, 0);
    P.hv_max_io_slots = 0;
    std::vector<int32_t> pio;                        // IO columns (indices into hio) of the current point, ascending
    std::vector<int32_t> grp_slot((size_t)ng, -1);   // slot of the current point in a group (-1: not touched)
    std::vector<int32_t> touched;
    std::vector<int32_t> io_slot_grp;               // group of every IO slot
    for (int32_t i = 0; i < nhp; ++i) {
        pio.clear(); touched.clear();
        for (int64_t o = pobs[i]; o < pobs[i + 1]; ++o) {
            const int32_t c = P.o_cam[o];
            const int g = cam_idx[c] / 8;
            if (grp_slot[g] < 0) { grp_slot[g] = cnt[g]++; touched.push_back(g); }
            for (int q = 6; q < P.cam_ncol[c]; ++q) pio.push_back(io_idx[P.cam_col[(size_t)c * MAXCOL + q] - 6 * nc]);
        }
        std::sort(pio.begin(), pio.end());
        pio.erase(std::unique(pio.begin(), pio.end()), pio.end());
        if ((int)pio.size() > Plan::HV_MAXSLOTS) return;          // (hv_ok stays false: the column-list kernels take these points)
        P.hv_max_io_slots = std::max(P.hv_max_io_slots, (int)pio.size());
        for (int32_t q : pio) { const int g = ngc + q / 48; if (grp_slot[g] < 0) { grp_slot[g] = cnt[g]++; touched.push_back(g); } }
        if (pobs[i + 1] > pobs[i] && grp_slot[yg] < 0) { grp_slot[yg] = cnt[yg]++; touched.push_back(yg); }
        std::sort(touched.begin(), touched.end());
        for (int32_t g : touched) { pg_g.push_back(g); pg_slot.push_back(grp_slot[g]); }
        pg0[i + 1] = (int64_t)pg_g.size();
        // (the Zs indices need the groups' offsets, which need the final slot counts: second pass below; here the
        // group-local parts)
        for (int64_t o = pobs[i]; o < pobs[i + 1]; ++o) {
            const int32_t c = P.o_cam[o];
            const int g = cam_idx[c] / 8, ld = 16 * P.hv_grp_nb[g];
            P.hv_obs_dst[o - ho0] = grp_slot[g] * 3 * ld + 6 * (cam_idx[c] % 8);
            P.hv_obs_ld[o - ho0] = (uint8_t)ld;
            for (int q = 6; q < P.cam_ncol[c]; ++q) {
                const int32_t io = io_idx[P.cam_col[(size_t)c * MAXCOL + q] - 6 * nc];
                P.hv_obs_ioloc[(size_t)(o - ho0) * Plan::HV_NIOC + (q - 6)] = (uint8_t)(std::lower_bound(pio.begin(), pio.end(), io) - pio.begin());
            }
        }
        for (int32_t q : pio) {
            const int g = ngc + q / 48, ld = 16 * P.hv_grp_nb[g];
            P.hv_io_dst.push_back(grp_slot[g] * 3 * ld + q % 48);
            P.hv_io_ld.push_back((uint8_t)ld); P.hv_io_pt.push_back(i); io_slot_grp.push_back(g);
        }
        P.hv_pt_io0[i + 1] = (int32_t)P.hv_io_dst.size();
        if (pobs[i + 1] > pobs[i]) { const int ld = 16 * P.hv_grp_nb[yg]; P.hv_pt_y[2 * i] = grp_slot[yg] * 3 * ld + yrow; P.hv_pt_y[2 * i + 1] = ld; }
        for (int32_t g : touched) grp_slot[g] = -1;
    }
    // offsets of the groups in Zs
    std::vector<int64_t> zoff((size_t)ng + 1, 0);
    for (int g = 0; g < ng; ++g) zoff[g + 1] = zoff[g] + (int64_t)cnt[g] * 3 * 16 * P.hv_grp_nb[g];
    if (zoff[ng] >= (int64_t)1 << 31) return;          // (32-bit indices)
    P.hv_z_doubles = zoff[ng];
    for (int32_t i = 0; i < nhp; ++i) {
        for (int64_t o = pobs[i]; o < pobs[i + 1]; ++o) P.hv_obs_dst[o - ho0] += (int32_t)zoff[cam_idx[P.o_cam[o]] / 8];
        if (pobs[i + 1] > pobs[i]) P.hv_pt_y[2 * i] += (int32_t)zoff[yg];
    }
    for (size_t q = 0; q < P.hv_io_dst.size(); ++q) P.hv_io_dst[q] += (int32_t)zoff[io_slot_grp[q]];
    // pair lists: every point contributes (slot_i, slot_j) to every pair of its groups, i >= j
    std::unordered_map<uint64_t, int32_t> pair_id;
    std::vector<int32_t> pair_g;                      // [pairs][2]
    std::vector<int64_t> pair_n;
    for (int32_t i = 0; i < nhp; ++i)
        for (int64_t a = pg0[i]; a < pg0[i + 1]; ++a)
            for (int64_t b = pg0[i]; b <= a; ++b) {
                const uint64_t key = (uint64_t)pg_g[a] * (uint64_t)ng + (uint64_t)pg_g[b];
                auto it = pair_id.find(key);
                if (it == pair_id.end()) { it = pair_id.emplace(key, (int32_t)pair_n.size()).first; pair_g.push_back(pg_g[a]); pair_g.push_back(pg_g[b]); pair_n.push_back(0); }
                ++pair_n[it->second];
            }
    const size_t npairs = pair_n.size();
    // a fixed order of the pairs (the hash map's is not one): by (group i, group j)
    std::vector<int32_t> pord(npairs);
    std::iota(pord.begin(), pord.end(), 0);
    std::sort(pord.begin(), pord.end(), [&](int32_t a, int32_t b) {
        return pair_g[2 * a] != pair_g[2 * b] ? pair_g[2 * a] < pair_g[2 * b] : pair_g[2 * a + 1] < pair_g[2 * b + 1];
    });
    std::vector<int32_t> prank(npairs);
    for (size_t r = 0; r < npairs; ++r) prank[pord[r]] = (int32_t)r;
    std::vector<int64_t> pk0(npairs + 1, 0);          // first k-step of every pair (in the fixed order)
    for (size_t r = 0; r < npairs; ++r) pk0[r + 1] = pk0[r] + (3 * pair_n[pord[r]] + 3) / 4;
    const int64_t total_ks = pk0[npairs];
    if (total_ks * 8 >= (int64_t)1 << 31) return;
    P.hv_ops.assign((size_t)total_ks * 8, -1);
    std::vector<int64_t> fill(npairs, 0);             // points placed so far
    std::vector<std::vector<int32_t>> pair_first(npairs);      // the point of every entry of a pair's list (ascending)
    for (size_t r = 0; r < npairs; ++r) pair_first[r].reserve((size_t)pair_n[pord[r]]);
    for (int32_t i = 0; i < nhp; ++i)
        for (int64_t a = pg0[i]; a < pg0[i + 1]; ++a)
            for (int64_t b = pg0[i]; b <= a; ++b) {
                const int32_t r = prank[pair_id[(uint64_t)pg_g[a] * (uint64_t)ng + (uint64_t)pg_g[b]]];
                const int64_t e = fill[r]++;
                pair_first[r].push_back(i);
                const int gi = pg_g[a], gj = pg_g[b];
                const int ldi = 16 * P.hv_grp_nb[gi], ldj = 16 * P.hv_grp_nb[gj];
                for (int c = 0; c < 3; ++c) {
                    const int64_t q = 3 * e + c;      // k-column of the pair: k-step q / 4, lane group q % 4
                    int32_t *op = P.hv_ops.data() + ((pk0[r] + q / 4) * 4 + q % 4) * 2;
                    op[0] = (int32_t)(zoff[gi] + ((int64_t)pg_slot[a] * 3 + c) * ldi);
                    op[1] = (int32_t)(zoff[gj] + ((int64_t)pg_slot[b] * 3 + c) * ldj);
                }
            }
    // tasks: the pairs' k-steps in pieces of about equal length.  Enough of them to occupy the chip, long enough to
    // pay for the flush (one atomic per element of the pair's blocks)
    int ksmax = env_int("DBAT_HIP_HEAVY_KS", 0);
    if (ksmax <= 0) ksmax = (int)std::min<int64_t>(96, std::max<int64_t>(12, (total_ks + 4095) / 4096));
    P.hv_ks_per_task = ksmax;
    P.hv_task.clear(); P.hv_mfma = 0;
    std::vector<int64_t> task_key;                    // position of the task's first k-step in its pair's list, as a fraction of the point range
    for (size_t r = 0; r < npairs; ++r) {
        const int gi = pair_g[2 * pord[r]], gj = pair_g[2 * pord[r] + 1];
        const int64_t nks = pk0[r + 1] - pk0[r];
        const int64_t nt = (nks + ksmax - 1) / ksmax;
        const int nbi = P.hv_grp_nb[gi], nbj = P.hv_grp_nb[gj];
        P.hv_mfma += nks * (gi == gj ? nbi * (nbi + 1) / 2 : nbi * nbj);
        for (int64_t q = 0; q < nt; ++q) {
            const int64_t k0 = q * nks / nt, k1 = (q + 1) * nks / nt;
            const int32_t tk[4] = {gi, gj, (int32_t)(pk0[r] + k0), (int32_t)(k1 - k0)};
            P.hv_task.insert(P.hv_task.end(), tk, tk + 4);
            task_key.push_back(pair_first[r][(size_t)std::min<int64_t>(4 * k0 / 3, (int64_t)pair_first[r].size() - 1)]);
        }
    }
    {   // Task order: by the first point of the task, then by group pair -- tasks that read the same slots (one run of points,
        // every pair of its groups) are neighbours.  k_heavy_syrk maps consecutive workgroups onto ONE XCD (its blockIdx
        // remap), so a slot is fetched into that XCD's L2 once and read from there by its other partner groups (dense
        // scene, 48 images x 16 384 points: every slot has seven partners; without the order the kernel pulled 784 MB
        // per launch through the fabric, bound by that and not by the matrix pipe).  The tasks are about equal in length
        // (hv_ks_per_task), so nothing is lost by giving up "longest first".
        const size_t nt = P.hv_task.size() / 4;
        std::vector<int32_t> ord(nt);
        std::iota(ord.begin(), ord.end(), 0);
        std::stable_sort(ord.begin(), ord.end(), [&](int32_t a, int32_t b) {
            if (task_key[a] != task_key[b]) return task_key[a] < task_key[b];
            if (P.hv_task[4 * a] != P.hv_task[4 * b]) return P.hv_task[4 * a] < P.hv_task[4 * b];
            return P.hv_task[4 * a + 1] < P.hv_task[4 * b + 1];
        });
        std::vector<int32_t> tk(P.hv_task.size());
        for (size_t a = 0; a < nt; ++a) std::copy(P.hv_task.begin() + 4 * ord[a], P.hv_task.begin() + 4 * ord[a] + 4, tk.begin() + 4 * a);
        P.hv_task.swap(tk);
    }
    P.hv_ntasks = (int32_t)(P.hv_task.size() / 4);
    P.hv_obs0 = ho0; P.hv_pt0 = hp0; P.hv_npts = nhp;
    P.hv_max_batch_slots = 0;
    for (int64_t b = P.nb_tiled; b < nb; ++b) {
        const int64_t b0 = P.batch_start[b], b1 = P.batch_start[b + 1];
        if (b1 <= b0) continue;
        const int32_t p0 = P.o_pt[b0] - hp0, p1 = P.o_pt[b1 - 1] - hp0 + 1;
        P.hv_max_batch_slots = std::max(P.hv_max_batch_slots, P.hv_pt_io0[p1] - P.hv_pt_io0[p0]);
    }
    if (P.hv_max_batch_slots > 2048) { P.hv_ok = false; return; }      // (48 KB of LDS for the sums: beyond that the column lists)
    P.hv_alg_flops = 0;
    for (int32_t i = 0; i < nhp; ++i) { const int64_t k = pobs[i + 1] - pobs[i]; P.hv_alg_flops += 108 * k + 216 * k * k; }
    P.hv_ok = P.hv_ntasks > 0;
    if (env_on("DBAT_HIP_PLAN_STATS"))
        fprintf(stderr, "[plan] heavy / giant points on the matrix cores: %d points, %lld observations, %d row groups (%zu cameras, %zu IO columns), "
                        "%zu group pairs, %lld k-steps in %d tasks (<= %d each), Zs %.2f MB\n",
                nhp, (long long)nho, ng, hcams.size(), hio.size(), npairs, (long long)total_ks, P.hv_ntasks, ksmax, P.hv_z_doubles * 8e-6);
}

// ---- plan reuse (dbat_hip_structure_key / dbat_hip_set_values) --------------------------------------------------------
// What a plan depends on -- everything of the problem but the parameter VALUES (IO, EO, OP) and the values / standard
// deviations of the prior observations: sizes, lens model, visibility (ip_cam, ip_pt), the image observations and their
// standard deviations, pixel sizes, the estimation masks, the block structure, which parameters have prior observations,
// the shard, the device, and the DBAT_HIP_* switches of the environment.  Two problems with the same key differ at most in
// those values: the reference re-enters bundle() from any s at no set-up cost (misc/deserialize.m:31-46); here a handle is
// re-used through dbat_hip_set_values.  128 bits, the same for any number of threads (fixed blocks, combined in order).
inline void hash_words(const void *data, size_t bytes, uint64_t &h0, uint64_t &h1) {
    const unsigned char *p = static_cast<const unsigned char *>(data);
    uint64_t a = h0 ^ 0x9E3779B97F4A7C15ull, b = h1 ^ 0xC2B2AE3D27D4EB4Full, c = h0 + 0x165667B19E3779F9ull, d = h1 + 0x27D4EB2F165667C5ull;
    auto mix = [](uint64_t h, uint64_t w, uint64_t k) { h ^= w * k; h = ((h << 31) | (h >> 33)) * 0x9FB21C651E98DF25ull; return h; };
    size_t i = 0;
    for (; i + 16 <= bytes; i += 16) {
        uint64_t w0, w1;
        memcpy(&w0, p + i, 8); memcpy(&w1, p + i + 8, 8);
        a = mix(a, w0, 0xA0761D6478BD642Full); b = mix(b, w0, 0xE7037ED1A0B428DBull);
        c = mix(c, w1, 0x8EBC6AF09C88C6E3ull); d = mix(d, w1, 0x589965CC75374CC3ull);
    }
    uint64_t tail[2] = {0, 0};
    if (i < bytes) memcpy(tail, p + i, bytes - i);
    a = mix(a, tail[0] ^ (uint64_t)bytes, 0xA0761D6478BD642Full); b = mix(b, tail[1] + (uint64_t)bytes, 0xE7037ED1A0B428DBull);
    h0 = mix(a, c, 0x1D8E4E27C47D124Full); h1 = mix(b, d, 0xEB44ACCAB455D165ull);
    h0 ^= h0 >> 29; h1 ^= h1 >> 32;
}
inline void structure_key(const dbat_hip_problem &pb, uint64_t key[2]) {
    uint64_t h0 = 0x0123456789ABCDEFull, h1 = 0xFEDCBA9876543210ull;
    const Par par{Par::default_threads()};
    auto field = [&](const void *data, size_t bytes, uint64_t tag) {
        const uint64_t hdr[3] = {tag, (uint64_t)bytes, data ? 1ull : 0ull};
        hash_words(hdr, sizeof(hdr), h0, h1);
        if (!data || !bytes) return;
        constexpr size_t BLK = (size_t)1 << 20;
        const int64_t nblk = (int64_t)((bytes + BLK - 1) / BLK);
        if (nblk <= 1) { hash_words(data, bytes, h0, h1); return; }
        std::vector<uint64_t> bh((size_t)2 * nblk);
        par.run(nblk, [&](int64_t lo, int64_t hi, int) {
            for (int64_t q = lo; q < hi; ++q) {
                uint64_t a = (uint64_t)q, b = ~(uint64_t)q;
                hash_words(static_cast<const unsigned char *>(data) + (size_t)q * BLK, std::min(BLK, bytes - (size_t)q * BLK), a, b);
                bh[2 * q] = a; bh[2 * q + 1] = b;
            }
        }, 1);
        hash_words(bh.data(), bh.size() * 8, h0, h1);
    };
    const int64_t nc = pb.n_images, np = pb.n_points, no = pb.n_obs, R = 5 + (int64_t)pb.nK + pb.nP;
    const int64_t hdr[10] = {pb.abi_version, nc, np, no, pb.dist_model, pb.nK, pb.nP, pb.device, pb.shard_rank, pb.shard_count};
    field(hdr, sizeof(hdr), 1);
    if (nc > 0 && np > 0 && no >= 0 && R >= 5 && R <= MAXIO) {
        field(pb.ip_cam, (size_t)no * 4, 2); field(pb.ip_pt, (size_t)no * 4, 3);
        field(pb.ip_val, (size_t)no * 16, 4); field(pb.ip_std, (size_t)no * 16, 5);
        field(pb.px_size, (size_t)nc * 16, 6);
        field(pb.est_IO, (size_t)(R * nc), 7); field(pb.est_EO, (size_t)(6 * nc), 8); field(pb.est_OP, (size_t)(3 * np), 9);
        field(pb.IO_block, (size_t)(R * nc) * 4, 10); field(pb.EO_block, (size_t)(6 * nc) * 4, 11);
        field(pb.prior_IO_use, (size_t)(R * nc), 12); field(pb.prior_EO_use, (size_t)(6 * nc), 13); field(pb.prior_OP_use, (size_t)(3 * np), 14);
    }
    // the switches of the environment select layouts and kernels: part of the structure
    std::vector<std::string> envs;
    for (char **e = environ; e && *e; ++e) if (strncmp(*e, "DBAT_HIP_", 9) == 0) envs.emplace_back(*e);
    std::sort(envs.begin(), envs.end());
    for (const std::string &e : envs) field(e.data(), e.size(), 15);
    key[0] = h0; key[1] = h1;
}

// New parameter values and prior observations for a plan of the same structure: z0, the fixed IO values, the prior
// weights and values -- exactly what build_plan derives from IO_val, EO_val, OP_val and prior_*_val / prior_*_std.
inline bool plan_set_values(const dbat_hip_problem &pb, Plan &P) {
    const int nc = P.nc, np = P.np, R = P.nIOrows;
    const Par par{Par::default_threads()};
    for (size_t e = 0; e < (size_t)6 * nc; ++e) P.z0[e] = pb.EO_val[e];
    for (int k = 0; k < P.nIOu; ++k) P.z0[6 * (int64_t)nc + k] = pb.IO_val[P.lead_io[k]];
    par.run(np, [&](int64_t plo, int64_t phi, int) {
        for (int64_t p = plo; p < phi; ++p)
            for (int d = 0; d < 3; ++d) P.z0[P.NS + 3 * (int64_t)P.pt_rank[p] + d] = pb.OP_val[3 * p + d];
    });
    P.io_fixed.assign(pb.IO_val, pb.IO_val + (size_t)R * nc);
    bool bad = false;
    auto prior = [&](int64_t z, double val, double std_) {
        const double w = 1.0 / (std_ * std_);
        if (!(w > 0) || !std::isfinite(w)) bad = true;
        P.z_prw[z] = w; P.z_prv[z] = val;
    };
    // (which entries have prior observations is structure: z_prw > 0 marks them)
    for (int k = 0; k < P.nIOu; ++k) {
        const int64_t e = P.lead_io[k], z = 6 * (int64_t)nc + k;
        if (P.z_prw[z] > 0) prior(z, pb.prior_IO_val[e], pb.prior_IO_std[e]);
    }
    for (size_t e = 0; e < (size_t)6 * nc; ++e) if (P.z_prw[e] > 0) prior((int64_t)e, pb.prior_EO_val[e], pb.prior_EO_std[e]);
    if (P.n_prior[2] > 0)
        for (int64_t p = 0; p < np; ++p)
            for (int d = 0; d < 3; ++d) {
                const int64_t z = P.NS + 3 * (int64_t)P.pt_rank[p] + d;
                if (P.z_prw[z] > 0) prior(z, pb.prior_OP_val[3 * p + d], pb.prior_OP_std[3 * p + d]);
            }
    if (bad) return fail(P, "prior observation with zero/invalid std");
    return true;
}

inline bool build_plan(const dbat_hip_problem &pb, Plan &P, bool with_obs) {
    // DBAT_HIP_PLAN_STATS=2: wall time of every section of the plan (stderr)
    const bool plan_clock = env_int("DBAT_HIP_PLAN_STATS", 0) >= 2;
    auto plan_t0 = std::chrono::steady_clock::now();
    auto lapt = [&](const char *what) {
        if (!plan_clock) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[plan clock] %-60s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - plan_t0).count());
        plan_t0 = now;
    };
    if (pb.abi_version != DBAT_HIP_ABI_VERSION) return fail(P, "ABI version mismatch");
    { std::string env_err; if (!env_validate(env_err)) return fail(P, env_err.c_str()); }
    if (pb.n_images <= 0 || pb.n_points <= 0 || pb.n_obs < 0) return fail(P, "empty problem");
    if (with_obs) structure_key(pb, P.key);
    if (pb.dist_model < 2 || pb.dist_model > 5)
        return fail(P, "lens distortion model must be 2..5 (brown_euler_cam4.m:122-130)");
    if (pb.nK < 0 || pb.nK > MAXK || pb.nP < 0 || pb.nP > MAXP || pb.nP == 1)
        return fail(P, "nK/nP out of supported range");
    P.nc = pb.n_images; P.np = pb.n_points; P.no = pb.n_obs;
    P.nK = pb.nK; P.nP = pb.nP; P.model = pb.dist_model;
    P.nIOrows = 5 + pb.nK + pb.nP;
    P.rank = pb.shard_rank; P.nranks = std::max(1, pb.shard_count);
    if (P.rank < 0 || P.rank >= P.nranks) return fail(P, "bad shard rank");
    const int nc = P.nc, np = P.np, R = P.nIOrows;
    const Par par{Par::default_threads()};

    // ---- est / prior masks; bundle.m:137-154
    std::vector<uint8_t> estIO(pb.est_IO, pb.est_IO + (size_t)R * nc);
    std::vector<uint8_t> estEO(pb.est_EO, pb.est_EO + (size_t)6 * nc);
    std::vector<uint8_t> estOP(pb.est_OP, pb.est_OP + (size_t)3 * np);
    // K estimation mask must be a leading run; P1,P2 together (multi_res.m:179-186,202-209)
    for (int c = 0; c < nc; ++c) {
        const uint8_t *e = &estIO[(size_t)c * R];
        bool gap = false;
        for (int j = 0; j < P.nK; ++j) { if (!e[5 + j]) gap = true; else if (gap) return fail(P, "Illegal cK vector"); }
        gap = false;
        for (int j = 0; j < P.nP; ++j) { if (!e[5 + P.nK + j]) gap = true; else if (gap) return fail(P, "Illegal cP vector"); }
        if (P.nP >= 2 && (e[5 + P.nK] != e[5 + P.nK + 1])) return fail(P, "Illegal cP vector");
    }
    lapt("est / prior masks; bundle.m:137-154");
    // ---- serial indices
    std::vector<int32_t> distIO, distEO;
    std::vector<int64_t> leadIO, leadEO;
    bool simpleIO, simpleEO;
    serialize_block(R, nc, pb.IO_block, estIO.data(), distIO, leadIO, simpleIO);
    serialize_block(6, nc, pb.EO_block, estEO.data(), distEO, leadEO, simpleEO);
    P.shared_eo = !simpleEO;
    // z slot of every EO entry: its own, or -- buildserialindices.m:204-221 -- that of the leading
    // entry of its block (the first estimated element with the same block id in the same row)
    std::vector<int64_t> eo_z((size_t)6 * nc);
    std::vector<uint8_t> eo_lead((size_t)6 * nc, 0);
    for (size_t e = 0; e < eo_z.size(); ++e) {
        eo_z[e] = distEO[e] >= 0 ? leadEO[distEO[e]] : (int64_t)e;
        eo_lead[e] = distEO[e] >= 0 && leadEO[distEO[e]] == (int64_t)e;
    }
    for (size_t e = 0; e < estEO.size(); ++e)
        if (estEO[e] && pb.EO_block[e] == 0) return fail(P, "estimated EO element with block id 0");
    for (size_t e = 0; e < estIO.size(); ++e)
        if (estIO[e] && pb.IO_block[e] == 0) return fail(P, "estimated IO element with block id 0");
    P.nIOu = (int)leadIO.size();
    P.lead_io = leadIO;
    P.nIO = P.nIOu; P.nEO = (int64_t)leadEO.size();
    P.NS = (int64_t)6 * nc + P.nIOu;
    P.NZ = P.NS + (int64_t)3 * np;
    P.io_src = distIO;
    P.io_fixed.assign(pb.IO_val, pb.IO_val + (size_t)R * nc);
    P.z_est.assign(P.NZ, 0); P.z_prw.assign(P.NZ, 0.0); P.z_prv.assign(P.NZ, 0.0);
    P.z0.assign(P.NZ, 0.0);
    for (size_t e = 0; e < (size_t)6 * nc; ++e) { P.z0[e] = pb.EO_val[e]; P.z_est[e] = eo_lead[e]; }
    for (int k = 0; k < P.nIOu; ++k) { P.z0[6 * (int64_t)nc + k] = pb.IO_val[leadIO[k]]; P.z_est[6 * (int64_t)nc + k] = 1; }
    par.run((int64_t)3 * np, [&](int64_t lo_, int64_t hi_, int) {
        for (int64_t e = lo_; e < hi_; ++e) { P.z0[P.NS + e] = pb.OP_val[e]; P.z_est[P.NS + e] = estOP[e]; }
    });
    // x order: IO leading (column-major), EO est (column-major), OP est (column-major)
    P.x2z.clear();
    P.x2z.reserve((size_t)P.NZ);
    for (int k = 0; k < P.nIOu; ++k) P.x2z.push_back(6 * (int64_t)nc + k);
    for (size_t e = 0; e < (size_t)6 * nc; ++e) if (eo_lead[e]) P.x2z.push_back((int64_t)e);
    {   // the estimated OP coordinates in order: counts per range, then every range fills its slice
        const int nr = par.ranges((int64_t)3 * np, 1 << 16);
        std::vector<int64_t> cnt((size_t)nr + 1, 0);
        const Par pr{nr};
        pr.run((int64_t)3 * np, [&](int64_t lo_, int64_t hi_, int tid) {
            int64_t c = 0;
            for (int64_t e = lo_; e < hi_; ++e) c += estOP[e] != 0;
            cnt[tid + 1] = c;
        }, 1);
        const int64_t base = (int64_t)P.x2z.size();
        for (int t = 0; t < nr; ++t) cnt[t + 1] += cnt[t];
        P.x2z.resize((size_t)(base + cnt[nr]));
        pr.run((int64_t)3 * np, [&](int64_t lo_, int64_t hi_, int tid) {
            int64_t q = base + cnt[tid];
            for (int64_t e = lo_; e < hi_; ++e) if (estOP[e]) P.x2z[q++] = P.NS + e;
        }, 1);
    }
    P.n = (int64_t)P.x2z.size();
    P.nOP = P.n - P.nIO - P.nEO;
    // priors: use & est & leading (bundle.m:137-154, buildserialindices.m:138-139)
    P.prior_z.clear();
    auto add_prior = [&](int64_t z, double val, double std_) {
        P.z_prw[z] = 1.0 / (std_ * std_); P.z_prv[z] = val; P.prior_z.push_back(z);
    };
    for (int k = 0; k < P.nIOu; ++k) {
        const int64_t e = leadIO[k];
        if (pb.prior_IO_use && pb.prior_IO_use[e]) { add_prior(6 * (int64_t)nc + k, pb.prior_IO_val[e], pb.prior_IO_std[e]); P.n_prior[0]++; }
    }
    for (size_t e = 0; e < (size_t)6 * nc; ++e)
        if (eo_lead[e] && pb.prior_EO_use && pb.prior_EO_use[e]) { add_prior((int64_t)e, pb.prior_EO_val[e], pb.prior_EO_std[e]); P.n_prior[1]++; }
    for (size_t e = 0; e < (size_t)3 * np; ++e)
        if (estOP[e] && pb.prior_OP_use && pb.prior_OP_use[e]) { add_prior(P.NS + (int64_t)e, pb.prior_OP_val[e], pb.prior_OP_std[e]); P.n_prior[2]++; }
    for (int64_t z : P.prior_z)
        if (!(P.z_prw[z] > 0) || !std::isfinite(P.z_prw[z])) return fail(P, "prior observation with zero/invalid std");
    P.m = 2 * P.no + P.n_prior[0] + P.n_prior[1] + P.n_prior[2];

    lapt("serial indices");
    // ---- per-camera column lists
    P.px.assign(pb.px_size, pb.px_size + (size_t)2 * nc);
    P.cam_ncol.assign(nc, 6); P.cam_col.assign((size_t)nc * MAXCOL, -1);
    P.cam_iorow.assign((size_t)nc * MAXIO, 0); P.cam_eo_est.assign(nc, 0);
    // shared EO elements take the general (column-list) kernels, as estimated IO does
    P.ncolmax = 6; P.with_io = P.nIOu > 0 || P.shared_eo;
    for (int c = 0; c < nc; ++c) {
        uint32_t m = 0;
        for (int k = 0; k < 6; ++k) {
            P.cam_col[(size_t)c * MAXCOL + k] = (int32_t)eo_z[(size_t)6 * c + k];
            if (estEO[(size_t)c * 6 + k]) m |= 1u << k;
        }
        P.cam_eo_est[c] = m;
        int ncol = 6;
        for (int r = 0; r < R; ++r) {
            const int32_t src = distIO[(size_t)c * R + r];
            if (src >= 0) {
                P.cam_col[(size_t)c * MAXCOL + ncol] = 6 * nc + src;
                P.cam_iorow[(size_t)c * MAXIO + (ncol - 6)] = r;
                ++ncol;
            }
        }
        P.cam_ncol[c] = ncol;
        P.ncolmax = std::max(P.ncolmax, ncol);
    }
    {   // the usual self-calibration in every camera?  (same test as k_cam_prep's bit 8 of eo_est)
        static const int std8[8] = {0, 1, 2, 5, 6, 7, 8, 9};
        P.all_std8 = P.nIOu > 0 && P.nK == 3 && P.nP == 2;
        for (int c = 0; c < nc && P.all_std8; ++c) {
            if (P.cam_ncol[c] != 14) { P.all_std8 = false; break; }
            for (int k = 0; k < 8; ++k) if (P.cam_iorow[(size_t)c * MAXIO + k] != std8[k]) P.all_std8 = false;
        }
    }

    lapt("per-camera column lists");
    // ---- observations: validate order, weights (parallel over ranges of the IP columns; the first offence in
    // column order is the one reported, as a single pass would)
    std::vector<int32_t> k_pt(np, 0), n_cam(nc, 0);
    {
        std::atomic<int64_t> first_range{INT64_MAX}, first_order{INT64_MAX}, first_std{INT64_MAX};
        auto lower = [](std::atomic<int64_t> &a, int64_t v) { int64_t c = a.load(); while (v < c && !a.compare_exchange_weak(c, v)) {} };
        par.run(P.no, [&](int64_t lo, int64_t hi, int) {
            int32_t run_cam = -1; int32_t run_n = 0;
            for (int64_t o = lo; o < hi; ++o) {
                const int32_t c = pb.ip_cam[o], p = pb.ip_pt[o];
                if (c < 0 || c >= nc || p < 0 || p >= np) { lower(first_range, o); break; }
                if (o > 0) {
                    const int32_t c0 = pb.ip_cam[o - 1], p0 = pb.ip_pt[o - 1];
                    if (c < c0 || (c == c0 && p <= p0)) lower(first_order, o);
                }
                if (!(pb.ip_std[2 * o] > 0) || !(pb.ip_std[2 * o + 1] > 0)) lower(first_std, o);
                __atomic_fetch_add(&k_pt[p], 1, __ATOMIC_RELAXED);
                if (c != run_cam) { if (run_n) __atomic_fetch_add(&n_cam[run_cam], run_n, __ATOMIC_RELAXED); run_cam = c; run_n = 0; }
                ++run_n;
            }
            if (run_n) __atomic_fetch_add(&n_cam[run_cam], run_n, __ATOMIC_RELAXED);
        });
        const int64_t fr = first_range.load(), fo = first_order.load();
        if (fr <= fo && fr != INT64_MAX) return fail(P, "IP.cam / IP.pt out of range");
        if (fo != INT64_MAX) return fail(P, "IP columns must be image-major with ascending OP index (prob2dbatstruct.m:349-365)");
        P.max_k = 0;
        for (int p = 0; p < np; ++p) P.max_k = std::max(P.max_k, k_pt[p]);
        // uniform sigma per camera?  sigma_mm = IP.std .* pxSize(:,cam)  (buildweightmatrix.m:20).  The columns are
        // image-major: a camera's first observation is the start of its range.
        std::vector<int64_t> cam_obs_first(nc + 1, 0);
        for (int c = 0; c < nc; ++c) cam_obs_first[c + 1] = cam_obs_first[c] + n_cam[c];
        std::atomic<int64_t> first_diff{INT64_MAX};
        par.run(P.no, [&](int64_t lo, int64_t hi, int) {
            for (int64_t o = lo; o < hi; ++o) {
                const int32_t c = pb.ip_cam[o];
                const int64_t f = cam_obs_first[c];
                const double wu = 1.0 / (pb.ip_std[2 * o] * P.px[2 * c]), wv = 1.0 / (pb.ip_std[2 * o + 1] * P.px[2 * c + 1]);
                const double fu = 1.0 / (pb.ip_std[2 * f] * P.px[2 * c]), fv = 1.0 / (pb.ip_std[2 * f + 1] * P.px[2 * c + 1]);
                if (wu != fu || wv != fv) { lower(first_diff, o); break; }
            }
        });
        const int64_t fd = first_diff.load();
        P.uniform_w = fd == INT64_MAX;
        P.cam_w.assign((size_t)2 * nc, 0.0);
        for (int c = 0; c < nc; ++c) {               // (cameras first seen before the first differing observation, as a single pass leaves it)
            const int64_t f = cam_obs_first[c];
            if (n_cam[c] > 0 && f < fd) { P.cam_w[2 * c] = 1.0 / (pb.ip_std[2 * f] * P.px[2 * c]); P.cam_w[2 * c + 1] = 1.0 / (pb.ip_std[2 * f + 1] * P.px[2 * c + 1]); }
        }
        if (first_std.load() != INT64_MAX) return fail(P, "IP.std must be positive");
    }

    lapt("observations: validate order, weights");
    // ---- structural rank test (sprank(J), gauss_newton_armijo.m:132-142).  First the cheap
    // necessary conditions on the natural parameter groups; the exact matching follows below.
    P.rank_ok = P.m >= P.n;
    for (int p = 0; p < np && P.rank_ok; ++p) {
        int e = 0, pr = 0;
        for (int d = 0; d < 3; ++d) { e += estOP[(size_t)3 * p + d]; pr += P.z_prw[P.NS + 3 * (int64_t)p + d] > 0; }
        if (e > 2 * k_pt[p] + pr) P.rank_ok = false;
        if (e > 0 && k_pt[p] == 0 && pr < e) P.rank_ok = false;
    }
    for (int c = 0; c < nc && P.rank_ok; ++c) {
        int e = 0, pr = 0;
        for (int d = 0; d < 6; ++d) { e += estEO[(size_t)6 * c + d]; pr += P.z_prw[6 * (int64_t)c + d] > 0; }
        if (e > 2 * n_cam[c] + pr) P.rank_ok = false;
    }
    {
        std::vector<int64_t> rows(P.nIOu, 0);
        for (int c = 0; c < nc; ++c)
            for (int r = 0; r < R; ++r) { const int32_t s = distIO[(size_t)c * R + r]; if (s >= 0) rows[s] += 2 * (int64_t)n_cam[c]; }
        for (int k = 0; k < P.nIOu && P.rank_ok; ++k)
            if (rows[k] == 0 && !(P.z_prw[6 * (int64_t)nc + k] > 0)) P.rank_ok = false;
    }

    lapt("structural rank test (sprank(J), gauss_newton_armijo.m:132");
    // ---- processing order of the object points
    std::vector<int64_t> pstart(np + 1, 0);
    for (int p = 0; p < np; ++p) pstart[p + 1] = pstart[p] + k_pt[p];
    // by_pt: the IP columns of every point, ascending (image-major scan => cameras ascend inside a point); cbp: their
    // cameras.  A stable counting sort in parallel: every thread owns a range of POINTS and scans all columns.
    uvec<int64_t> by_pt(P.no);
    uvec<int32_t> cbp(P.no);
    par.run(np, [&](int64_t plo, int64_t phi, int) {
        std::vector<int32_t> cur((size_t)(phi - plo), 0);
        for (int64_t o = 0; o < P.no; ++o) {
            const int32_t p = pb.ip_pt[o];
            if (p < plo || p >= phi) continue;
            const int64_t q = pstart[p] + cur[p - plo]++;
            by_pt[q] = o; cbp[q] = pb.ip_cam[o];
        }
    }, 1 << 15);
    lapt("observations by point (counting sort)");
    // ---- exact structural rank (sprank(J) < n  =>  code -4): when the counting conditions
    // above hold, a maximum matching of the unknowns to the rows of J decides.  An unknown
    // with a prior observation owns that row.  The others are matched to image rows (two per
    // observation) -- IO columns first, then EO, then OP, each to the first free row it has
    // ("cheap assignment", which settles all of them in a healthy network) and otherwise
    // along an augmenting path.  J is never formed: the rows of an EO or IO column are the
    // contiguous observation ranges of its cameras, those of an OP column the observations
    // of its point.
    // The healthy network first, in parallel: every object point claims as many rows among its OWN observations as
    // it has unknowns without a prior (rows of different points are disjoint: a bitmap of taken rows and an atomic
    // OR per claim are all it takes; the observations are tried from a point-dependent start, so that the claims spread
    // over the point's cameras), then the IO and EO columns take the first rows that are still free in their cameras.
    // If everybody is served that IS a perfect matching, hence sprank = n.  Anything else is left to the exact
    // matching below, from scratch: the answer is the existence of a perfect matching either way.
    bool sprank_settled = false;
    if (P.rank_ok && !env_on("DBAT_HIP_SPRANK_OFF") && !P.shared_eo) {
        std::vector<int64_t> cam_obs0(nc + 1, 0);
        for (int c = 0; c < nc; ++c) cam_obs0[c + 1] = cam_obs0[c] + n_cam[c];
        std::vector<uint64_t> taken((size_t)(2 * P.no + 63) / 64 + 1, 0);
        auto wanted = [&](int64_t col) -> bool { return P.z_est[col] && !(P.z_prw[col] > 0); };
        std::atomic<int> short_pts{0};
        par.run(np, [&](int64_t plo, int64_t phi, int) {
            for (int64_t p = plo; p < phi; ++p) {
                int want = 0;
                for (int d = 0; d < 3; ++d) want += wanted(P.NS + 3 * p + d) ? 1 : 0;
                const int k = k_pt[p];
                for (int dd = 0; dd < 2 * k && want > 0; ++dd) {         // one row per observation first, then the second rows
                    const int j = (int)((p + dd) % k);
                    const int64_t row = 2 * by_pt[pstart[p] + j] + (dd >= k ? 1 : 0);
                    __atomic_fetch_or(&taken[row >> 6], 1ull << (row & 63), __ATOMIC_RELAXED);
                    --want;
                }
                if (want > 0) { short_pts.fetch_add(1); return; }
            }
        });
        bool cheap_ok = short_pts.load() == 0;
        auto claim = [&](int c, int64_t &row) -> bool {                  // the next free row of camera c at or after `row`
            while (row < 2 * cam_obs0[c + 1] && ((taken[row >> 6] >> (row & 63)) & 1)) ++row;
            if (row >= 2 * cam_obs0[c + 1]) return false;
            taken[row >> 6] |= 1ull << (row & 63);
            return true;
        };
        if (cheap_ok) {      // IO columns: a free row of the first of their cameras that has one
            std::vector<uint8_t> io_done(std::max(1, P.nIOu), 0);
            int64_t io_left = 0;
            for (int k = 0; k < P.nIOu; ++k) { io_done[k] = !wanted(6 * (int64_t)nc + k); io_left += !io_done[k]; }
            for (int c = 0; c < nc && io_left > 0; ++c) {
                int64_t row = 2 * cam_obs0[c];
                for (int r = 0; r < R && io_left > 0; ++r) {
                    const int32_t k = distIO[(size_t)c * R + r];
                    if (k < 0 || io_done[k]) continue;
                    if (!claim(c, row)) break;
                    io_done[k] = 1; --io_left;
                }
            }
            cheap_ok = io_left == 0;
        }
        for (int c = 0; c < nc && cheap_ok; ++c) {
            int64_t row = 2 * cam_obs0[c];
            for (int d = 0; d < 6 && cheap_ok; ++d)
                if (wanted(6 * (int64_t)c + d)) cheap_ok = claim(c, row);
        }
        sprank_settled = cheap_ok;
    }
    lapt("structural rank: cheap assignment in parallel");
    if (plan_clock) fprintf(stderr, "[plan clock] cheap assignment settled the structural rank: %s\n", sprank_settled ? "yes" : "no");
    if (P.rank_ok && !sprank_settled && !env_on("DBAT_HIP_SPRANK_OFF") && !P.shared_eo) {     // (shared EO: the counting conditions only)
        const int64_t ncol_all = P.NZ;
        std::vector<int64_t> cam_obs0(nc + 1, 0);
        for (int c = 0; c < nc; ++c) cam_obs0[c + 1] = cam_obs0[c] + n_cam[c];
        std::vector<std::vector<int32_t>> io_cams(P.nIOu);     // cameras of each IO column
        std::vector<std::vector<int64_t>> io_cum(P.nIOu);      // cumulative row counts over them
        for (int c = 0; c < nc; ++c)
            for (int r = 0; r < R; ++r) {
                const int32_t s = distIO[(size_t)c * R + r];
                if (s >= 0 && n_cam[c] > 0) io_cams[s].push_back(c);
            }
        for (int k = 0; k < P.nIOu; ++k) {
            io_cum[k].assign(io_cams[k].size() + 1, 0);
            for (size_t i = 0; i < io_cams[k].size(); ++i) io_cum[k][i + 1] = io_cum[k][i] + 2 * (int64_t)n_cam[io_cams[k][i]];
        }
        const int64_t eo_end = 6 * (int64_t)nc, io_end = P.NS;
        auto nrows = [&](int64_t col) -> int64_t {
            if (col < eo_end) return 2 * (int64_t)n_cam[col / 6];
            if (col < io_end) return io_cum[col - eo_end].back();
            return 2 * (int64_t)k_pt[(col - io_end) / 3];
        };
        auto row_at = [&](int64_t col, int64_t i) -> int64_t {
            if (col < eo_end) return 2 * cam_obs0[col / 6] + i;
            if (col < io_end) {
                const auto &cum = io_cum[col - eo_end];
                const size_t q = std::upper_bound(cum.begin(), cum.end(), i) - cum.begin() - 1;
                return 2 * cam_obs0[io_cams[col - eo_end][q]] + (i - cum[q]);
            }
            const int64_t p = (col - io_end) / 3;
            return 2 * by_pt[pstart[p] + (i >> 1)] + (i & 1);
        };
        std::vector<int32_t> row_owner((size_t)2 * P.no, -1);
        std::vector<int64_t> cheap(ncol_all, 0), iter(ncol_all, 0), take(ncol_all, -1);
        std::vector<int32_t> stamp(ncol_all, -1);
        std::vector<int64_t> stack;
        int32_t gen = 0;
        auto match_column = [&](int64_t k) -> bool {
            stack.assign(1, k); iter[k] = 0; stamp[k] = ++gen;
            while (!stack.empty()) {
                const int64_t j = stack.back(), nr = nrows(j);
                bool found = false;
                while (cheap[j] < nr) {
                    const int64_t r = row_at(j, cheap[j]++);
                    if (row_owner[r] < 0) { take[j] = r; found = true; break; }
                }
                if (found) {
                    for (int64_t c : stack) row_owner[take[c]] = (int32_t)c;
                    return true;
                }
                bool deeper = false;
                while (iter[j] < nr) {
                    const int64_t r = row_at(j, iter[j]++), j2 = row_owner[r];
                    if (stamp[j2] != gen) {
                        stamp[j2] = gen; take[j] = r; iter[j2] = 0;
                        stack.push_back(j2); deeper = true;
                        break;
                    }
                }
                if (!deeper) stack.pop_back();
            }
            return false;
        };
        auto wanted = [&](int64_t col) -> bool { return P.z_est[col] && !(P.z_prw[col] > 0); };
        for (int64_t col = eo_end; col < io_end && P.rank_ok; ++col) if (wanted(col) && !match_column(col)) P.rank_ok = false;
        for (int64_t col = 0; col < eo_end && P.rank_ok; ++col) if (wanted(col) && !match_column(col)) P.rank_ok = false;
        for (int64_t col = io_end; col < ncol_all && P.rank_ok; ++col) if (wanted(col) && !match_column(col)) P.rank_ok = false;
    }

    lapt("structural rank: exact matching (only when the cheap assignment fails)");
    // camera co-visibility: the envelope of the reduced system (chol.hpp) and the full graph -- the pattern of the
    // reduced system (ordering + symbolic factorisation of the tile Cholesky, chol_df.hpp).  Every thread collects
    // the points of its range in its own copy (min / bit-or: exact, order-free), the copies are merged.
    P.cam_first.resize(nc);
    for (int c = 0; c < nc; ++c) P.cam_first[c] = c;
    P.cam_adj_words = (nc + 63) / 64;
    P.cam_adj.assign((size_t)nc * P.cam_adj_words, 0);
    {
        const size_t aw = (size_t)nc * P.cam_adj_words;
        const int nr = (int)std::min<int64_t>(par.ranges(np, 1 << 14), std::max<int64_t>(1, (int64_t)(256u << 20) / (int64_t)std::max<size_t>(aw * 8, 1)));
        std::vector<std::vector<uint64_t>> adj_t(nr);
        std::vector<std::vector<int32_t>> first_t(nr);
        const Par par_adj{nr};
        par_adj.run(np, [&](int64_t plo, int64_t phi, int tid) {
            std::vector<uint64_t> &adj = adj_t[tid];
            std::vector<int32_t> &first = first_t[tid];
            adj.assign(aw, 0); first.resize(nc);
            for (int c = 0; c < nc; ++c) first[c] = c;
            for (int64_t p = plo; p < phi; ++p) {
                const int k = k_pt[p];
                if (!k) continue;
                const int32_t *cp = cbp.data() + pstart[p];
                const int32_t c0 = cp[0];                                 // cameras ascend inside a point
                for (int j = 1; j < k; ++j) if (c0 < first[cp[j]]) first[cp[j]] = c0;
                for (int a = 0; a < k; ++a) {
                    uint64_t *row = adj.data() + (size_t)cp[a] * P.cam_adj_words;
                    for (int b = 0; b < k; ++b) row[cp[b] >> 6] |= 1ull << (cp[b] & 63);
                }
            }
        }, 1);
        for (int t = 0; t < nr; ++t) {
            if (adj_t[t].empty()) continue;
            for (size_t i = 0; i < aw; ++i) P.cam_adj[i] |= adj_t[t][i];
            for (int c = 0; c < nc; ++c) P.cam_first[c] = std::min(P.cam_first[c], first_t[t][c]);
        }
    }
    lapt("camera co-visibility graph");
    // ---- nested dissection (nd.hpp); with several ranks its first levels are the ranks' domains
    {
        const bool nd_off = env_on("DBAT_HIP_ND_OFF");
        const bool permuted_ok = !P.shared_eo;
        P.mg_subtree = P.nranks > 1 && permuted_ok && !nd_off && !env_on("DBAT_HIP_MG_REPLICATED");
        std::vector<double> xyz((size_t)3 * nc), wcam(nc);
        for (int c = 0; c < nc; ++c) {
            for (int k = 0; k < 3; ++k) { const double v = pb.EO_val[(size_t)6 * c + k]; xyz[(size_t)3 * c + k] = std::isfinite(v) ? v : 0.0; }   // only steers the bisection
            wcam[c] = (double)n_cam[c];
        }
        const int leaf = std::max(8, env_int("DBAT_HIP_ND_LEAF", 32));
        nd_build(nc, P.cam_adj.data(), P.cam_adj_words, xyz.data(), wcam.data(), P.mg_subtree ? P.nranks : 1, leaf, nd_off, P.nd);
    }
    lapt("nested dissection");
    // Points that fit a tile of the MFMA Schur kernel (at most CMAX cameras and
    // IOT estimated IO columns) are processed first; "heavy" points (e.g. control
    // points seen in very many images) follow and go through k_build.
    P.CMAX = env_int("DBAT_HIP_CMAX", P.with_io ? Plan::SIG_IO_CMAX : 21);
    if (P.shared_eo) P.CMAX = 0;                    // the tile kernels address camera rows as 6*camera + k
    if (P.CMAX < 0 || P.CMAX > (P.with_io ? 18 : 21)) P.CMAX = P.with_io ? 18 : 21;   // 6*CMAX (+IOT) <= 128 rows of the MFMA tile
    if (P.ncolmax > 15) P.CMAX = 0;                  // the tile kernels hold at most 9 IO columns per camera (DBAT's usual self-calibration:
                                                     // cc, pp, aspect, K1-K3, P1-P2); beyond that: untiled (k_build)
    {   // batch size: whole points, at most BT observations
        P.BT = env_int("DBAT_HIP_BT", 256);
        if (P.BT != 128 && P.BT != 256) P.BT = 256;
        if ((size_t)P.BT * P.ncolmax * 3 * 8 + (size_t)P.BT * 18 * 8 > 150 * 1024) P.BT = 128;
        if (P.BT != 256) P.CMAX = 0;                // the tile kernels are written for four waves of observations
    }
    std::vector<uint8_t> heavy(np, 0), giant(np, 0);
    par.run(np, [&](int64_t plo, int64_t phi, int) {
    for (int64_t p = plo; p < phi; ++p) giant[p] = k_pt[p] > P.BT;
    for (int64_t p = plo; p < phi && P.CMAX; ++p) {
        if (k_pt[p] > P.CMAX) { heavy[p] = 1; continue; }
        if (P.with_io) {
            int32_t seen[Plan::IOT + 1]; int ns = 0;
            for (int j = 0; j < k_pt[p] && ns <= Plan::IOT; ++j) {
                const int32_t c = cbp[pstart[p] + j];
                for (int q = 6; q < P.cam_ncol[c] && ns <= Plan::IOT; ++q) {
                    const int32_t io = P.cam_col[(size_t)c * MAXCOL + q];
                    bool f = false;
                    for (int t = 0; t < ns; ++t) if (seen[t] == io) { f = true; break; }
                    if (!f) seen[ns++] = io;
                }
            }
            if (ns > Plan::IOT) heavy[p] = 1;
        }
    }
    });
    {   // A tile kernel launched for a handful of points costs its fixed 40 ... 50 us whatever it does (the four control
        // points of the camcal demo beside 96 points that every image sees): where heavy points exist anyway and the others
        // are few, they all take the matrix-core path of the heavy points (heavy.hpp), whose cost per point is small.
        int64_t n_heavy = 0, n_tiled = 0;
        for (int p = 0; p < np; ++p) { if (giant[p]) continue; if (heavy[p]) n_heavy += k_pt[p]; else n_tiled += k_pt[p]; }
        const bool hv_can = P.CMAX && P.BT == 256 && !P.shared_eo && P.ncolmax <= 6 + Plan::HV_NIOC && env_int("DBAT_HIP_HEAVY", 1) != 0;
        if (hv_can && n_heavy > 0 && n_tiled <= 2048)
            for (int p = 0; p < np; ++p) if (k_pt[p] > 0 && !giant[p]) heavy[p] = 1;
    }
    lapt("tile-fit classification of the points");
    // Key = Morton code of the point's initial coordinates in the principal axes of the
    // point cloud: points that are close in object space are seen by the same cameras, so
    // neighbouring points touch the same blocks of the reduced system (tiles below).  A
    // flat cloud (terrain under an aerial block, a facade: smallest principal extent
    // below a fifth of the largest) is ordered along its plane only -- interleaving the
    // thin direction would widen the footprint of a run of points, and with it the set
    // of cameras a tile has to hold.  Unobserved points sort last.
    std::vector<uint64_t> key(np);
    {
        double mean[3] = {0, 0, 0}, cov[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        int64_t nfin = 0;
        for (int p = 0; p < np; ++p) {
            const double *q = pb.OP_val + (size_t)3 * p;
            if (std::isfinite(q[0]) && std::isfinite(q[1]) && std::isfinite(q[2])) { ++nfin; for (int d = 0; d < 3; ++d) mean[d] += q[d]; }
        }
        for (int d = 0; d < 3; ++d) mean[d] /= (double)std::max<int64_t>(nfin, 1);
        for (int p = 0; p < np; ++p) {
            const double *q = pb.OP_val + (size_t)3 * p;
            if (!(std::isfinite(q[0]) && std::isfinite(q[1]) && std::isfinite(q[2]))) continue;
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) cov[i][j] += (q[i] - mean[i]) * (q[j] - mean[j]);
        }
        double ax[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};          // columns = principal axes (cyclic Jacobi)
        for (int sweep = 0; sweep < 30; ++sweep) {
            const double off = std::fabs(cov[0][1]) + std::fabs(cov[0][2]) + std::fabs(cov[1][2]);
            if (!(off > 1e-14 * (std::fabs(cov[0][0]) + std::fabs(cov[1][1]) + std::fabs(cov[2][2])))) break;
            for (int i = 0; i < 2; ++i)
                for (int j = i + 1; j < 3; ++j) {
                    if (cov[i][j] == 0.0) continue;
                    const double th = 0.5 * std::atan2(2 * cov[i][j], cov[j][j] - cov[i][i]);
                    const double cs = std::cos(th), sn = std::sin(th);
                    for (int k = 0; k < 3; ++k) {                      // cov <- G' cov G, ax <- ax G
                        const double a = cov[k][i], b = cov[k][j];
                        cov[k][i] = cs * a - sn * b; cov[k][j] = sn * a + cs * b;
                    }
                    for (int k = 0; k < 3; ++k) {
                        const double a = cov[i][k], b = cov[j][k];
                        cov[i][k] = cs * a - sn * b; cov[j][k] = sn * a + cs * b;
                    }
                    for (int k = 0; k < 3; ++k) {
                        const double a = ax[k][i], b = ax[k][j];
                        ax[k][i] = cs * a - sn * b; ax[k][j] = sn * a + cs * b;
                    }
                }
        }
        auto coord = [&](int p, int a) {
            const double *q = pb.OP_val + (size_t)3 * p;
            return (q[0] - mean[0]) * ax[0][a] + (q[1] - mean[1]) * ax[1][a] + (q[2] - mean[2]) * ax[2][a];
        };
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        for (int p = 0; p < np; ++p)
            for (int d = 0; d < 3; ++d) {
                const double v = coord(p, d);
                if (std::isfinite(v)) { lo[d] = std::min(lo[d], v); hi[d] = std::max(hi[d], v); }
            }
        int axis[3] = {0, 1, 2};
        std::sort(axis, axis + 3, [&](int a, int b) { return hi[a] - lo[a] > hi[b] - lo[b]; });
        double ext = hi[axis[0]] - lo[axis[0]];
        if (!(ext > 0)) ext = 1;
        const int dims = (hi[axis[2]] - lo[axis[2]]) < 0.2 * ext ? 2 : 3;
        P.order_dims = dims;
        auto spread = [](uint64_t v) {      // 21 bits -> every third bit
            v &= 0x1FFFFF;
            v = (v | v << 32) & 0x1F00000000FFFFull;
            v = (v | v << 16) & 0x1F0000FF0000FFull;
            v = (v | v << 8) & 0x100F00F00F00F00Full;
            v = (v | v << 4) & 0x10C30C30C30C30C3ull;
            v = (v | v << 2) & 0x1249249249249249ull;
            return v;
        };
        par.run(np, [&](int64_t plo, int64_t phi, int) {
            for (int64_t p = plo; p < phi; ++p) {
                uint64_t k = 0;
                for (int d = 0; d < dims; ++d) {
                    double v = (coord((int)p, axis[d]) - lo[axis[d]]) / ext;
                    if (!(v >= 0)) v = 0;
                    if (v > 1) v = 1;
                    k |= spread((uint64_t)(v * 2097151.0)) << d;
                }
                // tiled points, then heavy points, then giant points; unobserved points last
                key[p] = !k_pt[p] ? ~0ull
                         : giant[p] ? ((k >> 2) | (3ull << 62))
                         : heavy[p] ? ((k >> 2) | (1ull << 63)) : (k >> 1);
            }
        });
    }
    P.porder.resize(np);
    std::iota(P.porder.begin(), P.porder.end(), 0);
    {
        // Inside a coarse cell of the curve (about 512 points) the points are ordered by their
        // camera list, so that points seen by exactly the same cameras -- they update the same
        // rows and columns of the reduced system -- follow each other (signature groups of
        // k_build_sig).  Across cells the order stays the curve's.
        int64_t nobs_pts = 0;
        for (int p = 0; p < np; ++p) nobs_pts += k_pt[p] > 0;
        int cell_bits = 0;
        while (((int64_t)512 << cell_bits) < nobs_pts && cell_bits < 60) ++cell_bits;
        // the code interleaves `order_dims` of three bit lanes into the top 62 bits of the key
        const int eff_bits = (cell_bits * 3 + P.order_dims - 1) / P.order_dims;
        const int low_bits = std::max(0, 62 - eff_bits);
        // one record per point: (cell of the curve, camera signature inside a cell of tiled points, key, point) -- a
        // strict total order whose last member reproduces the stable sort by the first three
        struct Rec { uint64_t cell, sig, key; int32_t p; };
        std::vector<Rec> recs(np);
        par.run(np, [&](int64_t plo, int64_t phi, int) {
            for (int64_t p = plo; p < phi; ++p) {
                uint64_t sg = 0;
                const bool in_cell = low_bits > 0 && key[p] < (1ull << 62);
                if (in_cell) {
                    // first four cameras (16 bits each) lead, so that groups with neighbouring camera sets stay close
                    uint64_t lead = 0, h = 1469598103934665603ull;
                    for (int j = 0; j < k_pt[p]; ++j) {
                        const uint64_t c = (uint64_t)cbp[pstart[p] + j];
                        if (j < 3) lead |= (c & 0xFFFF) << (48 - 16 * j);
                        h = (h ^ c) * 1099511628211ull;
                    }
                    sg = lead | (h & 0xFFFF);
                }
                recs[p] = Rec{in_cell ? key[p] >> low_bits : key[p], sg, key[p], (int32_t)p};
            }
        });
        par.sort(recs, [](const Rec &a, const Rec &b) {
            if (a.cell != b.cell) return a.cell < b.cell;
            if (a.sig != b.sig) return a.sig < b.sig;
            if (a.key != b.key) return a.key < b.key;
            return a.p < b.p;
        });
        par.run(np, [&](int64_t lo_, int64_t hi_, int) { for (int64_t i = lo_; i < hi_; ++i) P.porder[i] = recs[i].p; });
    }
    if (P.mg_subtree) {
        // Domain sharding: a point goes to the rank whose domain holds its interior cameras (nd.hpp: all
        // of them lie in ONE domain); points that only see top-separator cameras go wherever the load is
        // lowest.  The processing order becomes owner-major (the order inside a rank stays the curve's), so
        // a rank's points are one contiguous range of it.
        std::vector<int32_t> owner(np, -1);
        std::vector<int64_t> load(P.nranks, 0);
        for (int p = 0; p < np; ++p)
            for (int j = 0; j < k_pt[p] && owner[p] < 0; ++j) owner[p] = P.nd.cam_owner[cbp[pstart[p] + j]];
        for (int p = 0; p < np; ++p) if (owner[p] >= 0) load[owner[p]] += k_pt[p];
        for (int p = 0; p < np; ++p)
            if (owner[p] < 0) {
                int best = 0;
                for (int r = 1; r < P.nranks; ++r) if (load[r] < load[best]) best = r;
                owner[p] = best; load[best] += k_pt[p];
            }
        {   // stable sort by owner: a counting sort
            std::vector<int64_t> cnt((size_t)P.nranks + 1, 0);
            for (int p = 0; p < np; ++p) ++cnt[(size_t)owner[p] + 1];
            for (int r = 0; r < P.nranks; ++r) cnt[r + 1] += cnt[r];
            P.pt_lo = cnt[P.rank]; P.pt_hi = cnt[P.rank + 1];
            std::vector<int32_t> sorted(np);
            for (int i = 0; i < np; ++i) sorted[cnt[owner[P.porder[i]]]++] = P.porder[i];
            P.porder.swap(sorted);
        }
        if (P.pt_hi <= P.pt_lo) P.pt_lo = P.pt_hi = 0;
    } else
    // shard = contiguous range of the processing order balanced by observation count
    {
        std::vector<int64_t> cum(np + 1, 0);
        for (int i = 0; i < np; ++i) cum[i + 1] = cum[i] + k_pt[P.porder[i]];
        auto cut = [&](int r) -> int64_t {
            if (r <= 0) return 0;
            if (r >= P.nranks) return np;
            const int64_t target = (cum[np] * r) / P.nranks;
            return std::lower_bound(cum.begin(), cum.end(), target) - cum.begin();
        };
        P.pt_lo = cut(P.rank); P.pt_hi = cut(P.rank + 1);
        if (P.pt_hi < P.pt_lo) P.pt_hi = P.pt_lo;
    }
    lapt("processing order (keys, sort), shards");
    // the point part of z follows the processing order: permute everything that was laid out by point id
    P.pt_rank.assign(np, 0);
    for (int i = 0; i < np; ++i) P.pt_rank[P.porder[i]] = i;
    {
        auto zperm = [&](int64_t zi) -> int64_t {
            if (zi < P.NS) return zi;
            const int64_t e = zi - P.NS;
            return P.NS + 3 * (int64_t)P.pt_rank[e / 3] + e % 3;
        };
        uvec<double> a((size_t)P.NZ), b((size_t)P.NZ), c((size_t)P.NZ);
        uvec<uint8_t> e((size_t)P.NZ);
        par.run(np, [&](int64_t plo, int64_t phi, int) {
            for (int64_t zi = P.NS + 3 * plo; zi < P.NS + 3 * phi; ++zi) { a[zi] = P.z0[zi]; b[zi] = P.z_prw[zi]; c[zi] = P.z_prv[zi]; e[zi] = P.z_est[zi]; }
        });
        par.run(np, [&](int64_t plo, int64_t phi, int) {        // (whole points per thread: z_est is a byte array)
            for (int64_t zi = P.NS + 3 * plo; zi < P.NS + 3 * phi; ++zi) {
                const int64_t q = zperm(zi);
                P.z0[q] = a[zi]; P.z_prw[q] = b[zi]; P.z_prv[q] = c[zi]; P.z_est[q] = e[zi];
            }
        });
        par.run((int64_t)P.x2z.size(), [&](int64_t lo_, int64_t hi_, int) { for (int64_t i = lo_; i < hi_; ++i) P.x2z[i] = zperm(P.x2z[i]); });
        for (auto &v : P.prior_z) v = zperm(v);
    }
    // z_mine: who counts an entry in the sums over z (and supplies it to the gathered result).  OP: the
    // owning shard.  EO/IO: rank 0 -- with domain sharding the EO of a domain's cameras belongs to that
    // domain's rank (nobody else ever computes their step), the top separators' and the IO to rank 0
    P.z_mine.assign(P.NZ, 0);
    if (P.mg_subtree) {
        for (int c = 0; c < nc; ++c) {
            const int ow = P.nd.cam_owner[c] < 0 ? 0 : P.nd.cam_owner[c];
            if (ow == P.rank) for (int k = 0; k < 6; ++k) P.z_mine[(size_t)6 * c + k] = 1;
        }
        if (P.rank == 0) for (int64_t z = 6 * (int64_t)nc; z < P.NS; ++z) P.z_mine[z] = 1;
    } else if (P.rank == 0) for (int64_t z = 0; z < P.NS; ++z) P.z_mine[z] = 1;
    for (int64_t i = P.pt_lo; i < P.pt_hi; ++i)
        for (int d = 0; d < 3; ++d) P.z_mine[P.NS + 3 * i + d] = 1;
    if (!with_obs) return true;

    lapt("permutation of the point arrays, z_mine");
    // ---- batches of whole points, at most BT observations each; tiles of
    // batches touching at most CMAX cameras (fixed-IO path only).
    // Two stages.  DECISIONS (one thread; it only reads the camera lists of the points): where every point's
    // observations start, batch / tile / signature-group boundaries, the chunk list.  COPIES (all threads): the
    // per-observation arrays in processing order, the tile-local camera indices, the slot-major and the camera-major
    // copies of the image coordinates -- everything that is proportional to the number of observations.
    P.batch_start.clear(); P.batch_start.push_back(0);
    P.giant_start.clear();
    // Where a point's observations start does not depend on any decision: a prefix sum over the processing order.
    // So the arrays that only depend on it are copied FIRST, by all threads -- and the decisions then read the
    // cameras of the points as one contiguous stream (P.o_cam) instead of chasing the points through memory.
    const int64_t n_sh = P.pt_hi - P.pt_lo;
    uvec<int32_t> kk((size_t)n_sh);                  // observations of the point at processing index pt_lo + is
    uvec<uint8_t> pfl((size_t)n_sh);                 // 1: giant, 2: heavy
    uvec<int64_t> pt_pos((size_t)n_sh + 1);          // first observation
    par.run(n_sh, [&](int64_t lo_, int64_t hi_, int) {
        for (int64_t is = lo_; is < hi_; ++is) {
            const int32_t p = P.porder[P.pt_lo + is];
            kk[is] = k_pt[p]; pfl[is] = (uint8_t)((giant[p] ? 1 : 0) | (heavy[p] ? 2 : 0));
        }
    });
    pt_pos[0] = 0;
    for (int64_t is = 0; is < n_sh; ++is) pt_pos[is + 1] = pt_pos[is] + kk[is];
    const int64_t nobs_shard = pt_pos[n_sh];
    P.o_cam.resize(nobs_shard); P.o_pt.resize(nobs_shard); P.o_uv.resize(2 * nobs_shard);
    P.o_seg.resize(nobs_shard); P.o_row.resize(nobs_shard); P.o_lc.resize(nobs_shard);
    P.o_pidx.resize(nobs_shard);
    if (!P.uniform_w) P.o_w.resize(2 * nobs_shard);
    par.run(n_sh, [&](int64_t lo_, int64_t hi_, int) {
        for (int64_t is = lo_; is < hi_; ++is) {
            int64_t q = pt_pos[is];
            const int32_t p = P.porder[P.pt_lo + is];
            const int k = kk[is];
            const int32_t rk = P.pt_rank[p];
            for (int j = 0; j < k; ++j, ++q) {
                const int64_t o = by_pt[pstart[p] + j];
                const int32_t c = cbp[pstart[p] + j];
                P.o_cam[q] = c; P.o_pt[q] = rk;
                P.o_uv[2 * q] = pb.ip_val[2 * o]; P.o_uv[2 * q + 1] = pb.ip_val[2 * o + 1];
                P.o_row[q] = o;
                if (!P.uniform_w) {
                    P.o_w[2 * q] = 1.0 / (pb.ip_std[2 * o] * P.px[2 * c]);
                    P.o_w[2 * q + 1] = 1.0 / (pb.ip_std[2 * o + 1] * P.px[2 * c + 1]);
                }
            }
        }
    }, 1024);
    lapt("copies: observations in processing order");
    P.tile_batch.clear(); P.tile_cam_start.clear(); P.tile_cams.clear();
    P.tile_batch.push_back(0); P.tile_cam_start.push_back(0);
    std::vector<int32_t> stamp(nc, -1);              // tile id in which a camera was last seen
    std::vector<int32_t> io_stamp(std::max(1, P.nIOu), -1);
    std::vector<int32_t> cur_cams, cur_io;
    P.tile_io_start.clear(); P.tile_iocols.clear(); P.tile_cam_io.clear(); P.tile_io_simple.clear();
    P.tile_io_start.push_back(0);
    int64_t pos = 0, bstart = 0, tile_first_obs = 0;
    // what the later copies need to know about a point of the shard (index i - pt_lo)
    uvec<uint32_t> pt_seg((size_t)n_sh);             // segment word (0: a giant point)
    uvec<uint8_t> pt_pidx((size_t)n_sh);             // ordinal inside its batch
    std::vector<int64_t> tile_obs0;                  // [ntiles + 1] observation range of every tile
    struct GroupRec { int64_t pos0, npts; int k; };
    std::vector<GroupRec> groups;                    // signature groups (slot-major copy of their image coordinates)
    // signature groups of the current tile
    P.sg_chunk.clear(); P.sg_tile_chunk0.assign(1, 0); P.sg_lc.clear(); P.sg_gcam.clear(); P.sg_kmax = 0; P.sg_rows_max = 0; P.sg_ngroups = 0; P.sg_npoints = 0;
    std::vector<int32_t> sg_gc;                      // global camera of every sg_lc entry (converted when the tile closes)
    size_t sg_lc_tile0 = 0;
    std::vector<int32_t> g_cams;                     // cameras of the open group
    int64_t g_pos0 = 0, g_rank0 = 0, g_npts = 0;
    auto close_group = [&]() {
        if (g_npts == 0) return;
        const int k = (int)g_cams.size();
        // chunks of equal length (a group of 70 points: 35 + 35, not 64 + 6)
        const int64_t nchunks = (g_npts + Plan::SG_CHUNK - 1) / Plan::SG_CHUNK;
        for (int64_t q = 0; q < nchunks; ++q) {
            const int64_t c0 = q * g_npts / nchunks, c1 = (q + 1) * g_npts / nchunks;
            const int32_t ch[8] = {(int32_t)(g_rank0 + c0), (int32_t)(c1 - c0), k,
                                   (int32_t)(g_pos0 + c0 * k), (int32_t)g_npts, (int32_t)c0, (int32_t)g_pos0, 0};
            P.sg_chunk.insert(P.sg_chunk.end(), ch, ch + 8);
            for (int j = 0; j < 16; ++j) sg_gc.push_back(j < k ? g_cams[j] : -1);      // 16 camera slots per chunk
        }
        groups.push_back(GroupRec{g_pos0, g_npts, k});
        P.sg_kmax = std::max(P.sg_kmax, k);
        ++P.sg_ngroups; P.sg_npoints += g_npts;
        g_npts = 0; g_cams.clear();
    };
    auto close_tile = [&](int64_t end_obs) {
        close_group();
        // local indices in ascending global camera order
        std::sort(cur_cams.begin(), cur_cams.end());
        std::vector<int32_t> &loc = stamp;            // reuse as cam -> local index (restored below)
        for (size_t l = 0; l < cur_cams.size(); ++l) loc[cur_cams[l]] = (int32_t)l;
        if (tile_obs0.empty()) tile_obs0.push_back(tile_first_obs);
        tile_obs0.push_back(end_obs);                 // (o_lc of the tile's observations: in the copy stage)
        // IO columns of the tile, ascending; every camera's IO columns -> local IO rows
        std::sort(cur_io.begin(), cur_io.end());
        // k_build_sig's per-point IO rows: 1 = one IO block (IO column q = tile IO row q, every camera the same
        // number of columns), 2 = two blocks of 8 rows each (camera columns q -> base + q, base 0 or 8)
        bool io_one = !cur_cams.empty(), io_two = !cur_cams.empty() && cur_io.size() == 16;
        for (int32_t c : cur_cams) {
            uint8_t rows16[16] = {0};
            for (int q = 6; q < P.cam_ncol[c]; ++q) {
                const int32_t io = P.cam_col[(size_t)c * MAXCOL + q] - 6 * nc;
                rows16[q - 6] = (uint8_t)(std::lower_bound(cur_io.begin(), cur_io.end(), io) - cur_io.begin());
                io_one = io_one && rows16[q - 6] == q - 6;
                io_two = io_two && (rows16[0] == 0 || rows16[0] == 8) && rows16[q - 6] == rows16[0] + (q - 6);
            }
            io_one = io_one && P.cam_ncol[c] == P.cam_ncol[cur_cams[0]] && (size_t)(P.cam_ncol[c] - 6) == cur_io.size();
            io_two = io_two && P.cam_ncol[c] == 14;
            P.tile_cam_io.insert(P.tile_cam_io.end(), rows16, rows16 + 16);
        }
        P.tile_io_simple.push_back(io_one ? 1 : (io_two ? 2 : 0));
        for (int32_t io : cur_io) { P.tile_iocols.push_back(io); io_stamp[io] = -1; }
        P.tile_io_start.push_back((int32_t)P.tile_iocols.size());
        cur_io.clear();
        P.sg_lc.resize(sg_gc.size());
        for (size_t e = sg_lc_tile0; e < sg_gc.size(); ++e) P.sg_lc[e] = sg_gc[e] >= 0 ? (uint8_t)loc[sg_gc[e]] : 0;
        {   // the tile's chunks longest first: its waves take them in this order and finish together
            const size_t q0 = (size_t)P.sg_tile_chunk0.back(), q1 = P.sg_chunk.size() / 8;
            std::vector<int32_t> ord(q1 - q0);
            std::iota(ord.begin(), ord.end(), 0);
            std::stable_sort(ord.begin(), ord.end(), [&](int32_t a, int32_t b) {
                return (int64_t)P.sg_chunk[8 * (q0 + a) + 1] * P.sg_chunk[8 * (q0 + a) + 2] >
                       (int64_t)P.sg_chunk[8 * (q0 + b) + 1] * P.sg_chunk[8 * (q0 + b) + 2];
            });
            std::vector<int32_t> cd(P.sg_chunk.begin() + 8 * q0, P.sg_chunk.end());
            std::vector<uint8_t> lcs(P.sg_lc.begin() + 16 * q0, P.sg_lc.end());
            for (size_t a = 0; a < ord.size(); ++a) {
                std::copy(cd.begin() + 8 * ord[a], cd.begin() + 8 * ord[a] + 8, P.sg_chunk.begin() + 8 * (q0 + a));
                std::copy(lcs.begin() + 16 * ord[a], lcs.begin() + 16 * ord[a] + 16, P.sg_lc.begin() + 16 * (q0 + a));
            }
        }
        for (size_t q = (size_t)P.sg_tile_chunk0.back(); q < P.sg_chunk.size() / 8; ++q)
            P.sg_rows_max = std::max(P.sg_rows_max, 6 * P.sg_chunk[8 * q + 2] + (P.tile_io_start.back() - P.tile_io_start[P.tile_io_start.size() - 2]) + 1);
        P.sg_gcam.resize(P.sg_lc.size());
        for (size_t e = sg_lc_tile0; e < sg_gc.size(); ++e) P.sg_gcam[e] = cur_cams.empty() ? 0 : cur_cams[P.sg_lc[e]];
        sg_lc_tile0 = sg_gc.size();
        P.sg_tile_chunk0.push_back((int32_t)(P.sg_chunk.size() / 8));
        for (int32_t c : cur_cams) { P.tile_cams.push_back(c); loc[c] = -1; }
        P.tile_cam_start.push_back((int32_t)P.tile_cams.size());
        P.tile_batch.push_back((int32_t)P.batch_start.size() - 1);   // = number of closed batches
        cur_cams.clear();
        tile_first_obs = end_obs;
    };
    int32_t tile_id = 0;
    int pidx = 0;
    bool in_heavy = false;
    lapt("batches, tiles, signature groups");
    // batches per tile: long tiles amortise the flush of the tile into S, but a small
    // problem must still break into enough tiles to occupy the 256 CUs twice over
    const int64_t shard_obs = nobs_shard;
    const int bmax_floor = std::max(1, env_int("DBAT_HIP_TILE_BMIN", 2));     // (round 4: 4 -> 2; the reference's roma project: tile kernel 0.136 -> 0.084 ms)
    const int bmax_auto = (int)std::min<int64_t>(48, std::max<int64_t>(bmax_floor, (shard_obs / std::max(1, P.BT) + 511) / 512));
    const int tile_bmax = bmax_auto;
    for (int64_t i = P.pt_lo; i < P.pt_hi; ++i) {
        const int64_t is = i - P.pt_lo;
        const int k = kk[is];
        pt_seg[is] = 0; pt_pidx[is] = 0;
        if (k == 0) continue;
        const int32_t *cp = P.o_cam.data() + pos;    // (pos == pt_pos[is])
        if (pfl[is] & 1) {
            if (P.giant_start.empty()) {
                // first giant point: close the last batch (and tile); what follows is outside the batches
                if (pos > bstart) { P.batch_start.push_back(pos); bstart = pos; }
                if (P.CMAX && !in_heavy) {
                    for (int32_t c : cur_cams) stamp[c] = -1;
                    close_tile(pos);
                    in_heavy = true;
                    P.nb_tiled = (int64_t)P.batch_start.size() - 1;
                }
            }
            P.giant_start.push_back(pos);            // (segment word 0, ordinal 0)
            pos += k;
            bstart = pos;
            continue;
        }
        if (P.CMAX && (pfl[is] & 2) && !in_heavy) {
            // first heavy point: close the last tile; the remaining batches are not tiled
            if (pos > bstart) { P.batch_start.push_back(pos); bstart = pos; }
            for (int32_t c : cur_cams) stamp[c] = -1;
            close_tile(pos);
            in_heavy = true;
            P.nb_tiled = (int64_t)P.batch_start.size() - 1;
        }
        if (P.CMAX && !in_heavy) {
            int fresh = 0, fresh_io = 0;
            int32_t fio[Plan::IOT + 1];
            for (int j = 0; j < k; ++j) {
                const int32_t c = cp[j];
                if (stamp[c] == tile_id) continue;
                ++fresh;
                for (int q = 6; q < P.cam_ncol[c]; ++q) {
                    const int32_t io = P.cam_col[(size_t)c * MAXCOL + q] - 6 * nc;
                    if (io_stamp[io] == tile_id) continue;
                    bool f = false;
                    for (int t = 0; t < fresh_io; ++t) if (fio[t] == io) { f = true; break; }
                    if (!f && fresh_io <= Plan::IOT) fio[fresh_io++] = io;
                }
            }
            // a tile is one workgroup's work: cap its length, otherwise a long run of points seen
            // by the same few cameras becomes the tail of the whole launch
            const int64_t tile_nb = (int64_t)P.batch_start.size() - 1 - P.tile_batch.back();   // closed batches
            const bool too_long = pos - bstart + k > P.BT && tile_nb + 1 >= tile_bmax;
            const bool over = (int)cur_cams.size() + fresh > P.CMAX ||
                              (int)cur_io.size() + fresh_io > Plan::IOT || too_long;
            if (over && pos > tile_first_obs) {
                // close the current batch and tile before this point
                P.batch_start.push_back(pos); bstart = pos;
                for (int32_t c : cur_cams) stamp[c] = -1;
                close_tile(pos);
                ++tile_id;
            }
        }
        if (pos - bstart + k > P.BT) { P.batch_start.push_back(pos); bstart = pos; }
        if (pos == bstart) pidx = 0;
        if (P.CMAX && !in_heavy) {                   // same cameras as the previous point of the tile: same group
            bool same = g_npts > 0 && (int)g_cams.size() == k && g_npts < (1 << 20);
            for (int j = 0; same && j < k; ++j) same = g_cams[j] == cp[j];
            if (!same) {
                close_group();
                g_pos0 = pos; g_rank0 = i;
                g_cams.assign(cp, cp + k);
            }
            ++g_npts;
        }
        pt_seg[is] = (uint32_t)(pos - bstart) | ((uint32_t)k << 16);
        pt_pidx[is] = (uint8_t)pidx;
        if (P.CMAX && !in_heavy)
            for (int j = 0; j < k; ++j) {
                const int32_t c = cp[j];
                if (stamp[c] != tile_id) {
                    stamp[c] = tile_id; cur_cams.push_back(c);
                    for (int q = 6; q < P.cam_ncol[c]; ++q) {
                        const int32_t io = P.cam_col[(size_t)c * MAXCOL + q] - 6 * nc;
                        if (io_stamp[io] != tile_id) { io_stamp[io] = tile_id; cur_io.push_back(io); }
                    }
                }
            }
        pos += k;
        ++pidx;
    }
    if (!P.giant_start.empty()) P.giant_start.push_back(pos);
    if (pos > bstart || (P.batch_start.size() == 1 && P.giant_start.empty())) P.batch_start.push_back(pos);
    if (P.CMAX && !in_heavy) {
        for (int32_t c : cur_cams) stamp[c] = -1;
        close_tile(pos);
        P.nb_tiled = (int64_t)P.batch_start.size() - 1;
    }
    if (!P.CMAX) P.nb_tiled = 0;
    lapt("decisions: batches, tiles, signature groups (one thread)");
    // ---- the copies that depend on the decisions
    par.run(n_sh, [&](int64_t lo_, int64_t hi_, int) {
        for (int64_t is = lo_; is < hi_; ++is) {
            const uint32_t seg = pt_seg[is];
            const uint8_t pi = pt_pidx[is];
            for (int64_t q = pt_pos[is]; q < pt_pos[is + 1]; ++q) { P.o_seg[q] = seg; P.o_pidx[q] = pi; P.o_lc[q] = 0; }
        }
    }, 1024);
    {   // tile-local camera index of every tiled observation
        const int64_t nt_ = (int64_t)tile_obs0.size() - 1;
        par.run(std::max<int64_t>(nt_, 0), [&](int64_t lo_, int64_t hi_, int) {
            std::vector<int32_t> loc(nc, 0);
            for (int64_t t = lo_; t < hi_; ++t) {
                for (int32_t l = P.tile_cam_start[t]; l < P.tile_cam_start[t + 1]; ++l) loc[P.tile_cams[l]] = l - P.tile_cam_start[t];
                for (int64_t o = tile_obs0[t]; o < tile_obs0[t + 1]; ++o) P.o_lc[o] = (uint8_t)loc[P.o_cam[o]];
            }
        }, 4);
    }
    {   // slot-major copy of every signature group's image coordinates (zeros where there is no group)
        P.sg_uv.resize(P.o_uv.size());
        if (!P.uniform_w) P.sg_w.resize(P.o_w.size()); else P.sg_w.clear();
        const int64_t tiled_end = tile_obs0.empty() ? 0 : tile_obs0.back();
        par.run(nobs_shard - tiled_end, [&](int64_t lo_, int64_t hi_, int) {
            std::fill(P.sg_uv.begin() + 2 * (tiled_end + lo_), P.sg_uv.begin() + 2 * (tiled_end + hi_), 0.0);
            if (!P.uniform_w) std::fill(P.sg_w.begin() + 2 * (tiled_end + lo_), P.sg_w.begin() + 2 * (tiled_end + hi_), 0.0);
        }, 1 << 16);
        par.run((int64_t)groups.size(), [&](int64_t lo_, int64_t hi_, int) {
            for (int64_t g = lo_; g < hi_; ++g) {
                const int64_t g0 = groups[g].pos0, m = groups[g].npts;
                const int k = groups[g].k;
                for (int64_t i = 0; i < m; ++i)
                    for (int j = 0; j < k; ++j) {
                        const int64_t src = g0 + i * k + j, dst = g0 + (int64_t)j * m + i;
                        P.sg_uv[2 * dst] = P.o_uv[2 * src]; P.sg_uv[2 * dst + 1] = P.o_uv[2 * src + 1];
                        if (!P.uniform_w) { P.sg_w[2 * dst] = P.o_w[2 * src]; P.sg_w[2 * dst + 1] = P.o_w[2 * src + 1]; }
                    }
            }
        }, 64);
    }
    lapt("copies: segment words, tile-local cameras, slot-major coordinates");
    {   // launch the longest tiles first
        const int nt = (int)P.tile_batch.size() - 1;
        P.tile_order.resize(std::max(nt, 0));
        std::iota(P.tile_order.begin(), P.tile_order.end(), 0);
        std::stable_sort(P.tile_order.begin(), P.tile_order.end(), [&](int32_t a, int32_t b) {
            return P.batch_start[P.tile_batch[a + 1]] - P.batch_start[P.tile_batch[a]] >
                   P.batch_start[P.tile_batch[b + 1]] - P.batch_start[P.tile_batch[b]];
        });
        P.n_tiles_io_simple = 0;
        if (env_on("DBAT_HIP_SIG_IOS_OFF")) std::fill(P.tile_io_simple.begin(), P.tile_io_simple.end(), 0);
        for (uint8_t f : P.tile_io_simple) P.n_tiles_io_simple += f ? 1 : 0;
    }
    {   // camera-major copy of the observations (stable counting sort by camera): first the tiled
        // ones (chunks [0, n_cm_chunks_tiled): k_cam_normal), then the rest (heavy / giant points);
        // k_residual_cm runs over all chunks.  In parallel: the range is cut into blocks, every block counts its
        // cameras (nc counters per block: small), a prefix over (camera, block) gives every block its place.
        const int64_t ntiled = P.nb_tiled > 0 ? P.batch_start[P.nb_tiled] : 0;
        const int64_t nall = (int64_t)P.o_cam.size();
        P.cm_pt.resize(nall); P.cm_uv.resize(2 * nall);
        if (!P.uniform_w) P.cm_w.resize(2 * nall); else P.cm_w.clear();
        P.cm_chunk_cam.clear(); P.cm_chunk_start.clear();
        auto part = [&](int64_t lo, int64_t hi) {
            const int nbk = par.ranges(hi - lo, 1 << 15);
            std::vector<std::vector<int64_t>> cntb(nbk, std::vector<int64_t>((size_t)nc, 0));
            const Par pk{nbk};
            pk.run(hi - lo, [&](int64_t a_, int64_t b_, int tid) {
                std::vector<int64_t> &cn = cntb[tid];
                for (int64_t o = lo + a_; o < lo + b_; ++o) ++cn[P.o_cam[o]];
            }, 1);
            std::vector<int64_t> cnt((size_t)nc + 1, 0);
            for (int c = 0; c < nc; ++c) {
                int64_t run = cnt[c];
                for (int t = 0; t < nbk; ++t) { const int64_t v = cntb[t][c]; cntb[t][c] = run; run += v; }
                cnt[c + 1] = run;
            }
            pk.run(hi - lo, [&](int64_t a_, int64_t b_, int tid) {
                std::vector<int64_t> &fillc = cntb[tid];
                for (int64_t o = lo + a_; o < lo + b_; ++o) {
                    const int64_t q = lo + fillc[P.o_cam[o]]++;
                    P.cm_pt[q] = P.o_pt[o];
                    P.cm_uv[2 * q] = P.o_uv[2 * o]; P.cm_uv[2 * q + 1] = P.o_uv[2 * o + 1];
                    if (!P.uniform_w) { P.cm_w[2 * q] = P.o_w[2 * o]; P.cm_w[2 * q + 1] = P.o_w[2 * o + 1]; }
                }
            }, 1);
            for (int c = 0; c < nc; ++c)
                for (int64_t s0 = cnt[c]; s0 < cnt[c + 1]; s0 += Plan::CM_CHUNK) {
                    P.cm_chunk_cam.push_back(c); P.cm_chunk_start.push_back(lo + s0);
                }
        };
        part(0, ntiled);
        P.n_cm_chunks_tiled = (int64_t)P.cm_chunk_cam.size();
        part(ntiled, nall);
        P.cm_chunk_start.push_back(nall);
        // a chunk ends where the next one starts (same camera + CM_CHUNK, the next camera's first
        // observation, or the end of its part)
    }
    lapt("tile bookkeeping");
    // k_build_sig takes the tiled points when every one of them fits its five row blocks, the
    // interior orientation is fixed and the groups are long enough to fill a wave's lanes
    // every chunk's rows (6k + the IO columns of its tile + the right-hand-side row) fit five 16-row blocks
    const bool sg_can = P.CMAX > 0 && P.nb_tiled > 0 && P.sg_kmax > 0 && P.sg_rows_max <= 80 && P.ncolmax <= 14 && P.BT == 256;
    // the build kernel pays from about four points per group on; so does the back-substitution since round 6 (several
    // lanes per point of a short chunk; until then from about eight: C1 / C2 have 6 ... 7 points per group)
    P.sg_ok = sg_can && P.sg_npoints >= 4 * P.sg_ngroups;
    // (self-calibration: k_backsub_sig knows the usual eight IO columns only)
    const bool bs_can = sg_can && (P.ncolmax <= 6 || P.all_std8);
    // (round 6: short chunks take several lanes per point.  Self-calibration: from four points per group on -- C2: k_backsub
    // 0.066 -> k_backsub_sig 0.049 ms; fixed IO: the column-list kernel stays ahead below eight -- C1: 0.009 against 0.012 ms)
    P.sg_backsub_ok = bs_can && P.sg_npoints >= (P.ncolmax > 6 ? 4 : 8) * P.sg_ngroups;
    if (const char *e = env_get("DBAT_HIP_SIG")) {     // 0 off, 2 whenever possible
        P.sg_ok = atoi(e) == 0 ? false : (atoi(e) >= 2 ? sg_can : P.sg_ok);
        P.sg_backsub_ok = atoi(e) == 0 ? false : (atoi(e) >= 2 ? bs_can : P.sg_backsub_ok);
    }
    if (env_on("DBAT_HIP_PLAN_STATS") && P.sg_ngroups > 0)
        fprintf(stderr, "[plan] %lld signature groups, %.1f points/group, %zu chunks, k max %d, sig kernel %s\n",
                (long long)P.sg_ngroups, (double)P.sg_npoints / P.sg_ngroups, P.sg_chunk.size() / 8, P.sg_kmax,
                P.sg_ok ? "on" : "off");
    if (env_on("DBAT_HIP_PLAN_STATS") && P.tile_batch.size() > 1) {      // tile size distribution
        std::vector<int> nbt;
        for (size_t i = 0; i + 1 < P.tile_batch.size(); ++i) nbt.push_back(P.tile_batch[i + 1] - P.tile_batch[i]);
        std::sort(nbt.begin(), nbt.end());
        double s = 0; for (int v : nbt) s += v;
        fprintf(stderr, "[plan] points ordered along a %d-D curve in the principal axes of the cloud\n", P.order_dims);
        fprintf(stderr, "[plan] %zu tiles, batches/tile min %d median %d mean %.1f p90 %d p99 %d max %d; cams/tile mean %.1f\n",
                nbt.size(), nbt.front(), nbt[nbt.size() / 2], s / nbt.size(), nbt[nbt.size() * 9 / 10],
                nbt[nbt.size() * 99 / 100], nbt.back(), (double)P.tile_cams.size() / nbt.size());
    }
    lapt("kernel choice");
    build_heavy_plan(P);
    lapt("heavy / giant points: row groups, slots, pair tasks");
    return true;
}

}  // namespace dbat
