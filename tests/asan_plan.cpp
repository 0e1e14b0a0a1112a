// Host logic of the library under AddressSanitizer + UBSan: the plan (plan.hpp: serial indices,
// weights, structural rank matching, point ordering, batches, tiles, signature chunks, camera-major
// copy) is pure host code; this harness includes it directly and runs it on problems dumped by
// tests/helpers.py::dump_problem (the format of tests/abi_c_driver.c).  Built and run by
// tests/test_abi_cpu.py::test_plan_under_sanitizers; no GPU, no HIP runtime calls.
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../dbat_amd/csrc/plan.hpp"

static std::vector<std::vector<char>> g_keep;
static const void *slurp(FILE *f, size_t bytes) {
    g_keep.emplace_back(bytes ? bytes : 1);
    if (fread(g_keep.back().data(), 1, bytes, f) != bytes) { fprintf(stderr, "short read\n"); exit(3); }
    return g_keep.back().data();
}

int main(int argc, char **argv) {
    if (argc < 2) return 1;
    for (int a = 1; a < argc; ++a) {
        FILE *f = fopen(argv[a], "rb");
        int64_t h[8];
        if (!f || fread(h, sizeof h, 1, f) != 1) { fprintf(stderr, "cannot read %s\n", argv[a]); return 1; }
        dbat_hip_problem pb;
        memset(&pb, 0, sizeof pb);
        pb.abi_version = DBAT_HIP_ABI_VERSION;
        pb.n_images = (int32_t)h[0]; pb.n_points = (int32_t)h[1]; pb.n_obs = h[2];
        pb.dist_model = (int32_t)h[3]; pb.nK = (int32_t)h[4]; pb.nP = (int32_t)h[5];
        const size_t R = (size_t)h[6], nc = (size_t)h[0], np = (size_t)h[1], no = (size_t)h[2];
        pb.ip_cam = (const int32_t *)slurp(f, 4 * no); pb.ip_pt = (const int32_t *)slurp(f, 4 * no);
        pb.ip_val = (const double *)slurp(f, 16 * no); pb.ip_std = (const double *)slurp(f, 16 * no);
        pb.IO_val = (const double *)slurp(f, 8 * R * nc); pb.px_size = (const double *)slurp(f, 16 * nc);
        pb.EO_val = (const double *)slurp(f, 48 * nc); pb.OP_val = (const double *)slurp(f, 24 * np);
        pb.est_IO = (const uint8_t *)slurp(f, R * nc); pb.est_EO = (const uint8_t *)slurp(f, 6 * nc);
        pb.est_OP = (const uint8_t *)slurp(f, 3 * np);
        pb.IO_block = (const int32_t *)slurp(f, 4 * R * nc); pb.EO_block = (const int32_t *)slurp(f, 24 * nc);
        pb.prior_IO_use = (const uint8_t *)slurp(f, R * nc); pb.prior_IO_val = (const double *)slurp(f, 8 * R * nc);
        pb.prior_IO_std = (const double *)slurp(f, 8 * R * nc);
        pb.prior_EO_use = (const uint8_t *)slurp(f, 6 * nc); pb.prior_EO_val = (const double *)slurp(f, 48 * nc);
        pb.prior_EO_std = (const double *)slurp(f, 48 * nc);
        pb.prior_OP_use = (const uint8_t *)slurp(f, 3 * np); pb.prior_OP_val = (const double *)slurp(f, 24 * np);
        pb.prior_OP_std = (const double *)slurp(f, 24 * np);
        fclose(f);
        for (int nranks = 1; nranks <= 3; nranks += 2)
            for (int rank = 0; rank < nranks; ++rank) {
                pb.shard_rank = rank; pb.shard_count = nranks;
                dbat::Plan P;
                if (!dbat::build_plan(pb, P, true)) { printf("%s: rejected: %s\n", argv[a], P.err.c_str()); continue; }
                int64_t sig_pts = 0;
                for (size_t q = 0; q < P.sg_chunk.size() / 8; ++q) sig_pts += P.sg_chunk[8 * q + 1];
                printf("%s rank %d/%d: n %" PRId64 " m %" PRId64 " obs %zu batches %zu tiles %zu chunks %zu (%" PRId64 " pts) rank_ok %d\n",
                       argv[a], rank, nranks, P.n, P.m, P.o_cam.size(), P.batch_start.size() - 1,
                       P.tile_batch.size() - 1, P.sg_chunk.size() / 8, sig_pts, (int)P.rank_ok);
            }
        g_keep.clear();
    }
    return 0;
}
