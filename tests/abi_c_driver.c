/* A compiled, non-Python caller of the C ABI (plain C11, `gcc -std=c11 -Iinclude`).
 *
 *   abi_c_driver layout          print sizeof/offsetof of the three ABI structs as JSON
 *                                (tests/test_abi_cpu.py compares them with the ctypes
 *                                mirror in dbat_amd/_hip.py)
 *   abi_c_driver plan  FILE      host-only entry points on a problem read from FILE
 *   abi_c_driver solve FILE DAMP create -> solve -> final residuals -> destroy on the GPU
 *
 * FILE is a flat little-endian dump written by tests/helpers.py::dump_problem:
 *   int64 header[8] = {n_images, n_points, n_obs, dist_model, nK, nP, nIOrows, 0}
 *   then the arrays of dbat_hip_problem in declaration order.
 * This is what a MEX gateway (mex/dbat_hip_mex.cpp) or any other host language does with
 * the header: nothing but the declarations of include/dbat_hip.h. */
#include <inttypes.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dbat_hip.h"

#define FIELD(T, f) printf("%s\"%s\": %zu", first++ ? ", " : "", #f, offsetof(T, f))

static void layout(void) {
    int first = 0;
    printf("{\"dbat_hip_problem\": {\"sizeof\": %zu, \"offsets\": {", sizeof(dbat_hip_problem));
    FIELD(dbat_hip_problem, abi_version); FIELD(dbat_hip_problem, n_images); FIELD(dbat_hip_problem, n_points);
    FIELD(dbat_hip_problem, n_obs); FIELD(dbat_hip_problem, dist_model); FIELD(dbat_hip_problem, nK);
    FIELD(dbat_hip_problem, nP); FIELD(dbat_hip_problem, ip_cam); FIELD(dbat_hip_problem, ip_pt);
    FIELD(dbat_hip_problem, ip_val); FIELD(dbat_hip_problem, ip_std); FIELD(dbat_hip_problem, IO_val);
    FIELD(dbat_hip_problem, px_size); FIELD(dbat_hip_problem, EO_val); FIELD(dbat_hip_problem, OP_val);
    FIELD(dbat_hip_problem, est_IO); FIELD(dbat_hip_problem, est_EO); FIELD(dbat_hip_problem, est_OP);
    FIELD(dbat_hip_problem, IO_block); FIELD(dbat_hip_problem, EO_block);
    FIELD(dbat_hip_problem, prior_IO_use); FIELD(dbat_hip_problem, prior_IO_val); FIELD(dbat_hip_problem, prior_IO_std);
    FIELD(dbat_hip_problem, prior_EO_use); FIELD(dbat_hip_problem, prior_EO_val); FIELD(dbat_hip_problem, prior_EO_std);
    FIELD(dbat_hip_problem, prior_OP_use); FIELD(dbat_hip_problem, prior_OP_val); FIELD(dbat_hip_problem, prior_OP_std);
    FIELD(dbat_hip_problem, device); FIELD(dbat_hip_problem, shard_rank); FIELD(dbat_hip_problem, shard_count);
    first = 0;
    printf("}}, \"dbat_hip_options\": {\"sizeof\": %zu, \"offsets\": {", sizeof(dbat_hip_options));
    FIELD(dbat_hip_options, damping); FIELD(dbat_hip_options, max_iter); FIELD(dbat_hip_options, conv_tol);
    FIELD(dbat_hip_options, abs_term); FIELD(dbat_hip_options, singular_test); FIELD(dbat_hip_options, store_trace);
    FIELD(dbat_hip_options, mu); FIELD(dbat_hip_options, alpha_min); FIELD(dbat_hip_options, lambda0);
    FIELD(dbat_hip_options, lambda_min); FIELD(dbat_hip_options, rho_bad); FIELD(dbat_hip_options, rho_good);
    FIELD(dbat_hip_options, delta0);
    FIELD(dbat_hip_options, term_fun); FIELD(dbat_hip_options, term_user);
    FIELD(dbat_hip_options, veto_fun); FIELD(dbat_hip_options, veto_user);
    FIELD(dbat_hip_options, trace_fun); FIELD(dbat_hip_options, trace_user);
    first = 0;
    printf("}}, \"dbat_hip_result\": {\"sizeof\": %zu, \"offsets\": {", sizeof(dbat_hip_result));
    FIELD(dbat_hip_result, code); FIELD(dbat_hip_result, iters); FIELD(dbat_hip_result, n_res);
    FIELD(dbat_hip_result, n_damp); FIELD(dbat_hip_result, n_trace); FIELD(dbat_hip_result, sigma0);
    FIELD(dbat_hip_result, time_s); FIELD(dbat_hip_result, n_residual_evals);
    FIELD(dbat_hip_result, n_linearizations); FIELD(dbat_hip_result, n_solves);
    FIELD(dbat_hip_result, n_trace_only); FIELD(dbat_hip_result, stage_s);
    printf("}}, \"abi_version\": %d, \"unique_id_bytes\": %d}\n", DBAT_HIP_ABI_VERSION, DBAT_HIP_UNIQUE_ID_BYTES);
}

static void *slurp(FILE *f, size_t bytes) {
    void *p = malloc(bytes ? bytes : 1);
    if (!p || fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "short read\n"); exit(3); }
    return p;
}

static int load(const char *path, dbat_hip_problem *pb, int *nIOrows) {
    FILE *f = fopen(path, "rb");
    int64_t h[8];
    if (!f || fread(h, sizeof h, 1, f) != 1) return -1;
    memset(pb, 0, sizeof *pb);
    pb->abi_version = DBAT_HIP_ABI_VERSION;
    pb->n_images = (int32_t)h[0]; pb->n_points = (int32_t)h[1]; pb->n_obs = h[2];
    pb->dist_model = (int32_t)h[3]; pb->nK = (int32_t)h[4]; pb->nP = (int32_t)h[5];
    const size_t R = (size_t)h[6], nc = (size_t)h[0], np = (size_t)h[1], no = (size_t)h[2];
    *nIOrows = (int)R;
    pb->ip_cam = slurp(f, 4 * no); pb->ip_pt = slurp(f, 4 * no);
    pb->ip_val = slurp(f, 16 * no); pb->ip_std = slurp(f, 16 * no);
    pb->IO_val = slurp(f, 8 * R * nc); pb->px_size = slurp(f, 16 * nc);
    pb->EO_val = slurp(f, 48 * nc); pb->OP_val = slurp(f, 24 * np);
    pb->est_IO = slurp(f, R * nc); pb->est_EO = slurp(f, 6 * nc); pb->est_OP = slurp(f, 3 * np);
    pb->IO_block = slurp(f, 4 * R * nc); pb->EO_block = slurp(f, 24 * nc);
    pb->prior_IO_use = slurp(f, R * nc); pb->prior_IO_val = slurp(f, 8 * R * nc); pb->prior_IO_std = slurp(f, 8 * R * nc);
    pb->prior_EO_use = slurp(f, 6 * nc); pb->prior_EO_val = slurp(f, 48 * nc); pb->prior_EO_std = slurp(f, 48 * nc);
    pb->prior_OP_use = slurp(f, 3 * np); pb->prior_OP_val = slurp(f, 24 * np); pb->prior_OP_std = slurp(f, 24 * np);
    pb->device = 0; pb->shard_rank = 0; pb->shard_count = 1;
    fclose(f);
    return 0;
}

#define CHECK(call)                                                                   \
    do {                                                                              \
        int rc_ = (call);                                                             \
        if (rc_ != DBAT_HIP_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, dbat_hip_last_error()); return 2; } \
    } while (0)

int main(int argc, char **argv) {
    if (argc >= 2 && !strcmp(argv[1], "layout")) { layout(); return 0; }
    if (argc < 3) { fprintf(stderr, "usage: %s layout | plan FILE | solve FILE DAMPING\n", argv[0]); return 1; }
    dbat_hip_problem pb;
    int nIOrows = 0;
    if (load(argv[2], &pb, &nIOrows)) { fprintf(stderr, "cannot read %s\n", argv[2]); return 1; }
    if (dbat_hip_abi_version() != DBAT_HIP_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 1; }
    int64_t n = 0, m = 0, nIO = 0, nEO = 0, nOP = 0, lo = 0, hi = 0;
    CHECK(dbat_hip_plan(&pb, &n, &m, &nIO, &nEO, &nOP, &lo, &hi));
    int32_t rank_ok = 0;
    CHECK(dbat_hip_plan_structural_rank_ok(&pb, &rank_ok));
    double *x = malloc(sizeof(double) * (size_t)(n ? n : 1));
    CHECK(dbat_hip_plan_serialize(&pb, x));
    double sx = 0;
    for (int64_t i = 0; i < n; ++i) sx += x[i];
    if (!strcmp(argv[1], "plan")) {
        printf("{\"n\": %" PRId64 ", \"m\": %" PRId64 ", \"nIO\": %" PRId64 ", \"nEO\": %" PRId64 ", \"nOP\": %" PRId64
               ", \"rank_ok\": %d, \"sum_x0\": %.17g}\n", n, m, nIO, nEO, nOP, rank_ok, sx);
        return 0;
    }
    if (argc < 4) return 1;
    dbat_hip_handle *h = NULL;
    CHECK(dbat_hip_create(&pb, &h));
    dbat_hip_options opt;
    CHECK(dbat_hip_default_options((int32_t)atoi(argv[3]), &opt));
    opt.store_trace = 0;
    dbat_hip_result res;
    double *rr = malloc(sizeof(double) * (size_t)(opt.max_iter + 3));
    double *damp = malloc(sizeof(double) * (size_t)(2 * opt.max_iter + 4));
    CHECK(dbat_hip_solve(h, &opt, x, &res, rr, damp, NULL, NULL));
    double *ru = calloc((size_t)m, sizeof(double)), *rw = calloc((size_t)m, sizeof(double));
    CHECK(dbat_hip_final_residuals(h, ru, rw));
    double f = 0;
    for (int64_t i = 0; i < m; ++i) f += rw[i] * rw[i];
    sx = 0;
    for (int64_t i = 0; i < n; ++i) sx += x[i];
    printf("{\"code\": %d, \"iters\": %d, \"sigma0\": %.17g, \"sum_x\": %.17g, \"rtr\": %.17g, \"x\": [", res.code, res.iters,
           res.sigma0, sx, f);
    for (int64_t i = 0; i < n; ++i) printf("%s%.17g", i ? ", " : "", x[i]);
    printf("]}\n");
    dbat_hip_destroy(h);
    return 0;
}
