/*
 * abi_harness.c -- include/aligner_hip.h compiled as C99 and called the way a C / Rust-FFI consumer would call it.
 *
 * The GPU test-suite reaches the library through ctypes, which re-declares the structs by hand (aligner_amd/_ffi.py); this
 * program is the check that the HEADER ITSELF is a usable C interface: it is built with `gcc -std=c99 -Wall -Werror -Iinclude`
 * (tests/test_host_logic.py builds it on every CPU run; no GPU is needed to compile or link), pins the record layouts at
 * compile time, and -- run on a GPU box by tests/test_gpu_parity.py -- aligns one pair through aln_align_pair
 * (AlignerTrait::perform_alignment, aligner-core/src/lib.rs:27-40) and three pairs through aln_align_batch (the batch site
 * statistics/mod.rs:255-286), printing everything it gets back; the Python test compares the printout with the oracle.
 *
 * usage: abi_harness <case file>          (written by the test)
 *   line 1: semantics del ext rows cols
 *   then rows*cols matrix values, then N and N query codes, then M and M target codes
 * output: one `key value...` line per fact; exit status 0 iff every call returned ALN_OK.
 */
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "aligner_hip.h"

/* compile-time layout pins (C99 has no _Static_assert): the sizes and offsets every binding relies on */
#define PIN(name, cond) typedef char pin_##name[(cond) ? 1 : -1]
PIN(result_size, sizeof(aln_pair_result) == 48);
PIN(result_f, offsetof(aln_pair_result, f) == 0);
PIN(result_score, offsetof(aln_pair_result, score) == 8);
PIN(result_end_y, offsetof(aln_pair_result, end_y) == 16);
PIN(result_start_y, offsetof(aln_pair_result, start_y) == 24);
PIN(result_aln_len, offsetof(aln_pair_result, aln_len) == 32);
PIN(result_status, offsetof(aln_pair_result, status) == 36);
PIN(result_passes, offsetof(aln_pair_result, passes) == 40);
PIN(result_flags, offsetof(aln_pair_result, flags) == 44);
PIN(params_size, sizeof(aln_params) == 64);
PIN(params_del, offsetof(aln_params, del) == 8);
PIN(params_matrix, offsetof(aln_params, matrix) == 24);
PIN(params_rows, offsetof(aln_params, rows) == 32);
PIN(params_stride, offsetof(aln_params, row_stride) == 40);
PIN(params_outputs, offsetof(aln_params, outputs) == 48);
PIN(params_blank, offsetof(aln_params, blank_code) == 52);
PIN(params_passes, offsetof(aln_params, max_passes) == 56);

static void print_codes(const char *key, const uint8_t *v, uint32_t n)
{
    uint32_t i;
    printf("%s", key);
    for (i = 0; i < n; ++i) printf(" %u", (unsigned)v[i]);
    printf("\n");
}

static void print_result(const char *key, const aln_pair_result *r)
{
    printf("%s status %d f %.17g score %.17g end %u %u start %u %u len %u\n", key, (int)r->status, r->f, r->score, (unsigned)r->end_y,
           (unsigned)r->end_x, (unsigned)r->start_y, (unsigned)r->start_x, (unsigned)r->aln_len);
}

int main(int argc, char **argv)
{
    FILE *fp;
    int semantics, st = 0, rc = 0;
    double del, ext, *matrix;
    unsigned rows, cols, N, M, i;
    uint8_t *q, *t;
    aln_ctx *ctx;
    aln_params p;

    printf("abi_version_header %d sizeof_result %u sizeof_params %u\n", ALN_ABI_VERSION, (unsigned)sizeof(aln_pair_result), (unsigned)sizeof(aln_params));
    if (argc < 2) {          /* layout only (what the CPU test runs: the library loads and reports the header's ABI version) */
        printf("abi_version_library %d\n", aln_abi_version());
        return aln_abi_version() == ALN_ABI_VERSION ? 0 : 1;
    }
    fp = fopen(argv[1], "r");
    if (!fp) { perror(argv[1]); return 2; }
    if (fscanf(fp, "%d %lf %lf %u %u", &semantics, &del, &ext, &rows, &cols) != 5) return 2;
    matrix = (double *)malloc(sizeof(double) * rows * cols);
    for (i = 0; i < rows * cols; ++i) if (fscanf(fp, "%lf", &matrix[i]) != 1) return 2;
    if (fscanf(fp, "%u", &N) != 1) return 2;
    q = (uint8_t *)malloc(N + 1);
    for (i = 0; i < N; ++i) { unsigned c; if (fscanf(fp, "%u", &c) != 1) return 2; q[i] = (uint8_t)c; }
    if (fscanf(fp, "%u", &M) != 1) return 2;
    t = (uint8_t *)malloc(M + 1);
    for (i = 0; i < M; ++i) { unsigned c; if (fscanf(fp, "%u", &c) != 1) return 2; t[i] = (uint8_t)c; }
    fclose(fp);

    ctx = aln_create(0, &st);
    if (!ctx) { printf("aln_create failed: %d %s\n", st, aln_last_error()); return 3; }
    printf("devices %d\n", aln_device_count(ctx));

    memset(&p, 0, sizeof p);
    p.semantics = semantics; p.del = del; p.ext = ext;
    p.matrix = matrix; p.rows = rows; p.cols = cols; p.row_stride = cols;
    p.outputs = ALN_OUT_SCORE | ALN_OUT_TRACEBACK | ALN_OUT_DIRECTIONS | ALN_OUT_H_MATRIX;
    p.blank_code = 98;

    {   /* ---- one perform_alignment call, every output */
        aln_pair_result r;
        const size_t cap = (size_t)N + M + 2, cells = (size_t)(N + 1) * (M + 1);
        uint8_t *qa = (uint8_t *)malloc(cap), *ta = (uint8_t *)malloc(cap), *dirs = (uint8_t *)malloc(cells);
        double *h = (double *)malloc(sizeof(double) * cells);
        size_t c;
        st = aln_align_pair(ctx, &p, q, N, t, M, &r, qa, ta, dirs, h);
        if (st != ALN_OK) { printf("aln_align_pair failed: %d %s\n", st, aln_last_error()); rc = 4; }
        print_result("pair", &r);
        if (st == ALN_OK) {
            print_codes("pair_q_aln", qa, r.aln_len);
            print_codes("pair_t_aln", ta, r.aln_len);
            print_codes("pair_dirs", dirs, (uint32_t)cells);
            printf("pair_h");
            for (c = 0; c < cells; ++c) printf(" %.17g", h[c]);
            printf("\n");
        }
        free(qa); free(ta); free(dirs); free(h);
    }
    {   /* ---- the batch driver: (q, t), (t, q), (q, q) out of one residue buffer, strings in the cumulative layout */
        const uint64_t q_off[3] = {0, N, 0}, q_len[3] = {N, M, N}, t_off[3] = {N, 0, 0}, t_len[3] = {M, N, N};
        uint64_t tb_off[3], total = 0;
        aln_pair_result res[3];
        uint8_t *seqs = (uint8_t *)malloc((size_t)N + M + 1), *tb;
        int j;
        memcpy(seqs, q, N); memcpy(seqs + N, t, M);
        for (j = 0; j < 3; ++j) { tb_off[j] = total; total += 2 * (q_len[j] + t_len[j] + 2); }
        tb = (uint8_t *)malloc((size_t)total);
        p.outputs = ALN_OUT_SCORE | ALN_OUT_TRACEBACK;
        st = aln_align_batch(ctx, &p, seqs, q_off, q_len, t_off, t_len, 3, res, tb, tb_off);
        if (st != ALN_OK) { printf("aln_align_batch failed: %d %s\n", st, aln_last_error()); rc = 5; }
        for (j = 0; j < 3 && st == ALN_OK; ++j) {
            char key[32];
            const uint64_t cap = q_len[j] + t_len[j] + 2;
            sprintf(key, "batch%d", j); print_result(key, &res[j]);
            if (res[j].status != ALN_OK) continue;
            sprintf(key, "batch%d_q_aln", j); print_codes(key, tb + tb_off[j], res[j].aln_len);
            sprintf(key, "batch%d_t_aln", j); print_codes(key, tb + tb_off[j] + cap, res[j].aln_len);
        }
        free(seqs); free(tb);
    }
    aln_destroy(ctx);
    free(matrix); free(q); free(t);
    return rc;
}
