/*
 * aligner_oracle.c -- CPU restatement of ikramanop/aligner's DP matrix fill + traceback.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under aligner_amd/ (the product path) may link, import or
 * call this file.  Allowed users: tests/, __graft_entry__.smoke() (as the checker) and
 * bench.py's cpu_baseline leg (as the thing timed *beside* the GPU, never instead of it).
 *
 * The reference is Rust and cannot be built here (no cargo/rustc in the image), so this is a
 * restatement, function by function, of:
 *   CORE_GLOBAL   aligner-core/src/simple/mod.rs:42-145   (SimpleGlobalAligner::perform_alignment)
 *   CORE_LOCAL    aligner-core/src/simple/mod.rs:168-264  (SimpleLocalAligner::perform_alignment)
 *   tie rules     aligner-core/src/enums.rs:17-47         (Direction::get_direction[_with_beginning])
 *   LEGACY_GLOBAL src/align/aligner_core.rs:96-183        (SimpleAligner::global_alignment, i32)
 *   LEGACY_LOCAL  src/align/aligner_core.rs:185-269       (SimpleAligner::local_alignment, i32)
 *   midline/freq  aligner-core/src/alignment.rs:12-43
 *   PWM_LOCAL     aligner-core/src/pwm/mod.rs:29-126     (PWMAligner::perform_alignment)
 * Third-party arithmetic not in the reference tree: ndarray-stats 0.5.0 (Cargo.lock:1291)
 * QuantileExt::argmax / ::max, used at simple/mod.rs:212,247 -- restated as "first maximum in
 * logical row-major order, replace only on strictly greater" (published behaviour of 0.5.0).
 *
 * Parity pinning: LEGACY_* and CORE_GLOBAL(del==ext) are pinned by the reference's own golden
 * matrices (src/tests/test_alignment.rs:14-67,106-159 -> tests/golden/legacy_kat.json).  CORE_LOCAL has no
 * reference test: its end-cell tie rule and row-1 penalty carry-over are "parity unpinned" beyond
 * source reading.
 *
 * The memory behaviour is layout-faithful on purpose (dense row-major (M+1)x(N+1) f64 H plus a
 * 1-byte direction matrix, filled query-outer / target-inner, then separate argmax and max passes)
 * so that timing this file is timing the reference's algorithm, not a tuned variant.
 */
#include <math.h>
#include <float.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "aligner_oracle.h"

/* enums.rs:18-28 */
static inline uint8_t pick(double top, double left, double diag, double *out)
{
    double m = fmax(fmax(top, left), diag);
    *out = m;
    if (fabs(m - top) < DBL_EPSILON) return ORC_TOP;
    if (fabs(m - left) < DBL_EPSILON) return ORC_LEFT;
    return ORC_DIAGONAL;
}

/* enums.rs:30-46 */
static inline uint8_t pick_b(double top, double left, double diag, double *out)
{
    double m = fmax(fmax(top, left), diag);
    *out = m;
    if (m == 0.0) return ORC_BEGINNING;
    if (fabs(m - top) < DBL_EPSILON) return ORC_TOP;
    if (fabs(m - left) < DBL_EPSILON) return ORC_LEFT;
    return ORC_DIAGONAL;
}

/* shared traceback loop: simple/mod.rs:107-127, :220-242; src/align/aligner_core.rs:153-173,:238-258.
 * qa/ta already hold the seed pair; returns the new length; (cy,cx) updated to the stop cell. */
static size_t walk(const uint8_t *D, size_t W, const uint8_t *q, const uint8_t *t, uint8_t blank,
                   size_t *cy, size_t *cx, uint8_t *qa, uint8_t *ta, size_t len)
{
    for (;;) {
        uint8_t d = D[*cy * W + *cx];
        if (d == ORC_BEGINNING) break;
        if (d == ORC_TOP) {
            qa[len] = blank; ta[len] = t[*cy - 1]; (*cy)--;
        } else if (d == ORC_LEFT) {
            qa[len] = q[*cx - 1]; ta[len] = blank; (*cx)--;
        } else {
            qa[len] = q[*cx - 1]; ta[len] = t[*cy - 1]; (*cx)--; (*cy)--;
        }
        len++;
    }
    return len;
}

static void reverse(uint8_t *a, size_t n)
{
    for (size_t i = 0, j = n ? n - 1 : 0; i < j; i++, j--) { uint8_t x = a[i]; a[i] = a[j]; a[j] = x; }
}

static int check_codes(const uint8_t *s, size_t n, uint32_t dim)
{
    for (size_t i = 0; i < n; i++) if (s[i] >= dim) return 0;
    return 1;
}

int orc_align(const orc_params *p, const uint8_t *q, size_t N, const uint8_t *t, size_t M,
              orc_result *res, uint8_t *qa, uint8_t *ta, double *H_out, uint8_t *D_out)
{
    memset(res, 0, sizeof(*res));
    /* simple/mod.rs:49-51,175-177 */
    if (p->heuristics_present && (p->semantics == ORC_CORE_GLOBAL || p->semantics == ORC_CORE_LOCAL))
        return res->status = ORC_ERR_UNNECESSARY_ARGUMENT;
    /* reference panics: last().unwrap() on empty (simple/mod.rs:103-104), index underflow (:214-215) */
    if (N == 0 || M == 0) return res->status = ORC_ERR_EMPTY_SEQUENCE;
    /* ndarray OOB panic at simple/mod.rs:85,198 */
    if (!check_codes(t, M, p->rows) || !check_codes(q, N, p->cols)) return res->status = ORC_ERR_CODE_OUT_OF_RANGE;

    const size_t W = N + 1, Hh = M + 1;
    const double *S = p->matrix;
    const int64_t rs = p->row_stride;
    uint8_t *D = (uint8_t *)malloc(Hh * W);
    if (!D) return res->status = ORC_ERR_OOM;
    /* from_shape_fn(dim, |_| Direction::Beginning) */
    memset(D, ORC_BEGINNING, Hh * W);

    if (p->semantics == ORC_CORE_GLOBAL || p->semantics == ORC_CORE_LOCAL) {
        const int local = p->semantics == ORC_CORE_LOCAL;
        const double del = p->del, ext = p->ext;
        double *H = (double *)calloc(Hh * W, sizeof(double));
        if (!H) { free(D); return res->status = ORC_ERR_OOM; }
        if (!local) {
            /* simple/mod.rs:59-70 */
            for (size_t x = 1; x < W; x++) { H[x] = -(double)x * del; D[x] = ORC_LEFT; }
            for (size_t y = 1; y < Hh; y++) { H[y * W] = -(double)y * del; D[y * W] = ORC_TOP; }
            H[N] = -((double)N + 1.0) * del;
            H[M * W] = -((double)M + 1.0) * del;
        }
        double penalty = del; /* :72 / :185 */
        for (size_t x = 1; x <= N; x++) {         /* :74 / :187 query outer */
            const size_t qc = q[x - 1];
            for (size_t y = 1; y <= M; y++) {     /* :75 / :188 target inner */
                const size_t tc = t[y - 1];
                double v;
                uint8_t d;
                const double top = H[(y - 1) * W + x] - penalty;
                const double left = H[y * W + x - 1] - penalty;
                const double diag = H[(y - 1) * W + x - 1] + S[(int64_t)tc * rs + (int64_t)qc];
                d = local ? pick_b(top, left, diag, &v) : pick(top, left, diag, &v);
                penalty = (d != ORC_BEGINNING) ? ext : del; /* :88-92 / :201-205 */
                H[y * W + x] = v;
                D[y * W + x] = d;
            }
        }
        size_t cx, cy, len = 0;
        if (!local) {
            /* :99-143 */
            cx = N; cy = M;
            qa[0] = q[N - 1]; ta[0] = t[M - 1]; len = 1;
            len = walk(D, W, q, t, p->blank_code, &cy, &cx, qa, ta, len);
            res->f = 0.0;
            res->score = H[M * W + N];
            res->end_y = (uint32_t)M; res->end_x = (uint32_t)N;
            res->start_y = (uint32_t)cy; res->start_x = (uint32_t)cx;
            res->coords[0] = 1; res->coords[1] = N; res->coords[2] = 1; res->coords[3] = M;
        } else {
            /* :212 argmax (ndarray-stats: first max, row-major, whole array incl. borders) */
            size_t best = 0;
            for (size_t i = 1; i < Hh * W; i++) if (H[i] > H[best]) best = i;
            const size_t my = best / W, mx = best % W;
            if (my == 0 || mx == 0) { /* :214-215 usize underflow / OOB panic */
                free(H); free(D);
                return res->status = ORC_ERR_NO_POSITIVE_CELL;
            }
            qa[0] = q[mx - 1]; ta[0] = t[my - 1]; len = 1;
            cx = mx; cy = my;
            len = walk(D, W, q, t, p->blank_code, &cy, &cx, qa, ta, len);
            /* :247 second full pass */
            double f = H[0];
            for (size_t i = 1; i < Hh * W; i++) if (H[i] > f) f = H[i];
            res->f = f;
            res->score = f;
            res->end_y = (uint32_t)my; res->end_x = (uint32_t)mx;
            res->start_y = (uint32_t)cy; res->start_x = (uint32_t)cx;
            /* :255-258 */
            res->coords[0] = cx + 1; res->coords[1] = mx + 1; res->coords[2] = cy + 1; res->coords[3] = my + 1;
        }
        reverse(qa, len); reverse(ta, len);
        res->aln_len = (uint32_t)len;
        if (H_out) memcpy(H_out, H, Hh * W * sizeof(double));
        if (D_out) memcpy(D_out, D, Hh * W);
        free(H); free(D);
        return res->status = ORC_OK;
    }

    /* legacy: i32, linear gap (src/align/aligner_core.rs) */
    {
        const int local = p->semantics == ORC_LEGACY_LOCAL;
        const int32_t del = (int32_t)p->del;
        int32_t *H = (int32_t *)calloc(Hh * W, sizeof(int32_t));
        if (!H) { free(D); return res->status = ORC_ERR_OOM; }
        int32_t max_f = 0; size_t max_x = 0, max_y = 0;
        if (!local) {
            /* :104-117 */
            for (size_t x = 1; x < W; x++) { H[x] = -(int32_t)x * del; D[x] = ORC_LEFT; }
            for (size_t y = 1; y < Hh; y++) { H[y * W] = -(int32_t)y * del; D[y * W] = ORC_TOP; }
            H[M * W] = -((int32_t)M + 1) * del;
            H[N] = -((int32_t)N + 1) * del;
        }
        for (size_t x = 1; x <= N; x++) {
            const size_t qc = q[x - 1];
            for (size_t y = 1; y <= M; y++) {
                const size_t tc = t[y - 1];
                const int32_t top = H[(y - 1) * W + x] - del;
                const int32_t left = H[y * W + x - 1] - del;
                const int32_t diag = H[(y - 1) * W + x - 1] + (int32_t)S[(int64_t)tc * rs + (int64_t)qc];
                int32_t m = top > left ? top : left;
                if (diag > m) m = diag;
                if (local && m < 0) m = 0; /* :210 */
                H[y * W + x] = m;
                if (local && m == 0) D[y * W + x] = ORC_BEGINNING;      /* :214 */
                else if (m == top) D[y * W + x] = ORC_TOP;              /* :136 / :216 */
                else if (m == left) D[y * W + x] = ORC_LEFT;
                else if (m == diag) D[y * W + x] = ORC_DIAGONAL;
                if (local && m >= max_f) { max_f = m; max_x = x - 1; max_y = y - 1; } /* :224-228 */
            }
        }
        size_t cx, cy, len;
        if (!local) {
            cx = N - 1; cy = M - 1;                 /* :146-147 */
            qa[0] = q[N - 1]; ta[0] = t[M - 1];     /* :148-151 */
            res->score = H[M * W + N];
            res->end_y = (uint32_t)M; res->end_x = (uint32_t)N;
        } else {
            cx = max_x; cy = max_y;                 /* :232-233 */
            qa[0] = q[max_x]; ta[0] = t[max_y];     /* :234-237 */
            res->score = max_f;
            res->end_y = (uint32_t)(max_y + 1); res->end_x = (uint32_t)(max_x + 1);
        }
        len = 1;
        len = walk(D, W, q, t, p->blank_code, &cy, &cx, qa, ta, len);
        reverse(qa, len); reverse(ta, len);
        res->f = res->score;
        res->start_y = (uint32_t)cy; res->start_x = (uint32_t)cx;
        res->coords[0] = cx + 1; res->coords[1] = res->end_x; res->coords[2] = cy + 1; res->coords[3] = res->end_y;
        res->aln_len = (uint32_t)len;
        if (H_out) for (size_t i = 0; i < Hh * W; i++) H_out[i] = (double)H[i];
        if (D_out) memcpy(D_out, D, Hh * W);
        free(H); free(D);
        return res->status = ORC_OK;
    }
}

/* pwm/mod.rs:29-126 */
int orc_align_pwm(const orc_params *p, const uint8_t *seq, size_t Q, orc_result *res, uint32_t *numbered, uint8_t *qal,
                  double *H_out, uint8_t *D_out)
{
    memset(res, 0, sizeof(*res));
    if (p->heuristics_present) return res->status = ORC_ERR_UNNECESSARY_ARGUMENT;   /* :36-38 */
    if (p->rows != 4) return res->status = ORC_ERR_MATRIX_SHAPE;                    /* :40-42 */
    for (size_t i = 0; i < Q; i++) if (seq[i] >= p->rows) return res->status = ORC_ERR_CODE_OUT_OF_RANGE;
    const size_t Wd = p->cols;            /* numbered_sequence = 1..=W */
    const size_t W = Wd + 1, Hh = Q + 1;  /* dim = (Q+1, W+1) */
    const double *S = p->matrix;
    const int64_t rs = p->row_stride;
    const double del = p->del, ext = p->ext;
    double *H = (double *)calloc(Hh * W, sizeof(double));
    uint8_t *D = (uint8_t *)malloc(Hh * W);
    if (!H || !D) { free(H); free(D); return res->status = ORC_ERR_OOM; }
    memset(D, ORC_BEGINNING, Hh * W);
    double penalty = del;                 /* :50 */
    for (size_t x = 1; x <= Wd; x++) {    /* :52 columns outer */
        for (size_t y = 1; y <= Q; y++) { /* :53 query inner */
            double v;
            const double top = H[(y - 1) * W + x] - penalty;
            const double left = H[y * W + x - 1] - penalty;
            const double diag = H[(y - 1) * W + x - 1] + S[(int64_t)seq[y - 1] * rs + (int64_t)(x - 1)];
            const uint8_t d = pick_b(top, left, diag, &v);
            penalty = (d != ORC_BEGINNING) ? ext : del;
            H[y * W + x] = v;
            D[y * W + x] = d;
        }
    }
    size_t best = 0;                      /* :76 argmax, first maximum in row-major order */
    for (size_t i = 1; i < Hh * W; i++) if (H[i] > H[best]) best = i;
    const size_t my = best / W, mx = best % W;
    size_t cy = my, cx = mx, len = 0;
    for (;;) {                            /* :81-103, no seed pair */
        const uint8_t d = D[cy * W + cx];
        if (d == ORC_BEGINNING) break;
        if (d == ORC_TOP) { numbered[len] = 0; qal[len] = seq[cy - 1]; cy--; }
        else if (d == ORC_LEFT) { numbered[len] = (uint32_t)cx; qal[len] = p->blank_code; cx--; }
        else { numbered[len] = (uint32_t)cx; qal[len] = seq[cy - 1]; cx--; cy--; }
        len++;
    }
    for (size_t i = 0, j = len ? len - 1 : 0; i < j; i++, j--) {
        uint32_t a = numbered[i]; numbered[i] = numbered[j]; numbered[j] = a;
        uint8_t b = qal[i]; qal[i] = qal[j]; qal[j] = b;
    }
    double f = H[0];                      /* :108 */
    for (size_t i = 1; i < Hh * W; i++) if (H[i] > f) f = H[i];
    res->f = f; res->score = f;
    res->end_y = (uint32_t)my; res->end_x = (uint32_t)mx;
    res->start_y = (uint32_t)cy; res->start_x = (uint32_t)cx;
    res->coords[0] = cx + 1; res->coords[1] = mx + 1; res->coords[2] = cy + 1; res->coords[3] = my + 1;   /* :118-121 */
    res->aln_len = (uint32_t)len;
    if (H_out) memcpy(H_out, H, Hh * W * sizeof(double));
    if (D_out) memcpy(D_out, D, Hh * W);
    free(H); free(D);
    return res->status = ORC_OK;
}

/* alignment.rs:25-42 */
void orc_midline(const uint8_t *qa, const uint8_t *ta, size_t len, const double *S, int64_t row_stride,
                 uint8_t blank, uint8_t pos, uint8_t *out)
{
    for (size_t i = 0; i < len; i++) {
        uint8_t x = qa[i], y = ta[i];
        if (x == y) out[i] = x;
        else if (x != blank && y != blank && S[(int64_t)y * row_stride + x] >= 0.0) out[i] = pos;
        else out[i] = blank;
    }
}

/* alignment.rs:13-23 */
void orc_frequency_matrix(const uint8_t *qa, const uint8_t *ta, size_t len, uint8_t blank, uint32_t volume, double *out)
{
    memset(out, 0, sizeof(double) * volume * volume);
    for (size_t i = 0; i < len; i++) {
        uint8_t x = qa[i], y = ta[i];
        if (x != blank && y != blank) out[(size_t)y * volume + x] += 1.0;
    }
}

/* ---- batch driver: static partition over threads, mirroring the reference's only batch site
 * (aligner-core/src/statistics/mod.rs:255-286: THREADS std::threads, each a contiguous share). ---- */
typedef struct {
    const orc_params *p; const uint8_t *seqs; const uint64_t *q_off, *q_len, *t_off, *t_len;
    size_t lo, hi; orc_result *res; uint8_t *tb; const uint64_t *tb_off;
} job_t;

static void *worker(void *arg)
{
    job_t *j = (job_t *)arg;
    for (size_t i = j->lo; i < j->hi; i++) {
        size_t N = j->q_len[i], M = j->t_len[i];
        uint8_t *qa, *ta, *tmp = NULL;
        if (j->tb) { qa = j->tb + j->tb_off[i]; ta = qa + (M + N + 2); }
        else { tmp = (uint8_t *)malloc(2 * (M + N + 2)); qa = tmp; ta = tmp + (M + N + 2); }
        orc_align(j->p, j->seqs + j->q_off[i], N, j->seqs + j->t_off[i], M, &j->res[i], qa, ta, NULL, NULL);
        free(tmp);
    }
    return NULL;
}

int orc_align_batch(const orc_params *p, const uint8_t *seqs, const uint64_t *q_off, const uint64_t *q_len,
                    const uint64_t *t_off, const uint64_t *t_len, size_t n_pairs, int n_threads,
                    orc_result *res, uint8_t *tb, const uint64_t *tb_off)
{
    if (n_threads < 1) n_threads = 1;
    if ((size_t)n_threads > n_pairs) n_threads = (int)(n_pairs ? n_pairs : 1);
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * n_threads);
    job_t *jobs = (job_t *)malloc(sizeof(job_t) * n_threads);
    size_t per = n_pairs / n_threads, extra = n_pairs % n_threads, lo = 0;
    for (int k = 0; k < n_threads; k++) {
        size_t cnt = per + ((size_t)k < extra ? 1 : 0);
        jobs[k] = (job_t){p, seqs, q_off, q_len, t_off, t_len, lo, lo + cnt, res, tb, tb_off};
        lo += cnt;
        if (n_threads == 1) worker(&jobs[k]);
        else pthread_create(&th[k], NULL, worker, &jobs[k]);
    }
    if (n_threads > 1) for (int k = 0; k < n_threads; k++) pthread_join(th[k], NULL);
    free(th); free(jobs);
    return 0;
}
