/* aligner_oracle.h -- CPU oracle for the DP hot path.  TEST INFRASTRUCTURE ONLY (see aligner_oracle.c). */
#ifndef ALIGNER_ORACLE_H
#define ALIGNER_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* aligner-core/src/enums.rs:9-15 discriminants */
enum { ORC_TOP = 0, ORC_LEFT = 1, ORC_DIAGONAL = 2, ORC_BEGINNING = 3 };
enum { ORC_CORE_GLOBAL = 0, ORC_CORE_LOCAL = 1, ORC_LEGACY_GLOBAL = 2, ORC_LEGACY_LOCAL = 3, ORC_PWM_LOCAL = 4 };
enum {
    ORC_OK = 0,
    ORC_ERR_UNNECESSARY_ARGUMENT = 1, /* Error::UnnecessaryArgument, lib.rs:51 */
    ORC_ERR_EMPTY_SEQUENCE = 2,       /* reference panics */
    ORC_ERR_CODE_OUT_OF_RANGE = 3,    /* reference panics (ndarray OOB) */
    ORC_ERR_NO_POSITIVE_CELL = 4,     /* reference panics (argmax on a border) */
    ORC_ERR_OOM = 6,
    ORC_ERR_MATRIX_SHAPE = 9         /* Error::MatrixShapeError, pwm/mod.rs:40-42 */
};

typedef struct {
    int32_t semantics;
    int32_t heuristics_present; /* Some(Heuristics) -> Err(UnnecessaryArgument) */
    double del, ext;
    const double *matrix;       /* matrix[[target_code, query_code]] */
    uint32_t rows, cols;
    int64_t row_stride;         /* in elements */
    uint8_t blank_code;         /* Protein::Blank / DNA::Blank = 98 */
} orc_params;

typedef struct {
    double f;                   /* Alignment.f: 0.0 for core global, H max for local */
    double score;               /* H[M][N] (global) or H max (local) */
    uint32_t end_y, end_x;      /* traceback start cell */
    uint32_t start_y, start_x;  /* cell where the traceback loop stopped */
    uint64_t coords[4];         /* ((c0,c1),(c2,c3)) of Alignment.coords */
    uint32_t aln_len;
    int32_t status;
} orc_result;

/* PWMAligner::perform_alignment (aligner-core/src/pwm/mod.rs:29-126): seq = the aligner's `query` (rows, length Q),
 * matrix = 4 x W position-weight matrix (columns 1..W).  numbered[i] = PWM column or 0 (capacity Q+W+2),
 * qal[i] = residue code or blank.  res->coords as Alignment.coords; no seed pair, no panic on an empty result. */
int orc_align_pwm(const orc_params *p, const uint8_t *seq, size_t Q, orc_result *res, uint32_t *numbered, uint8_t *qal,
                  double *H_out, uint8_t *D_out);

/* qa / ta: caller buffers of capacity M+N+2 each. H_out ((M+1)*(N+1) doubles) and D_out (bytes) optional. */
int orc_align(const orc_params *p, const uint8_t *q, size_t N, const uint8_t *t, size_t M,
              orc_result *res, uint8_t *qa, uint8_t *ta, double *H_out, uint8_t *D_out);

void orc_midline(const uint8_t *qa, const uint8_t *ta, size_t len, const double *S, int64_t row_stride,
                 uint8_t blank, uint8_t pos, uint8_t *out);
void orc_frequency_matrix(const uint8_t *qa, const uint8_t *ta, size_t len, uint8_t blank, uint32_t volume, double *out);

/* pair i: query = seqs[q_off[i] .. +q_len[i]), target likewise; tb (optional) receives qa at tb_off[i] and
 * ta at tb_off[i] + (M+N+2). */
int orc_align_batch(const orc_params *p, const uint8_t *seqs, const uint64_t *q_off, const uint64_t *q_len,
                    const uint64_t *t_off, const uint64_t *t_len, size_t n_pairs, int n_threads,
                    orc_result *res, uint8_t *tb, const uint64_t *tb_off);

#ifdef __cplusplus
}
#endif
#endif
