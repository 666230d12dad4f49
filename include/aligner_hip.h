/*
 * aligner_hip.h -- C ABI of the MI355X-native DP matrix-fill + traceback path.
 *
 * This is the drop-in boundary for ONE hot path of ikramanop/aligner: the pairwise-alignment DP fill and
 * traceback behind `AlignerTrait::perform_alignment` (aligner-core/src/lib.rs:27-40), implemented by
 * `SimpleGlobalAligner` / `SimpleLocalAligner` (aligner-core/src/simple/mod.rs:42-145, :168-264) and by the
 * legacy `SimpleAligner::{global,local}_alignment` (src/align/aligner_core.rs:96-183, :185-269).
 * The reference has no FFI of its own; a thin Rust shim implements `AlignerTrait` for
 * `Hip{Global,Local}Aligner<T>` on top of these entry points.  Plain pointers and sizes only; the library never
 * frees caller memory and never returns owned pointers other than the opaque handles below.
 *
 * Conventions (same as the reference): query = columns (x, length N), target = rows (y, length M); the substitution
 * lookup is matrix[[target_code, query_code]] (simple/mod.rs:85,198); residues are one byte holding the enum
 * discriminant (`Into<usize>`, enums.rs:98-102,149-153); direction codes are the `Direction` discriminants
 * Top=0, Left=1, Diagonal=2, Beginning=3 (enums.rs:9-15).
 */
#ifndef ALIGNER_HIP_H
#define ALIGNER_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ALN_ABI_VERSION 2

/* Which reference routine is reproduced bit-for-bit. */
enum aln_semantics {
    ALN_CORE_GLOBAL = 0,   /* SimpleGlobalAligner::perform_alignment   simple/mod.rs:42-145  */
    ALN_CORE_LOCAL = 1,    /* SimpleLocalAligner::perform_alignment    simple/mod.rs:168-264 */
    ALN_LEGACY_GLOBAL = 2, /* SimpleAligner::global_alignment          src/align/aligner_core.rs:96-183  */
    ALN_LEGACY_LOCAL = 3,  /* SimpleAligner::local_alignment           src/align/aligner_core.rs:185-269 */
    ALN_PWM_LOCAL = 4      /* PWMAligner::perform_alignment            aligner-core/src/pwm/mod.rs:29-126:
                            * `target` = the aligner's query (rows), matrix = 4 x W position-weight matrix whose column x scores
                            * PWM position x; `query` is ignored (N = W).  Aligned output: q_aln holds uint32_t PWM column
                            * numbers (0 = gap; capacity N+M+2 entries, 4-byte aligned), t_aln the residue codes / blank. */
};

/* Status codes.  1 mirrors `Err(Error::UnnecessaryArgument)` (lib.rs:51; simple/mod.rs:49-51,175-177).
 * 2..4 are conditions on which the reference PANICS; a drop-in shim maps them back to panic!(). */
enum aln_status {
    ALN_OK = 0,
    ALN_ERR_UNNECESSARY_ARGUMENT = 1,
    ALN_ERR_EMPTY_SEQUENCE = 2,      /* last().unwrap() on an empty Vec      simple/mod.rs:103-104        */
    ALN_ERR_CODE_OUT_OF_RANGE = 3,   /* ndarray index out of bounds          simple/mod.rs:85,198         */
    ALN_ERR_NO_POSITIVE_CELL = 4,    /* argmax on a border -> usize underflow simple/mod.rs:214-215       */
    ALN_ERR_DEVICE = 5,              /* HIP runtime / kernel failure (aln_last_error() has the text)      */
    ALN_ERR_OOM = 6,
    ALN_ERR_INVALID_ARGUMENT = 7,
    ALN_ERR_UNSUPPORTED = 8,
    ALN_ERR_MATRIX_SHAPE = 9         /* Err(Error::MatrixShapeError): PWM without exactly 4 rows, pwm/mod.rs:40-42 */
};

/* What the caller wants back (bitmask in aln_params.outputs). */
enum aln_outputs {
    ALN_OUT_SCORE = 1,       /* f / score / end cell */
    ALN_OUT_TRACEBACK = 2,   /* aligned code strings, start cell, aln_len */
    ALN_OUT_DIRECTIONS = 4,  /* (M+1)x(N+1) Direction bytes = AlignmentResult.direction_matrix (pair API only) */
    ALN_OUT_H_MATRIX = 8     /* (M+1)x(N+1) f64 = AlignmentResult.alignment_matrix (pair API only; generic kernels: 1.2 ms for a 1k x 1k pair against 0.3 ms without) */
};

/* Arguments of perform_alignment(del, ext, matrix, heuristics) + the output selection. */
typedef struct aln_params {
    int32_t semantics;          /* enum aln_semantics */
    int32_t heuristics_present; /* Some(Heuristics) -> ALN_ERR_UNNECESSARY_ARGUMENT for the core semantics */
    double del;                 /* gap opening ("deletions"); the single i32 gap cost for the legacy semantics */
    double ext;                 /* gap extension; ignored by the legacy semantics */
    const double *matrix;       /* host pointer, element [t][q] at matrix[t*row_stride + q] */
    uint32_t rows, cols;        /* matrix shape; codes >= shape -> ALN_ERR_CODE_OUT_OF_RANGE */
    int64_t row_stride;         /* in elements (ndarray stride of axis 0) */
    uint32_t outputs;           /* bitmask of enum aln_outputs; 0 = SCORE|TRACEBACK */
    uint8_t blank_code;         /* T::blank(): 98 for Protein and DNA (enums.rs:81,144) */
    uint8_t force_f64;          /* 1: run the f64 kernels even if the inputs are integral (testing) */
    uint8_t force_serial;       /* 1: run the strict reference-order kernel (one lane per pair; testing/fallback) */
    uint8_t force_generic;      /* 1: run the generic (non-profiled) integer kernels (testing) */
    uint32_t max_passes;        /* CORE_LOCAL with del != ext: cap on speculative fills before the serial kernel; 0 = 4 */
} aln_params;

/* Fixed-size per-pair summary (48 bytes); also the record gathered across GPUs. */
typedef struct aln_pair_result {
    double f;                   /* Alignment.f: 0.0 for CORE_GLOBAL (simple/mod.rs:139), H max for local (:247) */
    double score;               /* H[M][N] for the global semantics, H max for the local ones */
    uint32_t end_y, end_x;      /* cell the traceback starts from (1-based matrix coordinates) */
    uint32_t start_y, start_x;  /* cell where the traceback loop stopped */
    uint32_t aln_len;           /* length of both aligned strings (includes the reference's duplicated seed pair) */
    int32_t status;             /* enum aln_status for this pair */
    uint32_t passes;            /* bits 0-6: full fill passes (1 unless CORE_LOCAL with del != ext); bit 7: strict-order fallback;
                                   bits 8-15: localized repairs of strip 0; bits 16-19: 1 + checkpoint at which the last one re-converged;
                                   bits 20-23: why a repair escalated to a full pass (1 hazard beyond the last checkpoint, 2 strip 0's
                                   bottom row moved, 3 no re-convergence, 4 repair limit); diagnostics only */
    uint32_t flags;             /* bit0: integer kernels were used (also for a real-valued scheme whose numbers are all multiples of
                                   2^-k, k <= 8: it is filled as the integer scheme times 2^k and f / score are scaled back, exactly); bit1: the strip-pipelined single-pair route;
                                   bit2: generic kernels, one workgroup per pair (real-valued matrix / H output, <= 4 pairs per call) */
} aln_pair_result;

typedef struct aln_ctx aln_ctx;      /* one per process: one GPU or a list of GPUs; thread-safe */
typedef struct aln_batch aln_batch;  /* a batch of pairs staged in HBM */

/* ---- context.  The reference runs its aligners on the caller's threads with no shared state (ten std::threads in
 * statistics/mod.rs:255-286); here every thread of the process shares one context.  A context owns, per GPU, a pool of
 * slots (device buffers, a stream, pinned staging) that calls lease -- nothing is allocated per call once the pool is warm.
 * aln_create: one GPU.  aln_create_multi: the listed GPUs of this process (n_devices = 0: every visible one); single calls go
 * to the devices in turn, a batch call is cut into chunks that the devices take from a common queue, and every device writes
 * its chunks' summaries and strings straight into the caller's host buffers over its own PCIe link (the host array IS the
 * gather; the multi-process form gathers device-side with RCCL, aligner_amd/distributed.py).
 * Creation warms the context (~0.1 s per GPU in all): the code objects are loaded and one synthetic 1000 x 1000 pair goes through
 * aln_align_pair, so that the first real call finds pool slot 0, its buffers and every kernel of that route in place (a first
 * 1000 x 1000 pair: 0.4 ms instead of 26-60; aligner-cli aligns one pair per process).  ALN_NO_WARMUP=1 leaves it to the first call. ---- */
aln_ctx *aln_create(int device_id, int *status);
aln_ctx *aln_create_multi(int n_devices, const int *device_ids, int *status);
int aln_device_count(const aln_ctx *ctx);
void aln_destroy(aln_ctx *ctx);
const char *aln_last_error(void);        /* thread-local text of the last ALN_ERR_DEVICE / _OOM */
int aln_abi_version(void);
int aln_device_info(aln_ctx *ctx, int *compute_units, size_t *hbm_bytes, char *name, size_t name_cap);   /* first device */

/* ---- one pair, blocking (replaces one perform_alignment call).  q_aln / t_aln: capacity N+M+2 bytes each.
 * directions: optional (M+1)*(N+1) bytes; h_matrix: optional (M+1)*(N+1) doubles. ---- */
int aln_align_pair(aln_ctx *ctx, const aln_params *params, const uint8_t *query, size_t N, const uint8_t *target,
                   size_t M, aln_pair_result *out, uint8_t *q_aln, uint8_t *t_aln, uint8_t *directions,
                   double *h_matrix);

/* ---- batch driver: semantics == map of aln_align_pair over independent pairs (the reference's only batch site is
 * statistics/mod.rs:255-286).  Pair i: query = seqs[q_off[i] .. +q_len[i]), target = seqs[t_off[i] .. +t_len[i]).
 * All pointers are HOST memory (any malloc'ed / Vec memory; nothing has to be pinned).  The call is a pipeline: the batch is
 * cut into chunks of the caller's pair order and chunk i+1 is uploaded, chunk i filled and traced back, chunk i-1 downloaded
 * at the same time; residues are checked against the matrix shape on the device.
 * tb_buf (optional): pair i's aligned query at tb_off[i], aligned target at tb_off[i] + q_len[i] + t_len[i] + 2; bytes of a
 * string's capacity beyond aln_len are unspecified.  Any tb_off is accepted; the cumulative layout
 * tb_off[i+1] = tb_off[i] + 2 * (q_len[i] + t_len[i] + 2) is copied back without a per-pair scatter.
 * ALN_PWM_LOCAL: q_len[i] is ignored (N = matrix cols); tb_off[i] must be a multiple of 4; pair i's uint32 column numbers
 * start at tb_off[i], its residue string at tb_off[i] + 4 * (cols + t_len[i] + 2). ---- */
int aln_align_batch(aln_ctx *ctx, const aln_params *params, const uint8_t *seqs, const uint64_t *q_off,
                    const uint64_t *q_len, const uint64_t *t_off, const uint64_t *t_len, size_t n_pairs,
                    aln_pair_result *results, uint8_t *tb_buf, const uint64_t *tb_off);

/* The chunks aln_align_batch cuts these pairs into on a context of n_devices GPUs (host arithmetic only; no GPU needed):
 * returns their number and writes up to cap (first pair, pair count) entries.  Chunks are ranges of the caller's pair order of
 * 5e9 .. 1.6e10 cells; the devices take them from a common queue. */
size_t aln_plan_chunks(const aln_params *params, const uint64_t *q_len, const uint64_t *t_len, size_t n_pairs, int n_devices,
                       uint64_t *first, uint64_t *count, size_t cap);

/* ---- staged form of the batch driver (one device: the context's first): inputs resident in HBM, results left in HBM
 * until fetched.
 * create = validate + H2D + allocate; run = fill (+ exact re-fills) + traceback, asynchronous on `stream`
 * (a hipStream_t passed as void*, NULL = the context's own stream); fetch = D2H. ---- */
aln_batch *aln_batch_create(aln_ctx *ctx, const aln_params *params, const uint8_t *seqs, const uint64_t *q_off,
                            const uint64_t *q_len, const uint64_t *t_off, const uint64_t *t_len, size_t n_pairs,
                            int *status);
int aln_batch_run(aln_batch *b, void *stream);
int aln_batch_sync(aln_batch *b);
int aln_batch_fetch(aln_batch *b, aln_pair_result *results, uint8_t *tb_buf, const uint64_t *tb_off);
void aln_batch_destroy(aln_batch *b);
uint64_t aln_batch_cells(const aln_batch *b);                 /* sum of M_i * N_i over valid pairs */
size_t aln_batch_size(const aln_batch *b);
void *aln_batch_results_device(aln_batch *b);                 /* device pointer: aln_pair_result[n_pairs] (for RCCL) */
uint64_t aln_batch_direction_bytes(const aln_batch *b);       /* bytes of packed directions one run writes */
/* mean kernel time per run since aln_batch_enable_timing(b, 1), from HIP events recorded on the launch stream
 * around the fill kernel and the traceback kernel (last 256 runs; call after aln_batch_sync) */
int aln_batch_timing(aln_batch *b, double *fill_ms, double *traceback_ms, uint32_t *fill_launches);
void aln_batch_enable_timing(aln_batch *b, int on);

#ifdef __cplusplus
}
#endif
#endif /* ALIGNER_HIP_H */
