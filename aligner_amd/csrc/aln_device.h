// aln_device.h -- structures shared by the host driver and the gfx950 kernels.
//
// HBM data layout of a staged batch (all offsets in bytes unless noted):
//   seqs    : u8 residue codes of every query / target, packed back to back
//   descs   : PairDesc[n]           one record per pair
//   order   : u32[n]                pair indices sorted by M*N descending (LPT order for the work queue)
//   dirs    : packed 2-bit directions, one region per pair (PairDesc.dir_off), "skewed strip" layout:
//               strip s = rows s*512+1 .. (s+1)*512 of H, handled by one wave; lane l owns R consecutive rows
//               (R = 8 for every full strip; the last strip picks the smallest R in {1,2,4,8} covering its rows);
//               lane l computes column x at wave step k = (x-1) + l (anti-diagonal skew);
//               one u32 per lane holds R rows x SPB = 16/R consecutive steps (a "block"); a lane keeps FOUR blocks
//               in registers and stores them as one 16-byte quad: word index of step k =
//               ((kb >> 2) * 64 + lane) * 4 + (kb & 3), kb = k / SPB (aln_dir_word_index).  A wave therefore stores
//               1 KiB contiguous per 4*SPB steps (global_store_dwordx4, fully coalesced, 0.25 B per cell) and one
//               128-byte line holds 8 lanes x 4 blocks, i.e. a 64-row x 8..64-step tile -- the traceback walk stays
//               inside a line for several steps.  Cells are shifted in from the top (v_alignbit), so the cell of
//               (row r, step k) sits at bits 30 - 2*((e - k)*R + (R-1-r)), e = last step of that block in which the
//               lane is active (aln_dir_bitpos).  The 2-bit value is the priority tag of the max: 0 Diagonal, 1 Left, 2 Top,
//               3 Beginning (aln_tag_to_dir maps it to the reference's Direction discriminant).
//             serial-order fallback (layout 1): plain row-major, row stride (N+4)/4 bytes, 4 cells per byte.
//   results : aln_pair_result[n]
//   tb      : aligned code strings, pair i at tb_off: query string then (M+N+2 bytes later) target string -- cumulative
//             2 * (N + M + 2) per pair in the caller's pair order, i.e. exactly the layout aln_align_batch documents for tb_buf,
//             so a chunk's strings go back to the caller with ONE copy when the caller uses that layout
//   tags    : the walk's 2-bit tag string (one byte per step), pair i at tag_off, N + M + 2 bytes
//   scratch : per wave: strip boundary row (score type, N+66 entries), advice bytes, bottom-row zero bytes
#pragma once
#include <stdint.h>

#include "../../include/aligner_hip.h"

#ifndef ALN_FULL_R
#define ALN_FULL_R 8             // rows per lane in a full strip of the batch kernels
#endif
#define ALN_STRIP_ROWS (64 * ALN_FULL_R)   // rows per full strip
#define ALN_LAYOUT_SKEW 0u      // 512-row strips, R = 8 except the last strip (batch kernels)
#define ALN_LAYOUT_ROWMAJOR 1u  // serial-order fallback
#define ALN_LAYOUT_UNIFORM 2u   // every strip has 64*R rows, R in bits 8..15 (single-pair kernel; walked by the aln_tb_single_* kernels)
#define ALN_LAYOUT_UBATCH 3u    // the same uniform-R strips written by the batch kernel (a cooperative re-fill, see CoopRec): walked by
                                // the batch traceback like the skewed layout

struct PairDesc {
    uint64_t q_off, t_off;   // into seqs
    uint32_t N, M;           // query / target length
    uint64_t dir_off;        // into dirs, multiple of 256
    uint64_t tb_off;         // into tb
    uint64_t tag_off;        // into tags
    uint64_t h_off;          // element offset into the optional H matrix buffer
    int32_t status;          // pre-validation status; != 0 -> kernels skip the pair
    uint32_t layout;         // written by the fill kernel
};
// pre-validation status "nothing to align, and the reference returns Ok": PWMAligner with an empty sequence (pwm/mod.rs:52-108:
// the loops do not run, argmax is (0, 0), the traceback stops at once, f = 0).  The kernels skip the pair and report ALN_OK.
#define ALN_PRE_EMPTY_OK (-1)

struct FillArgs {
    const uint8_t *seqs;
    PairDesc *descs;
    const uint32_t *order;
    uint32_t n_pairs;
    uint32_t *counter;        // zeroed before every launch: [0] work-queue head (generic kernels), [1] tail of the completion queue, [2] the walk
                              // kernel's head, [4..5] the fast kernels' two-ended queue word (next_pair2)
    uint8_t *dirs;
    aln_pair_result *results;
    uint8_t *scratch;         // per-wave scratch base
    uint64_t scratch_stride;  // bytes per wave
    uint32_t max_len;         // max over pairs of max(N, M): sizes the scratch arrays
    uint32_t *doneq;          // optional (overlapped traceback): completion queue, pair + 1 per entry in the order the pairs finish;
                              // zeroed with the counter; its tail is counter[1]
    uint64_t max_cells;       // max over the queue's pairs of N * M (the first pair of the LPT order): scales the wave priorities
    const void *matrix;       // device copy, contiguous rows x cols, int32 or double
    uint32_t prof_stride;     // fast kernels: bytes of one wave's LDS query profile (cols * 512)
    uint32_t rows, cols;
    double del, ext;
    int32_t semantics;
    uint32_t max_passes;
    uint32_t force_serial;
    uint32_t store_dirs;      // 0: score-only run (no direction stores, no traceback)
    uint32_t pwm;             // 1: position-weight-matrix scoring: S[t[y-1]][x-1], the query codes are the column indices
    const uint32_t *pwm_words;// fast path: per column the four int8 scores 4*s - 2 of residues 0..3, packed
    uint32_t no_repair;       // 1: disable the localized strip-0 repair (testing: full re-fills only)
    uint32_t cascade_rows;    // fast path: boundary rows (8-byte granules) in each wave's scratch.  A strip never writes the row it reads,
                              // so rows alternate: 2 = non-hazard semantics; 3 (ALN_CASCADE_ROWS) = hazard pairs, whose strip 0 keeps
                              // its bottom row to itself (the localized repair compares against it) while strips 1.. alternate in the other two
    uint32_t zrow_bytes;      // bytes of the bottom-row record in each wave's scratch (bytes per column, or one direction word per block)
    void *hmat;               // optional: H dump, score type, (M+1)x(N+1) row-major per pair
    uint8_t blank;
    // cooperative passes (fast kernels, see CoopRec): control words, one claim word and one record per fill wave; null = off
    uint32_t *coop;
    uint32_t n_descs;                // descriptors behind `descs` (a claimed strip names its pair by index)
    uint32_t coop_waves;             // fill waves = claim words = records (the claim words are padded to a multiple of 64)
    uint32_t back_waves;             // fast kernels: 1 = the last third of the grid (the youngest wave of every SIMD of a full grid) takes its
                                     // pairs from the back of the queue, the shortest first (next_pair2)
    uint32_t salt;                   // tag salt of this launch (aln_coop_tag)
    uint32_t coop_tail;              // first passes are opened for the pairs from this queue position on (the last ones taken; earlier
                                     // pairs finish while every wave still has pairs of its own to take)
    uint32_t coop_linger;            // 1: waves that find the queue dry stay and take strips until every pair has finished
    uint32_t coop_debug;             // testing (ALN_COOP_DEBUG): bit 0 first passes are not opened; bit 1 no hints are posted (the owner
                                     // claims every strip of an open pass itself); bit 2 re-fills keep the skewed layout
    uint32_t duo_qo;                 // aln_fill_duo_kernel: u16 entries of ONE staged query in a wave's LDS (longest query + 72, rounded up to 8)
    uint32_t ck_last;                // fast kernels: the last checkpoint step of strip 0 of a hazard pair (512 or ALN_CK_LAST = 1024)
    uint32_t claim;                  // fast kernels without cooperative passes: queue positions a wave takes per atomic (1..4): batches of
                                     // many equal short pairs keep their waves in step, and 3000 waves at one counter within a microsecond
                                     // wait for each other (C3: 14 of the 47 us a wave spends per pair lay between two pairs)
    uint32_t fair;                   // fast kernels: 0, or log2 of the time slice (10 ns ticks) in which the waves of a SIMD take turns at stepping down (FastStrip::fair_prio)
    uint32_t f64_old;                // generic f64 kernels: 1 = run_strip's all-options loop instead of the lean f64 strip (ALN_F64_OLD, testing)
};

// ---- cooperative passes of the fast batch kernel
// One wave per pair leaves the tail of a small batch to whoever got the last large pair -- or a pair whose row-1 advice did not
// survive and has to be filled a second time (0.8 % of C5; the 8-way shard ended 1.2x after its ideal time because of them).  A
// pass over a pair is a chain of strips; strip s + 1 only needs the bottom row of strip s, 64 columns at a time, so the strips of
// ONE pass can run on different waves as a pipeline.  The wave that owns the pair opens its record (CoopRec, one per fill wave),
// runs strip 0 itself and then claims further strips like everybody else.
//   claim word   one per fill wave, in a dense array: {sequence number, kind (0 closed, 1 lazy, 2 urgent), strips, strips left}.
//                A claim is a compare-and-swap that takes `left` down by one (contended by the few waves that found this record,
//                and the sequence number makes a stale value fail); strips are claimed in order, and a claimed strip is being run
//                by a resident wave, so the wait of strip s for strip s - 1's columns always ends.
//   finding work two counters of unclaimed strips, urgent (re-fills: the pair is late already) and lazy (first passes).  A wave
//                between two pairs reads the urgent one -- one load of a line its L2 holds -- and looks for the record only when
//                it is not zero: first in a short log of the last urgent opens, then by scanning the claim words, 64 per load.
//                A wave that finds the queue dry does the same for both kinds.  No queue of hints: a queue's head is one word
//                every idle wave fights for (measured: 3000 waves popping 20 000 entries by compare-and-swap took 0.6 s).
//   hand-over    bottom rows, progress words, candidates and the record's fields are stored write-through (agent-scope atomics)
//                and read the same way (cdna_hip_programming.md, Guideline 16); the direction quads of such a kernel are
//                stored write-through as well (FastIn::wt_dirs).
// Every wait is bounded and raises the record's abort word; the owner then redoes the pass alone.
#define ALN_COOP_MAX_NS 64u
// Everything one wave hands to another is either inside a word that only atomic read-modify-writes touch (the claim word) or an
// 8-byte GRANULE {value, tag}, stored and loaded whole (agent-scope atomics = write-through / past L1): the reader polls until the
// tag is the one it expects, so neither the order in which stores become visible nor a copy of an older pass in some cache can be
// taken for the data (cdna_hip_programming.md, Guideline 16, form R2).  Tag of strip s of a pass: aln_coop_tag(salt, seq) | s --
// salt: a per-launch number of the slot that owns the scratch rows (10 bits; the host clears the rows when it wraps), seq: the
// owner wave's count of multi-strip passes in this launch (12 bits; the wave clears its rows when it wraps).  Bit 31 set and
// bit 30 clear: no T value (|T| < 2^28) looks like a tag, whatever else a row was used for.
struct CoopRec {
    unsigned long long pairg;                      // {pair, tag | 127}
    unsigned long long cand[ALN_COOP_MAX_NS][5];   // per strip {., tag | strip}: end-cell candidate bv, by, bx (T form), corner, status (1: gave up)
};
__host__ __device__ inline uint32_t aln_coop_tag(uint32_t salt, uint32_t seq) { return 0x80000000u | ((salt & 0x3ffu) << 19) | ((seq & 0xfffu) << 7); }
// claim word: bits 0..6 strips left, 7..13 strips, 14..15 kind (0: only the owner claims), 16 strip 0's bottom row keeps a row of
// its own, 17..19 rows per lane of uniform strips (0: the skewed layout), 20..31 seq
#define ALN_COOP_KIND_LAZY 1u
#define ALN_COOP_KIND_URGENT 2u
__host__ __device__ inline uint32_t aln_coop_word(uint32_t seq, uint32_t R, uint32_t own, uint32_t kind, uint32_t ns, uint32_t left)
{
    return ((seq & 0xfffu) << 20) | (R << 17) | (own << 16) | (kind << 14) | (ns << 7) | left;
}
// FillArgs::coop: control words, then the claim words (one per fill wave, padded to a multiple of 64), then the records
#define ALN_COOP_UOPEN 0         // unclaimed strips of urgent passes
#define ALN_COOP_LOPEN 1         // unclaimed strips of lazy passes
#define ALN_COOP_ULOGW 2         // urgent opens so far = write index of the log
#define ALN_COOP_FINISHED 3      // pairs whose summary is written (the four words above are what an idle wave polls: one 16-byte load)
#define ALN_COOP_HELPED 5        // diagnostics: strips run by a wave that does not own the pair
#define ALN_COOP_ABORTS 6
#define ALN_COOP_SCANS 7         // diagnostics: scans of the claim words
#define ALN_COOP_ULOG 32         // .. 47: the waves (+ 1) that opened the last 16 urgent passes
#define ALN_COOP_CTL_WORDS 256u   // (128 ..: diagnostics)

// one large pair, one wave per strip, strips pipelined through granule rows in HBM/L2
struct SingleArgs {
    const uint8_t *seqs;
    PairDesc *descs;
    uint32_t pair;
    uint8_t *dirs;
    aln_pair_result *results;
    uint32_t *granules;       // ns rows of gstride 4-byte granules, zeroed before every pass
    uint64_t gstride;
    uint8_t *advice, *zrow;   // N + 66 bytes each
    int32_t *cand;            // per strip: {bv (L form), by, bx, corner (L form)}
    uint32_t *ctrl;           // [0] abort, [1 + p] "pass p is needed", [15] "serial fallback needed"
    const void *matrix;
    uint32_t rows, cols;
    double del, ext;
    int32_t semantics;
    uint32_t R, ns, pass, max_passes;
    uint32_t hazard;
    uint32_t store_dirs;
    uint32_t test_drop;       // fault injection (env ALN_TEST_DROP_STRIP = s + 1): strip s never runs -- the run must end
                              // poisoned (ALN_ERR_DEVICE) within the polls' bounds, not hang
    // Localized repair of the row-1 hazard (see aln_single_repair_finalize_kernel): when pass 0's advice turns out wrong in the
    // leading columns only, the first rep_S strips re-run their leading columns -- strip s up to step rep_K + 64 (rep_S - 1 - s),
    // where pass 0 saved its lane state -- instead of the whole pipeline running a second time.
    int *ckpt;                // lane state of strips < rep_S at their stop steps (18 x 64 ints per strip), saved by pass 0
    uint32_t *rgranules;      // scratch granule rows of the repair run (rep_S rows of gstride), zeroed by the kernel that arms it
    int32_t *rcand;           // per repaired strip {new prefix candidate (bv, by, bx), old prefix candidate (bv, by, bx), converged, 0}
    uint32_t rep_S, rep_K;    // strips re-run (0: this pair is never repaired), stop step of the last of them
    uint32_t mode;            // 0 = a full pass (`pass`, gated by ctrl[1 + pass]); 1 = the repair run (gated by ctrl[8])
};

// strip 0 of a hazard pair checkpoints its lane state at steps max(16, one quad), then doubling, up to 1024 (r02: 512 -- enough for
// BLOSUM62 11 / 2, whose bottom-row zeros sit next to the left border; under costlier gaps -- BLOSUM62 x 2 with 46 / 9, the integer
// form of the dyadic scheme x 0.5 with 11.5 / 2.25 -- 41 % of C5's pairs flipped an advice bit beyond column 512 and were re-filled)
#define ALN_CK_SLOTS 7
#define ALN_CK_LAST 1024u
#define ALN_CASCADE_ROWS 3u
#define ALN_CK_FIRST 16u

struct TraceArgs {
    const uint8_t *seqs;
    const PairDesc *descs;
    uint32_t n_pairs;
    const uint8_t *dirs;
    aln_pair_result *results;
    uint8_t *tb;
    uint8_t *tags;
    int32_t semantics;
    uint8_t blank;
    uint8_t pwm;
    // overlapped traceback (aln_traceback_overlap_kernel runs beside the fill kernel; aln_traceback_kernel then sweeps up)
    uint32_t *walked;         // per pair: the epoch of the run in which the overlap kernel walked it (null: no overlap)
    uint32_t epoch;
    const uint32_t *doneq;    // the fill kernel's completion queue (pair + 1, zero = not yet)
    uint32_t *head;           // next entry of the completion queue to hand to a walk wave
    uint32_t n_order;         // entries the queue will hold (the fill queue's pairs)
    uint64_t wait_ticks;      // give up on a chunk after this many 100 MHz ticks (the sweep walks what is left)
};

// One pair of the generic kernels (f64, or int32 without the fast path's conditions), filled by ONE workgroup: wave s owns strip s
// (64 R rows), the strips are pipelined through LDS rings.  Directions in the uniform-R layout of the single-pair route.
struct WgArgs {
    const uint8_t *seqs;
    PairDesc *descs;
    uint32_t pair;
    uint8_t *dirs;
    aln_pair_result *results;
    const void *matrix;       // device copy, contiguous rows x cols, int32 or double
    uint32_t rows, cols;
    double del, ext;
    int32_t semantics;
    uint32_t R, ns;           // rows per lane (1 or 2), strips = waves of the workgroup (<= 16)
    uint32_t max_passes;
    uint32_t store_dirs;
    uint8_t *scratch;         // (max(N, M) + 66) scores: row 1 as the pass computed it (adopt_advice_checked); then the column of the
                              // strict-order routine, should the passes not converge
    void *hmat;               // optional: H dump, score type, (M+1)x(N+1) row-major per pair (desc.h_off elements in)
    uint32_t pwm;             // 1: position-weight-matrix scoring: S[t[y-1]][x-1], the column index instead of a query residue
};
#define ALN_WG_RING 256u      // entries of a hand-off ring (a strip's bottom row, by column & 255)
__host__ __device__ inline uint32_t aln_wg_lds_bytes(uint32_t rows, uint32_t cols, uint32_t sc_size, uint32_t ns, uint32_t N)
{
    return ((rows * cols * sc_size + 15u) & ~15u) + ns * ALN_WG_RING * sc_size + 3u * ((N + 66u + 15u) & ~15u) + 64u * 4u + ns * 32u +
           (N <= 2048u ? (((N + 66u) * sc_size + 15u) & ~15u) : 0u);      // row 1 of the pass: in LDS for short queries, else in the scratch
}

// Parallel traceback of one large pair (uniform-R layout): per strip and entry column an "exit map", then a short
// serial chain through the maps, then one walker per strip that writes its segment of the tag string.
struct TraceSingleArgs {
    const uint8_t *seqs;
    const PairDesc *descs;
    uint32_t pair;
    const uint8_t *dirs;
    aln_pair_result *results;
    uint8_t *tb;
    uint8_t *tags;
    int32_t semantics;
    uint32_t R, ns;
    uint4 *map;               // ns x (N + 1) entries {exit cx, exit cy, steps, stopped}
    uint4 *seg;               // per strip {entry cy, entry cx, tag-string offset, valid}
    uint32_t pwm;             // PWMAligner: no duplicated seed pair (pwm/mod.rs:76-108)
};

__host__ __device__ constexpr inline uint32_t aln_spb(uint32_t R) { return R >= 5 ? 2u : R >= 3 ? 4u : 16u / R; }
// stored tag -> Direction discriminant (enums.rs:9-15: Top=0 Left=1 Diagonal=2 Beginning=3)
__host__ __device__ inline int aln_tag_to_dir(int t) { return t == 3 ? 3 : 2 - t; }
__host__ __device__ inline int aln_dir_to_tag(int d) { return d == 3 ? 3 : 2 - d; }
// bit position of cell (row r of the lane, wave step k) inside its packed word; lane l is active for steps l .. l+N-1
__host__ __device__ inline uint32_t aln_dir_bitpos(uint32_t k, uint32_t r, uint32_t lane, uint32_t N, int R)
{
    const uint32_t spb = aln_spb((uint32_t)R);
    const uint32_t bend = (k / spb) * spb + spb - 1, lend = lane + N - 1;
    const uint32_t e = bend < lend ? bend : lend;
    return 30u - 2u * ((e - k) * (uint32_t)R + ((uint32_t)R - 1u - r));
}
// rows handled per lane in the strip that starts `rem` rows before the end of the target: the smallest R with 64 R >= rem.
// (A strip costs R cells of issue per step whatever its number of busy lanes; with R in {1,2,4,8} only, the last strips of C5
// were rounded up by 8 % of the batch's work; every R in 1..8 leaves 3 %.)  Steps per direction word: aln_spb(R), the
// largest power of two <= 16 / R (16, 8, 4, 4, 2, 2, 2, 2): the step arithmetic of fill and walk stays shifts and masks, and
// the words of R = 3, 5, 6, 7 simply leave their low bits unused (12, 10, 12, 14 of 16 cell slots filled).
__host__ __device__ inline int aln_pick_r(uint32_t rem)
{
    const int r = (int)((rem + 63u) / 64u);
    return r > ALN_FULL_R ? ALN_FULL_R : (r < 1 ? 1 : r);
}
__host__ __device__ inline uint32_t aln_num_strips(uint32_t M) { return (M + ALN_STRIP_ROWS - 1) / ALN_STRIP_ROWS; }
// blocks (of SPB steps) a strip of `nsteps` steps stores, padded to whole quads
__host__ __device__ inline uint32_t aln_strip_blocks(uint32_t nsteps, uint32_t spb) { return (((nsteps + spb - 1) / spb) + 3u) & ~3u; }
// word index of wave step k of lane `lane` inside its strip region
__host__ __device__ inline uint64_t aln_dir_word_index(uint32_t k, uint32_t lane, uint32_t spb)
{
    const uint32_t kb = k / spb;
    return ((uint64_t)(kb >> 2) * 64u + lane) * 4u + (kb & 3u);
}
// bytes of one full (R = 8) strip region: steps N+63, 2 steps per block, 256 B per block
__host__ __device__ inline uint64_t aln_strip_bytes(uint32_t N) { return (uint64_t)aln_strip_blocks(N + 63, aln_spb(ALN_FULL_R)) * 256u; }
// bytes of one strip of the uniform-R layout (single-pair kernel)
__host__ __device__ inline uint64_t aln_uniform_strip_bytes(uint32_t N, uint32_t R) { return (uint64_t)aln_strip_blocks(N + 63, aln_spb(R)) * 256u; }
__host__ __device__ inline uint64_t aln_rowmajor_bytes(uint32_t N, uint32_t M)
{
    return (uint64_t)(M + 1) * ((N + 4) / 4);
}
// rows per lane of a cooperative re-fill's uniform strips: at most eight strips up to 2048 rows; 0 = keep the skewed layout
__host__ __device__ inline uint32_t aln_coop_uniform_r(uint32_t M)
{
    if (M <= 64u || M > 2048u) return 0u;
    return M <= 512u ? 1u : M <= 1024u ? 2u : 4u;
}
__host__ __device__ inline uint64_t aln_dir_bytes(uint32_t N, uint32_t M)
{
    uint64_t a = (uint64_t)aln_num_strips(M) * aln_strip_bytes(N);
    uint64_t b = aln_rowmajor_bytes(N, M);
    uint64_t m = a > b ? a : b;
    const uint32_t R = aln_coop_uniform_r(M);
    if (R) {
        const uint64_t c = (uint64_t)((M + 64u * R - 1u) / (64u * R)) * aln_uniform_strip_bytes(N, R);
        m = c > m ? c : m;
    }
    return (m + 255) & ~(uint64_t)255;
}
