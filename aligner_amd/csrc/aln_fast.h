// aln_fast.h -- the fast integer fill path (included by aln_kernels.hip inside its anonymous namespace).
//
// Same recurrence as the generic path, reformulated so that ONE v_max3_i32 yields value AND direction:
//   carried state   T = 4*H + 2                              (H exact in the upper 30 bits)
//   Top  key = Ttop  - 4p       (tag 2)      Left key = Tleft - 4p - 1 (tag 1)      Diag key = Tdiag + (4s - 2) (tag 0)
//   key = max3(..)  ->  H' = key >> 2,  tag = key & 3, in the reference's tie order Top > Left > Diagonal (enums.rs:18-28)
//   T' = (key & ~3) | 2          Beginning (H' == 0, local, enums.rs:37) is tag 3; the legacy clamp at zero
//                                (aligner_core.rs:210) is max(key, 3).
// Substitution scores come from a per-strip query profile in LDS, P[c][row] = 4*S[t[row]][c] - 2 as int8, row-contiguous
// per code: one ds_read of R bytes per step feeds the lane's R cells (SDWA byte adds).  Not conflict-free: at a step the
// lanes read the rows of up to 24 different codes, 512 B apart (measured SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.39);
// LDS instructions are 1.4 % of the VALU count, so it does not bind.
// The query code flows down the lanes with the same DPP wave_shr:1 that carries the boundary cell.
// Local end cell: per row one packed register  (T' << 11) | f(step)  updated with one v_lshl_add per cell and one v_max3
// per two steps, folded every 2048 steps with the exact tie rule (first in row-major order: core; last in column-major
// order: legacy).
// Per cell (core local): v_cmp, v_cndmask (penalty), v_add, v_add3, v_add_sdwa, v_max3, v_and_or, v_max_u32 (tag 3),
// v_alignbit, v_lshl_add, half a v_max3 = 10.5 VALU instructions; core global / legacy global: 6.
// Two kernels use this file: the batch kernel (one wave per pair, strips in sequence, C++ step below) and the single-pair
// kernel (one wave per strip; its steady state is the generated asm of aln_single_unit.inc, see steady_run).
#pragma once

// ---- single instructions the compiler would otherwise re-associate into longer sequences
__device__ __forceinline__ int v_max3(int a, int b, int c)
{
    int d;
    asm("v_max3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ int v_tform(int key)               // (key & ~3) | 2
{
    int d;
    asm("v_and_or_b32 %0, %1, -4, 2" : "=v"(d) : "v"(key));
    return d;
}
__device__ __forceinline__ int v_add3_m1(int a, int b)         // a + b - 1
{
    int d;
    asm("v_add3_u32 %0, %1, %2, -1" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ int v_pack11(int t, int kterm)      // (t << 11) + kterm, kterm wave-uniform
{
    // as asm: left to itself the compiler folds the computation of kterm into every cell (v_lshlrev + v_bitop3 against
    // the step counter), one VALU instruction more per cell than this
    int d;
    asm("v_lshl_add_u32 %0, %1, 11, %2" : "=v"(d) : "v"(t), "s"(kterm));
    return d;
}
// The heart of a cell as ONE statement (hipcc pads every asm statement with a wait state, so one statement per cell,
// not one per instruction).  pw holds four int8 profile scores; B selects this row's byte:
//   c = diag + sext(pw.byte[B]);  key = max3(top + negp, left + negp - 1, c);  nt = (key & ~3) | 2
#define ALN_CELL_ASM(BYTE)                                                                                    \
    asm("v_add_u32_sdwa %0, %8, sext(%9) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:" BYTE "\n\t" \
        "v_add_u32 %1, %5, %7\n\t"                                                                            \
        "v_add3_u32 %2, %6, %7, -1\n\t"                                                                       \
        "v_max3_i32 %3, %1, %2, %0\n\t"                                                                       \
        "v_and_or_b32 %4, %3, -4, 2"                                                                          \
        : "=&v"(c), "=&v"(a), "=&v"(b), "=&v"(key), "=v"(nt)                                                  \
        : "v"(top), "v"(left), "v"(negp), "v"(diag), "v"(pw))
template <int B>
__device__ __forceinline__ void v_cell(int top, int left, int negp, int diag, uint32_t pw, int &key, int &nt)
{
    int a, b, c;
    if constexpr (B == 0) ALN_CELL_ASM("BYTE_0");
    else if constexpr (B == 1) ALN_CELL_ASM("BYTE_1");
    else if constexpr (B == 2) ALN_CELL_ASM("BYTE_2");
    else ALN_CELL_ASM("BYTE_3");
}
#undef ALN_CELL_ASM

// a lane's R profile bytes are read as one LDS word of RP = 1, 2, 4 or 8 bytes (R rounded up), RP-aligned
template <int RP> struct ProfWordP;
template <> struct ProfWordP<8> { using T = uint2; };
template <> struct ProfWordP<4> { using T = uint32_t; };
template <> struct ProfWordP<2> { using T = uint16_t; };
template <> struct ProfWordP<1> { using T = uint8_t; };
template <int R> struct ProfWord {
    static constexpr int RP = R > 4 ? 8 : R > 2 ? 4 : R;
    using T = typename ProfWordP<RP>::T;
};

// the 32-bit word of the profile read that holds row r's byte
template <int R>
__device__ __forceinline__ uint32_t prof_word(const typename ProfWord<R>::T &pw, int r)
{
    if constexpr (R > 4) return r < 4 ? pw.x : pw.y;
    else return (uint32_t)pw;
}
template <int R>
__device__ __forceinline__ int prof_byte(const typename ProfWord<R>::T &pw, int r)
{
    if constexpr (R > 4) return (int)(int8_t)(((r < 4 ? pw.x : pw.y) >> (8 * (r & 3))) & 0xff);
    else return (int)(int8_t)(((uint32_t)pw >> (8 * r)) & 0xff);
}

__device__ __forceinline__ int shr1_i(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, 0x138, 0xf, 0xf, false); }

// Inter-strip hand-off of the single-pair kernel: every boundary cell travels as one naturally aligned 4-byte granule, the
// T value itself -- T = 4H + 2 is never zero, and the buffer is zeroed before every pass, so "non-zero" means "produced".
// Written by ONE write-through (sc1) store and polled with sc1 loads: the data is the flag, no fence
// (cdna_hip_programming.md G16 "R2").
__device__ __forceinline__ void granule_store(uint32_t *p, int v)
{
    __hip_atomic_store(p, (uint32_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t granule_load(const uint32_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- single-pair kernel, steady state of the core-local fill: one asm statement per 16-column unit, generated by
// tools/gen_single_asm.py (see there for the schedule and the hazards it honours).
#include "aln_single_unit.inc"

// LDS hand-off ring between two waves of a workgroup: T values (never zero) by column & (ALN_RING - 1); a slot is zeroed
// by the consumer when it has taken it, and written by the producer only when it is zero.
#define ALN_RING 4096u

// Everything a strip needs that is uniform over the pair.  Passed BY VALUE so that it lives in (scalar) registers.
struct FastIn {
    int lane;
    uint32_t N, M;
    const uint8_t *q, *t;
    const int *S;             // LDS, [t][q]
    uint32_t cols;
    uint8_t *prof;            // LDS, this wave's profile
    int nd4, ne4;             // -4*del, -4*ext
    uint32_t *dirw;
    unsigned long long *brow_in, *brow_out;   // batch kernels: the row above this strip / this strip's bottom row, as 8-byte granules
                              // {T value, tag of the strip that wrote it} (see row_tag below) (one row used in place; for hazard
                              // pairs strip 0's bottom row keeps a row of its own: the repair compares against it)
    uint8_t *advice, *zrow;
    int *ckpt;
    bool hazard;
    bool adv_any;             // single-pair kernel: the advice may hold non-zero bytes (second and later passes)
    bool store_dirs;          // false: score-only (the packed directions are not written)
    bool wt_dirs;             // batch kernels: direction quads are stored write-through (a walk kernel on another XCD reads them while this kernel
                              // runs; with cooperative passes a re-fill's strips, run on other XCDs, rewrite the first pass's lines: a plain store would
                              // leave them dirty in two L2s, and which write-back wins is anybody's guess)
    bool pwm;                 // position-weight-matrix scoring (batch kernels only)
    const uint32_t *pwm_words;// per column: int8 scores 4*s - 2 of residues 0..3
    uint32_t ck_stop;         // single-pair kernel: the step at which pass 0 saves the strip's lane state (ck_mode 1) / the repair run
                              // stops and compares (ck_mode 2); 0 = none
    int ck_mode;              // 0 plain, 1 save checkpoints, 2 repair
    uint32_t last_flip;       // repair: last column whose input to this strip (row-1 advice, or the row above) differs from the checkpointed pass
    uint16_t *qo_pad;         // single-pair kernel: LDS, q[x] * 64R at index x + 63, zeros elsewhere (N + 192 entries)
    int *bring;               // single-pair kernel: LDS, 2 x 64 ints
    const uint32_t *gin;      // single-pair kernel: granule rows
    uint32_t *gout;
    uint32_t lds_scratch;           // single-pair kernel: LDS address of 512 scratch bytes (256-aligned), see steady_run
    uint32_t *ring_in, *ring_out;   // single-pair kernel: LDS rings (ALN_RING entries, aligned to their size) when the
                                    // strip above / below is a wave of this workgroup, else null (granule rows)
    uint32_t *abort_flag;
    // batch kernels: where a strip sits in the pair (skewed layout: 512 rows, aln_strip_bytes; a cooperative re-fill's uniform
    // layout: 64 R rows, aln_uniform_strip_bytes)
    uint32_t strip_rows, strip_q16;
    // batch kernels: tag of this pass's granules without the strip number (row_tag); every boundary cell, every bottom-row record
    // word and every end-cell candidate a strip hands on carries tag_base | strip, unique for (launch, owner's pass, strip): the
    // reader polls until it sees that tag, whoever ran the strip and whatever copies of older passes caches still hold
    uint32_t tag_base;
    // batch kernels: 0, or (log2 of the time slice in 10 ns ticks) << 8 | the wave's slot in its SIMD (fair_prio)
    uint32_t fair;
    uint32_t ck_last;         // batch kernels: the last checkpoint step (FillArgs::ck_last)
};

// The lane's running end-cell candidate (T form) + outcome flags; threaded through the strips by value.
struct FastOut {
    int bv;
    uint32_t by, bx;
    int corner;
    bool repaired, brow_bad, aborted;
    uint32_t ck_slot;         // repair pass: index of the checkpoint (step 16, 32, 64, ... 512 for R = 8) at which the lane state re-converged
    uint32_t c_out;           // repair pass: last column in which this strip's bottom row changed (0: it did not)
};

// a better-than-b for the local end cell, values in any monotone form
template <int SEM>
__device__ __forceinline__ bool better_i(int v, uint32_t y, uint32_t x, int bv, uint32_t by, uint32_t bx)
{
    if (v > bv) return true;
    if (v < bv) return false;
    if (SEM == ALN_CORE_LOCAL) return y < by || (y == by && x < bx);     // first in row-major order (simple/mod.rs:212)
    return x > bx || (x == bx && y > by);                                 // last in column-major order (aligner_core.rs:224)
}

__device__ __forceinline__ uint32_t uniform32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t uniform64(uint64_t v)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t uniform64(const void *p) { return uniform64((uint64_t)(uintptr_t)p); }

// PWM (position-weight-matrix scoring, batch kernels only) is a template parameter: a run-time test of it sat in the
// hot loop of every batch fill.
template <int SEM, int R, bool SINGLE, bool FIRST, bool LAST, bool PWM = false>
struct FastStrip {
    static constexpr int SPB = (int)aln_spb(R);
    static constexpr int RP = ProfWord<R>::RP;             // bytes between two lanes' profile words
    // steps per chunk of the end-cell tracker: the largest multiple of a quad's steps that 11 bits can number
    static constexpr uint32_t CHUNK = (2048u / (4u * SPB)) * (4u * SPB);
    static constexpr uint32_t STRIP_ROWS = SINGLE ? 64u * R : (uint32_t)ALN_STRIP_ROWS;
    static constexpr bool LOCAL = (SEM == ALN_CORE_LOCAL || SEM == ALN_LEGACY_LOCAL);
    using PW = typename ProfWord<R>::T;
    const FastIn in;
    const uint32_t strip;
    static constexpr bool last = LAST;
    const int lane;
    const uint32_t N;
    uint32_t lb, rb, yb;
    uint32_t chunk0;           // first step of the tracker's current chunk
    uint32_t cchg;             // repair: a column of this lane's whose bottom-row cell changed (0: none)
    bool zsel_on, brow_bad, aborted;
    int Tl[R], rbv[R];
    int hdiag, bottom, qoff, inchunk, qchunk, outq;
    uint32_t advchunk, dw;
    PW pw;
    const uint8_t *prow;       // this lane's column of the profile: prof + lane*R

    // single-pair kernel: query offsets and the incoming boundary row are staged in LDS
    const uint8_t *qo_lane;    // &qo_pad[63 - lane] (u16 entries: q[x] * 64R, zero padded on both sides)
    int *bring;                // 2 x 64 T values of the row above this strip
    int qv, top0v;
    uint32_t gpre;             // prefetched granule of the next 16-column group (lanes 0..15)
    // hand-written steady state (single-pair kernel, core local, R <= 2)
    static constexpr bool GLOBAL_ASM = (SEM == ALN_CORE_GLOBAL);   // six-instruction cells, border group instead of the constant 2
    static constexpr bool ASMPATH = SINGLE && (SEM == ALN_CORE_LOCAL || SEM == ALN_CORE_GLOBAL) && (R == 1 || R == 2) && !(FIRST && LAST);
    int twov;                  // 2; lane 0 of strip 0: a value no cell takes (row 1's penalty never follows the border)
    PW pw1;                    // steady format (see steady_enter)
    uint32_t qv2, qv3, gA, gB;
    bool insteady;
    bool ring_staged;          // LDS hand-off: the C++ step has already taken the next group out of the ring (it is in bring)

    __device__ __forceinline__ FastStrip(const FastIn &i, uint32_t s)
        : in(i), strip(s), lane(i.lane), N(i.N), cchg(0), brow_bad(false), aborted(false) {}

    // PWM scoring: the flowing register holds the column's four packed scores; one v_perm per four rows picks each
    // row's byte by its residue code -- the result has the layout of a profile read
    uint32_t psel_lo, psel_hi;
    __device__ __forceinline__ PW pwm_select(uint32_t w4) const
    {
        if constexpr (R > 4) return make_uint2(__builtin_amdgcn_perm(w4, w4, psel_lo), __builtin_amdgcn_perm(w4, w4, psel_hi));
        else return (PW)__builtin_amdgcn_perm(w4, w4, psel_lo);
    }

    // batch kernels: next 64 columns of the row above this strip (T form), one per lane
    // batch kernels: the next 64 columns of the row above this strip, one per lane.  The strip above may be running right now on
    // another wave (a cooperative pass): every granule carries the tag of the strip that wrote it, and the load is repeated until
    // all 64 carry the expected one (bounded; giving up poisons the pass, which the pair's owner then redoes alone).
    __device__ __forceinline__ int fetch_above(uint32_t xi)
    {
        const uint32_t want = in.tag_base | (strip - 1u);
        const unsigned long long *p = in.brow_in + xi + 1;
        unsigned long long g = 0;
        uint32_t spins = 0;
        for (;;) {
            if (xi < N) g = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all(xi >= N || (uint32_t)(g >> 32) == want)) break;
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1u << 17)) { aborted = true; break; }
        }
        return (xi < N) ? (int)(uint32_t)g : 2;
    }
    // batch kernels: lane 63's newest bottom cell enters a 64-deep lane shift register every step (DPP wave_shl:1, as in the
    // single-pair kernel), and every 64 steps the wave stores 64 consecutive columns of its bottom row with ONE coalesced store of
    // 8-byte granules (a store per step by lane 63 alone was 64 partial-line writes instead of four whole lines).  Before step k
    // lane l holds column k - 126 + l.  Repair run of strip 0: nothing is stored, the columns are compared with the row the
    // checkpointed pass wrote (a cell that comes out different ends the repair).
    __device__ __forceinline__ void flush_below(uint32_t k, uint32_t beyond)
    {
        const uint32_t x = k - 126u + (uint32_t)lane;
        if (x - 1u < N && x > beyond) {
            unsigned long long *p = in.brow_out + x;
            if (FIRST && SEM == ALN_CORE_LOCAL && in.ck_mode == 2) {
                if ((int)(uint32_t)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != outq) cchg = x;
            } else
                __hip_atomic_store(p, ((unsigned long long)(in.tag_base | strip) << 32) | (uint32_t)outq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // the columns the shift register still holds after `ksteps` steps (end of the strip, or the checkpoint at which a repair ends)
    __device__ __forceinline__ void flush_tail(uint32_t ksteps)
    {
        const uint32_t kb_last = ((ksteps - 1u) / 64u) * 64u;          // the last boundary flush_below ran at (from step 64 on)
        flush_below(ksteps, kb_last >= 64u ? kb_last - 63u : 0u);
    }

    // The SIMD's arbiter serves the two waves of highest priority -- oldest first among equals -- and the third only with what they
    // leave: three VALU-bound waves run at 1.6 / 0.9 / 0.4 GCUPS by age.  A chain of strips is as slow as its slowest wave.  With
    // `fair` set the three waves of a SIMD take turns at being the third: every 64 steps a wave looks at the 100 MHz clock and
    // steps down to priority 0 during its own slice ((slice + slot) mod 3 == 0), priority 1 otherwise.  A wave never has to run to
    // be promoted over the others (a starved wave would be late to do so -- rotating 0 / 1 / 2 left the youngest wave stale at
    // 0); the slice (in.fair >> 8, log2 of 10 ns ticks) is about what a stepped-down wave needs for its next 64 steps.
    __device__ __forceinline__ void fair_prio()
    {
        const uint32_t slice = (uint32_t)(wall_clock64() >> (in.fair >> 8));
        if ((slice + (in.fair & 255u)) % 3u == 0u) __builtin_amdgcn_s_setprio(0);
        else __builtin_amdgcn_s_setprio(1);
    }

    template <bool MASKED>
    __device__ __forceinline__ void step(const uint32_t k)
    {
        if ((k & 63u) == 0) {                                   // wave-uniform: refill the 64-column input chunks
            const uint32_t xi = k + (uint32_t)lane;             // 0-based column
            if (!SINGLE && in.fair) fair_prio();
            if (!SINGLE && !LAST && k >= 64u) flush_below(k, 0u);
            if (!FIRST && !SINGLE) inchunk = fetch_above(xi);
            if (SEM == ALN_CORE_LOCAL && FIRST && in.hazard) advchunk = (xi < N) ? in.advice[xi + 1] : 0u;
            if (!SINGLE) qchunk = (xi + 1 < N) ? (PWM ? (int)in.pwm_words[xi + 1] : (int)in.q[xi + 1] * (64 * RP)) : 0;
        }
        const int sel = (int)(k & 63u);
        int top0;
        if (FIRST) top0 = LOCAL ? 2 : ((k + 1 == N) ? 2 + (int)(N + 1) * in.nd4 : 2 + (int)(k + 1) * in.nd4);
        else if (SINGLE) top0 = top0v;                          // read from the LDS ring one step ago
        else top0 = __builtin_amdgcn_readlane(inchunk, sel);
        const int topIn = shr1_i(top0, bottom);                 // lane 0 <- row above the strip, lane l <- lane l-1
        if (SINGLE && !FIRST) top0v = bring[(k + 1) & 127u];    // next step's boundary cell (broadcast read)
        // cross-lane reads stay in wave-uniform control flow: inside a divergent branch the compiler may compute
        // their operand for the active lanes only
        const uint32_t adv = (SEM == ALN_CORE_LOCAL && FIRST) ? (uint32_t)__builtin_amdgcn_readlane((int)advchunk, sel) : 0u;
        const PW pwc = pw;                                      // profile bytes of THIS step (loaded one step ago)
        if constexpr (SINGLE) {
            pw = *reinterpret_cast<const PW *>(prow + qv);                           // step k+1: column k+1-lane
            qv = *reinterpret_cast<const uint16_t *>(qo_lane + 2 * (k + 2));         // step k+2
        } else {
            qoff = shr1_i(__builtin_amdgcn_readlane(qchunk, sel), qoff);             // next step's query code reaches every lane
            if constexpr (PWM) pw = pwm_select((uint32_t)qoff); else pw = *reinterpret_cast<const PW *>(prow + qoff);
        }
        // end-cell tie-break term of this step: earlier steps win (core) / later steps win (legacy)
        static_assert(CHUNK == 2048u, "the tracker's chunks start at multiples of 2048 steps");
        const uint32_t kc = k & 2047u;                                      // step within the tracker's chunk (scalar arithmetic: k is uniform)
        const int kterm = (SEM == ALN_CORE_LOCAL) ? (int)(2047u - kc) : (int)kc;
        const uint32_t xm1 = k - (uint32_t)lane;
        if (!MASKED || xm1 < N) {
            int top = topIn, diag = hdiag;
            // "cell above is Beginning" -> penalty del.  Row 1 (lane 0 of strip 0) sits under the border, T = 2 always:
            // its carried penalty comes from the bottom cell of the previous column instead (the advice), so lane 0
            // compares against 2 when the advice says "zero" and against a value no T takes when it does not -- one
            // v_writelane instead of a select chain
            int cmpv = 2;
            if (SEM == ALN_CORE_LOCAL && FIRST) {
                const int l0 = __builtin_amdgcn_readfirstlane(((k == 0) || (adv != 0)) ? 2 : 1);   // wave-uniform, in an SGPR
                asm("v_writelane_b32 %0, %1, 0" : "+v"(cmpv) : "s"(l0));
            }
            bool zr = (topIn == cmpv);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                int negp;
                if (SEM == ALN_CORE_LOCAL) negp = zr ? in.nd4 : in.ne4;
                else if (SEM == ALN_CORE_GLOBAL) negp = (r == 0 && FIRST && lane == 0 && k == 0) ? in.nd4 : in.ne4;
                else negp = in.nd4;
                int key, nt;
                if (SEM == ALN_LEGACY_LOCAL) {
                    const int c = diag + prof_byte<R>(pwc, r);
                    key = max(v_max3(top + negp, v_add3_m1(Tl[r], negp), c), 3);
                    nt = v_tform(key);
                } else {
                    const uint32_t w32 = prof_word<R>(pwc, r);
                    switch (r & 3) {                       // constant after unrolling: picks the SDWA byte select
                    case 0: v_cell<0>(top, Tl[r], negp, diag, w32, key, nt); break;
                    case 1: v_cell<1>(top, Tl[r], negp, diag, w32, key, nt); break;
                    case 2: v_cell<2>(top, Tl[r], negp, diag, w32, key, nt); break;
                    default: v_cell<3>(top, Tl[r], negp, diag, w32, key, nt); break;
                    }
                }
                uint32_t stored = (uint32_t)key;
                if (SEM == ALN_CORE_LOCAL) {
                    zr = (nt == 2);
                    // Beginning (H == 0) is tag 3: H == 0 <=> key in {0,1,2} <=> (unsigned)key < 4, so one unsigned max
                    // sets the tag without a mask (negative keys are huge as unsigned and pass through)
                    stored = max((uint32_t)key, 3u);
                }
                dw = __builtin_amdgcn_alignbit(stored, dw, 2);
                if (LOCAL) rbv[r] = max(rbv[r], v_pack11(nt, kterm));
                diag = Tl[r];
                Tl[r] = nt;
                top = nt;
            }
            hdiag = topIn;
            bottom = Tl[R - 1];
        }
        // bottom row to the strip below: lane 63's newest cell enters a 64-deep lane shift register (DPP wave_shl:1)
        if (!LAST) outq = __builtin_amdgcn_update_dpp(bottom, outq, 0x130, 0xf, 0xf, false);
    }

    // single-pair kernel: after step k lanes 48..63 hold the bottom-row cells of columns c-15..c, c = k - 63; one
    // 128-byte write-through store publishes them to the strip below
    __device__ __forceinline__ void publish(const uint32_t k)
    {
        const uint32_t c = k - 63u;
        const uint32_t col = c - 63u + (uint32_t)lane;              // wraps for lanes that hold nothing yet
        const bool mine = lane >= 48 && col < N;
        if (in.ring_out) {                                          // the consumer is a wave of this workgroup
            uint32_t *slot = in.ring_out + (col & (ALN_RING - 1u));
            uint32_t spins = 0;
            while (__any(mine && __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0)) {
                __builtin_amdgcn_s_sleep(1);                        // ring full: the consumer is a whole lap behind
                if (++spins > (1u << 22)) { aborted = true; break; }
            }
            if (mine) __hip_atomic_store(slot, (uint32_t)outq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return;
        }
        if (mine) granule_store(in.gout + col, outq);
    }

    // single-pair kernel: boundary cells arrive 16 columns at a time.  The granules of group j+1 are requested when
    // group j is staged (lanes 0..15, non-blocking), so in steady state staging never waits on HBM/L2; only a consumer
    // that has caught up with its producer spins (bounded) until the tags appear.
    __device__ __forceinline__ void stage_boundary16(const uint32_t j)
    {
        const uint32_t col = 16u * j + (uint32_t)lane;
        const bool need = lane < 16 && col < N;
        if (in.ring_in) {                                           // the producer is a wave of this workgroup
            uint32_t *slot = in.ring_in + (col & (ALN_RING - 1u));
            uint32_t g = 0, spins = 0;
            for (;;) {
                if (need) g = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (__all(!need || g != 0)) break;
                __builtin_amdgcn_s_sleep(1);
                ++spins;
                if (spins > (1u << 22) ||
                    ((spins & 1023u) == 0 && __hip_atomic_load(in.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                    if (lane == 0) __hip_atomic_store(in.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    aborted = true;
                    break;
                }
            }
            if (need) __hip_atomic_store(slot, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // consumed
            if (lane < 16) bring[col & 127u] = need ? (int)g : 2;
            return;
        }
        uint32_t g = gpre;
        uint32_t spins = 0;
        while (!__all(!need || g != 0)) {
            __builtin_amdgcn_s_sleep(1);
            if (need) g = granule_load(in.gin + col);
            ++spins;
            if (spins > (1u << 22) ||
                ((spins & 1023u) == 0 && __hip_atomic_load(in.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                // the producer never arrived: poison the run instead of hanging the GPU
                if (lane == 0) __hip_atomic_store(in.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                aborted = true;
                break;
            }
        }
        if (lane < 16) bring[col & 127u] = need ? (int)g : 2;
        const uint32_t ncol = col + 16u;
        gpre = (lane < 16 && ncol < N) ? granule_load(in.gin + ncol) : 0u;       // group j+1, consumed next time
    }

    // four blocks of SPB steps -> one 16-byte store per lane (1 KiB per wave, coalesced).  The block loop is a real
    // loop (not unrolled): unrolling 4*SPB steps makes the scheduler hoist every step's uniform values and spill.
    // the bottom-row zero flags (row-1 hazard) travel as the direction words of the lane that owns row M (tag 3 =
    // Beginning <=> H == 0): one dword per block instead of a select chain and a byte store per step
    __device__ __forceinline__ void store_zdw(uint32_t block)
    {
        if (SEM == ALN_CORE_LOCAL && zsel_on && (uint32_t)lane == lb) {
            if (SINGLE) reinterpret_cast<uint32_t *>(in.zrow)[block] = dw;
            else   // a granule like the boundary cells: read by the pair's owner, which may be another wave
                __hip_atomic_store(reinterpret_cast<unsigned long long *>(in.zrow) + block, ((unsigned long long)(in.tag_base | strip) << 32) | dw,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    // ---- steady state of the single-pair core-local kernel (ASMPATH): whole quads through the generated asm loop.
    // Lane state in "steady format": pw / pw1 = profile bytes of the next two steps, qv2 / qv3 = query offsets of the two
    // after them (the C++ step keeps pw of this step and qv of the next one only); gA / gB = boundary groups of the
    // next even / odd 16-column unit, in the lane order row_ror hands them to lane 0 (lane i of every row of 16 holds
    // column ku + (16 - i) % 16).
    __device__ __forceinline__ void steady_enter(const uint32_t k)
    {
        pw1 = *reinterpret_cast<const PW *>(prow + qv);
        qv2 = *reinterpret_cast<const uint16_t *>(qo_lane + 2 * (k + 2));
        qv3 = *reinterpret_cast<const uint16_t *>(qo_lane + 2 * (k + 3));
        gA = 0; gB = 0;
        if (!FIRST && !in.ring_in) {
            const uint32_t *src = in.gin + k + ((16u - (uint32_t)lane) & 15u);
            const uint32_t g0 = granule_load(src), g1 = granule_load(src + 16);
            if ((k >> 4) & 1u) { gB = g0; gA = g1; } else { gA = g0; gB = g1; }
        }
        insteady = true;
        ring_staged = true;
    }
    // back to the C++ step at step k (a multiple of 16): its LDS ring needs column group k / 16 and its prefetch register
    // the group after that
    __device__ __forceinline__ void steady_leave(const uint32_t k)
    {
        qv = *reinterpret_cast<const uint16_t *>(qo_lane + 2 * (k + 1));
        if (!FIRST) { gpre = 0; stage_boundary16(k >> 4); top0v = bring[k & 127u]; }
        if (FIRST && SEM == ALN_CORE_LOCAL && in.hazard) {       // the step reloads its advice chunk every 64 steps only
            const uint32_t xi = (k & ~63u) + (uint32_t)lane;
            advchunk = (xi < N) ? in.advice[xi + 1] : 0u;
        }
        insteady = false;
    }
    // quads [kb, kb_end) -- see tools/gen_single_asm.py for what the statement does and why it is one statement
    // MASKED: quads in which some lane's column does not exist (k - lane < 0 or >= N): the cell update runs under exec =
    // the lanes whose column exists; hand-over and LDS prefetch run for every lane
    template <bool MASKED>
    __device__ __forceinline__ void steady_run(const uint32_t kb, const uint32_t kb_end)
    {
        uint32_t ku = uniform32(kb * SPB);
        const uint32_t kend = uniform32(kb_end * SPB);
        uint32_t qop = (uint32_t)(uintptr_t)qo_lane + 2u * ku;
        const uint32_t prow32 = (uint32_t)(uintptr_t)prow;
        // core local: tracker term of the unit (a step adds 15 - i as an immediate).  Core global, strip 0: what the border
        // H[0][x] = -x del moves by per unit of 16 columns, and G = the border above the unit's columns (lane i: column
        // ku + (16 - i) % 16, i.e. x = that + 1); the corner column N never falls into this loop (see run()).
        int kt = (int)uniform32(GLOBAL_ASM ? (uint32_t)(16 * in.nd4) : 2032u - (ku & 2047u));
        int G = (GLOBAL_ASM && FIRST) ? 2 + (int)(ku + ((16u - (uint32_t)lane) & 15u) + 1u) * in.nd4 : 2;
        int X1, O1, np, ta, tb, c0, c1, k0, p0, u0, u1;
        uint64_t um;
        uint32_t P0 = (uint32_t)pw, P1 = (uint32_t)pw1, P2, P3, Q0, Q1, Q2 = qv2, Q3 = qv3, la, w0, w1, w2, w3, st, spin;
        uint32_t vsrc = 4u * (ku + ((16u - (uint32_t)lane) & 15u));
        uint32_t vpub = 4u * (ku + (uint32_t)lane - 111u);          // 4 * column (negative while lanes 48..63 hold nothing)
        uint32_t vdir = (kb >> 2) * 1024u + (uint32_t)lane * 16u;
        uint32_t vz = 4u * kb;
        const uint32_t vzero = 0;
        // LDS hand-off (mode 1): ring slot addresses of this lane's column of the current group / of its published column
        uint32_t amode = uniform32(in.ring_in ? 1u : 0u), pmode = uniform32(in.ring_out ? 1u : 0u);
        asm volatile("" : "+s"(amode), "+s"(pmode));          // keep them in SGPRs even when they fold to constants
        // The asm issues the same LDS operations whatever the mode (its lgkmcnt counts depend on it): with the global
        // hand-off the ring reads / writes go to scratch words and the step is 0, so the addresses stay put.
        const uint32_t vrmask = 4u * ALN_RING - 1u;
        const uint32_t vrbin = in.ring_in ? (uint32_t)(uintptr_t)in.ring_in : in.lds_scratch;
        const uint32_t vrbout = in.ring_out ? (uint32_t)(uintptr_t)in.ring_out : in.lds_scratch + 256u;
        uint32_t vrin = in.ring_in ? (vrbin | (vsrc & vrmask)) : vrbin;
        uint32_t vrout = in.ring_out ? (vrbout | (vpub & vrmask)) : vrbout + 4u * (uint32_t)lane;
        uint32_t astep = uniform32(in.ring_in ? 64u : 0u), pstep = uniform32(in.ring_out ? 64u : 0u);
        asm volatile("" : "+s"(astep), "+s"(pstep));
        // lanes whose column exists at step ku: lane <= ku and ku - lane < N
        uint64_t em = (ku >= 63u ? ~0ull : ((2ull << ku) - 1ull)) & (ku >= N ? (ku - N + 1u >= 64u ? 0ull : (~0ull << (ku - N + 1u))) : ~0ull), et;
        em = uniform64(em);
        int krem = (int)uniform32(N - 1u - ku);
        const uint32_t n4 = uniform32(4u * N);
        uint32_t gL = (in.ring_in && ring_staged) ? (uint32_t)bring[(ku + ((16u - (uint32_t)lane) & 15u)) & 127u] : 0u, chk;
        ring_staged = false;
        const uint64_t m48 = uniform64(0xffff000000000000ull), zmask = uniform64(zsel_on ? (1ull << lb) : 0ull);
        const uint32_t sdirs = uniform32(in.store_dirs ? 1u : 0u);
        uint8_t *dbase = reinterpret_cast<uint8_t *>(in.dirw) + (size_t)strip * (size_t)aln_uniform_strip_bytes(N, R);
        int dummyT = 2, dummyR = INT_MIN;                     // R = 1 has one row: its asm never touches T0 / r1
        int &t0ref = (R == 1) ? dummyT : Tl[0];
        int &r1ref = (R == 1) ? dummyR : rbv[R - 1];
        // wave-uniform addresses, pinned to SGPR pairs (the "s" constraint alone does not move a value out of VGPRs)
        const uint64_t sgin = uniform64(in.gin), sgout = uniform64(in.gout), sdbase = uniform64(dbase),
                       szbase = uniform64(in.zrow), sabort = uniform64(in.abort_flag);
#define ALN_STEADY_OPERANDS                                                                                    \
        : [T0] "+&v"(t0ref), [TL] "+&v"(Tl[R - 1]), [r0] "+&v"(rbv[0]), [r1] "+&v"(r1ref), [X0] "+&v"(hdiag),       \
          [O0] "+&v"(outq), [P0] "+&v"(P0), [P1] "+&v"(P1), [Q2] "+&v"(Q2), [Q3] "+&v"(Q3), [gA] "+&v"(gA), [gB] "+&v"(gB),  \
          [G] "+&v"(G), [qop] "+&v"(qop), [vsrc] "+&v"(vsrc), [vpub] "+&v"(vpub), [vdir] "+&v"(vdir), [vz] "+&v"(vz),       \
          [vrin] "+&v"(vrin), [vrout] "+&v"(vrout), [gL] "+&v"(gL), [chk] "=&v"(chk),                                     \
          [kt] "+&s"(kt), [ku] "+&s"(ku), [st] "=&s"(st), [spin] "=&s"(spin), [em] "+&s"(em), [et] "=&s"(et), [krem] "+&s"(krem), [w0] "=&v"(w0), [w1] "=&v"(w1), \
          [w2] "=&v"(w2), [w3] "=&v"(w3), [P2] "=&v"(P2), [P3] "=&v"(P3), [Q0] "=&v"(Q0), [Q1] "=&v"(Q1),             \
          [X1] "=&v"(X1), [O1] "=&v"(O1), [np] "=&v"(np), [ta] "=&v"(ta), [tb] "=&v"(tb), [c0] "=&v"(c0),             \
          [c1] "=&v"(c1), [k0] "=&v"(k0), [p0] "=&v"(p0), [la] "=&v"(la), [u0] "=&v"(u0), [u1] "=&v"(u1), [um] "=&s"(um)                                              \
        : [two] "v"(twov), [prow] "v"(prow32), [ne] "v"(in.ne4), [nd] "v"(in.nd4), [vzero] "v"(vzero),               \
          [kend] "s"(kend), [gin] "s"(sgin), [gout] "s"(sgout), [dbase] "s"(sdbase), [zbase] "s"(szbase),         \
          [abortp] "s"(sabort), [m48] "s"(m48), [zmask] "s"(zmask), [sdirs] "s"(sdirs), [amode] "s"(amode), [n4] "s"(n4), [astep] "s"(astep), [pstep] "s"(pstep), \
          [pmode] "s"(pmode), [vrmask] "v"(vrmask), [vrbin] "v"(vrbin), [vrbout] "v"(vrbout)                          \
        : "vcc", "scc", "memory"
#define ALN_PICK(L1F, L1M, L1L, L2F, L2M, L2L)                                                                  \
        if constexpr (R == 1) {                                                                                \
            if constexpr (FIRST) asm volatile(L1F ALN_STEADY_OPERANDS);                                        \
            else if constexpr (LAST) asm volatile(L1L ALN_STEADY_OPERANDS);                                    \
            else asm volatile(L1M ALN_STEADY_OPERANDS);                                                        \
        } else {                                                                                               \
            if constexpr (FIRST) asm volatile(L2F ALN_STEADY_OPERANDS);                                        \
            else if constexpr (LAST) asm volatile(L2L ALN_STEADY_OPERANDS);                                    \
            else asm volatile(L2M ALN_STEADY_OPERANDS);                                                        \
        }
        if constexpr (MASKED) {
            static_assert(!FIRST, "strip 0 runs its ends in C++ (its first column takes del, and nothing waits on its start)");
            // (the FIRST slots of the masked picks are never instantiated)
            if constexpr (GLOBAL_ASM) { ALN_PICK(ALN_GMASKED_ASM_R1_MID, ALN_GMASKED_ASM_R1_MID, ALN_GMASKED_ASM_R1_LAST, ALN_GMASKED_ASM_R2_MID, ALN_GMASKED_ASM_R2_MID, ALN_GMASKED_ASM_R2_LAST) }
            else { ALN_PICK(ALN_MASKED_ASM_R1_MID, ALN_MASKED_ASM_R1_MID, ALN_MASKED_ASM_R1_LAST, ALN_MASKED_ASM_R2_MID, ALN_MASKED_ASM_R2_MID, ALN_MASKED_ASM_R2_LAST) }
        } else if constexpr (GLOBAL_ASM) {
            ALN_PICK(ALN_GSTEADY_ASM_R1_FIRST, ALN_GSTEADY_ASM_R1_MID, ALN_GSTEADY_ASM_R1_LAST, ALN_GSTEADY_ASM_R2_FIRST, ALN_GSTEADY_ASM_R2_MID, ALN_GSTEADY_ASM_R2_LAST)
        } else {
            ALN_PICK(ALN_STEADY_ASM_R1_FIRST, ALN_STEADY_ASM_R1_MID, ALN_STEADY_ASM_R1_LAST, ALN_STEADY_ASM_R2_FIRST, ALN_STEADY_ASM_R2_MID, ALN_STEADY_ASM_R2_LAST)
        }
#undef ALN_PICK
#undef ALN_STEADY_OPERANDS
        pw = (PW)P0; pw1 = (PW)P1; qv2 = Q2; qv3 = Q3;
        bottom = Tl[R - 1];
        if (st != 0) {                                        // the producer never arrived: poison the run
            if (lane == 0) __hip_atomic_store(in.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            aborted = true;
        }
    }
    // strip 0, second and later passes: first quad in [kb, kb_end) one of whose columns carries a non-zero advice byte
    // (kb_end if none).  Lane j scans the 16 bytes of columns base + 16 j ..: 1024 columns per sweep.
    __device__ __forceinline__ uint32_t advice_free_until(const uint32_t kb, const uint32_t kb_end)
    {
        const uint32_t k_end = kb_end * SPB;
        for (uint32_t k = kb * SPB; k < k_end; k += 1024u) {
            uint32_t any = 0;
            const uint32_t x0 = k + 16u * (uint32_t)lane;     // step k of lane 0 is column x = k + 1
#pragma unroll
            for (int i = 0; i < 16; ++i) any |= (x0 + i < N) ? (uint32_t)in.advice[x0 + i + 1] : 0u;
            const uint64_t m = __ballot(any != 0);
            if (m) {
                const uint32_t kq = ((k + 16u * (uint32_t)__builtin_ctzll(m)) / (4u * SPB)) * 4u;   // block index of that quad
                return kq < kb_end ? kq : kb_end;
            }
        }
        return kb_end;
    }

    template <bool MASKED>
    __device__ __forceinline__ void quad(uint4 *dirq, const uint32_t kb)
    {
        uint4 v = make_uint4(0, 0, 0, 0);
#pragma unroll 1
        for (uint32_t j = 0; j < 4; ++j) {
            // wave-uniform by construction; say so (in strip 0 of a hazard pair, and wherever the strip's shape comes out of
            // a vector load -- the batch kernels -- the compiler otherwise carries the step counter in a VGPR and pays for
            // it in every step: four VALU instructions per step, measured as 5 % of the C5 fill)
            const uint32_t k0 = (FIRST || !SINGLE) ? (uint32_t)__builtin_amdgcn_readfirstlane((int)((kb + j) * SPB)) : (kb + j) * SPB;
            if (SINGLE && !FIRST && ((k0 + SPB) & 15u) == 0) stage_boundary16((k0 + SPB) >> 4);   // one block ahead
#pragma unroll
            for (int kk = 0; kk < SPB; ++kk) step<MASKED>(k0 + kk);
            if (SINGLE && !LAST && k0 + SPB >= 64u && ((k0 + SPB) & 15u) == 0) publish(k0 + SPB - 1);
            store_zdw(kb + j);
            if (j == 0) v.x = dw;
            else if (j == 1) v.y = dw;
            else if (j == 2) v.z = dw;
            else v.w = dw;
        }
        if (in.store_dirs) {
            uint4 *p = dirq + (size_t)(kb >> 2) * 64;
            if (!SINGLE && in.wt_dirs) {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 vv = {v.x, v.y, v.z, v.w};
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(vv) : "memory");
            } else *p = v;
        }
    }

    // folds the packed per-row candidates of the 2048-step chunk that starts at step `base` into the lane candidate
    __device__ __forceinline__ void fold(FastOut &o, uint32_t base)
    {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int v = rbv[r];
            const uint32_t y = yb + 1 + r;
            if (v != INT_MIN && y <= in.M) {
                const int t = v >> 11;
                const uint32_t kk = (uint32_t)v & 2047u;
                const uint32_t k = base + ((SEM == ALN_CORE_LOCAL) ? 2047u - kk : kk);
                const uint32_t x = k - (uint32_t)lane + 1;
                if (o.bx == 0 || better_i<SEM>(t, y, x, o.bv, o.by, o.bx)) { o.bv = t; o.by = y; o.bx = x; }
            }
            rbv[r] = INT_MIN;
        }
    }

    // the same for the tracker registers a checkpoint saved (the candidates of the prefix the checkpointed pass computed)
    __device__ __forceinline__ void fold_saved(FastOut &o, uint32_t slot)
    {
        const int *base = in.ckpt + slot * (18 * 64) + lane;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int v = base[(2 * r + 1) * 64];
            const uint32_t y = yb + 1 + r;
            if (v != INT_MIN && y <= in.M) {
                const int t = v >> 11;
                const uint32_t kk = (uint32_t)v & 2047u;
                const uint32_t k = (SEM == ALN_CORE_LOCAL) ? 2047u - kk : kk;
                const uint32_t x = k - (uint32_t)lane + 1;
                if (o.bx == 0 || better_i<SEM>(t, y, x, o.bv, o.by, o.bx)) { o.bv = t; o.by = y; o.bx = x; }
            }
        }
    }

    // Lane state at a block boundary (direction word flushed, input chunks about to be reloaded): everything the
    // rest of the strip depends on besides the inputs.  save = store it; !save = "is the DP state identical to the stored
    // one".  The end-cell tracker's registers need not re-converge: `tracker` says whether they differ from the stored
    // ones, and `stale` whether a stored candidate that the repaired prefix no longer produces is this lane's running
    // winner (then the prefix's contribution cannot be taken back out of `o`, and the caller re-fills the pair).
    __device__ __forceinline__ bool checkpoint(uint32_t slot, bool save, const FastOut &o, bool &tracker, bool &stale)
    {
        int *base = in.ckpt + slot * (18 * 64) + lane;
        bool same = true;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (save) { base[(2 * r) * 64] = Tl[r]; base[(2 * r + 1) * 64] = rbv[r]; }
            else {
                same = same && base[(2 * r) * 64] == Tl[r];
                const int old = base[(2 * r + 1) * 64];
                if (old != rbv[r]) {
                    tracker = true;
                    const uint32_t y = yb + 1 + r;
                    if (old != INT_MIN && y <= in.M) {
                        const uint32_t kk = (uint32_t)old & 2047u;
                        const uint32_t k = (SEM == ALN_CORE_LOCAL) ? 2047u - kk : kk;      // checkpoints lie in the tracker's first chunk
                        if ((old >> 11) == o.bv && y == o.by && k - (uint32_t)lane + 1 == o.bx) stale = true;
                    }
                }
            }
        }
        if (save) { base[16 * 64] = hdiag; base[17 * 64] = bottom; }
        else same = same && base[16 * 64] == hdiag && base[17 * 64] == bottom;
        return same;
    }

    __device__ __forceinline__ FastOut run(FastOut o)
    {
        const uint32_t M = in.M;
        const uint32_t y0 = strip * (SINGLE ? STRIP_ROWS : in.strip_rows);
        const uint32_t rows = min(M - y0, (uint32_t)(64 * R));
        const uint32_t L = (rows + R - 1) / R;
        const uint32_t nsteps = (SINGLE && !LAST) ? N + 63 : N + L - 1;
        yb = y0 + (uint32_t)lane * R;
        lb = (rows - 1) / R; rb = (rows - 1) % R;
        zsel_on = (SEM == ALN_CORE_LOCAL) && LAST && in.hazard;
        prow = in.prof + lane * RP;

        // ---- query profile of this strip's rows: P[c][row] = 4*S[t[row]][c] - 2  (int8), row-contiguous per code
        int tc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t y = yb + 1 + r;
            tc[r] = (y <= M) ? (int)in.t[y - 1] * (int)in.cols : 0;
            // left border H[y][0] in T form (simple/mod.rs:64-70)
            Tl[r] = LOCAL ? 2 : (y == M ? 2 + (int)(M + 1) * in.nd4 : 2 + (int)y * in.nd4);
            rbv[r] = INT_MIN;
        }
        psel_lo = 0; psel_hi = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {                       // PWM: byte selectors = the rows' residue codes (0..3)
            const uint32_t y = yb + 1 + r;
            const uint32_t code = (y <= M) ? (uint32_t)in.t[y - 1] & 3u : 0u;
            if (r < 4) psel_lo |= code << (8 * r); else psel_hi |= code << (8 * (r - 4));
        }
        for (uint32_t c = 0; c < (PWM ? 0u : in.cols); ++c) {
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t bte = (uint32_t)(4 * in.S[tc[r] + c] - 2) & 0xffu;
                if (r < 4) lo |= bte << (8 * r); else hi |= bte << (8 * (r - 4));
            }
            uint8_t *dst = in.prof + c * (64 * RP) + lane * RP;
            if constexpr (RP == 8) *reinterpret_cast<uint2 *>(dst) = make_uint2(lo, hi);
            else if constexpr (RP == 4) *reinterpret_cast<uint32_t *>(dst) = lo;
            else if constexpr (RP == 2) *reinterpret_cast<uint16_t *>(dst) = (uint16_t)lo;
            else *dst = (uint8_t)lo;
        }
        hdiag = LOCAL || yb == 0 ? 2 : 2 + (int)yb * in.nd4;            // H[yb][0]; yb < M always for valid lanes
        bottom = Tl[R - 1];
        inchunk = 2; qchunk = 0; advchunk = 0; dw = 0; outq = 0; qv = 0; top0v = 2;
        twov = (FIRST && lane == 0) ? 1 : 2;                 // T is always 2 (mod 4)
        pw1 = PW{}; qv2 = 0; qv3 = 0; gA = 0; gB = 0; insteady = false; ring_staged = false;
        if constexpr (SINGLE) {
            qo_lane = reinterpret_cast<const uint8_t *>(in.qo_pad + 63 - lane);
            bring = in.bring;
            gpre = (!FIRST && !in.ring_in && lane < 16 && (uint32_t)lane < N) ? granule_load(in.gin + lane) : 0u;
            if (!FIRST) { stage_boundary16(0); top0v = bring[0]; }
            qoff = *reinterpret_cast<const uint16_t *>(qo_lane);                      // step 0: column -lane
            qv = *reinterpret_cast<const uint16_t *>(qo_lane + 2);                    // step 1
        } else {
            qoff = (lane == 0) ? (PWM ? (int)in.pwm_words[0] : (int)in.q[0] * (64 * RP)) : 0;
        }
        if (!SINGLE && PWM) pw = pwm_select((uint32_t)qoff);
        else pw = *reinterpret_cast<const PW *>(prow + qoff);

        // directions: four blocks per lane per 16-byte store (aln_device.h); all segment ends are whole quads
        uint4 *dirq = reinterpret_cast<uint4 *>(in.dirw) +
                      (size_t)strip * (SINGLE ? (size_t)(aln_uniform_strip_bytes(N, R) / 16) : (size_t)in.strip_q16) + lane;
        const uint32_t nkb = aln_strip_blocks(nsteps, SPB);
        // ramp-up (some lanes not started) | steady state (every lane active, no exec masking) | ramp-down
        // (block counts: a lane's 64th step is in block 63 / SPB; segment ends are whole quads)
        const uint32_t kb_steady0 = min(nkb, ((uint32_t)((63 + SPB) / SPB) + 3u) & ~3u);
        uint32_t kb_steady1 = max(kb_steady0, min(nkb, (N / (4 * SPB)) * 4u));
        // core global, strip 0: step N - 1 reads the overwritten corner H[0][N] = -(N + 1) del (simple/mod.rs:62), which the
        // border group of the asm loop does not know: that quad stays with the C++ step
        if (ASMPATH && GLOBAL_ASM && FIRST && kb_steady1 * SPB >= N && kb_steady1 >= kb_steady0 + 4u) kb_steady1 -= 4u;
        // Segment ends: the 2048-step chunks of the end-cell tracker and, for strip 0 of a hazard pair, the
        // checkpoint steps 16 .. 1024.
        // (single-pair kernel: ONE checkpoint, at in.ck_stop -- saved by pass 0, the end of the repair run)
        const bool cks = SINGLE && SEM == ALN_CORE_LOCAL && in.ck_mode != 0 && in.ck_stop != 0;
        const bool ckmode = cks || (FIRST && !SINGLE && SEM == ALN_CORE_LOCAL && in.ck_mode != 0);
        // checkpoints sit on quad boundaries: the first at max(16, one quad of 4 * SPB steps), then doubling up to ALN_CK_LAST
        uint32_t next_ck = cks ? in.ck_stop : ckmode ? ((ALN_CK_FIRST + 4u * SPB - 1u) / (4u * SPB)) * (4u * SPB) : 0xffffffffu, slot = 0, chunk_base = 0;
        chunk0 = 0;
        uint32_t kb = 0;
        while (kb < nkb) {
            uint32_t seg_end = min(nkb, (chunk_base + CHUNK) / SPB);
            if (next_ck != 0xffffffffu) seg_end = min(seg_end, next_ck / SPB);
            const uint32_t e0 = min(kb_steady0, seg_end), e1 = min(kb_steady1, seg_end);
            if constexpr (ASMPATH && !FIRST) {
                // a strip with a strip above it never leaves the asm: masked quads at both ends (ramp-up, and the tail =
                // what is left of N after the last full quad + ramp-down), plain quads in between
                if (!aborted) {
                    if (!insteady) steady_enter(kb * SPB);
                    if (kb < e0) { steady_run<true>(kb, e0); kb = e0; }
                    if (kb < e1 && !aborted) { steady_run<false>(kb, e1); kb = e1; }
                    if (kb < seg_end && !aborted) { steady_run<true>(kb, seg_end); kb = seg_end; }
                }
                if (aborted) { o.aborted = true; return o; }
            }
            for (; kb < e0; kb += 4) quad<true>(dirq, kb);
            if constexpr (ASMPATH) {
                while (kb < e1 && !aborted) {
                    uint32_t kb_to = e1;
                    if (FIRST && in.adv_any) {
                        kb_to = advice_free_until(kb, e1);
                        if (kb_to == kb) {                       // this quad has advice: the C++ step handles it
                            if (insteady) steady_leave(kb * SPB);
                            quad<false>(dirq, kb);
                            kb += 4;
                            continue;
                        }
                    }
                    if (!insteady) steady_enter(kb * SPB);
                    steady_run<false>(kb, kb_to);
                    kb = kb_to;
                }
                if (aborted) { o.aborted = true; return o; }
                if (FIRST && insteady && kb >= kb_steady1) steady_leave(kb * SPB);   // strip 0 runs its tail in C++
            }
            for (; kb < e1; kb += 4) quad<false>(dirq, kb);
            for (; kb < seg_end; kb += 4) quad<true>(dirq, kb);
            if (cks && kb < nkb && kb * SPB == next_ck) {
                bool tracker = false, stale = false;
                if (in.ck_mode == 1) { checkpoint(0, true, o, tracker, stale); next_ck = 0xffffffffu; }
                else {
                    // the repair run ends here: has every lane rejoined the state pass 0 had at this step?  (The tracker
                    // registers hold the re-run prefix's candidates; the kernel folds them and pass 0's saved ones.)
                    o.repaired = __all(checkpoint(0, false, o, tracker, stale));
                    o.aborted = o.aborted || aborted;
                    return o;
                }
            } else if (ckmode && kb < nkb && kb * SPB == next_ck) {
                bool tracker = false, stale = false;
                if (in.ck_mode == 1) checkpoint(slot, true, o, tracker, stale);
                else if (__all(checkpoint(slot, false, o, tracker, stale)) && in.last_flip <= next_ck) {
                    // every lane's DP state is exactly what the checkpointed pass had here and no input differs from here
                    // on: the rest of this strip is unchanged.  The end-cell candidates of the repaired prefix replace the
                    // checkpointed pass's -- possible unless one of those is a lane's running winner.
                    if (!__any(stale)) {
                        if (__any(tracker)) fold(o, 0);
                        o.repaired = true;
                        o.ck_slot = slot;
                        if (!LAST) flush_tail(kb * SPB);       // the columns still in the shift register are compared too
                        o.c_out = __any(cchg != 0u) ? 1u : 0u;
                    }
                    return o;
                }
                ++slot;
                // (a repair whose last flipped bit lies within the first 512 columns gives up at step 512 as it always did: on C5 the
                // pairs that have not re-converged by then do not at 1024 either, and the longer try delayed their re-fill --
                // 6.75 -> 7.0 ms on the 8-way shard; the checkpoint at 1024 is for flips beyond column 512)
                const uint32_t lim = (in.ck_mode == 2 && in.last_flip <= 512u) ? 512u : in.ck_last;
                next_ck = next_ck < lim ? next_ck * 2u : 0xffffffffu;
                // a repair that has run out of checkpoints cannot succeed any more: stop here (the caller re-fills the pair)
                if (in.ck_mode == 2 && next_ck == 0xffffffffu) return o;
            }
            if (kb * SPB == chunk_base + CHUNK) {         // every semantics advances the chunk; only the local ones track an end cell
                if (LOCAL) fold(o, chunk_base);
                chunk_base += CHUNK;
                chunk0 = uniform32(chunk_base);              // (an SGPR: the tracker term of every cell is built from it)
            }
        }
        if (SINGLE && !LAST && !(ASMPATH && !FIRST)) publish(nkb * SPB - 1);     // the last (up to 15) columns (the asm publishes after every unit)
        if (!SINGLE && !LAST && !(ckmode && in.ck_mode == 2)) flush_tail(nkb * SPB);
        o.brow_bad = o.brow_bad || brow_bad;
        o.aborted = o.aborted || aborted;
        if (ckmode && in.ck_mode == 2) return o;         // ran out of checkpoints: the caller escalates to a full pass
        if (LOCAL) fold(o, chunk_base);
        if (last) {
            int hb = Tl[0];
#pragma unroll
            for (int r = 1; r < R; ++r) if ((uint32_t)r == rb) hb = Tl[r];
            o.corner = __builtin_amdgcn_readlane(hb, (int)lb);
        }
        return o;
    }
};

// Batch kernels, skewed layout: every strip but the last has 512 rows (R = 8), the last one picks R by its row count.  A
// cooperative re-fill (core local, no PWM scoring) instead cuts the pair into uniform strips of 64 R rows, R = 1, 2 or 4.
// Strip 0 and the strips below it are separate functions: a call site instantiates only the strips it can reach.
#define ALN_STRIP_CASE(RR, FIRSTV, LASTV) case RR: { FastStrip<SEM, RR, false, FIRSTV, LASTV, PWM> f(in, s); return f.run(o); }
template <int SEM, bool PWM, bool FIRSTV>
__device__ __forceinline__ FastOut fast_strip_of(const FastIn &in, FastOut o, uint32_t s, bool last, int R)
{
    if (!last) {
        if constexpr (SEM == ALN_CORE_LOCAL && !PWM) {
            switch (R) {
            ALN_STRIP_CASE(1, FIRSTV, false) ALN_STRIP_CASE(2, FIRSTV, false) ALN_STRIP_CASE(4, FIRSTV, false)
            default: break;
            }
        }
        FastStrip<SEM, ALN_FULL_R, false, FIRSTV, false, PWM> f(in, s);
        return f.run(o);
    }
    switch (R) {
    ALN_STRIP_CASE(1, FIRSTV, true) ALN_STRIP_CASE(2, FIRSTV, true) ALN_STRIP_CASE(3, FIRSTV, true) ALN_STRIP_CASE(4, FIRSTV, true)
    ALN_STRIP_CASE(5, FIRSTV, true) ALN_STRIP_CASE(6, FIRSTV, true) ALN_STRIP_CASE(7, FIRSTV, true)
    default: { FastStrip<SEM, 8, false, FIRSTV, true, PWM> f(in, s); return f.run(o); }
    }
}
#undef ALN_STRIP_CASE
template <int SEM, bool PWM>
__device__ __forceinline__ FastOut fast_strip_first(const FastIn &in, FastOut o, bool last, int R) { return fast_strip_of<SEM, PWM, true>(in, o, 0, last, R); }
template <int SEM, bool PWM>
__device__ __forceinline__ FastOut fast_strip_next(const FastIn &in, FastOut o, uint32_t s, bool last, int R) { return fast_strip_of<SEM, PWM, false>(in, o, s, last, R); }

// butterfly reduction of the per-lane end-cell candidates with the exact tie rule
template <int SEM>
__device__ __forceinline__ void reduce_best(FastOut &o)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const int ov = __shfl_xor(o.bv, m);
        const uint32_t oy = (uint32_t)__shfl_xor((int)o.by, m), ox = (uint32_t)__shfl_xor((int)o.bx, m);
        if (ox != 0 && (o.bx == 0 || better_i<SEM>(ov, oy, ox, o.bv, o.by, o.bx))) { o.bv = ov; o.by = oy; o.bx = ox; }
    }
}
