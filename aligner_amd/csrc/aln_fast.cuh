// aln_fast.cuh -- the fast integer fill path (included by aln_kernels.hip inside its anonymous namespace).
//
// Same recurrence as the generic path, reformulated so that ONE v_max3_i32 yields value AND direction:
//   carried state   T = 4*H + 2                              (H exact in the upper 30 bits)
//   Top  key = Ttop  - 4p       (tag 2)      Left key = Tleft - 4p - 1 (tag 1)      Diag key = Tdiag + (4s - 2) (tag 0)
//   key = max3(..)  ->  H' = key >> 2,  tag = key & 3, in the reference's tie order Top > Left > Diagonal (enums.rs:18-28)
//   T' = (key & ~3) | 2          Beginning (H' == 0, local, enums.rs:37) is tag 3; the legacy clamp at zero
//                                (aligner_core.rs:210) is max(key, 3).
// Substitution scores come from a per-strip query profile in LDS, P[c][row] = 4*S[t[row]][c] - 2 as int8, row-contiguous
// per code: one ds_read of R bytes per step feeds the lane's R cells (SDWA byte adds), conflict-free by construction.
// The query code flows down the lanes with the same DPP wave_shr:1 that carries the boundary cell.
// Local end cell: per row one packed register  (T' << 11) | f(step)  updated with ONE v_max per cell, folded every 2048
// steps with the exact tie rule (first in row-major order: core; last in column-major order: legacy).
// Per cell (core local): v_cmp, v_cndmask (penalty), v_add, v_add3, v_add_sdwa, v_max3, v_and_or, v_cndmask (tag 3),
// v_alignbit, v_lshl_add, v_max = 11 VALU ops; core global / legacy global: 6.
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
    return (int)(((uint32_t)t << 11) + (uint32_t)kterm);      // the compiler emits v_lshl_add_u32
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

template <int R> struct ProfWord;
template <> struct ProfWord<8> { using T = uint2; };
template <> struct ProfWord<4> { using T = uint32_t; };
template <> struct ProfWord<2> { using T = uint16_t; };
template <> struct ProfWord<1> { using T = uint8_t; };

// the 32-bit word of the profile read that holds row r's byte
template <int R>
__device__ __forceinline__ uint32_t prof_word(const typename ProfWord<R>::T &pw, int r)
{
    if constexpr (R == 8) return r < 4 ? pw.x : pw.y;
    else return (uint32_t)pw;
}
template <int R>
__device__ __forceinline__ int prof_byte(const typename ProfWord<R>::T &pw, int r)
{
    if constexpr (R == 8) return (int)(int8_t)(((r < 4 ? pw.x : pw.y) >> (8 * (r & 3))) & 0xff);
    else return (int)(int8_t)(((uint32_t)pw >> (8 * r)) & 0xff);
}

__device__ __forceinline__ int shr1_i(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, 0x138, 0xf, 0xf, false); }

// Inter-strip hand-off of the single-pair kernel: every boundary cell travels as one naturally aligned 8-byte granule
// {tag = 1, value = T} written by ONE write-through (sc1) store and polled with sc1 loads -- the data is the flag, no
// fence (cdna_hip_programming.md G16 "R2"); the buffer is zeroed before every launch so tag 0 = not yet produced.
__device__ __forceinline__ void granule_store(uint64_t *p, int v)
{
    __hip_atomic_store(p, (1ull << 32) | (uint32_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t granule_load(const uint64_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}


// ---- single-pair kernel, steady state of the core-local fill: ONE asm statement per step.  A lone wave issues one
// instruction per ~4-5 cycles whatever its kind (VALU, SALU, LDS, s_nop), so the step is written by hand to the minimum
// instruction count, with every gfx950 hazard slot (VALU writes VCC -> VALU reads it: 2 wait states; VALU write -> DPP
// read: 2) filled by useful work:
//   top-in   lanes 0..3 <- the boundary group G rotated by (k & 15) (row_ror; only lane 0 matters), then lanes 1..63 <-
//            lane-1's bottom cell (wave_shr:1)
//   LDS      query offset of step k+2, profile bytes of step k+1 (both complete before the statement ends)
//   per cell penalty select, three candidate keys, v_max3, T form, Beginning tag, direction bits, end-cell tracker
//   bottom   cell -> 64-deep lane shift register towards lane 63's publisher (wave_shl:1)
// `two` is 2 in every lane except lane 0 of strip 0, which holds an impossible T: row 1's carried penalty comes from the
// advice (not from the border above), and this path only runs over column groups whose advice bits are all zero.
#define ALN_S_HEAD(ROR, PWREAD)                                                                                \
    "v_mov_b32_dpp %[tin], %[G] " ROR " row_mask:0x1 bank_mask:0x1\n\t"                                      \
    "v_mov_b32_dpp %[tin], %[TL] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"                                  \
    "v_cmp_eq_u32 vcc, %[two], %[tin]\n\t"                                                                    \
    "ds_read_u16 %[qvn], %[qop] offset:%[qoff]\n\t"                                                          \
    "v_add_u32 %[la], %[prow], %[qvc]\n\t"                                                                   \
    PWREAD " %[pwn], %[la]\n\t"
#define ALN_S_TAIL                                                                                             \
    "v_mov_b32_dpp %[on], %[oo] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"                                    \
    "s_waitcnt lgkmcnt(0)"
#define ALN_S_BODY1                                                                                            \
    "v_add_u32_sdwa %[c0], %[hd], sext(%[pwc]) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t" \
    "v_cndmask_b32 %[np], %[ne], %[nd], vcc\n\t"                                                              \
    "v_add_u32 %[ta], %[tin], %[np]\n\t"                                                                      \
    "v_add3_u32 %[tb], %[TL], %[np], -1\n\t"                                                                  \
    "v_max3_i32 %[k0], %[ta], %[tb], %[c0]\n\t"                                                               \
    "v_and_or_b32 %[TL], %[k0], -4, 2\n\t"                                                                    \
    "v_max_u32 %[k0], %[k0], 3\n\t"                                                                           \
    "v_mov_b32 %[on], %[TL]\n\t"                                                                           \
    "v_lshl_add_u32 %[p0], %[TL], 11, %[kt]\n\t"                                                              \
    "v_alignbit_b32 %[dw], %[k0], %[dw], 2\n\t"                                                               \
    "v_max_i32 %[r0], %[r0], %[p0]\n\t"
// R = 2: row 1's diagonal is row 0's previous cell, its top is row 0's new cell
#define ALN_S_BODY2                                                                                            \
    "v_add_u32_sdwa %[c0], %[hd], sext(%[pwc]) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t" \
    "v_add_u32_sdwa %[c1], %[T0], sext(%[pwc]) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t" \
    "v_cndmask_b32 %[np], %[ne], %[nd], vcc\n\t"                                                              \
    "v_add_u32 %[ta], %[tin], %[np]\n\t"                                                                      \
    "v_add3_u32 %[tb], %[T0], %[np], -1\n\t"                                                                  \
    "v_max3_i32 %[k0], %[ta], %[tb], %[c0]\n\t"                                                               \
    "v_cmp_gt_u32 vcc, 3, %[k0]\n\t"                                                                          \
    "v_and_or_b32 %[T0], %[k0], -4, 2\n\t"                                                                    \
    "v_max_u32 %[k0], %[k0], 3\n\t"                                                                           \
    "v_cndmask_b32 %[np], %[ne], %[nd], vcc\n\t"                                                              \
    "v_add_u32 %[ta], %[T0], %[np]\n\t"                                                                       \
    "v_add3_u32 %[tb], %[TL], %[np], -1\n\t"                                                                  \
    "v_max3_i32 %[c1], %[ta], %[tb], %[c1]\n\t"                                                               \
    "v_alignbit_b32 %[dw], %[k0], %[dw], 2\n\t"                                                               \
    "v_and_or_b32 %[TL], %[c1], -4, 2\n\t"                                                                    \
    "v_max_u32 %[c1], %[c1], 3\n\t"                                                                           \
    "v_mov_b32 %[on], %[TL]\n\t"                                                                           \
    "v_lshl_add_u32 %[p0], %[T0], 11, %[kt]\n\t"                                                              \
    "v_alignbit_b32 %[dw], %[c1], %[dw], 2\n\t"                                                               \
    "v_max_i32 %[r0], %[r0], %[p0]\n\t"                                                                       \
    "v_lshl_add_u32 %[p0], %[TL], 11, %[kt]\n\t"                                                              \
    "v_max_i32 %[r1], %[r1], %[p0]\n\t"

// lane-local state of the asm step (all VGPRs) -- the same quantities FastStrip carries through its C++ step
struct SingleRegs {
    int T0, TL;               // this lane's cells of the previous column (T form); R = 1 uses TL only
    int r0, r1;               // packed end-cell trackers
    int hd;                   // diagonal of row 0: the cell that came in from above one step ago
    uint32_t dw;              // direction bits
    uint32_t pw, qv;          // profile bytes of this step / query offset of the next step's profile read
    int outq;                 // publisher shift register
};
template <int R, int NR>
__device__ __forceinline__ void single_step_asm(SingleRegs &s, int G, int two, uint32_t qop, uint32_t prow, int ne, int nd, int kt)
{
    int tin, np, ta, tb, c0, c1, k0, p0, on;
    uint32_t qvn, pwn, la;
#define ALN_S_OPERANDS                                                                                         \
    : [tin] "=&v"(tin), [np] "=&v"(np), [ta] "=&v"(ta), [tb] "=&v"(tb), [c0] "=&v"(c0), [c1] "=&v"(c1),      \
      [k0] "=&v"(k0), [p0] "=&v"(p0), [on] "=&v"(on), [qvn] "=&v"(qvn), [pwn] "=&v"(pwn), [la] "=&v"(la),     \
      [T0] "+v"(s.T0), [TL] "+v"(s.TL), [r0] "+v"(s.r0), [r1] "+v"(s.r1), [dw] "+v"(s.dw)                     \
    : [G] "v"(G), [two] "v"(two), [qop] "v"(qop), [prow] "v"(prow), [qvc] "v"(s.qv), [pwc] "v"(s.pw),         \
      [hd] "v"(s.hd), [ne] "v"(ne), [nd] "v"(nd), [oo] "v"(s.outq), [kt] "s"(kt), [qoff] "n"(2 * NR + 4),      \
      [nr] "n"(NR)                                                                                             \
    : "vcc"
    if constexpr (R == 1) {
        if constexpr (NR == 0) asm volatile(ALN_S_HEAD("quad_perm:[0,1,2,3]", "ds_read_u8") ALN_S_BODY1 ALN_S_TAIL ALN_S_OPERANDS);
        else asm volatile(ALN_S_HEAD("row_ror:%[nr]", "ds_read_u8") ALN_S_BODY1 ALN_S_TAIL ALN_S_OPERANDS);
    } else {
        if constexpr (NR == 0) asm volatile(ALN_S_HEAD("quad_perm:[0,1,2,3]", "ds_read_u16") ALN_S_BODY2 ALN_S_TAIL ALN_S_OPERANDS);
        else asm volatile(ALN_S_HEAD("row_ror:%[nr]", "ds_read_u16") ALN_S_BODY2 ALN_S_TAIL ALN_S_OPERANDS);
    }
#undef ALN_S_OPERANDS
    s.hd = tin; s.pw = pwn; s.qv = qvn; s.outq = on;
}

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
    int *brow, *brow0;        // boundary row (in place) / copy of strip 0's bottom row from the checkpointed pass
    uint8_t *advice, *zrow;
    int *ckpt;
    bool hazard;
    bool store_dirs;          // false: score-only (the packed directions are not written)
    bool pwm;                 // position-weight-matrix scoring (batch kernels only)
    const uint32_t *pwm_words;// per column: int8 scores 4*s - 2 of residues 0..3
    int ck_mode;              // 0 plain, 1 save checkpoints, 2 repair
    uint32_t last_flip;
    uint16_t *qo_pad;         // single-pair kernel: LDS, q[x] * 64R at index x + 63, zeros elsewhere (N + 192 entries)
    int *bring;               // single-pair kernel: LDS, 2 x 64 ints
    const uint64_t *gin;      // single-pair kernel: granule rows
    uint64_t *gout;
    uint32_t *abort_flag;
};

// The lane's running end-cell candidate (T form) + outcome flags; threaded through the strips by value.
struct FastOut {
    int bv;
    uint32_t by, bx;
    int corner;
    bool repaired, brow_bad, aborted;
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

template <int SEM, int R, bool SINGLE, bool FIRST, bool LAST>
struct FastStrip {
    static constexpr int SPB = 16 / R;
    static constexpr uint32_t STRIP_ROWS = SINGLE ? 64u * R : (uint32_t)ALN_STRIP_ROWS;
    static constexpr bool LOCAL = (SEM == ALN_CORE_LOCAL || SEM == ALN_LEGACY_LOCAL);
    using PW = typename ProfWord<R>::T;
    const FastIn in;
    const uint32_t strip;
    static constexpr bool last = LAST;
    const int lane;
    const uint32_t N;
    uint32_t lb, rb, yb;
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
    uint64_t gpre;             // prefetched granule of the next 16-column group (lanes 0..15)
    // hand-written steady state (single-pair kernel, core local, R <= 2)
    static constexpr bool ASMPATH = SINGLE && SEM == ALN_CORE_LOCAL && (R == 1 || R == 2);
    int twov;                  // 2; lane 0 of strip 0: a value no cell takes (row 1's penalty never follows the border)
    uint64_t advmask;          // strip 0: which of the 64 advice bytes in advchunk are set

    __device__ __forceinline__ FastStrip(const FastIn &i, uint32_t s)
        : in(i), strip(s), lane(i.lane), N(i.N), brow_bad(false), aborted(false) {}

    // PWM scoring: the flowing register holds the column's four packed scores; one v_perm per four rows picks each
    // row's byte by its residue code -- the result has the layout of a profile read
    uint32_t psel_lo, psel_hi;
    __device__ __forceinline__ PW pwm_select(uint32_t w4) const
    {
        if constexpr (R == 8) return make_uint2(__builtin_amdgcn_perm(w4, w4, psel_lo), __builtin_amdgcn_perm(w4, w4, psel_hi));
        else return (PW)__builtin_amdgcn_perm(w4, w4, psel_lo);
    }

    // next 64 columns of the row above this strip (T form), one per lane
    __device__ __forceinline__ int load_boundary(uint32_t xi)
    {
        if constexpr (!SINGLE) {
            return (xi < N) ? in.brow[xi + 1] : 2;
        } else {
            const uint64_t *src = in.gin + xi;
            uint64_t g = 0;
            uint32_t spins = 0;
            for (;;) {
                const bool need = xi < N;
                if (need) g = granule_load(src);
                if (__all(!need || (g >> 32) != 0)) break;
                __builtin_amdgcn_s_sleep(1);
                ++spins;
                if (spins > (1u << 22) ||
                    ((spins & 1023u) == 0 && __hip_atomic_load(in.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                    // the producer never arrived: poison the run instead of hanging the GPU
                    if (lane == 0) __hip_atomic_store(in.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    aborted = true;
                    break;
                }
            }
            return (xi < N) ? (int)(uint32_t)g : 2;
        }
    }

    template <bool MASKED>
    __device__ __forceinline__ void step(const uint32_t k)
    {
        if ((k & 63u) == 0) {                                   // wave-uniform: refill the 64-column input chunks
            const uint32_t xi = k + (uint32_t)lane;             // 0-based column
            if (!FIRST && !SINGLE) inchunk = load_boundary(xi);
            if (SEM == ALN_CORE_LOCAL && FIRST && in.hazard) advchunk = (xi < N) ? in.advice[xi + 1] : 0u;
            if (!SINGLE) qchunk = (xi + 1 < N) ? (in.pwm ? (int)in.pwm_words[xi + 1] : (int)in.q[xi + 1] * (64 * R)) : 0;
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
            pw = in.pwm ? pwm_select((uint32_t)qoff) : *reinterpret_cast<const PW *>(prow + qoff);
        }
        // end-cell tie-break term of this step: earlier steps win (core) / later steps win (legacy)
        const int kterm = (SEM == ALN_CORE_LOCAL) ? (int)(2047u - (k & 2047u)) : (int)(k & 2047u);
        const uint32_t xm1 = k - (uint32_t)lane;
        if (!MASKED || xm1 < N) {
            int top = topIn, diag = hdiag;
            bool zr = (topIn == 2);                             // "cell above is Beginning" -> penalty del
            if (SEM == ALN_CORE_LOCAL && FIRST && lane == 0) {
                // row 1: the carried penalty comes from the bottom cell of the previous column (advice)
                zr = (k == 0) || (adv != 0);
            }
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
            if (!SINGLE && !last && lane == 63) {
                const uint32_t x = xm1 + 1;
                if (FIRST && SEM == ALN_CORE_LOCAL && in.ck_mode == 2) { if (in.brow0[x] != bottom) brow_bad = true; }
                else {
                    in.brow[x] = bottom;
                    if (FIRST && SEM == ALN_CORE_LOCAL && in.ck_mode == 1) in.brow0[x] = bottom;
                }
            }
            if (SEM == ALN_CORE_LOCAL && !SINGLE && zsel_on && (uint32_t)lane == lb) {
                int hb = Tl[0];
#pragma unroll
                for (int r = 1; r < R; ++r) if ((uint32_t)r == rb) hb = Tl[r];
                in.zrow[xm1 + 1] = (hb == 2) ? 1 : 0;
            }
        }
        // bottom row to the strip below: lane 63's newest cell enters a 64-deep lane shift register (DPP wave_shl:1)
        if (SINGLE && !LAST) outq = __builtin_amdgcn_update_dpp(bottom, outq, 0x130, 0xf, 0xf, false);
    }

    // single-pair kernel: after step k lanes 48..63 hold the bottom-row cells of columns c-15..c, c = k - 63; one
    // 128-byte write-through store publishes them to the strip below
    __device__ __forceinline__ void publish(const uint32_t k)
    {
        const uint32_t c = k - 63u;
        const uint32_t col = c - 63u + (uint32_t)lane;              // wraps for lanes that hold nothing yet
        if (lane >= 48 && col < N) granule_store(in.gout + col, outq);
    }

    // single-pair kernel: boundary cells arrive 16 columns at a time.  The granules of group j+1 are requested when
    // group j is staged (lanes 0..15, non-blocking), so in steady state staging never waits on HBM/L2; only a consumer
    // that has caught up with its producer spins (bounded) until the tags appear.
    __device__ __forceinline__ void stage_boundary16(const uint32_t j)
    {
        const uint32_t col = 16u * j + (uint32_t)lane;
        const bool need = lane < 16 && col < N;
        uint64_t g = gpre;
        uint32_t spins = 0;
        while (!__all(!need || (g >> 32) != 0)) {
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
        if (lane < 16) bring[col & 127u] = need ? (int)(uint32_t)g : 2;
        const uint32_t ncol = col + 16u;
        gpre = (lane < 16 && ncol < N) ? granule_load(in.gin + ncol) : 0ull;       // group j+1, consumed next time
    }

    // four blocks of SPB steps -> one 16-byte store per lane (1 KiB per wave, coalesced).  The block loop is a real
    // loop (not unrolled): unrolling 4*SPB steps makes the scheduler hoist every step's uniform values and spill.
    // the single-pair kernel's bottom-row zero flags travel as the direction words of the lane that owns row M (tag 3 =
    // Beginning <=> H == 0): one dword per block instead of one byte store per step
    __device__ __forceinline__ void store_zdw(uint32_t block)
    {
        if (SINGLE && SEM == ALN_CORE_LOCAL && zsel_on && (uint32_t)lane == lb) reinterpret_cast<uint32_t *>(in.zrow)[block] = dw;
    }

    // 16 steady-state steps through the asm step (R = 1: one block, R = 2: two blocks), w0 / w1 = their direction words
    template <int B>
    __device__ __forceinline__ void asm_block(SingleRegs &sr, const int G, const uint32_t qop, const uint32_t prow32, const int kt0)
    {
        if constexpr (R == 1) {
            single_step_asm<1, 0>(sr, G, twov, qop, prow32, in.ne4, in.nd4, kt0 - 0);
            single_step_asm<1, 1>(sr, G, twov, qop, prow32, in.ne4, in.nd4, kt0 - 1);
            single_step_asm<1, 2>(sr, G, twov, qop, prow32, in.ne4, in.nd4, kt0 - 2);
            single_step_asm<1, 3>(sr, G, twov, qop, prow32, in.ne4, in.nd4, kt0 - 3);
            single_step_asm<1, 4>(sr, G, twov, qop, prow32, in.ne4, in.nd4, kt0 - 4);
            single_step_asm<1, 5>(sr, G, twov, qop, prow32, in.ne4, in.nd4, kt0 - 5);
            single_step_asm<1, 6>(sr, G, twov, qop, prow32, in.ne4, in.nd4, kt0 - 6);
            single_step_asm<1, 7>(sr, G, twov, qop, prow32, in.ne4, in.nd4, kt0 - 7);
            single_step_asm<1, 8>(sr, G, twov, qop, prow32, in.ne4, in.nd4, kt0 - 8);
            single_step_asm<1, 9>(sr, G, twov, qop, prow32, in.ne4, in.nd4, kt0 - 9);
            single_step_asm<1, 10>(sr, G, twov, qop, prow32, in.ne4, in.nd4, kt0 - 10);
            single_step_asm<1, 11>(sr, G, twov, qop, prow32, in.ne4, in.nd4, kt0 - 11);
            single_step_asm<1, 12>(sr, G, twov, qop, prow32, in.ne4, in.nd4, kt0 - 12);
            single_step_asm<1, 13>(sr, G, twov, qop, prow32, in.ne4, in.nd4, kt0 - 13);
            single_step_asm<1, 14>(sr, G, twov, qop, prow32, in.ne4, in.nd4, kt0 - 14);
            single_step_asm<1, 15>(sr, G, twov, qop, prow32, in.ne4, in.nd4, kt0 - 15);
        } else {
            single_step_asm<2, B + 0>(sr, G, twov, qop, prow32, in.ne4, in.nd4, kt0 - B - 0);
            single_step_asm<2, B + 1>(sr, G, twov, qop, prow32, in.ne4, in.nd4, kt0 - B - 1);
            single_step_asm<2, B + 2>(sr, G, twov, qop, prow32, in.ne4, in.nd4, kt0 - B - 2);
            single_step_asm<2, B + 3>(sr, G, twov, qop, prow32, in.ne4, in.nd4, kt0 - B - 3);
            single_step_asm<2, B + 4>(sr, G, twov, qop, prow32, in.ne4, in.nd4, kt0 - B - 4);
            single_step_asm<2, B + 5>(sr, G, twov, qop, prow32, in.ne4, in.nd4, kt0 - B - 5);
            single_step_asm<2, B + 6>(sr, G, twov, qop, prow32, in.ne4, in.nd4, kt0 - B - 6);
            single_step_asm<2, B + 7>(sr, G, twov, qop, prow32, in.ne4, in.nd4, kt0 - B - 7);
        }
    }

    // steady-state quad of the single-pair core-local kernel: 16-column units through the asm step; a unit of strip 0
    // whose advice bits are not all zero takes the C++ step (rare: second and later passes only)
    __device__ __forceinline__ void quad_asm(uint4 *dirq, const uint32_t kb)
    {
        uint4 v = make_uint4(0, 0, 0, 0);
        const uint32_t prow32 = (uint32_t)(uintptr_t)prow;
#pragma unroll 1
        for (uint32_t u = 0; u < 4u / R; ++u) {
            const uint32_t ku = (kb + u * R) * SPB;                     // first step of the unit, a multiple of 16
            bool slow = false;
            if (FIRST && in.hazard) {
                if ((ku & 63u) == 0) {
                    const uint32_t xi = ku + (uint32_t)lane;
                    advchunk = (xi < N) ? in.advice[xi + 1] : 0u;
                    advmask = __ballot(advchunk != 0);
                }
                slow = ((advmask >> (ku & 63u)) & 0xffffull) != 0;
            }
            if (slow) {
                if (!FIRST) top0v = bring[ku & 127u];
#pragma unroll 1
                for (uint32_t jj = 0; jj < (uint32_t)R; ++jj) {
                    const uint32_t k0 = ku + jj * SPB;
                    if (!FIRST && ((k0 + SPB) & 15u) == 0) stage_boundary16((k0 + SPB) >> 4);
#pragma unroll
                    for (int kk = 0; kk < SPB; ++kk) step<false>(k0 + kk);
                    if (!LAST && ((k0 + SPB) & 15u) == 0) publish(k0 + SPB - 1);
                    const uint32_t w = u * R + jj;
                    if (w == 0) v.x = dw; else if (w == 1) v.y = dw; else if (w == 2) v.z = dw; else v.w = dw;
                    store_zdw(kb + w);
                }
                continue;
            }
            // boundary cells of columns ku .. ku+15 in the order row_ror hands them to lane 0
            int G = 2;
            if (!FIRST) G = bring[(ku + ((16u - (uint32_t)lane) & 15u)) & 127u];
            SingleRegs sr;
            sr.T0 = Tl[0]; sr.TL = Tl[R - 1]; sr.r0 = rbv[0]; sr.r1 = rbv[R - 1]; sr.hd = hdiag; sr.dw = dw;
            sr.pw = (uint32_t)pw; sr.qv = (uint32_t)qv; sr.outq = outq;
            const uint32_t qop = (uint32_t)(uintptr_t)qo_lane + 2u * ku;
            const int kt0 = (int)(2047u - (ku & 2047u));
            if constexpr (R == 1) {
                if (!FIRST) stage_boundary16((ku + 16u) >> 4);
                asm_block<0>(sr, G, qop, prow32, kt0);
                if (u == 0) v.x = sr.dw; else if (u == 1) v.y = sr.dw; else if (u == 2) v.z = sr.dw; else v.w = sr.dw;
            } else {
                asm_block<0>(sr, G, qop, prow32, kt0);
                if (u == 0) v.x = sr.dw; else v.z = sr.dw;
                if (zsel_on && (uint32_t)lane == lb) reinterpret_cast<uint32_t *>(in.zrow)[kb + 2 * u] = sr.dw;
                if (!FIRST) stage_boundary16((ku + 16u) >> 4);
                asm_block<8>(sr, G, qop, prow32, kt0);
                if (u == 0) v.y = sr.dw; else v.w = sr.dw;
            }
            if (R > 1) { Tl[0] = sr.T0; rbv[R - 1] = sr.r1; }
            Tl[R - 1] = sr.TL; rbv[0] = sr.r0; hdiag = sr.hd; dw = sr.dw;
            pw = (PW)sr.pw; qv = (int)sr.qv; outq = sr.outq; bottom = sr.TL;
            if (!LAST) publish(ku + 15u);
            store_zdw(kb + u * R + (R - 1));
        }
        if (in.store_dirs) dirq[(size_t)(kb >> 2) * 64] = v;
    }

    template <bool MASKED>
    __device__ __forceinline__ void quad(uint4 *dirq, const uint32_t kb)
    {
        if constexpr (!MASKED && ASMPATH) { quad_asm(dirq, kb); return; }
        uint4 v = make_uint4(0, 0, 0, 0);
#pragma unroll 1
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t k0 = (kb + j) * SPB;
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
        if (in.store_dirs) dirq[(size_t)(kb >> 2) * 64] = v;
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

    // Lane state at a block boundary (direction word flushed, input chunks about to be reloaded): everything the
    // rest of the strip depends on besides the inputs.  save = store it, !save = "is it identical to the stored one".
    __device__ __forceinline__ bool checkpoint(uint32_t slot, bool save)
    {
        int *base = in.ckpt + slot * (18 * 64) + lane;
        bool same = true;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (save) { base[(2 * r) * 64] = Tl[r]; base[(2 * r + 1) * 64] = rbv[r]; }
            else same = same && base[(2 * r) * 64] == Tl[r] && base[(2 * r + 1) * 64] == rbv[r];
        }
        if (save) { base[16 * 64] = hdiag; base[17 * 64] = bottom; }
        else same = same && base[16 * 64] == hdiag && base[17 * 64] == bottom;
        return same;
    }

    __device__ __forceinline__ FastOut run(FastOut o)
    {
        const uint32_t M = in.M;
        const uint32_t y0 = strip * STRIP_ROWS;
        const uint32_t rows = min(M - y0, (uint32_t)(64 * R));
        const uint32_t L = (rows + R - 1) / R;
        const uint32_t nsteps = (SINGLE && !LAST) ? N + 63 : N + L - 1;
        yb = y0 + (uint32_t)lane * R;
        lb = (rows - 1) / R; rb = (rows - 1) % R;
        zsel_on = (SEM == ALN_CORE_LOCAL) && LAST && in.hazard;
        prow = in.prof + lane * R;

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
        for (uint32_t c = 0; c < (in.pwm ? 0u : in.cols); ++c) {
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t bte = (uint32_t)(4 * in.S[tc[r] + c] - 2) & 0xffu;
                if (r < 4) lo |= bte << (8 * r); else hi |= bte << (8 * (r - 4));
            }
            uint8_t *dst = in.prof + c * (64 * R) + lane * R;
            if constexpr (R == 8) *reinterpret_cast<uint2 *>(dst) = make_uint2(lo, hi);
            else if constexpr (R == 4) *reinterpret_cast<uint32_t *>(dst) = lo;
            else if constexpr (R == 2) *reinterpret_cast<uint16_t *>(dst) = (uint16_t)lo;
            else *dst = (uint8_t)lo;
        }
        hdiag = LOCAL || yb == 0 ? 2 : 2 + (int)yb * in.nd4;            // H[yb][0]; yb < M always for valid lanes
        bottom = Tl[R - 1];
        inchunk = 2; qchunk = 0; advchunk = 0; dw = 0; outq = 0; qv = 0; top0v = 2;
        twov = (FIRST && lane == 0) ? 1 : 2;                 // T is always 2 (mod 4)
        advmask = 0;
        if constexpr (SINGLE) {
            qo_lane = reinterpret_cast<const uint8_t *>(in.qo_pad + 63 - lane);
            bring = in.bring;
            gpre = (!FIRST && lane < 16 && (uint32_t)lane < N) ? granule_load(in.gin + lane) : 0ull;
            if (!FIRST) { stage_boundary16(0); top0v = bring[0]; }
            qoff = *reinterpret_cast<const uint16_t *>(qo_lane);                      // step 0: column -lane
            qv = *reinterpret_cast<const uint16_t *>(qo_lane + 2);                    // step 1
        } else {
            qoff = (lane == 0) ? (in.pwm ? (int)in.pwm_words[0] : (int)in.q[0] * (64 * R)) : 0;
        }
        if (!SINGLE && in.pwm) pw = pwm_select((uint32_t)qoff);
        else pw = *reinterpret_cast<const PW *>(prow + qoff);

        // directions: four blocks per lane per 16-byte store (aln_device.h); all segment ends are whole quads
        uint4 *dirq = reinterpret_cast<uint4 *>(in.dirw) +
                      (size_t)strip * (SINGLE ? (size_t)(aln_uniform_strip_bytes(N, R) / 16) : (size_t)(aln_strip_bytes(N) / 16)) + lane;
        const uint32_t nkb = aln_strip_blocks(nsteps, SPB);
        // ramp-up (some lanes not started) | steady state (every lane active, no exec masking) | ramp-down
        const uint32_t kb_steady0 = min(nkb, (uint32_t)(64 / SPB));
        const uint32_t kb_steady1 = max(kb_steady0, min(nkb, (N / (4 * SPB)) * 4u));
        // Segment ends: the 2048-step chunks of the end-cell tracker and, for strip 0 of a hazard pair, the
        // checkpoint steps 64, 128, 256, 512.
        const bool ckmode = FIRST && !SINGLE && SEM == ALN_CORE_LOCAL && in.ck_mode != 0;
        uint32_t next_ck = ckmode ? 64u : 0xffffffffu, slot = 0, chunk_base = 0;
        uint32_t kb = 0;
        while (kb < nkb) {
            uint32_t seg_end = min(nkb, (chunk_base + 2048u) / SPB);
            if (next_ck != 0xffffffffu) seg_end = min(seg_end, next_ck / SPB);
            const uint32_t e0 = min(kb_steady0, seg_end), e1 = min(kb_steady1, seg_end);
            for (; kb < e0; kb += 4) quad<true>(dirq, kb);
            for (; kb < e1; kb += 4) quad<false>(dirq, kb);
            if (ASMPATH && !FIRST) top0v = bring[(kb * SPB) & 127u];     // the C++ step reads its boundary cell one step ahead
            for (; kb < seg_end; kb += 4) quad<true>(dirq, kb);
            if (ckmode && kb < nkb && kb * SPB == next_ck) {
                if (in.ck_mode == 1) checkpoint(slot, true);
                else if (__all(checkpoint(slot, false)) && in.last_flip <= next_ck) {
                    // every lane is in exactly the state the checkpointed pass had here and no advice differs from
                    // here on: the rest of this strip -- and so of the whole fill -- is unchanged
                    o.repaired = true;
                    o.brow_bad = o.brow_bad || brow_bad;
                    return o;
                }
                ++slot;
                next_ck = next_ck < 512u ? next_ck * 2u : 0xffffffffu;
            }
            if (LOCAL && kb * SPB == chunk_base + 2048u) { fold(o, chunk_base); chunk_base += 2048u; }
        }
        if (SINGLE && !LAST) publish(nkb * SPB - 1);     // the last (up to 15) columns
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

// Batch kernels: every strip but the last has 512 rows (R = 8); the last one picks R by its row count.
template <int SEM>
__device__ __forceinline__ FastOut fast_strip(const FastIn &in, FastOut o, uint32_t s, bool last, int R)
{
    if (!last) {
        if (s == 0) { FastStrip<SEM, ALN_FULL_R, false, true, false> f(in, s); return f.run(o); }
        FastStrip<SEM, ALN_FULL_R, false, false, false> f(in, s);
        return f.run(o);
    }
    if (s == 0) {
        if (R == 8) { FastStrip<SEM, 8, false, true, true> f(in, s); return f.run(o); }
        if (R == 4) { FastStrip<SEM, 4, false, true, true> f(in, s); return f.run(o); }
        if (R == 2) { FastStrip<SEM, 2, false, true, true> f(in, s); return f.run(o); }
        FastStrip<SEM, 1, false, true, true> f(in, s);
        return f.run(o);
    }
    if (R == 8) { FastStrip<SEM, 8, false, false, true> f(in, s); return f.run(o); }
    if (R == 4) { FastStrip<SEM, 4, false, false, true> f(in, s); return f.run(o); }
    if (R == 2) { FastStrip<SEM, 2, false, false, true> f(in, s); return f.run(o); }
    FastStrip<SEM, 1, false, false, true> f(in, s);
    return f.run(o);
}

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
