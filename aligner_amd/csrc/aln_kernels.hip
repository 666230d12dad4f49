// aln_kernels.hip -- gfx950 kernels of the DP matrix-fill + traceback path.
//
// What is computed (reference: aligner-core/src/simple/mod.rs:42-145 global, :168-264 local;
// tie rules aligner-core/src/enums.rs:17-47; legacy twin src/align/aligner_core.rs:96-269):
//   H[y][x] = max(H[y-1][x] - p, H[y][x-1] - p, H[y-1][x-1] + S[t[y-1]][q[x-1]])        (+ clamp at 0: legacy local)
//   D[y][x] = Beginning if H==0 (local) else Top / Left / Diagonal in that tie order
// with the reference's loop-carried penalty p (NOT Gotoh): see penalty<>() below.
//
// Mapping to the machine (CDNA4, wave64):
//   * one wave owns a strip of up to 512 target rows; lane l owns R consecutive rows (register blocked);
//   * the wave walks the query with an anti-diagonal skew: at step k lane l is at column x = k - l + 1, so the only
//     cross-lane traffic per step is ONE DPP wave_shr:1 (lane l-1's bottom cell -> lane l's top input);
//   * strip s+1 consumes the bottom row of strip s through a per-wave boundary row (HBM/L2 resident, read in
//     64-column chunks and fed to lane 0 with v_readlane);
//   * 2-bit directions are packed 16 per lane-register; four such blocks of a lane form a 16-byte quad and a wave stores
//     1 KiB contiguous per quad (aln_device.h);
//   * the local end cell is tracked per row in registers and reduced with DPP/shuffles at the end;
//   * one large pair instead runs one wave per strip, four strips per workgroup, pipelined (aln_fill_single_kernel).
// No MFMA: this is integer (or exact f64) max-plus DP.
#include <hip/hip_runtime.h>
#include <float.h>
#include <limits.h>
#include <math.h>

#include "aln_device.h"

// The file is compiled as several translation units in parallel (aligner_amd/build.py: -DALN_TU=<mask>), each instantiating one
// family of kernels together with the launch helpers that name them; the templates themselves are seen by every unit.
#define ALN_PART_GENERIC 1     // generic fill kernels (int32 without the profile, f64), one-workgroup route, validation
#define ALN_PART_FAST_CL 2     // fast integer batch kernel, core local, with the cooperative passes
#define ALN_PART_FAST_REST 4   // fast integer batch kernels with the cooperative passes: core local with PWM scoring, core global, legacy
#define ALN_PART_FAST_CL_SOLO 32      // the same two without (batches that have nothing to share)
#define ALN_PART_FAST_REST_SOLO 64
#define ALN_PART_SINGLE 8      // single-pair (strip-pipelined) route
#define ALN_PART_TB 16         // traceback + direction unpack
#ifndef ALN_TU
#define ALN_TU 127
#endif

namespace {

constexpr int D_TOP = 0, D_LEFT = 1, D_DIAG = 2, D_BEG = 3;

// ---------------------------------------------------------------- score-type helpers
template <typename SC> struct ScOps;

template <> struct ScOps<int> {
    static __device__ __forceinline__ bool eq(int m, int a) { return m == a; }
    static __device__ __forceinline__ int vmax(int a, int b) { return a > b ? a : b; }
    static __device__ __forceinline__ int lowest() { return INT_MIN; }
    static __device__ __forceinline__ int shr1(int old, int v)
    {   // lane l <- lane l-1; lane 0 keeps `old`  (DPP wave_shr:1, gfx9)
        return __builtin_amdgcn_update_dpp(old, v, 0x138, 0xf, 0xf, false);
    }
    static __device__ __forceinline__ int rdlane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
    static __device__ __forceinline__ int xshfl(int v, int m) { return __shfl_xor(v, m); }
    static __device__ __forceinline__ int from_double(double d) { return (int)d; }
};

template <> struct ScOps<double> {
    // enums.rs:21-25: (max - x).abs() < f64::EPSILON
    static __device__ __forceinline__ bool eq(double m, double a) { return fabs(m - a) < DBL_EPSILON; }
    static __device__ __forceinline__ double vmax(double a, double b) { return fmax(a, b); }
    static __device__ __forceinline__ double lowest() { return -DBL_MAX; }
    static __device__ __forceinline__ double shr1(double old, double v)
    {
        int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), 0x138, 0xf, 0xf, false);
        int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), 0x138, 0xf, 0xf, false);
        return __hiloint2double(hi, lo);
    }
    static __device__ __forceinline__ double rdlane(double v, int l)
    {
        int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
        int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
        return __hiloint2double(hi, lo);
    }
    static __device__ __forceinline__ double xshfl(double v, int m) { return __shfl_xor(v, m); }
    static __device__ __forceinline__ double from_double(double d) { return d; }
};

template <int SEM> constexpr bool is_local() { return SEM == ALN_CORE_LOCAL || SEM == ALN_LEGACY_LOCAL; }
template <int SEM> constexpr bool is_core() { return SEM == ALN_CORE_GLOBAL || SEM == ALN_CORE_LOCAL; }

// H[0][x]: simple/mod.rs:59-62 + the overwritten corner :69 (legacy :104-107, :116-117)
template <typename SC, int SEM>
__device__ __forceinline__ SC border_top(uint32_t x, uint32_t N, SC del)
{
    if (is_local<SEM>() || x == 0) return (SC)0;
    return x == N ? -((SC)N + (SC)1) * del : -(SC)x * del;
}
// H[y][0]: simple/mod.rs:64-67 + the overwritten corner :70 (legacy :109-115)
template <typename SC, int SEM>
__device__ __forceinline__ SC border_left(uint32_t y, uint32_t M, SC del)
{
    if (is_local<SEM>() || y == 0) return (SC)0;
    return y == M ? -((SC)M + (SC)1) * del : -(SC)y * del;
}

// One cell.  topH/leftH/diagH are neighbour H values, p the carried penalty, s the substitution score.
template <typename SC, int SEM>
__device__ __forceinline__ void cell(SC topH, SC leftH, SC diagH, SC s, SC p, SC &h, int &d)
{
    using O = ScOps<SC>;
    const SC a = topH - p, b = leftH - p, c = diagH + s;
    SC m = O::vmax(O::vmax(a, b), c);
    if (SEM == ALN_LEGACY_LOCAL) m = O::vmax(m, (SC)0);              // aligner_core.rs:210
    int dd = O::eq(m, a) ? D_TOP : (O::eq(m, b) ? D_LEFT : D_DIAG);  // Top > Left > Diagonal
    if (is_local<SEM>() && m == (SC)0) dd = D_BEG;                   // enums.rs:37 / aligner_core.rs:214
    h = m;
    d = dd;
}

// The same cell for the strip kernels, without a branch: the stored 2-bit tag (0 Diagonal, 1 Left, 2 Top, 3 Beginning) comes out
// of selects (the strip loops ran ~5 scalar branches per cell before; each costs a wave ~16 cycles whether taken or not).
template <typename SC, int SEM>
__device__ __forceinline__ uint32_t cell_tag(SC topH, SC leftH, SC diagH, SC s, SC p, SC &h)
{
    using O = ScOps<SC>;
    const SC a = topH - p, b = leftH - p, c = diagH + s;
    SC m = O::vmax(O::vmax(a, b), c);
    if (SEM == ALN_LEGACY_LOCAL) m = O::vmax(m, (SC)0);
    uint32_t tag = O::eq(m, b) ? 1u : 0u;
    tag = O::eq(m, a) ? 2u : tag;                                    // Top > Left > Diagonal
    if (is_local<SEM>()) tag = (m == (SC)0) ? 3u : tag;              // enums.rs:37 / aligner_core.rs:214
    h = m;
    return tag;
}

// The f64 cell, instruction by instruction (every real-valued caller runs it: each HeuristicAligner iteration, heuristic/mod.rs:58-77,
// the re-estimated PWMs of latent-repeat-search, engine/calc.rs:107-136).  Left to the compiler, `fmax` became a canonicalizing
// v_max_f64 pair per operand, fabs a v_and, every select a pair of v_cndmask on register halves it had just moved together, and
// the cell came to ~40 VALU instructions with 29 spilled registers.  Here: three adds, two maxes (NaN cannot occur: finite inputs, only
// add / max), the two distances m - a, m - b through the neg modifier (m >= a, b: no abs), three compares against constants and
// three selects of small constants for the tag.  The literal (max - x).abs() < f64::EPSILON of enums.rs:21-25 stays literal.
template <int SEM>
__device__ __forceinline__ uint32_t cell_tag_f64(double top, double left, double diag, double s, double negp, double &h, bool &zero)
{
    double a, b, c, m, da, db;
    asm("v_add_f64 %0, %6, %9\n\t"
        "v_add_f64 %1, %7, %9\n\t"
        "v_add_f64 %2, %8, %10\n\t"
        "v_max_f64 %3, %0, %1\n\t"
        "v_max_f64 %3, %3, %2\n\t"
        "v_add_f64 %4, %3, -%0\n\t"
        "v_add_f64 %5, %3, -%1"
        : "=&v"(a), "=&v"(b), "=&v"(c), "=v"(m), "=v"(da), "=v"(db)      // m, da, db are written after the last read of an input: they may
        : "v"(top), "v"(left), "v"(diag), "v"(negp), "v"(s));             // take an input's registers (the new H lands where the old one was)
    uint32_t tag = (db < DBL_EPSILON) ? 1u : 0u;
    tag = (da < DBL_EPSILON) ? 2u : tag;                              // Top > Left > Diagonal
    zero = (m == 0.0);
    if (is_local<SEM>()) tag = zero ? 3u : tag;                       // enums.rs:37
    h = m;
    return tag;
}

__device__ __forceinline__ double max_f64_raw(double a, double b)
{
    double d;
    asm("v_max_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// The R cells of one lane for one wave step, f64 (core semantics only: the legacy ones are i32).  zr0: the carried penalty of the
// lane's first row is `del` (local: the cell above is Beginning -- or, for row 1, what the advice says; global: the very first cell).
// The penalty of the rows below follows from the cell just computed (simple/mod.rs:88-92, :201-205).  tag0: the tag of the first row.
template <int SEM, int R>
__device__ __forceinline__ void cells_f64(double topIn, double &hdiag, double (&Hl)[R], const double (&sc)[R], bool zr0, double nd, double ne,
                                          uint32_t &dw, double (&rbv)[R], uint32_t (&rbx)[R], uint32_t x, uint32_t &tag0)
{
    double top = topIn, diag = hdiag;
    bool zr = zr0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const double negp = zr ? nd : ne;
        double h;
        bool zero;
        const uint32_t tag = cell_tag_f64<SEM>(top, Hl[r], diag, sc[r], negp, h, zero);
        zr = (SEM == ALN_CORE_LOCAL) ? zero : false;
        diag = Hl[r];
        Hl[r] = h;
        top = h;
        dw = __builtin_amdgcn_alignbit(tag, dw, 2);                   // (dw >> 2) | (tag << 30)
        if (SEM == ALN_CORE_LOCAL) {
            rbx[r] = (h > rbv[r]) ? x : rbx[r];                         // first maximum of the row (simple/mod.rs:212)
            rbv[r] = max_f64_raw(rbv[r], h);
        }
        if (r == 0) tag0 = tag;
    }
    hdiag = topIn;
}

// ---------------------------------------------------------------- per-wave state
template <typename SC>
struct Wave {
    int lane;
    uint32_t N, M;
    const uint8_t *q, *t;
    const SC *S;              // LDS copy of the substitution matrix, [t][q] contiguous
    uint32_t cols;
    SC del, ext;
    uint32_t *dirw;           // this pair's direction region
    SC *brow;                 // boundary row scratch (N + 66)
    uint8_t *advice, *zrow;   // row-1 penalty advice / observed bottom-row zeros
    SC *row1;                 // hazard pairs: H[1][x] and its direction tag as the pass computed them (adopt_advice_checked)
    uint8_t *row1tag;
    SC *hmat;                 // optional H dump for this pair
    bool hazard;
    bool store_dirs;          // false: score-only
    bool pwm;                 // position-weight-matrix scoring: column index instead of a query residue
    // running end-cell candidate of this lane (local semantics) and final corner value (global)
    SC bv; uint32_t by, bx;
    SC corner;
    SC bt;                    // lean f64 strip: the best H any lane of the wave has seen in this pass (the tracker's threshold)
};

// a better-than-b for the local end cell
template <typename SC, int SEM>
__device__ __forceinline__ bool better(SC v, uint32_t y, uint32_t x, SC bv, uint32_t by, uint32_t bx)
{
    if (v > bv) return true;
    if (v < bv) return false;
    if (SEM == ALN_CORE_LOCAL)   // first maximum in row-major order (ndarray-stats argmax, simple/mod.rs:212)
        return y < by || (y == by && x < bx);
    // legacy: last `>=` in column-major visiting order (aligner_core.rs:224-228)
    return x > bx || (x == bx && y > by);
}

// ---------------------------------------------------------------- one strip of rows, R rows per lane
template <typename SC, int SEM, int R>
__device__ __forceinline__ void run_strip(Wave<SC> &w, const uint32_t strip, const bool last)
{
    using O = ScOps<SC>;
    constexpr int SPB = (int)aln_spb(R);              // steps per packed direction word
    const int lane = w.lane;
    const uint32_t N = w.N, M = w.M;
    const uint32_t y0 = strip * ALN_STRIP_ROWS;
    const uint32_t rows = min(M - y0, (uint32_t)(64 * R));
    const uint32_t L = (rows + R - 1) / R;            // lanes holding at least one valid row
    const uint32_t nsteps = N + L - 1;
    const uint32_t yb = y0 + (uint32_t)lane * R;      // this lane's rows are yb+1 .. yb+R
    const uint32_t lb = (rows - 1) / R, rb = (rows - 1) % R;   // where the strip's last valid row lives
    const SC del = w.del, ext = w.ext;

    int tc[R];
    SC Hl[R], rbv[R];
    uint32_t rbx[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t y = yb + 1 + r;
        tc[r] = (y <= M) ? (int)w.t[y - 1] * (int)w.cols : 0;
        Hl[r] = border_left<SC, SEM>(y, M, del);
        rbv[r] = (SEM == ALN_LEGACY_LOCAL) ? (SC)-1 : O::lowest();
        rbx[r] = 0;
    }
    SC hdiag = border_left<SC, SEM>(yb, M, del);      // H[yb][0]
    SC bottom = Hl[R - 1];
    SC inchunk = (SC)0;
    uint32_t advchunk = 0;
    // The query code travels down the lanes like the boundary cell (lane 0 takes column k + 1's at step k; PWM scoring: the
    // column index itself), and the scores of the NEXT step are read from LDS while this one computes: no memory access sits in
    // a step's dependency chain (a byte load per step and the LDS read behind it did: 0.23 GCUPS per wave).
    int qchunk = 0, qoff = w.pwm ? 0 : ((lane == 0) ? (int)w.q[0] : 0);
    SC snext[R];
#pragma unroll
    for (int r = 0; r < R; ++r) snext[r] = w.S[tc[r] + qoff];

    uint32_t *dirw = w.dirw + (strip * aln_strip_bytes(N)) / 4;
    const uint32_t nkb = (nsteps + SPB - 1) / SPB;
    for (uint32_t kb = 0; kb < nkb; ++kb) {
        uint32_t dw = 0;
#pragma unroll
        for (int kk = 0; kk < SPB; ++kk) {
            const uint32_t k = kb * SPB + kk;
            if ((k & 63u) == 0) {                      // wave-uniform: next 64 columns of the incoming row
                const uint32_t xi = k + 1 + lane;
                if (strip > 0) inchunk = (xi <= N) ? w.brow[xi] : (SC)0;
                if (SEM == ALN_CORE_LOCAL && strip == 0 && w.hazard) advchunk = (xi <= N) ? w.advice[xi] : 0u;
                if (!w.pwm) qchunk = (xi < N) ? (int)w.q[xi] : 0;
            }
            SC top0;
            if (strip == 0) top0 = border_top<SC, SEM>(k + 1, N, del);
            else top0 = O::rdlane(inchunk, (int)(k & 63u));
            const SC topIn = O::shr1(top0, bottom);   // lane 0 <- top0, lane l <- lane l-1's bottom cell
            // cross-lane read kept in wave-uniform control flow (see FastStrip::step)
            const uint32_t adv = (SEM == ALN_CORE_LOCAL && strip == 0)
                                     ? (uint32_t)__builtin_amdgcn_readlane((int)advchunk, (int)(k & 63u)) : 0u;
            const uint32_t xm1 = k - (uint32_t)lane;  // x - 1 (wraps for lanes that have not started)
            SC scur[R];
#pragma unroll
            for (int r = 0; r < R; ++r) scur[r] = snext[r];
            if (w.pwm) qoff = (int)min(xm1 + 1u, N - 1u);                                   // next step's column (lanes not started yet: any valid one)
            else qoff = __builtin_amdgcn_update_dpp(__builtin_amdgcn_readlane(qchunk, (int)(k & 63u)), qoff, 0x138, 0xf, 0xf, false);
#pragma unroll
            for (int r = 0; r < R; ++r) snext[r] = w.S[tc[r] + qoff];
            if (xm1 < N) {
                const uint32_t x = xm1 + 1;
                if constexpr (sizeof(SC) == 8 && is_core<SEM>()) {
                    // row 1 (lane 0 of strip 0) takes its penalty from the advice / the first cell; every other first row from the cell above
                    const bool row1 = (strip == 0 && lane == 0);
                    const bool zr0 = (SEM == ALN_CORE_GLOBAL) ? (row1 && x == 1) : (row1 ? (x == 1 || adv != 0) : (topIn == (SC)0));
                    uint32_t tag0 = 0;
                    cells_f64<SEM, R>(topIn, hdiag, Hl, scur, zr0, -del, -ext, dw, rbv, rbx, x, tag0);
                    if (SEM == ALN_CORE_LOCAL && strip == 0 && w.hazard && lane == 0) { w.row1[x] = Hl[0]; w.row1tag[x] = (uint8_t)tag0; }
                } else {
                SC top = topIn, diag = hdiag;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const uint32_t y = yb + 1 + r;
                        const SC s = scur[r];
                        SC p;
                        if (SEM == ALN_CORE_GLOBAL) {
                            // penalty is `del` for the first visited cell only (simple/mod.rs:72,88-92)
                            p = (r == 0 && y == 1 && x == 1) ? del : ext;
                        } else if (SEM == ALN_CORE_LOCAL) {
                            // carried penalty = del iff the previously visited cell was Beginning (H == 0):
                            // the cell above for y >= 2; the BOTTOM cell of the previous column for y == 1 (advice)
                            p = (top == (SC)0) ? del : ext;
                            if (r == 0 && y == 1) p = (x == 1 || adv != 0) ? del : ext;
                        } else {
                            p = del;
                        }
                        SC h;
                        const uint32_t tag = cell_tag<SC, SEM>(top, Hl[r], diag, s, p, h);
                        diag = Hl[r];
                        Hl[r] = h;
                        top = h;
                        dw = (dw >> 2) | (tag << 30);                           // same packing as v_alignbit in the fast path
                        if (is_local<SEM>()) {
                            const bool upd = (SEM == ALN_CORE_LOCAL) ? (h > rbv[r]) : (h >= rbv[r]);
                            rbv[r] = upd ? h : rbv[r];
                            rbx[r] = upd ? x : rbx[r];
                        }
                        if (SEM == ALN_CORE_LOCAL && r == 0 && strip == 0 && w.hazard && lane == 0) { w.row1[x] = h; w.row1tag[x] = (uint8_t)tag; }
                    }
                }
                if (w.hmat != nullptr) {
#pragma unroll
                    for (int r = 0; r < R; ++r) if (yb + 1 + r <= M) w.hmat[(size_t)(yb + 1 + r) * (N + 1) + x] = Hl[r];
                }
                hdiag = topIn;
                bottom = Hl[R - 1];
                if (!last && lane == 63) w.brow[x] = bottom;          // hand the bottom row to the next strip
                if (SEM == ALN_CORE_LOCAL && last && w.hazard && (uint32_t)lane == lb) {
                    SC hb = Hl[0];
#pragma unroll
                    for (int r = 1; r < R; ++r) if ((uint32_t)r == rb) hb = Hl[r];
                    w.zrow[x] = (hb == (SC)0) ? 1 : 0;
                }
            }
        }
        if (w.store_dirs) dirw[aln_dir_word_index(kb * SPB, (uint32_t)lane, SPB)] = dw;   // quad layout of the fast path (aln_device.h)
    }

    // fold this strip's per-row candidates into the lane's running end-cell candidate
    if (is_local<SEM>()) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t y = yb + 1 + r;
            if (y <= M && rbx[r] != 0 && better<SC, SEM>(rbv[r], y, rbx[r], w.bv, w.by, w.bx)) {
                w.bv = rbv[r]; w.by = y; w.bx = rbx[r];
            }
        }
    }
    if (last) {                                        // H[M][N] lives in lane lb, row rb
        SC hb = Hl[0];
#pragma unroll
        for (int r = 1; r < R; ++r) if ((uint32_t)r == rb) hb = Hl[r];
        w.corner = O::rdlane(hb, (int)lb);
    }
}

// ---------------------------------------------------------------- the lean f64 strip (r03): core semantics, no H dump
// run_strip above carries every option of the generic kernels through its inner loop (H dump, int / f64, legacy semantics, PWM or
// not, first strip or not as run-time flags): 30.6 VALU instructions per cell for f64 in the R = 7 loop (20 in the cell, 74 per
// step around it, a dozen of them reloads of spilled scalars).  Every real-valued batch runs here instead:
//   * the cell is ONE asm block of 15 VALU instructions (core local; 11 core global): three adds, two maxes, the two distances of
//     enums.rs:21-25, their compares against f64::EPSILON and H == 0 as wave masks, the 2-bit tag shifted into the direction word
//     by two v_addc_co_u32 whose carry-in IS the mask (the word is filled from the bottom and bit-reversed once per block: the
//     same bits as v_alignbit from the top), the NEXT row's carried penalty selected from this cell's zero mask, and the end-cell
//     tracker reduced to one compare `h >= best of this lane` whose mask is looked at once per step (the update itself -- the
//     row-major-first rule of `better` -- runs only in the steps that have a candidate);
//   * H, the scores and the incoming cell are double-buffered by step parity (SPB is even), so nothing is moved between steps;
//   * the first-strip / PWM variants are template parameters, the bottom-row zeros of the last strip and row 1's record are
//     taken from the cells' masks in scalar code.
// Hazards the assembler does not see inside asm blocks (gfx940+: a VALU-written SGPR needs two wait states before a VALU reads
// it): every v_cmp result is read by a VALU instruction at least three instructions later, or by SALU (interlocked).  The blocks'
// s_or / s_andn2 write SCC: it is on every clobber list (left off, a scalar compare of the compiler's own was carried across a block and
// a record branch went the wrong way -- seen only as extra advice passes, the results stayed right).
typedef unsigned long long lmask_t;

#define ALN_F64_CELL_HEAD \
    "v_add_f64 %[a], %[top], %[np]\n\t" \
    "v_add_f64 %[b], %[left], %[np]\n\t" \
    "v_add_f64 %[c], %[diag], %[s]\n\t" \
    "v_max_f64 %[m], %[a], %[b]\n\t" \
    "v_max_f64 %[m], %[m], %[c]\n\t"
#define ALN_F64_CELL_LOCAL_TAIL \
    "v_cmp_eq_f64_e64 %[Z], 0, %[m]\n\t" \
    "v_add_f64 %[a], %[m], -%[a]\n\t" \
    "v_add_f64 %[b], %[m], -%[b]\n\t" \
    "v_cmp_gt_f64_e64 %[A], %[eps], %[a]\n\t" \
    "v_cmp_gt_f64_e64 %[B], %[eps], %[b]\n\t" \
    "v_cmp_ge_f64_e64 %[HM], %[m], %[bv]\n\t" \
    "s_andn2_b64 %[B], %[B], %[A]\n\t" \
    "s_or_b64 %[B], %[B], %[Z]\n\t" \
    "s_or_b64 %[A], %[A], %[Z]\n\t" \
    "v_addc_co_u32_e64 %[dw], vcc, %[dw], %[dw], %[B]\n\t" \
    "v_addc_co_u32_e64 %[dw], vcc, %[dw], %[dw], %[A]"
#define ALN_F64_CELL_SELNEXT \
    "\n\tv_cndmask_b32_e64 %[nl], %[nel], %[ndl], %[Z]\n\t" \
    "v_cndmask_b32_e64 %[nh], %[neh], %[ndh], %[Z]"

// one core-local cell.  negp: minus the carried penalty of THIS cell; negp_next: of the cell below (del iff this H == 0,
// simple/mod.rs:201-205).  hm: lanes whose H reaches bv (the caller passes the best of the whole wave).  The tag (3 <=> H == 0) is in dw's two lowest bits, bit 1 first.
template <bool SELNEXT>
__device__ __forceinline__ void cell_f64_local(const double top, const double left, const double diag, const double s, const double negp,
                                               const int ndl, const int ndh, const int nel, const int neh, const lmask_t eps, const double bv,
                                               double &h, uint32_t &dw, lmask_t &hm, double &negp_next)
{
    double a, b, c, m;
    lmask_t A, B, Z, HM;
    if constexpr (SELNEXT) {
        int nl, nh;
        asm(ALN_F64_CELL_HEAD ALN_F64_CELL_LOCAL_TAIL ALN_F64_CELL_SELNEXT
            : [a] "=&v"(a), [b] "=&v"(b), [c] "=&v"(c), [m] "=&v"(m), [dw] "+v"(dw), [Z] "=&s"(Z), [A] "=&s"(A), [B] "=&s"(B), [HM] "=&s"(HM),
              [nl] "=&v"(nl), [nh] "=&v"(nh)
            : [top] "v"(top), [left] "v"(left), [diag] "v"(diag), [s] "v"(s), [np] "v"(negp), [eps] "s"(eps), [bv] "v"(bv),
              [ndl] "v"(ndl), [ndh] "v"(ndh), [nel] "v"(nel), [neh] "v"(neh)
            : "vcc", "scc");
        negp_next = __hiloint2double(nh, nl);
    } else {
        asm(ALN_F64_CELL_HEAD ALN_F64_CELL_LOCAL_TAIL
            : [a] "=&v"(a), [b] "=&v"(b), [c] "=&v"(c), [m] "=&v"(m), [dw] "+v"(dw), [Z] "=&s"(Z), [A] "=&s"(A), [B] "=&s"(B), [HM] "=&s"(HM)
            : [top] "v"(top), [left] "v"(left), [diag] "v"(diag), [s] "v"(s), [np] "v"(negp), [eps] "s"(eps), [bv] "v"(bv)
            : "vcc", "scc");
    }
    h = m; hm = HM;
}

// The same cell software-pipelined over the rows of a lane (what ships): a wave issues in order, and in the block above every cell
// ends with v_cmp -> s_or -> v_addc_co, scalar results going straight back into vector instructions, before the next row's first
// add can issue (measured: 23.4 VALU instructions per cell instead of 30.7 and yet 12 % SLOWER, the VALU 73 % busy instead of 96).
// Here row r's block carries row r-1's tag work between its own dependent instructions: head0 (row 0), mid (rows 1..R-1), tail.
// pa / pb / pm: the previous row's a, b and H; zp its zero mask.  Hazard distances as above (v_cmp Z -> the next block's
// v_cndmask: three instructions; head0 ends in s_nop 1).
__device__ __forceinline__ void cell_f64_head0(const double top, const double left, const double diag, const double s, const double negp,
                                               double &a, double &b, double &h, lmask_t &z)
{
    double c;
    asm(ALN_F64_CELL_HEAD
        "v_cmp_eq_f64_e64 %[Z], 0, %[m]\n\t"
        "s_nop 1"
        : [a] "=&v"(a), [b] "=&v"(b), [c] "=&v"(c), [m] "=&v"(h), [Z] "=&s"(z)
        : [top] "v"(top), [left] "v"(left), [diag] "v"(diag), [s] "v"(s), [np] "v"(negp));
}
__device__ __forceinline__ void cell_f64_mid(const double pm, double &pa, double &pb, const lmask_t zp, const double left, const double diag, const double s,
                                             const int ndl, const int ndh, const int nel, const int neh, const lmask_t eps, const double bv,
                                             double &a, double &b, double &h, lmask_t &z, lmask_t &hm, uint32_t &dw)
{
    double c;
    lmask_t A, B;
    // this row's carried penalty (del iff the cell above is 0, simple/mod.rs:201-205) is a temporary of the block: the halves of a
    // 64-bit operand cannot be named in inline asm, so it lives in a fixed pair
    asm("v_cndmask_b32_e64 v126, %[nel], %[ndl], %[ZP]\n\t"
        "v_cndmask_b32_e64 v127, %[neh], %[ndh], %[ZP]\n\t"
        "v_add_f64 %[pa], %[pm], -%[pa]\n\t"
        "v_add_f64 %[a], %[pm], v[126:127]\n\t"
        "v_add_f64 %[b], %[left], v[126:127]\n\t"
        "v_add_f64 %[pb], %[pm], -%[pb]\n\t"
        "v_add_f64 %[c], %[diag], %[s]\n\t"
        "v_cmp_gt_f64_e64 %[A], %[eps], %[pa]\n\t"
        "v_max_f64 %[m], %[a], %[b]\n\t"
        "v_cmp_gt_f64_e64 %[B], %[eps], %[pb]\n\t"
        "v_cmp_ge_f64_e64 %[HM], %[pm], %[bv]\n\t"
        "v_max_f64 %[m], %[m], %[c]\n\t"
        "s_andn2_b64 %[B], %[B], %[A]\n\t"
        "s_or_b64 %[B], %[B], %[ZP]\n\t"
        "v_cmp_eq_f64_e64 %[Z], 0, %[m]\n\t"
        "s_or_b64 %[A], %[A], %[ZP]\n\t"
        "v_addc_co_u32_e64 %[dw], vcc, %[dw], %[dw], %[B]\n\t"
        "v_addc_co_u32_e64 %[dw], vcc, %[dw], %[dw], %[A]"
        : [a] "=&v"(a), [b] "=&v"(b), [c] "=&v"(c), [m] "=&v"(h), [pa] "+v"(pa), [pb] "+v"(pb), [dw] "+v"(dw), [Z] "=&s"(z), [A] "=&s"(A), [B] "=&s"(B),
          [HM] "=&s"(hm)
        : [pm] "v"(pm), [left] "v"(left), [diag] "v"(diag), [s] "v"(s), [ZP] "s"(zp), [eps] "s"(eps), [bv] "v"(bv),
          [ndl] "v"(ndl), [ndh] "v"(ndh), [nel] "v"(nel), [neh] "v"(neh)
        : "vcc", "scc", "v126", "v127");
}
__device__ __forceinline__ void cell_f64_tail(const double pm, double &pa, double &pb, const lmask_t zp, const lmask_t eps, const double bv,
                                              lmask_t &hm, uint32_t &dw)
{
    lmask_t A, B;
    asm("v_add_f64 %[pa], %[pm], -%[pa]\n\t"
        "v_add_f64 %[pb], %[pm], -%[pb]\n\t"
        "v_cmp_ge_f64_e64 %[HM], %[pm], %[bv]\n\t"
        "v_cmp_gt_f64_e64 %[A], %[eps], %[pa]\n\t"
        "v_cmp_gt_f64_e64 %[B], %[eps], %[pb]\n\t"
        "s_andn2_b64 %[B], %[B], %[A]\n\t"
        "s_or_b64 %[B], %[B], %[ZP]\n\t"
        "s_or_b64 %[A], %[A], %[ZP]\n\t"
        "v_addc_co_u32_e64 %[dw], vcc, %[dw], %[dw], %[B]\n\t"
        "v_addc_co_u32_e64 %[dw], vcc, %[dw], %[dw], %[A]"
        : [pa] "+v"(pa), [pb] "+v"(pb), [dw] "+v"(dw), [A] "=&s"(A), [B] "=&s"(B), [HM] "=&s"(hm)
        : [pm] "v"(pm), [ZP] "s"(zp), [eps] "s"(eps), [bv] "v"(bv)
        : "vcc", "scc");
}
// one core-global cell: no Beginning, no end-cell tracker (simple/mod.rs:72-97)
__device__ __forceinline__ void cell_f64_global(const double top, const double left, const double diag, const double s, const double negp,
                                                const lmask_t eps, double &h, uint32_t &dw)
{
    double a, b, c, m;
    lmask_t A, B;
    asm(ALN_F64_CELL_HEAD
        "v_add_f64 %[a], %[m], -%[a]\n\t"
        "v_add_f64 %[b], %[m], -%[b]\n\t"
        "v_cmp_gt_f64_e64 %[A], %[eps], %[a]\n\t"
        "v_cmp_gt_f64_e64 %[B], %[eps], %[b]\n\t"
        "s_andn2_b64 %[B], %[B], %[A]\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32_e64 %[dw], vcc, %[dw], %[dw], %[B]\n\t"
        "v_addc_co_u32_e64 %[dw], vcc, %[dw], %[dw], %[A]"
        : [a] "=&v"(a), [b] "=&v"(b), [c] "=&v"(c), [m] "=&v"(m), [dw] "+v"(dw), [A] "=&s"(A), [B] "=&s"(B)
        : [top] "v"(top), [left] "v"(left), [diag] "v"(diag), [s] "v"(s), [np] "v"(negp), [eps] "s"(eps)
        : "vcc", "scc");
    h = m;
}

// one wave step.  Hi / Ho: the lane's R cells of the previous / this column; si / so: this step's scores / the next step's
// (read from LDS while this one computes); tprev / tcur: the cell above the lane's first row, previous / this column.
template <int SEM, int R, bool FIRST, bool PWM>
__device__ __forceinline__ void f64_step(Wave<double> &w, const uint32_t k, const double (&Hi)[R], double (&Ho)[R], const double (&si)[R], double (&so)[R],
                                     const double tprev, double &tcur, const int (&tc)[R], double &inchunk, uint32_t &advchunk, int &qchunk, int &qoff,
                                     uint32_t &dw, const uint32_t yb, const uint32_t lb, const uint32_t rb, const double nd, const double ne,
                                     const int ndl, const int ndh, const int nel, const int neh, const lmask_t eps, const bool last)
{
    const uint32_t N = w.N, M = w.M;
    const int lane = w.lane;
    if ((k & 63u) == 0) {                          // wave-uniform: next 64 columns of the incoming row, of the advice, of the query
        const uint32_t xi = k + 1 + (uint32_t)lane;
        if (!FIRST) inchunk = (xi <= N) ? w.brow[xi] : 0.0;
        if (SEM == ALN_CORE_LOCAL && FIRST && w.hazard) advchunk = (xi <= N) ? w.advice[xi] : 0u;
        if (!PWM) qchunk = (xi < N) ? (int)w.q[xi] : 0;
    }
    const double top0 = FIRST ? border_top<double, SEM>(k + 1, N, w.del) : ScOps<double>::rdlane(inchunk, (int)(k & 63u));
    const double topIn = ScOps<double>::shr1(top0, Hi[R - 1]);
    const uint32_t adv = (SEM == ALN_CORE_LOCAL && FIRST) ? (uint32_t)__builtin_amdgcn_readlane((int)advchunk, (int)(k & 63u)) : 0u;
    const uint32_t xm1 = k - (uint32_t)lane;
    if (PWM) qoff = (int)min(xm1 + 1u, N - 1u);
    else qoff = __builtin_amdgcn_update_dpp(__builtin_amdgcn_readlane(qchunk, (int)(k & 63u)), qoff, 0x138, 0xf, 0xf, false);
#pragma unroll
    for (int r = 0; r < R; ++r) so[r] = w.S[tc[r] + qoff];
    lmask_t hit = 0;
    if (xm1 < N) {
        const bool row1 = FIRST && lane == 0;
        if constexpr (SEM == ALN_CORE_LOCAL) {
            const bool zr0 = row1 ? (k == 0 || adv != 0) : (topIn == 0.0);
            const double negp = zr0 ? nd : ne;
            lmask_t zp, hm;
            double pa, pb;
            cell_f64_head0(topIn, Hi[0], tprev, si[0], negp, pa, pb, Ho[0], zp);
#pragma unroll
            for (int r = 1; r < R; ++r) {
                double na, nb;
                lmask_t zn;
                cell_f64_mid(Ho[r - 1], pa, pb, zp, Hi[r], Hi[r - 1], si[r], ndl, ndh, nel, neh, eps, w.bt, na, nb, Ho[r], zn, hm, dw);
                hit |= hm;
                pa = na; pb = nb; zp = zn;
            }
            cell_f64_tail(Ho[R - 1], pa, pb, zp, eps, w.bt, hm, dw);
            hit |= hm;
            // this step's tags sit in dw's low 2 R bits, row r at bits 2 (R-1-r) .. +1 as (b1, b0) from the bottom: 3 <=> H == 0
            if (FIRST && w.hazard && k < N) {      // row 1 as this pass computed it (adopt_advice_checked); lane 0's x is k + 1
                const uint32_t t = (dw >> (2 * (R - 1))) & 3u;
                if (lane == 0) { w.row1[k + 1] = Ho[0]; w.row1tag[k + 1] = (uint8_t)(((t & 1u) << 1) | (t >> 1)); }
            }
            if (last && w.hazard && k - lb < N) {  // bottom-row zeros: row rb of lane lb, whose x is k - lb + 1
                if ((uint32_t)lane == lb) w.zrow[k - lb + 1] = (uint8_t)((((dw >> (2u * ((uint32_t)R - 1u - rb))) & 3u) == 3u) ? 1 : 0);
            }
        } else {
            double negp = (row1 && k == 0) ? nd : ne, top = topIn;      // `del` for the first visited cell only (simple/mod.rs:72,88-92)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                cell_f64_global(top, Hi[r], r ? Hi[r - 1] : tprev, si[r], negp, eps, Ho[r], dw);
                negp = ne;
                top = Ho[r];
            }
        }
        if (!last && k - 63u < N) {                // hand the bottom row to the next strip: lane 63's x is k - 62
            if (lane == 63) w.brow[k - 62u] = Ho[R - 1];
        }
    }
    if (SEM == ALN_CORE_LOCAL && hit != 0) {       // wave-uniform: some cell reaches the best the wave has seen (rare: records and ties).
        const uint32_t x = xm1 + 1;                // Outside the active-lane region: the reduction below is over all 64 lanes.
        const bool act = xm1 < N;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (__any(act && Ho[r] >= w.bt)) {
                const uint32_t y = yb + 1 + (uint32_t)r;
                // first maximum in row-major order, simple/mod.rs:212; every cell that has the final maximum passes here
                if (act && y <= M && Ho[r] >= w.bt && (w.bx == 0 || better<double, SEM>(Ho[r], y, x, w.bv, w.by, w.bx))) { w.bv = Ho[r]; w.by = y; w.bx = x; }
            }
        }
        double t = w.bv;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) t = max_f64_raw(t, ScOps<double>::xshfl(t, m));
        w.bt = max_f64_raw(w.bt, t);
    }
    tcur = topIn;
}

template <int SEM, int R, bool FIRST, bool PWM>
__device__ __forceinline__ void f64_strip(Wave<double> &w, const uint32_t strip, const bool last)
{
    const uint32_t N = w.N, M = w.M;
    const int lane = w.lane;
    int tc[R];
    double inchunk;
    uint32_t advchunk, dw;
    int qchunk, qoff;
    constexpr int SPB = (int)aln_spb(R);
    static_assert(SPB % 2 == 0, "steps are double-buffered in pairs");
    const uint32_t y0 = strip * ALN_STRIP_ROWS;
    const uint32_t rows = min(M - y0, (uint32_t)(64 * R));
    const uint32_t L = (rows + R - 1) / R;
    const uint32_t nsteps = N + L - 1;
    const uint32_t yb = y0 + (uint32_t)lane * R;
    const uint32_t lb = (rows - 1) / R, rb = (rows - 1) % R;
    double nd = -w.del, ne = -w.ext;
    // the penalties live in vector registers for the whole strip (left alone the compiler rebuilds them from scalars every step)
    int ndl = __double2loint(nd), ndh = __double2hiint(nd), nel = __double2loint(ne), neh = __double2hiint(ne);
    asm volatile("" : "+v"(ndl), "+v"(ndh), "+v"(nel), "+v"(neh));
    nd = __hiloint2double(ndh, ndl); ne = __hiloint2double(neh, nel);
    const lmask_t eps = (lmask_t)__double_as_longlong(DBL_EPSILON);
    double HA[R], HB[R], SA[R], SB[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t y = yb + 1 + r;
        tc[r] = (y <= M) ? (int)w.t[y - 1] * (int)w.cols : 0;
        HA[r] = HB[r] = border_left<double, SEM>(y, M, w.del);
    }
    double tA = border_left<double, SEM>(yb, M, w.del), tB = tA;      // H[yb][0]
    inchunk = 0.0; advchunk = 0; qchunk = 0;
    qoff = PWM ? 0 : ((lane == 0) ? (int)w.q[0] : 0);
#pragma unroll
    for (int r = 0; r < R; ++r) { SA[r] = w.S[tc[r] + qoff]; SB[r] = SA[r]; }
    uint32_t *dirw = w.dirw + (strip * aln_strip_bytes(N)) / 4;
    const uint32_t nkb = (nsteps + SPB - 1) / SPB;
    for (uint32_t kb = 0; kb < nkb; ++kb) {
        dw = 0;
#pragma unroll
        for (int kk = 0; kk < SPB; kk += 2) {
            f64_step<SEM, R, FIRST, PWM>(w, kb * SPB + kk, HA, HB, SA, SB, tA, tB, tc, inchunk, advchunk, qchunk, qoff, dw, yb, lb, rb, nd, ne, ndl, ndh, nel, neh, eps, last);
            f64_step<SEM, R, FIRST, PWM>(w, kb * SPB + kk + 1, HB, HA, SB, SA, tB, tA, tc, inchunk, advchunk, qchunk, qoff, dw, yb, lb, rb, nd, ne, ndl, ndh, nel, neh, eps, last);
        }
        if (w.store_dirs) dirw[aln_dir_word_index(kb * SPB, (uint32_t)lane, SPB)] = __builtin_bitreverse32(dw);
    }
    if (last) {                                    // H[M][N]: lane lb, row rb, written by that lane's last step lb + N - 1
        const bool inB = ((lb + N - 1) & 1u) == 0;  // even steps write HB
        double hb = inB ? HB[0] : HA[0];
#pragma unroll
        for (int r = 1; r < R; ++r) if ((uint32_t)r == rb) hb = inB ? HB[r] : HA[r];
        w.corner = ScOps<double>::rdlane(hb, (int)lb);
    }
}

template <int SEM, int R>
__device__ __forceinline__ void run_strip_f64(Wave<double> &w, const uint32_t strip, const bool last)
{
    if (strip == 0) {
        if (w.pwm) f64_strip<SEM, R, true, true>(w, strip, last);
        else f64_strip<SEM, R, true, false>(w, strip, last);
    } else {
        if (w.pwm) f64_strip<SEM, R, false, true>(w, strip, last);
        else f64_strip<SEM, R, false, false>(w, strip, last);
    }
}

// ---------------------------------------------------------------- strict reference order, one lane
// Exact for every input (it IS the reference's loop nest); used when the speculative fills do not converge
// and on request (aln_params.force_serial).  Directions go to the row-major layout.
template <typename SC, int SEM>
__device__ __noinline__ void serial_fill_impl(Wave<SC> &w)
{
    const uint32_t N = w.N, M = w.M;
    const SC del = w.del, ext = w.ext;
    SC *col = w.brow;                                  // H[.][x-1] on entry of column x, updated in place
    uint8_t *dirs = reinterpret_cast<uint8_t *>(w.dirw);
    const uint32_t rowbytes = (N + 4) / 4;
    for (uint32_t y = 0; y <= M; ++y) col[y] = border_left<SC, SEM>(y, M, del);
    SC p = del;
    SC bv = (SEM == ALN_LEGACY_LOCAL) ? (SC)-1 : ScOps<SC>::lowest();
    uint32_t by = 0, bx = 0;
    for (uint32_t x = 1; x <= N; ++x) {
        const int qc = w.pwm ? (int)(x - 1) : (int)w.q[x - 1];
        SC diag = col[0];
        col[0] = border_top<SC, SEM>(x, N, del);
        SC top = col[0];
        for (uint32_t y = 1; y <= M; ++y) {
            const SC left = col[y];
            const SC s = w.S[(int)w.t[y - 1] * (int)w.cols + qc];
            if (!is_core<SEM>()) p = del;
            SC h;
            int d;
            cell<SC, SEM>(top, left, diag, s, p, h, d);
            if (is_core<SEM>()) p = (d != D_BEG) ? ext : del;          // simple/mod.rs:88-92, :201-205
            diag = left;
            col[y] = h;
            top = h;
            uint8_t *bp = dirs + (size_t)y * rowbytes + (x >> 2);
            const uint32_t sh = 2 * (x & 3u);
            if (w.store_dirs) {
                uint8_t old = (x == 1 || (x & 3u) == 0) ? 0 : *bp;
                *bp = (uint8_t)(old | (d << sh));
            }
            if (is_local<SEM>() && better<SC, SEM>(h, y, x, bv, by, bx)) { bv = h; by = y; bx = x; }
            if (w.hmat != nullptr) w.hmat[(size_t)y * (N + 1) + x] = h;
        }
    }
    w.bv = bv; w.by = by; w.bx = bx;
    w.corner = col[M];
}

#include "aln_fast.h"

// The per-wave state must stay in registers on the hot path: the out-of-line serial routine gets its own copy so the
// caller's Wave object never has its address taken.
template <typename SC, int SEM>
__device__ __forceinline__ void serial_fill(Wave<SC> &w)
{
    Wave<SC> c;
    c.lane = 0; c.N = w.N; c.M = w.M; c.q = w.q; c.t = w.t; c.S = w.S; c.cols = w.cols; c.del = w.del; c.ext = w.ext;
    c.dirw = w.dirw; c.brow = w.brow; c.hmat = w.hmat; c.store_dirs = w.store_dirs; c.pwm = w.pwm;
    serial_fill_impl<SC, SEM>(c);
    w.bv = c.bv; w.by = c.by; w.bx = c.bx; w.corner = c.corner;
}

// adopts the observed bottom-row zeros as the new row-1 advice; true when nothing changed (self-consistent fill)
__device__ __forceinline__ bool adopt_advice(uint8_t *advice, const uint8_t *zrow, uint32_t N, int lane, uint32_t &last_flip)
{
    int mismatch = 0;
    uint32_t lf = 0;
    for (uint32_t x = 2 + lane; x <= N; x += 64) {
        const uint8_t z = zrow[x - 1];
        if (advice[x] != z) { mismatch = 1; advice[x] = z; }
        if (z != 0) lf = x;                          // advice differs from the all-"ext" first pass here
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) lf = max(lf, (uint32_t)__shfl_xor((int)lf, m));
    last_flip = lf;
    __threadfence_block();
    return !__any(mismatch);
}

// The same, and is the pass already the answer?  Only row 1's penalties depend on the advice.  If, at every column whose advice
// bit changes, cell (1, x) recomputed with the new penalty -- from the pass's own H[1][x-1] -- keeps its value and its direction,
// then (induction over the reference's visiting order) the fill under the new advice is cell for cell the fill just made: its
// bottom row is the observed one, the new advice is self-consistent, and no second pass is needed.  With real-valued scores
// the bottom row's exact zeros fall anywhere and nearly every pair changes some bit; two thirds of them change no cell.
template <typename SC, int SEM>
__device__ __forceinline__ bool adopt_advice_checked(uint8_t *advice, const uint8_t *zrow, const SC *row1, const uint8_t *row1tag, const SC *S,
                                                     uint32_t cols, const uint8_t *q, const uint8_t *t, bool pwm, SC del, SC ext, uint32_t N, int tid,
                                                     int nthreads, int &same_cells)
{
    int mismatch = 0, moved = 0;
    const int t0 = (int)t[0] * (int)cols;
    for (uint32_t x = 2 + (uint32_t)tid; x <= N; x += (uint32_t)nthreads) {
        const uint8_t z = zrow[x - 1];
        if (advice[x] != z) {
            mismatch = 1;
            SC h;
            const uint32_t tag = cell_tag<SC, SEM>((SC)0, row1[x - 1], (SC)0, S[t0 + (pwm ? (int)(x - 1) : (int)q[x - 1])], z ? del : ext, h);
            if (!(h == row1[x]) || tag != (uint32_t)row1tag[x]) moved = 1;
            advice[x] = z;
        }
    }
    same_cells = !moved;
    return !mismatch;
}

// the same for the fast path, whose last strip records the direction words of the lane that owns row M (one per block):
// H[M][x] == 0 <=> tag 3.  Ru = 0: the skewed layout (512-row strips, the last one by aln_pick_r); else uniform strips of 64 Ru rows.
// The record is a row of granules {direction word, tag of the strip that wrote it}: in a cooperative pass that was another wave.
// bad: a granule never showed the expected tag (cannot be: the strip's candidate, stored after its record has drained, has been seen).
__device__ __forceinline__ bool adopt_advice_zdw(uint8_t *advice, const unsigned long long *zdw, uint32_t ztag, uint32_t N, uint32_t M, uint32_t Ru,
                                                 int lane, uint32_t &last_flip, bool &bad)
{
    const uint32_t srows = Ru ? 64u * Ru : (uint32_t)ALN_STRIP_ROWS;
    const uint32_t ns = (M + srows - 1u) / srows, rows_last = M - (ns - 1) * srows;
    const uint32_t R = Ru ? Ru : (uint32_t)aln_pick_r(rows_last), lb = (rows_last - 1) / R, rb = (rows_last - 1) % R, spb = aln_spb(R);
    int mismatch = 0, lost = 0;
    uint32_t lf = 0;
    for (uint32_t x = 2 + lane; x <= N; x += 64) {
        const uint32_t k = x - 2 + lb;                                  // wave step of cell (M, x - 1)
        unsigned long long g = __hip_atomic_load(zdw + k / spb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (uint32_t spins = 0; (uint32_t)(g >> 32) != ztag && spins < 100000u; ++spins) {
            __builtin_amdgcn_s_sleep(2);
            g = __hip_atomic_load(zdw + k / spb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if ((uint32_t)(g >> 32) != ztag) lost = 1;
        const uint8_t z = (((uint32_t)g >> aln_dir_bitpos(k, rb, lb, N, (int)R)) & 3u) == 3u ? 1 : 0;
        if (advice[x] != z) { mismatch = 1; advice[x] = z; }
        if (z != 0) lf = x;
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) lf = max(lf, (uint32_t)__shfl_xor((int)lf, m));
    last_flip = lf;
    bad = __any(lost) != 0;
    __threadfence_block();
    return !__any(mismatch);
}

// writes the fill-side half of the pair's summary (the traceback kernel adds start cell and length)
template <int SEM>
__device__ __forceinline__ void write_result(aln_pair_result &res, double best, uint32_t by, uint32_t bx, double corner,
                                             uint32_t N, uint32_t M, uint32_t passes, uint32_t flags, bool pwm = false)
{
    res.passes = passes;
    res.flags = flags;
    res.start_y = res.start_x = 0;
    res.aln_len = 0;
    res.status = ALN_OK;
    if (is_local<SEM>()) {
        res.score = best;
        res.f = best;
        res.end_y = by; res.end_x = bx;
        // core local: the argmax runs over the zero borders too; a non-positive maximum sits on (0,0) (simple/mod.rs:212-215)
        if (SEM == ALN_CORE_LOCAL && !(best > 0.0)) {
            // PWMAligner has no seed pair: an all-non-positive matrix is an empty alignment with f = 0 (pwm/mod.rs:76-108)
            res.status = pwm ? ALN_OK : ALN_ERR_NO_POSITIVE_CELL;
            res.end_y = res.end_x = 0; res.score = res.f = 0.0;
        }
    } else {
        res.score = corner;
        res.f = (SEM == ALN_CORE_GLOBAL) ? 0.0 : corner;       // simple/mod.rs:139
        res.end_y = M; res.end_x = N;
    }
}

// ---------------------------------------------------------------- one pair, generic kernels (int32 without the profile, f64)
template <typename SC, int SEM, bool LEAN = false>
__device__ __forceinline__ void do_pair(Wave<SC> &w, const FillArgs &a, PairDesc &desc, aln_pair_result &res)
{
    using O = ScOps<SC>;
    const int lane = w.lane;
    const uint32_t N = desc.N, M = desc.M;
    w.N = N; w.M = M;
    w.q = a.seqs + desc.q_off;
    w.t = a.seqs + desc.t_off;
    w.dirw = reinterpret_cast<uint32_t *>(a.dirs + desc.dir_off);
    w.hmat = a.hmat ? reinterpret_cast<SC *>(a.hmat) + desc.h_off : nullptr;
    w.hazard = (SEM == ALN_CORE_LOCAL) && (w.del != w.ext) && N >= 2;
    w.store_dirs = a.store_dirs != 0;
    w.pwm = a.pwm != 0;
    if (w.hmat != nullptr) {   // borders of the optional H dump (simple/mod.rs:55-70)
        for (uint32_t x = lane; x <= N; x += 64) w.hmat[x] = border_top<SC, SEM>(x, N, w.del);
        for (uint32_t y = lane; y <= M; y += 64) w.hmat[(size_t)y * (N + 1)] = border_left<SC, SEM>(y, M, w.del);
    }
    if (w.hazard)
        for (uint32_t x = lane; x <= N + 1; x += 64) { w.advice[x] = 0; w.zrow[x] = 0; }
    // lanes exchange advice / boundary rows through the wave's scratch: make the stores above (and those of the
    // previous pair) visible before any lane loads them (s_waitcnt vmcnt(0); same-CU L1 is coherent)
    __threadfence_block();

    uint32_t passes = 0;
    bool converged = false;
    const uint32_t max_passes = a.max_passes ? a.max_passes : 4u;
    if (!a.force_serial) {
        const uint32_t ns = aln_num_strips(M);
        do {
            w.bv = (SEM == ALN_LEGACY_LOCAL) ? (SC)-1 : O::lowest();
            w.by = 0; w.bx = 0;
            w.bt = w.bv;
            for (uint32_t s = 0; s < ns; ++s) {
                const bool last = (s + 1 == ns);
                if (s > 0) __threadfence_block();      // strip s reads the boundary row strip s-1 stored
                const int R = last ? aln_pick_r(M - s * ALN_STRIP_ROWS) : ALN_FULL_R;
                if constexpr (LEAN) {                  // aln_fill_f64_kernel: the lean f64 strip only
                    {
                        switch (R) {
                        case 1: run_strip_f64<SEM, 1>(w, s, last); break;
                        case 2: run_strip_f64<SEM, 2>(w, s, last); break;
                        case 3: run_strip_f64<SEM, 3>(w, s, last); break;
                        case 4: run_strip_f64<SEM, 4>(w, s, last); break;
                        case 5: run_strip_f64<SEM, 5>(w, s, last); break;
                        case 6: run_strip_f64<SEM, 6>(w, s, last); break;
                        case 7: run_strip_f64<SEM, 7>(w, s, last); break;
                        default: run_strip_f64<SEM, 8>(w, s, last); break;
                        }
                        continue;
                    }
                }
                switch (R) {
                case 1: run_strip<SC, SEM, 1>(w, s, last); break;
                case 2: run_strip<SC, SEM, 2>(w, s, last); break;
                case 3: run_strip<SC, SEM, 3>(w, s, last); break;
                case 4: run_strip<SC, SEM, 4>(w, s, last); break;
                case 5: run_strip<SC, SEM, 5>(w, s, last); break;
                case 6: run_strip<SC, SEM, 6>(w, s, last); break;
                case 7: run_strip<SC, SEM, 7>(w, s, last); break;
                default: run_strip<SC, SEM, 8>(w, s, last); break;
                }
            }
            ++passes;
            __threadfence_block();
            if (!w.hazard) { converged = true; break; }
            int same = 0;
            const bool unchanged = adopt_advice_checked<SC, SEM>(w.advice, w.zrow, w.row1, w.row1tag, w.S, w.cols, w.q, w.t, w.pwm, w.del, w.ext,
                                                                 N, lane, 64, same);
            __threadfence_block();
            converged = !__any(!unchanged) || !__any(!same);
        } while (!converged && passes < max_passes);
    }
    uint32_t layout = ALN_LAYOUT_SKEW;
    if (!converged) {
        if (lane == 0) serial_fill<SC, SEM>(w);
        __threadfence_block();
        w.bv = O::rdlane(w.bv, 0);
        w.by = __builtin_amdgcn_readlane((int)w.by, 0);
        w.bx = __builtin_amdgcn_readlane((int)w.bx, 0);
        w.corner = O::rdlane(w.corner, 0);
        layout = ALN_LAYOUT_ROWMAJOR;
        passes |= 0x80u;
    } else if (is_local<SEM>()) {
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const SC ov = O::xshfl(w.bv, m);
            const uint32_t oy = (uint32_t)__shfl_xor((int)w.by, m), ox = (uint32_t)__shfl_xor((int)w.bx, m);
            if (ox != 0 && (w.bx == 0 || better<SC, SEM>(ov, oy, ox, w.bv, w.by, w.bx))) { w.bv = ov; w.by = oy; w.bx = ox; }
        }
    }
    if (lane == 0) {
        desc.layout = layout;
        write_result<SEM>(res, (double)w.bv, w.by, w.bx, (double)w.corner, N, M, passes, sizeof(SC) == 4 ? 1u : 0u, w.pwm);
    }
}

// ---------------------------------------------------------------- one pair, generic kernels, one WORKGROUP per pair
// The generic kernels above give a pair one wave: 4.4 ms for a 1000 x 1000 pair with a real-valued matrix (every iteration of
// HeuristicAligner, heuristic/mod.rs:58-77, is such a call).  Here wave s of one workgroup owns strip s (64 R rows) and the
// strips run as a pipeline: the bottom row of strip s travels to strip s+1 through an LDS ring, 16 columns at a time
// (prod[s] = columns published, cons[s] = the step whose 16 columns strip s has taken; both in LDS, workgroup-scope
// acquire / release).  Everything else -- cell, penalty rule, advice passes, end cell -- is run_strip's; the directions go
// to the uniform-R layout, so the parallel traceback of the single-pair route applies.  Every wait is bounded and raises the
// workgroup's abort flag; an aborted or non-converging fill ends in the strict-order routine like the other kernels.
template <typename SC> struct WgShared {
    const SC *S;
    SC *rings;
    uint8_t *advice, *zrow, *row1tag;
    SC *row1;                           // global scratch: H[1][x] of the pass
    uint32_t *prod, *cons, *flags;      // flags[0] abort, flags[1] advice mismatch
};

template <typename SC, int SEM, int R>
__device__ __forceinline__ void wg_strip(const WgArgs &a, const PairDesc &d, const WgShared<SC> &sh, const uint32_t strip, const bool last,
                                         const bool hazard, const int lane, SC &bv, uint32_t &by, uint32_t &bx, SC &corner)
{
    using O = ScOps<SC>;
    constexpr int SPB = (int)aln_spb(R);
    const uint32_t N = d.N, M = d.M;
    const uint8_t *q = a.seqs + d.q_off, *t = a.seqs + d.t_off;
    const uint32_t y0 = strip * 64u * R;
    const uint32_t rows = min(M - y0, (uint32_t)(64 * R));
    const uint32_t L = (rows + R - 1) / R;
    const uint32_t nsteps = last ? N + L - 1 : N + 63;
    const uint32_t yb = y0 + (uint32_t)lane * R;
    const uint32_t lb = (rows - 1) / R, rb = (rows - 1) % R;
    const SC del = O::from_double(a.del), ext = O::from_double(a.ext);
    SC *ring_in = sh.rings + (size_t)(strip ? strip - 1 : 0) * ALN_WG_RING, *ring_out = sh.rings + (size_t)strip * ALN_WG_RING;
    SC *hmat = a.hmat ? reinterpret_cast<SC *>(a.hmat) + d.h_off : nullptr;

    int tc[R];
    SC Hl[R], rbv[R];
    uint32_t rbx[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t y = yb + 1 + r;
        tc[r] = (y <= M) ? (int)t[y - 1] * (int)a.cols : 0;
        Hl[r] = border_left<SC, SEM>(y, M, del);
        rbv[r] = (SEM == ALN_LEGACY_LOCAL) ? (SC)-1 : O::lowest();
        rbx[r] = 0;
    }
    SC hdiag = border_left<SC, SEM>(yb, M, del);
    SC bottom = Hl[R - 1];
    SC inchunk = (SC)0;
    uint32_t advchunk = 0;
    bool dead = false;                                 // this wave gave up waiting (the abort flag is up)
    // the query code travels down the lanes like the boundary cell (lane 0 takes column k + 1's at step k), and the scores of
    // the NEXT step are read from LDS while this one computes: no memory access sits in a step's dependency chain
    const bool pwm = a.pwm != 0;                       // PWM scoring: the "code" is the column index itself
    int qchunk = 0, qoff = (lane == 0 && !pwm) ? (int)q[0] : 0;
    SC snext[R];
#pragma unroll
    for (int r = 0; r < R; ++r) snext[r] = sh.S[tc[r] + qoff];

    uint32_t *dirw = reinterpret_cast<uint32_t *>(a.dirs + d.dir_off + (uint64_t)strip * aln_uniform_strip_bytes(N, R));
    const uint32_t nkb = (nsteps + SPB - 1) / SPB;
    for (uint32_t kb = 0; kb < nkb && !dead; ++kb) {
        uint32_t dw = 0;
#pragma unroll
        for (int kk = 0; kk < SPB; ++kk) {
            const uint32_t k = kb * SPB + kk;
            if ((k & 15u) == 0) {                      // wave-uniform: the next 16 columns of the row above, lane j <- column k + 1 + j
                const uint32_t xi = k + 1 + (uint32_t)lane;
                if (strip > 0) {
                    const uint32_t need = min(N, k + 16u);
                    uint32_t spins = 0;
                    while (__hip_atomic_load(sh.prod + strip - 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > (1u << 22) || __hip_atomic_load(sh.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) { dead = true; break; }
                    }
                    inchunk = (lane < 16 && xi <= N && !dead) ? ring_in[xi & (ALN_WG_RING - 1u)] : (SC)0;
                    if (lane == 0) __hip_atomic_store(sh.cons + strip, k + 16u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);   // columns <= k + 16 are in registers
                }
                if (SEM == ALN_CORE_LOCAL && strip == 0 && hazard) advchunk = (lane < 16 && xi <= N) ? sh.advice[xi] : 0u;
                qchunk = (!pwm && lane < 16 && xi < N) ? (int)q[xi] : 0;
                if (!last && k + 16u > 62u + ALN_WG_RING) {
                    // the next 16 steps write columns up to k - 46 into slots the strip below must have emptied
                    const uint32_t x_max = k + 16u - 62u;
                    uint32_t spins = 0;
                    while (!dead && __hip_atomic_load(sh.cons + strip + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) + ALN_WG_RING < x_max) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > (1u << 22) || __hip_atomic_load(sh.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) dead = true;
                    }
                }
            }
            SC top0;
            if (strip == 0) top0 = border_top<SC, SEM>(k + 1, N, del);
            else top0 = O::rdlane(inchunk, (int)(k & 15u));
            const SC topIn = O::shr1(top0, bottom);
            const uint32_t adv = (SEM == ALN_CORE_LOCAL && strip == 0)
                                     ? (uint32_t)__builtin_amdgcn_readlane((int)advchunk, (int)(k & 15u)) : 0u;
            const uint32_t xm1 = k - (uint32_t)lane;
            SC scur[R];
#pragma unroll
            for (int r = 0; r < R; ++r) scur[r] = snext[r];
            qoff = __builtin_amdgcn_update_dpp(__builtin_amdgcn_readlane(qchunk, (int)(k & 15u)), qoff, 0x138, 0xf, 0xf, false);
            if (pwm) qoff = (int)min(xm1 + 1u, N - 1u);          // next step's column (a lane that has not started: any valid one)
#pragma unroll
            for (int r = 0; r < R; ++r) snext[r] = sh.S[tc[r] + qoff];
            if (xm1 < N) {
                const uint32_t x = xm1 + 1;
                if constexpr (sizeof(SC) == 8 && is_core<SEM>()) {
                    const bool row1 = (strip == 0 && lane == 0);
                    const bool zr0 = (SEM == ALN_CORE_GLOBAL) ? (row1 && x == 1) : (row1 ? (x == 1 || adv != 0) : (topIn == (SC)0));
                    uint32_t tag0 = 0;
                    cells_f64<SEM, R>(topIn, hdiag, Hl, scur, zr0, -del, -ext, dw, rbv, rbx, x, tag0);
                    if (SEM == ALN_CORE_LOCAL && strip == 0 && hazard && lane == 0) { sh.row1[x] = Hl[0]; sh.row1tag[x] = (uint8_t)tag0; }
                } else {
                SC top = topIn, diag = hdiag;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const uint32_t y = yb + 1 + r;
                        const SC sc = scur[r];
                        SC p;
                        if (SEM == ALN_CORE_GLOBAL) p = (r == 0 && y == 1 && x == 1) ? del : ext;
                        else if (SEM == ALN_CORE_LOCAL) {
                            p = (top == (SC)0) ? del : ext;
                            if (r == 0 && y == 1) p = (x == 1 || adv != 0) ? del : ext;
                        } else p = del;
                        SC h;
                        const uint32_t tag = cell_tag<SC, SEM>(top, Hl[r], diag, sc, p, h);
                        diag = Hl[r];
                        Hl[r] = h;
                        top = h;
                        dw = (dw >> 2) | (tag << 30);
                        if (is_local<SEM>()) {           // (a branch here: the CU is issue-bound with its 8+ waves, and most cells update nothing)
                            const bool upd = (SEM == ALN_CORE_LOCAL) ? (h > rbv[r]) : (h >= rbv[r]);
                            if (upd) { rbv[r] = h; rbx[r] = x; }
                        }
                        if (SEM == ALN_CORE_LOCAL && r == 0 && strip == 0 && hazard && lane == 0) { sh.row1[x] = h; sh.row1tag[x] = (uint8_t)tag; }
                    }
                }
                if (hmat != nullptr) {
#pragma unroll
                    for (int r = 0; r < R; ++r) if (yb + 1 + r <= M) hmat[(size_t)(yb + 1 + r) * (N + 1) + x] = Hl[r];
                }
                hdiag = topIn;
                bottom = Hl[R - 1];
                if (!last && lane == 63) ring_out[x & (ALN_WG_RING - 1u)] = bottom;
                if (SEM == ALN_CORE_LOCAL && last && hazard && (uint32_t)lane == lb) {
                    SC hb = Hl[0];
#pragma unroll
                    for (int r = 1; r < R; ++r) if ((uint32_t)r == rb) hb = Hl[r];
                    sh.zrow[x] = (hb == (SC)0) ? 1 : 0;
                }
            }
            // lane 63 has finished column k - 62: every 16 columns (and at the last one) they are published
            if (!last && k >= 62u) {
                const uint32_t done = k - 62u;
                if (done <= N && ((done & 15u) == 0 || done == N) && done != 0 && lane == 63)
                    __hip_atomic_store(sh.prod + strip, done, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        if (a.store_dirs && !dead) dirw[aln_dir_word_index(kb * SPB, (uint32_t)lane, SPB)] = dw;
    }
    if (dead && lane == 0) __hip_atomic_store(sh.flags, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);

    if (is_local<SEM>()) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t y = yb + 1 + r;
            if (y <= M && rbx[r] != 0 && better<SC, SEM>(rbv[r], y, rbx[r], bv, by, bx)) { bv = rbv[r]; by = y; bx = rbx[r]; }
        }
    }
    if (last) {
        SC hb = Hl[0];
#pragma unroll
        for (int r = 1; r < R; ++r) if ((uint32_t)r == rb) hb = Hl[r];
        corner = O::rdlane(hb, (int)lb);
    }
}

__device__ __forceinline__ void skip_invalid(aln_pair_result &res, int status, int lane)
{
    if (lane == 0) {
        res.f = 0.0; res.score = 0.0; res.end_y = res.end_x = res.start_y = res.start_x = 0;
        res.aln_len = 0; res.status = status == ALN_PRE_EMPTY_OK ? ALN_OK : status; res.passes = 0; res.flags = 0;
    }
}


// The SIMD's instruction arbiter is not fair: among waves of equal priority the oldest wave issues first, and with three
// VALU-bound waves per SIMD the youngest gets ~13 % of the issue slots (measured: 1.5 / 0.9 / 0.35 GCUPS for the three).
// Throughput does not care, the tail of a small batch does: a large pair taken at t = 0 by a youngest wave was still in its
// first pass when everything else had finished.  So the wave's priority follows the size of its pair (thirds of the
// queue's largest pair): in the LPT order this is "oldest pair first" -- a large pair is never starved by the smaller pairs
// the older waves of its SIMD move on to.
__device__ __forceinline__ void set_wave_priority(uint64_t cells, uint64_t max_cells)
{
    if (3 * cells > 2 * max_cells) __builtin_amdgcn_s_setprio(2);          // 3 belongs to the walk kernel that runs beside the fill
    else if (3 * cells > max_cells) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
}

// ---------------------------------------------------------------- one pair, fast integer kernels
// Core local with del != ext: the fill is speculative in the row-1 penalty ("advice") and exact once the
// advice equals the bottom row it produced.  Pass 1 runs with all-"ext" advice and checkpoints strip 0; bottom-row zeros
// sit next to the left border, so a mismatch is repaired by re-running only the leading columns of strip 0 until its
// lane state rejoins the checkpoint (localized repair); anything else escalates to full re-fills and finally to the
// strict reference-order routine.
// what a wave's scratch holds for the fast kernels: [boundary rows][advice][bottom-row record][checkpoint sets]
struct FastScratch {
    unsigned long long *rows; // boundary rows of granules, `nrows` of them, row_elems apart
    uint32_t row_elems, nrows;
    int *ckpt;                // checkpoint sets, one per row, ck_ints apart
    uint32_t ck_ints;
    uint8_t *advice, *zrow;
};
__device__ __forceinline__ FastScratch fast_scratch(const FillArgs &a, uint32_t wave)
{
    uint8_t *sc = a.scratch + (uint64_t)wave * a.scratch_stride;
    const uint64_t brow_bytes = ((uint64_t)(a.max_len + 66) * 8 + 63) & ~(uint64_t)63;
    const uint64_t adv_bytes = ((uint64_t)a.max_len + 66 + 63) & ~(uint64_t)63;
    FastScratch fs;
    fs.nrows = a.cascade_rows ? a.cascade_rows : 1u;
    fs.rows = reinterpret_cast<unsigned long long *>(sc);
    fs.row_elems = (uint32_t)(brow_bytes / 8);
    fs.advice = sc + (uint64_t)fs.nrows * brow_bytes;
    fs.zrow = fs.advice + adv_bytes;
    fs.ckpt = reinterpret_cast<int *>(fs.zrow + a.zrow_bytes);
    fs.ck_ints = ALN_CK_SLOTS * 18 * 64;
    return fs;
}

// ---- cooperative passes (CoopRec, aln_device.h)
struct CoopCtx {
    uint32_t *ctl;            // control words; null: off
    uint32_t *words;          // claim words, one per fill wave
    uint32_t nw, nw_pad;      // fill waves; claim words incl. padding (a multiple of 64)
    CoopRec *recs;
    uint32_t wave;            // this wave = its record
};
typedef unsigned long long gran_t;
__device__ __forceinline__ uint32_t coop_ld(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void coop_st(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ gran_t gran_ld(const gran_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void gran_st(gran_t *p, uint32_t v, uint32_t tag) { __hip_atomic_store(p, ((gran_t)tag << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ CoopCtx coop_ctx(const FillArgs &a, uint32_t wave)
{
    CoopCtx c;
    c.ctl = a.coop;
    c.nw = a.coop_waves; c.nw_pad = (a.coop_waves + 63u) & ~63u;
    c.words = a.coop ? a.coop + ALN_COOP_CTL_WORDS : nullptr;
    c.recs = a.coop ? reinterpret_cast<CoopRec *>(c.words + c.nw_pad) : nullptr;
    c.wave = wave;
    return c;
}
// a strip's end-cell candidate, corner and status as granules of its pass
__device__ __forceinline__ void coop_put_cand(CoopRec *r, uint32_t s, uint32_t tag, const FastOut &o, int lane)
{
    if (lane == 0) {
        gran_t *c = r->cand[s];
        gran_st(c, (uint32_t)o.bv, tag | s); gran_st(c + 1, o.by, tag | s); gran_st(c + 2, o.bx, tag | s); gran_st(c + 3, (uint32_t)o.corner, tag | s);
        gran_st(c + 4, o.aborted ? 1u : 0u, tag | s);
    }
}
// opens this wave's record: the pair's granule, then the claim word -- ns - 1 strips are up for grabs -- then the counters that
// make other waves look.  announce = false (testing): kind 0, nobody but the owner claims.
__device__ __forceinline__ void coop_open(const CoopCtx &c, int lane, uint32_t pair, uint32_t tag, uint32_t seq, uint32_t ns, uint32_t R, uint32_t own,
                                          bool urgent, bool announce)
{
    if (lane == 0) {
        gran_st(&c.recs[c.wave].pairg, pair, tag | 127u);
        coop_st(c.words + c.wave, aln_coop_word(seq, R, own, announce ? (urgent ? ALN_COOP_KIND_URGENT : ALN_COOP_KIND_LAZY) : 0u, ns, ns - 1u));
        if (announce) {
            __hip_atomic_fetch_add(c.ctl + (urgent ? ALN_COOP_UOPEN : ALN_COOP_LOPEN), ns - 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (urgent) {
                const uint32_t i = __hip_atomic_fetch_add(c.ctl + ALN_COOP_ULOGW, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                coop_st(c.ctl + ALN_COOP_ULOG + (i & 15u), c.wave + 1u);
            }
        }
    }
}
// The next unclaimed strip of wave w's open pass, or -1; word: the claim word the strip was taken from (wave-uniform).  own: the
// caller is w itself (it also claims from a pass it did not announce).
__device__ __forceinline__ int coop_claim(const CoopCtx &c, uint32_t w, int lane, bool own, uint32_t &word)
{
    int strip = -1;
    uint32_t v = 0;
    if (lane == 0) {
        uint32_t *p = c.words + w;
        v = coop_ld(p);
        for (;;) {
            const uint32_t left = v & 0x7fu, ns = (v >> 7) & 0x7fu, kind = (v >> 14) & 3u;
            if (left == 0u || (kind == 0u && !own)) break;
            if (__hip_atomic_compare_exchange_strong(p, &v, v - 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                strip = (int)(ns - left);
                if (kind) __hip_atomic_fetch_add(c.ctl + (kind == ALN_COOP_KIND_URGENT ? ALN_COOP_UOPEN : ALN_COOP_LOPEN), 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    strip = __builtin_amdgcn_readfirstlane(strip);
    word = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    return strip;
}
// A record with strips to give away, or false.  urgent_only: what a wave between two pairs asks (re-fills only).  Cheap when there
// is nothing (one 16-byte load of the counters' line); else the log of the last urgent opens, then a scan of the claim words.
__device__ __forceinline__ bool coop_find(const CoopCtx &c, bool urgent_only, int lane, uint32_t &wave_out, bool &all_done, uint32_t n_pairs)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 q;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(q) : "v"(c.ctl) : "memory");
    const uint32_t uopen = q.x, lopen = urgent_only ? 0u : q.y, wi = q.z;
    all_done = q.w >= n_pairs;
    // (the counters are exact only in the long run -- a claim takes the word before it takes the counter down -- hence "signed")
    if ((int32_t)uopen <= 0 && (int32_t)lopen <= 0) return false;
    const uint32_t kmin = ((int32_t)lopen > 0) ? ALN_COOP_KIND_LAZY : ALN_COOP_KIND_URGENT;
    if ((int32_t)uopen > 0) {                                // the last urgent opens
        uint32_t cand = 0, word = 0;
        if ((uint32_t)lane < 16u && (uint32_t)lane < wi) cand = coop_ld(c.ctl + ALN_COOP_ULOG + ((wi - 1u - (uint32_t)lane) & 15u));
        if (cand != 0 && cand <= c.nw) word = coop_ld(c.words + cand - 1u);
        const uint64_t m = __ballot((word & 0x7fu) != 0u && ((word >> 14) & 3u) == ALN_COOP_KIND_URGENT);
        if (m) { wave_out = (uint32_t)__builtin_amdgcn_readlane((int)cand, (int)__builtin_ctzll(m)) - 1u; return true; }
    }
    // scan, four loads of 64 words in flight; every wave starts somewhere else
    if (lane == 0) __hip_atomic_fetch_add(c.ctl + ALN_COOP_SCANS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t start = ((c.wave * 832u) % c.nw_pad) & ~63u;
    for (uint32_t i = 0; i < c.nw_pad; i += 256u) {
        uint32_t v[4], idx[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            idx[j] = (start + i + 64u * j + (uint32_t)lane) % c.nw_pad;
            v[j] = (i + 64u * j < c.nw_pad) ? coop_ld(c.words + idx[j]) : 0u;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t kind = (v[j] >> 14) & 3u;
            const uint64_t m = __ballot((v[j] & 0x7fu) != 0u && kind >= kmin && idx[j] != c.wave);
            if (m) { wave_out = (uint32_t)__builtin_amdgcn_readlane((int)idx[j], (int)__builtin_ctzll(m)); return true; }
        }
    }
    return false;
}

// Strip s >= 1 of a pass over `pair`: on the wave that owns the pair (open == false: the strips run one after the other and `o`
// carries the lane candidates along) or on any wave that claimed it (open: `o` starts empty and the strip's candidate goes to the
// owner's record).  `in` arrives with this wave's constants (lane, S, profile, penalties); everything about the pair is set here.
// Ru: 0 = skewed layout, else uniform strips of 64 Ru rows; own: strip 0's bottom row keeps row 0 to itself.
template <int SEM, bool PWM, bool COOP>
__device__ __forceinline__ void coop_run_strip(FastIn in, const FillArgs &a, uint32_t pair, uint32_t ns, uint32_t Ru, uint32_t own, uint32_t tag,
                                               uint32_t owner, CoopRec *rec, uint32_t s, FastOut &o, bool open, int del, int ext)
{
    const PairDesc &desc = a.descs[pair];
    const uint32_t N = desc.N, M = desc.M;
    const FastScratch fs = fast_scratch(a, owner);
    in.N = N; in.M = M;
    in.q = a.seqs + desc.q_off;
    in.t = a.seqs + desc.t_off;
    in.dirw = reinterpret_cast<uint32_t *>(a.dirs + desc.dir_off);
    in.hazard = (SEM == ALN_CORE_LOCAL) && (del != ext) && N >= 2;
    in.adv_any = false;
    in.ring_in = nullptr; in.ring_out = nullptr; in.lds_scratch = 0;
    in.ck_mode = 0; in.last_flip = 0; in.ck_stop = 0;
    in.store_dirs = a.store_dirs != 0;
    // direction quads of a shared pass are stored write-through: its strips run on other XCDs, and lines that sit dirty in two L2s
    // are written back in any order (fast_work fences before it opens a re-fill over a first pass stored the ordinary way)
    in.wt_dirs = a.doneq != nullptr || open;
    in.pwm = a.pwm != 0;
    in.pwm_words = a.pwm_words;
    in.advice = fs.advice; in.zrow = fs.zrow; in.ckpt = fs.ckpt;
    // rows: a strip never writes the row it reads.  Strip 0 writes row 0 -- which stays as it is when the pass may be repaired
    // (own) --, the strips below alternate between two rows (a strip is always behind the one whose row it overwrites: that one
    // waits, column by column, for the strip in between)
    const uint32_t r_out = own ? 1u + ((s - 1u) & 1u) : (s & 1u), r_in = s == 1u ? 0u : (own ? 1u + (s & 1u) : ((s - 1u) & 1u));
    in.brow_in = fs.rows + (size_t)r_in * fs.row_elems;
    in.brow_out = fs.rows + (size_t)r_out * fs.row_elems;
    in.strip_rows = Ru ? 64u * Ru : (uint32_t)ALN_STRIP_ROWS;
    in.strip_q16 = (uint32_t)((Ru ? aln_uniform_strip_bytes(N, Ru) : aln_strip_bytes(N)) / 16u);
    in.tag_base = tag;
    const bool last = s + 1u == ns;
    const int R = Ru ? (int)Ru : (last ? aln_pick_r(M - s * ALN_STRIP_ROWS) : ALN_FULL_R);
    if (open) { o.bv = INT_MIN; o.by = 0; o.bx = 0; o.corner = 0; o.repaired = false; o.brow_bad = false; o.aborted = false; o.ck_slot = 0; o.c_out = 0; }
    o = fast_strip_next<SEM, PWM>(in, o, s, last, R);
    if (COOP && o.aborted && a.coop && in.lane == 0) __hip_atomic_fetch_add(a.coop + 10, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (COOP && (a.coop_debug & 128u) && a.coop && !last) {        // testing: does the row this strip leaves behind carry its tag in every column?
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t bad = 0;
        for (uint32_t x = 1 + in.lane; x <= N; x += 64)
            if ((uint32_t)(gran_ld(in.brow_out + x) >> 32) != (tag | s)) ++bad;
        if (__any(bad != 0)) {
            uint32_t first = 0;
            if (in.lane == 0) first = __hip_atomic_fetch_add(a.coop + 14, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            first = (uint32_t)__builtin_amdgcn_readfirstlane((int)first);
            if (first == 0) {                            // the first strip that finds any: its shape and the columns, in order
                if (in.lane == 0) { coop_st(a.coop + 15, (s << 4) | (uint32_t)R); coop_st(a.coop + 16, N); coop_st(a.coop + 17, own | (open ? 2u : 0u) | (ns << 8)); }
                uint32_t slot = 0;
                for (uint32_t x0 = 1; x0 <= N && slot < 40; x0 += 64) {
                    const uint32_t x = x0 + in.lane;
                    const gran_t g = x <= N ? gran_ld(in.brow_out + x) : ((gran_t)(tag | s) << 32);
                    uint64_t m = __ballot((uint32_t)(g >> 32) != (tag | s));
                    while (m && slot < 40) {
                        const int l = __builtin_ctzll(m);
                        m &= m - 1;
                        if (in.lane == l) { coop_st(a.coop + 128 + slot, x); coop_st(a.coop + 192 + slot, (uint32_t)(g >> 32)); }
                        ++slot;
                    }
                }
            }
        }
    }
    if (!open) return;
    if (is_local<SEM>()) reduce_best<SEM>(o);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this strip's directions, row and bottom-row record are out: now say so
    coop_put_cand(rec, s, tag, o, in.lane);
}

// One pair on the wave that took it from the queue (helper == false), or every strip this wave can claim from wave w's open pass
// (helper == true: a pair somebody else owns).  ONE function, so that the strips below strip 0 are instantiated at one place
// whoever runs them.  epoch: this wave's count of multi-strip passes in this launch (the seq of their tags).
// Returns true when the pair's directions were written with ordinary stores (the strict-order fallback).
// COOP = false: the kernel variant for batches that have nothing to share (every pair one strip and no re-fill worth sharing: read
// pairs, PWM windows, the p-value batch) -- none of the machinery is compiled in, and the wave keeps far less state alive around
// a strip (with it, the 150 x 150 read pairs of C3 paid 100 scratch stores and loads per pair: 12 % of their fill).
template <int SEM, bool PWM, bool COOP>
__device__ __forceinline__ bool fast_work(FastIn in, const FastScratch &fs, const CoopCtx &cp, const FillArgs &a, const bool helper, const uint32_t w,
                                          uint32_t pair, const uint32_t qpos, uint32_t &epoch, int del, int ext)
{
    const int lane = in.lane;
    uint32_t N = 0, M = 0, ns_skew = 0, max_passes = 0;
    bool can_repair = false;
    if (!helper) {
        const PairDesc &desc = a.descs[pair];
        N = desc.N; M = desc.M;
        in.N = N; in.M = M;
        in.q = a.seqs + desc.q_off;
        in.t = a.seqs + desc.t_off;
        in.dirw = reinterpret_cast<uint32_t *>(a.dirs + desc.dir_off);
        in.hazard = (SEM == ALN_CORE_LOCAL) && (del != ext) && N >= 2;
        in.adv_any = false;
        in.ring_in = nullptr; in.ring_out = nullptr; in.lds_scratch = 0;
        in.ck_mode = 0; in.last_flip = 0; in.ck_stop = 0;
        in.store_dirs = a.store_dirs != 0;
        in.wt_dirs = a.doneq != nullptr;
        in.pwm = a.pwm != 0;
        in.pwm_words = a.pwm_words;
        in.advice = fs.advice; in.zrow = fs.zrow; in.ckpt = fs.ckpt;
        if (in.hazard)
            for (uint32_t x = lane; x <= N + 1; x += 64) in.advice[x] = 0;      // the bottom-row record is rewritten by every pass
        __threadfence_block();
        ns_skew = aln_num_strips(M);
        max_passes = a.max_passes ? a.max_passes : 4u;
        // repairable: strip 0's bottom row stays as the checkpointed pass wrote it (row 0); the strips below alternate between rows 1 and 2
        can_repair = in.hazard && !a.no_repair && (ns_skew == 1 || fs.nrows >= 3);
    }
    CoopRec *rec = (COOP && cp.ctl) ? cp.recs + (helper ? w : cp.wave) : nullptr;
    bool coop_ok = COOP && cp.ctl != nullptr;                    // cleared when a cooperative pass had to give up: the rest runs on this wave alone
    uint32_t passes = 0, Ru = 0, tag = 0, ns = 0, own = 0;
    bool converged = false, device_error = false, open = helper;
    FastOut o;
    o.bv = INT_MIN; o.by = 0; o.bx = 0; o.corner = 0; o.repaired = false; o.brow_bad = false; o.aborted = false; o.ck_slot = 0; o.c_out = 0;
    for (;;) {                                           // full passes (a helper: once through the strips)
        if (!helper) {
            // A re-fill (the pair is late: every other pair of the batch needs one pass) is cut into more, smaller strips than the
            // first pass, so that up to eight waves can share it
            // -- when that matters: while the queue still holds two pairs or more for every wave, the wave just fills it again by
            // itself (a shared re-fill takes other waves off their pairs and its smaller strips cost more instructions per cell:
            // 2 % of the C5 batch's fill when every re-fill was shared).  First passes: see FillArgs::coop_tail.
            unsigned long long qw = 0;                   // the queue word: pairs taken from the front | from the back (next_pair2)
            if (coop_ok && passes != 0) qw = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(a.counter + 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool share = coop_ok && (passes != 0 ? ((qw & 0xffffffffull) + (qw >> 32) + 2u * cp.nw >= a.n_pairs || (a.coop_debug & 8u))
                                                       : (qpos >= a.coop_tail && !(a.coop_debug & 1u)));
            Ru = (passes != 0 && share && SEM == ALN_CORE_LOCAL && !PWM && !(a.coop_debug & 4u)) ? aln_coop_uniform_r(M) : 0u;
            const uint32_t srows = Ru ? 64u * Ru : (uint32_t)ALN_STRIP_ROWS;
            ns = (M + srows - 1u) / srows;
            own = (passes == 0 && can_repair) ? 1u : 0u;
            // the tag of this pass's granules (rows, bottom-row record, candidates); when the 12 bits of the count wrap, tags of
            // this launch could come back: the rows are cleared first
            ++epoch;
            if ((epoch & 0xfffu) == 0u) {
                for (uint32_t x = lane; x < fs.nrows * fs.row_elems; x += 64) fs.rows[x] = 0ull;
                for (uint32_t x = lane; x < a.zrow_bytes / 8u; x += 64) reinterpret_cast<unsigned long long *>(fs.zrow)[x] = 0ull;
                __threadfence();
                ++epoch;
            }
            tag = aln_coop_tag(a.salt, epoch);
            open = share && ns >= 2 && ns <= ALN_COOP_MAX_NS;
            if (open) {
                // the strips of this pass may be stored from other XCDs: nothing the earlier passes of this pair stored the ordinary
                // way may still sit dirty in this XCD's L2 (it would be written back over them at some later time)
                if (passes != 0 && a.doneq == nullptr) __threadfence();
                coop_open(cp, lane, pair, tag, epoch, ns, Ru, own, passes != 0, !(a.coop_debug & 2u));
            }
            in.wt_dirs = a.doneq != nullptr || open;
            o.bv = INT_MIN; o.by = 0; o.bx = 0; o.corner = 0; o.repaired = false; o.brow_bad = false; o.aborted = false; o.ck_slot = 0; o.c_out = 0;
            in.brow_in = fs.rows; in.brow_out = fs.rows;
            in.strip_rows = srows;
            in.strip_q16 = (uint32_t)((Ru ? aln_uniform_strip_bytes(N, Ru) : aln_strip_bytes(N)) / 16u);
            in.tag_base = tag;
            in.ck_mode = (passes == 0 && can_repair) ? 1 : 0;                   // strip 0 of the first pass saves its checkpoints
            o = fast_strip_first<SEM, PWM>(in, o, ns == 1, Ru ? (int)Ru : (ns == 1 ? aln_pick_r(M) : ALN_FULL_R));
            in.ck_mode = 0;
            if (open) {
                if (is_local<SEM>()) reduce_best<SEM>(o);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                coop_put_cand(rec, 0u, tag, o, lane);
            }
        }
        // ---- the strips below strip 0: in order on this wave (not open), or whichever this wave can claim (the owner of an open pass
        // and every helper alike)
        for (uint32_t snext = 1;;) {
            uint32_t s = snext;
            if (open) {
                uint32_t word;
                const int c = coop_claim(cp, helper ? w : cp.wave, lane, !helper, word);
                if (c < 0) break;
                s = (uint32_t)c;
                if (helper) {
                    ns = (word >> 7) & 0x7fu; own = (word >> 16) & 1u; Ru = (word >> 17) & 7u; tag = aln_coop_tag(a.salt, word >> 20);
                    // the pair: a granule of the same pass (written before the claim word; the tag says it is not an older pass's)
                    gran_t g = gran_ld(&rec->pairg);
                    for (uint32_t spins = 0; (uint32_t)(g >> 32) != (tag | 127u) && spins < 100000u; ++spins) { __builtin_amdgcn_s_sleep(2); g = gran_ld(&rec->pairg); }
                    pair = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)g);
                    if ((uint32_t)(g >> 32) != (tag | 127u) || pair >= a.n_descs || s >= ns || Ru > 4u) {    // cannot be; never index by it
                        o.bv = INT_MIN; o.by = 0; o.bx = 0; o.corner = 0; o.aborted = true;
                        if (lane == 0) __hip_atomic_fetch_add(cp.ctl + 13, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        coop_put_cand(rec, s, tag, o, lane);
                        continue;
                    }
                    const PairDesc &d = a.descs[pair];
                    if (((word >> 14) & 3u) == ALN_COOP_KIND_URGENT) __builtin_amdgcn_s_setprio(2);       // a re-fill: the pair is late already
                    else set_wave_priority((uint64_t)d.N * d.M, a.max_cells);
                    if (lane == 0) __hip_atomic_fetch_add(cp.ctl + ALN_COOP_HELPED, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                if (s >= ns) break;
                ++snext;
            }
            coop_run_strip<SEM, PWM, COOP>(in, a, pair, ns, Ru, own, tag, helper ? w : cp.wave, rec, s, o, open, del, ext);
        }
        if (helper) return false;
        bool pass_bad = false;
        if (!open) pass_bad = o.aborted;                 // (a strip of this wave's own pass never waits: cannot be)
        else {
            // Every strip has been claimed; the waves that run the others are resident and their own waits are bounded.  Lane s
            // polls the five granules of strip s; all of them carry the pass's tag once the strip is through.
            gran_t g[5] = {0, 0, 0, 0, 0};
            uint32_t spins = 0;
            for (;;) {
                bool ok = true;
                if ((uint32_t)lane < ns) {
#pragma unroll
                    for (int i = 0; i < 5; ++i) { g[i] = gran_ld(rec->cand[lane] + i); ok = ok && (uint32_t)(g[i] >> 32) == (tag | (uint32_t)lane); }
                }
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(32);
                if (++spins > (1u << 18)) { device_error = true; break; }
            }
            if (device_error) { if (lane == 0) __hip_atomic_fetch_add(cp.ctl + 11, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            if (lane == 0 && spins) __hip_atomic_fetch_add(cp.ctl + 9, spins, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // the strips' candidates, merged with the exact tie rule; every lane ends up with the winner
            o.bv = INT_MIN; o.by = 0; o.bx = 0;
            if ((uint32_t)lane < ns) { o.bv = (int)(uint32_t)g[0]; o.by = (uint32_t)g[1]; o.bx = (uint32_t)g[2]; }
            pass_bad = __any((uint32_t)lane < ns && (uint32_t)g[4] != 0u) != 0;
            if (is_local<SEM>()) reduce_best<SEM>(o);
            o.corner = __builtin_amdgcn_readlane((int)(uint32_t)g[3], (int)(ns - 1u));
            o.aborted = false;
        }
        uint32_t last_flip = 0;
        bool zbad = false;
        if (!pass_bad && in.hazard) {
            converged = adopt_advice_zdw(in.advice, reinterpret_cast<const unsigned long long *>(in.zrow), tag | (ns - 1u), N, M, Ru, lane, last_flip, zbad);
            pass_bad = zbad;
            if (zbad && cp.ctl && lane == 0) __hip_atomic_fetch_add(cp.ctl + 12, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (pass_bad) {                                  // a strip gave up waiting (should not happen): this pass again, alone
            if (!coop_ok) { device_error = true; break; }
            coop_ok = false;
            if (lane == 0) __hip_atomic_fetch_add(cp.ctl + ALN_COOP_ABORTS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (in.hazard) for (uint32_t x = lane; x <= N + 1; x += 64) in.advice[x] = 0;     // (adopt may have touched it)
            __threadfence_block();
            if (passes != 0) { device_error = true; break; }               // a re-fill's advice cannot be rebuilt here: give the pair up
            continue;
        }
        ++passes;
        __threadfence_block();
        if (!in.hazard) { converged = true; break; }
        if (converged) break;
        if (passes == 1 && can_repair) {
            // Localized repair.  The new advice perturbs strip 0 from column last_flip's step on; strip 0 re-runs its leading
            // columns until its lane state rejoins a checkpoint (the end-cell candidates of the re-run prefix replace the
            // checkpointed ones).  If its bottom row changed on the way, the strips below would have to follow -- measured on
            // C5: a perturbation that reaches the bottom of strip 0 does not die out within the 512 checkpointed steps of any
            // strip (a cascade through per-strip checkpoints never once re-converged), so that case is re-filled in full.
            uint32_t repairs = 0;
            bool failed = false;
            in.strip_rows = (uint32_t)ALN_STRIP_ROWS;
            in.strip_q16 = (uint32_t)(aln_strip_bytes(N) / 16u);
            in.brow_in = fs.rows; in.brow_out = fs.rows;
            while (!converged && !failed && repairs < 8) {
                ++repairs;
                passes += 0x100u;                        // repair rounds are counted in bits 8..15
                if (last_flip > a.ck_last) { failed = true; passes |= 0x100000u; break; }    // beyond the last checkpoint
                in.ck_mode = 2;
                in.last_flip = last_flip;
                FastOut ro = o;
                ro.repaired = false; ro.c_out = 0;
                ro = fast_strip_first<SEM, PWM>(in, ro, ns_skew == 1, ns_skew == 1 ? aln_pick_r(M) : ALN_FULL_R);
                __threadfence_block();
                if (!__any(ro.repaired)) { failed = true; passes |= 0x300000u; break; }       // no re-convergence (or a stale end-cell candidate)
                if (ro.c_out != 0) { failed = true; passes |= 0x200000u; break; }            // strip 0's bottom row moved
                passes = (passes & ~0xf0000u) | ((ro.ck_slot + 1u) << 16);               // diagnostics: where the repair re-converged
                o = ro;
                if (ns_skew > 1) { converged = true; break; }                             // the bottom strip, hence z, is untouched
                // single strip: z may have moved (the repair run rewrote the record with this pass's tag)
                converged = adopt_advice_zdw(in.advice, reinterpret_cast<const unsigned long long *>(in.zrow), tag, N, M, 0u, lane, last_flip, zbad);
                if (zbad) { failed = true; break; }
            }
            in.ck_mode = 0; in.last_flip = 0;
            if (converged) break;
            if (!(passes & 0xf00000u)) passes |= 0x400000u;
        }
        if ((passes & 0xffu) >= max_passes) break;
    }
    PairDesc &desc = a.descs[pair];
    aln_pair_result &res = a.results[pair];
    if (device_error) {                                  // a strip handed to another wave never reported: nothing here may be trusted
        if (lane == 0) { skip_invalid(res, ALN_ERR_DEVICE, 0); desc.layout = ALN_LAYOUT_SKEW; }
        return true;
    }
    if (!converged) {                                    // strict reference order (exact for every input)
        Wave<int> c;
        c.lane = 0; c.N = N; c.M = M; c.q = in.q; c.t = in.t; c.S = in.S; c.cols = in.cols; c.del = del; c.ext = ext;
        c.dirw = in.dirw; c.brow = reinterpret_cast<int *>(fs.rows); c.hmat = nullptr; c.store_dirs = in.store_dirs; c.pwm = in.pwm;
        c.bv = 0; c.by = 0; c.bx = 0; c.corner = 0;
        if (lane == 0) serial_fill_impl<int, SEM>(c);
        __threadfence();                                 // (ordinary stores, also into the rows other waves' granules land in later: none stays dirty here)
        if (lane == 0) {
            desc.layout = ALN_LAYOUT_ROWMAJOR;
            write_result<SEM>(res, (double)c.bv, c.by, c.bx, (double)c.corner, N, M, passes | 0x80u, 1u, in.pwm);
        }
        return true;                                     // directions written with ordinary stores
    }
    if (is_local<SEM>()) reduce_best<SEM>(o);
    if (lane == 0) {
        desc.layout = Ru ? (ALN_LAYOUT_UBATCH | (Ru << 8)) : ALN_LAYOUT_SKEW;
        write_result<SEM>(res, (double)(o.bv >> 2), o.by, o.bx, (double)(o.corner >> 2), N, M, passes, 1u, in.pwm);
    }
    return false;
}

}  // namespace

// ---------------------------------------------------------------- fill kernels: persistent waves over a work queue
// The fast kernels' queue has two ends: {pairs taken from the front, pairs taken from the back} in ONE 64-bit word (counter[4..5]),
// so that a take knows both counts at once: it is valid iff front + back < n before it, and then nobody else has its pair.  The
// longest pairs sit at the front (LPT); the youngest wave of every SIMD -- which its two elders leave a quarter of the issue slots:
// 0.39 GCUPS against 1.6 and 0.9 -- takes from the back, so that it is never the one that holds a 2-million-cell pair (its first
// pair took it 5 of the 8-way shard's 6 ms) and the short pairs are out of the way before the queue runs dry.
__device__ __forceinline__ bool next_pair2(const FillArgs &a, int lane, bool from_back, uint32_t &pair, uint32_t &idx)
{
    unsigned long long v = 0;
    if (lane == 0) v = __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(a.counter + 4), from_back ? (1ull << 32) : 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t f = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), b = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    if ((unsigned long long)f + b >= a.n_pairs) return false;
    idx = from_back ? a.n_pairs - 1u - b : f;
    pair = a.order[idx];
    return true;
}
__device__ __forceinline__ bool next_pair(const FillArgs &a, int lane, uint32_t &pair, uint32_t &idx)
{
    idx = 0;
    if (lane == 0) idx = atomicAdd(a.counter, 1u);
    idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)idx);
    if (idx >= a.n_pairs) return false;
    pair = a.order[idx];
    return true;
}
// Overlapped traceback: this pair is complete -- hand it to the walk kernel.  The walk waves sit on other XCDs (own L2), and
// an agent-scope release fence here would write back this XCD's whole L2 (~90 us per pair, measured as 6 % of the fill).
// Instead everything a walk reads is stored write-through: the direction quads (FastIn::wt_dirs), and here the few
// summary fields; then only the stores' completion is awaited before the queue entry (itself write-through) goes out.
__device__ __forceinline__ void wt_store32(void *p, uint32_t v)
{
    __hip_atomic_store(reinterpret_cast<uint32_t *>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void pair_done(const FillArgs &a, int lane, uint32_t pair, bool plain_stores)
{
    if (!a.doneq) return;
    if (plain_stores) __threadfence();                   // strict-order fallback / generic kernels: ordinary stores, full release
    if (lane == 0) {
        aln_pair_result &res = a.results[pair];
        PairDesc &desc = a.descs[pair];
        wt_store32(&res.status, (uint32_t)res.status);
        wt_store32(&res.end_y, res.end_y);
        wt_store32(&res.end_x, res.end_x);
        wt_store32(&desc.layout, desc.layout);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
        const uint32_t pos = __hip_atomic_fetch_add(a.counter + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.doneq + pos, pair + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// The reference indexes the matrix with every residue (simple/mod.rs:85,198) and panics on a code outside it.  The wave that
// takes a pair scans both sequences first (the codes are on their way into L2 anyway; a few microseconds per pair) and turns
// that into the pair's status: the host never touches the residues of a batch.
__device__ __forceinline__ bool pair_codes_ok(const uint8_t *seqs, const PairDesc &d, uint32_t rows, uint32_t cols, bool pwm, int lane)
{
    const uint8_t *q = seqs + d.q_off, *t = seqs + d.t_off;
    uint32_t worst_q = 0, worst_t = 0;
    if (!pwm) for (uint32_t i = (uint32_t)lane; i < d.N; i += 64u) worst_q = max(worst_q, (uint32_t)q[i]);
    for (uint32_t i = (uint32_t)lane; i < d.M; i += 64u) worst_t = max(worst_t, (uint32_t)t[i]);
    return !__any(worst_q >= cols || worst_t >= rows);
}
// (f64: two workgroups per CU -- 256 registers per lane: eight rows of f64 state, scores and trackers spilled at 168)
template <typename SC, int SEM>
__global__ __launch_bounds__(256, 3) void aln_fill_kernel(FillArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    SC *S = reinterpret_cast<SC *>(smem);
    const SC *gm = reinterpret_cast<const SC *>(a.matrix);
    for (uint32_t i = threadIdx.x; i < a.rows * a.cols; i += blockDim.x) S[i] = gm[i];
    __syncthreads();

    Wave<SC> w;
    w.lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    uint8_t *sc = a.scratch + (uint64_t)wave * a.scratch_stride;
    const uint64_t brow_bytes = ((uint64_t)(a.max_len + 66) * sizeof(SC) + 63) & ~(uint64_t)63;
    const uint64_t adv_bytes = ((uint64_t)a.max_len + 66 + 63) & ~(uint64_t)63;
    w.brow = reinterpret_cast<SC *>(sc);
    w.advice = sc + brow_bytes;
    w.zrow = sc + brow_bytes + adv_bytes;                 // a.zrow_bytes >= adv_bytes
    w.row1 = reinterpret_cast<SC *>(w.zrow + a.zrow_bytes);
    w.row1tag = reinterpret_cast<uint8_t *>(w.row1) + brow_bytes;
    w.S = S;
    w.cols = a.cols;
    w.del = ScOps<SC>::from_double(a.del);
    w.ext = ScOps<SC>::from_double(a.ext);
    uint32_t pair, qpos;
    while (next_pair(a, w.lane, pair, qpos)) {
        PairDesc &desc = a.descs[pair];
        aln_pair_result &res = a.results[pair];
        if (desc.status != ALN_OK) skip_invalid(res, desc.status, w.lane);
        else if (!pair_codes_ok(a.seqs, desc, a.rows, a.cols, a.pwm != 0, w.lane)) skip_invalid(res, ALN_ERR_CODE_OUT_OF_RANGE, w.lane);
        else do_pair<SC, SEM>(w, a, desc, res);
        pair_done(a, w.lane, pair, true);
    }
}

// The real-valued batch (core semantics, no H dump): the same kernel around the lean f64 strip alone -- without run_strip's all-options
// loop in it the kernel fits 128 registers, four waves per SIMD instead of three (the f64 cell chain is latency-bound per wave).
template <int SEM>
__global__ __launch_bounds__(256, 4) void aln_fill_f64_kernel(FillArgs a)
{
    using SC = double;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    SC *S = reinterpret_cast<SC *>(smem);
    const SC *gm = reinterpret_cast<const SC *>(a.matrix);
    for (uint32_t i = threadIdx.x; i < a.rows * a.cols; i += blockDim.x) S[i] = gm[i];
    __syncthreads();

    Wave<SC> w;
    w.lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    uint8_t *sc = a.scratch + (uint64_t)wave * a.scratch_stride;
    const uint64_t brow_bytes = ((uint64_t)(a.max_len + 66) * sizeof(SC) + 63) & ~(uint64_t)63;
    const uint64_t adv_bytes = ((uint64_t)a.max_len + 66 + 63) & ~(uint64_t)63;
    w.brow = reinterpret_cast<SC *>(sc);
    w.advice = sc + brow_bytes;
    w.zrow = sc + brow_bytes + adv_bytes;                 // a.zrow_bytes >= adv_bytes
    w.row1 = reinterpret_cast<SC *>(w.zrow + a.zrow_bytes);
    w.row1tag = reinterpret_cast<uint8_t *>(w.row1) + brow_bytes;
    w.S = S;
    w.cols = a.cols;
    w.del = ScOps<SC>::from_double(a.del);
    w.ext = ScOps<SC>::from_double(a.ext);
    uint32_t pair, qpos;
    while (next_pair(a, w.lane, pair, qpos)) {
        PairDesc &desc = a.descs[pair];
        aln_pair_result &res = a.results[pair];
        if (desc.status != ALN_OK) skip_invalid(res, desc.status, w.lane);
        else if (!pair_codes_ok(a.seqs, desc, a.rows, a.cols, a.pwm != 0, w.lane)) skip_invalid(res, ALN_ERR_CODE_OUT_OF_RANGE, w.lane);
        else do_pair<SC, SEM, true>(w, a, desc, res);
        pair_done(a, w.lane, pair, true);
    }
}

template <int SEM, bool PWM, bool COOP>
// 160 VGPRs, not the 168 that three waves per SIMD would allow (the attribute counts pairs on gfx90a+): the 32 registers
// left over on every SIMD hold one wave of the walk kernel that runs beside the fill.
__global__ __attribute__((amdgpu_flat_work_group_size(256, 256), amdgpu_waves_per_eu(3, 3), amdgpu_num_vgpr(80)))
void aln_fill_fast_kernel(FillArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *S = reinterpret_cast<int *>(smem);
    const int *gm = reinterpret_cast<const int *>(a.matrix);
    for (uint32_t i = threadIdx.x; i < a.rows * a.cols; i += blockDim.x) S[i] = gm[i];
    __syncthreads();

    FastIn in;
    in.lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const FastScratch fs = fast_scratch(a, wave);
    CoopCtx cp;
    if constexpr (COOP) cp = coop_ctx(a, wave);
    else { cp.ctl = nullptr; cp.words = nullptr; cp.nw = 0; cp.nw_pad = 0; cp.recs = nullptr; cp.wave = wave; }
    in.brow_in = fs.rows; in.brow_out = fs.rows;
    in.advice = fs.advice;
    in.zrow = fs.zrow;
    in.ckpt = fs.ckpt;
    in.S = S;
    in.cols = a.cols;
    in.prof = smem + ((a.rows * a.cols * 4u + 15u) & ~15u) + (threadIdx.x >> 6) * a.prof_stride;
    in.nd4 = -4 * (int)a.del;
    in.ne4 = -4 * (int)a.ext;
    in.gin = nullptr; in.gout = nullptr; in.abort_flag = nullptr; in.qo_pad = nullptr; in.bring = nullptr;
    in.N = 0; in.M = 0; in.q = nullptr; in.t = nullptr; in.dirw = nullptr; in.hazard = false; in.adv_any = false; in.store_dirs = true; in.pwm = false; in.pwm_words = nullptr; in.ck_mode = 0; in.last_flip = 0;
    in.ring_in = nullptr; in.ring_out = nullptr; in.lds_scratch = 0; in.ck_stop = 0; in.wt_dirs = false;
    in.strip_rows = ALN_STRIP_ROWS; in.strip_q16 = 0; in.tag_base = 0;
    in.ck_last = a.ck_last;
    in.fair = a.fair ? (a.fair << 8) | (__builtin_amdgcn_s_getreg((3 << 11) | 4) & 15u) : 0u;      // HW_ID[3:0]: the wave's slot in its SIMD
    uint32_t pair = 0, qpos = 0, epoch = 0;
    bool dry = false;
    // the third workgroup of every CU = the youngest wave of every SIMD (a full grid only; FillArgs::back_waves)
    const bool from_back = a.back_waves != 0 && blockIdx.x * 3u >= gridDim.x * 2u;
    uint64_t idle_since = 0;
    uint32_t run_left = 0, run_pos = 0;                      // FillArgs::claim > 1: the rest of the run of queue positions this wave took
    for (;;) {
        // what next: strips other waves give away -- re-fills (urgent) before the next pair, first passes too once the queue is dry --
        // else the next pair of the queue, else (the queue is dry) wait for either
        bool all_done = false, helper = false;
        uint32_t w = 0;
        if (COOP && cp.ctl) helper = coop_find(cp, !dry, in.lane, w, all_done, a.n_pairs);
        if (!helper) {
            bool got = false;
            if (!COOP && a.claim > 1u) {
                if (!dry && run_left == 0u) {
                    unsigned long long v = 0;
                    if (in.lane == 0) v = __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(a.counter + 4), (unsigned long long)a.claim, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t f = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
                    if (f < a.n_pairs) { run_pos = f; run_left = min(a.claim, a.n_pairs - f); }
                }
                if (run_left != 0u) { qpos = run_pos++; --run_left; pair = a.order[qpos]; got = true; }
            } else if (!dry) got = next_pair2(a, in.lane, from_back, pair, qpos);
            if (!dry && !got) {
                dry = true;
                if (COOP && cp.ctl) continue;            // first passes of other waves' pairs next
            }
            if (dry) {
                // nobody has a strip to give away right now: stay while pairs are still being filled -- any of them may yet need a
                // second pass -- but look rarely (a few thousand idle waves polling one line every few microseconds slowed the
                // waves that still work by 20 %)
                if (!COOP || !cp.ctl || !a.coop_linger || all_done) break;
                if (idle_since == 0) idle_since = wall_clock64();
                else if (wall_clock64() - idle_since > 200000000ull) break;        // 2 s of nothing: leave
                __builtin_amdgcn_s_setprio(0);
                for (int i = 0; i < 8; ++i) __builtin_amdgcn_s_sleep(127);         // ~25 us
                continue;
            }
            PairDesc &desc = a.descs[pair];
            aln_pair_result &res = a.results[pair];
            const bool bad_shape = desc.status != ALN_OK;
            if (bad_shape || !pair_codes_ok(a.seqs, desc, a.rows, a.cols, a.pwm != 0, in.lane)) {
                skip_invalid(res, bad_shape ? desc.status : ALN_ERR_CODE_OUT_OF_RANGE, in.lane);
                pair_done(a, in.lane, pair, false);
                if (COOP && cp.ctl && in.lane == 0) __hip_atomic_fetch_add(cp.ctl + ALN_COOP_FINISHED, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                continue;
            }
            set_wave_priority((uint64_t)desc.N * desc.M, a.max_cells);
        }
        idle_since = 0;
#ifdef ALN_STAMPS                                        // tools/tail_timeline.py: when did which wave work on this pair (score-only runs)
        const uint64_t ts = wall_clock64();
#endif
        // (one call for both: the strips below strip 0 are the same code whoever runs them)
        const bool plain = fast_work<SEM, PWM, COOP>(in, fs, cp, a, COOP && helper, w, pair, qpos, epoch, (int)a.del, (int)a.ext);
        if (helper) continue;
#ifdef ALN_STAMPS
        if (in.lane == 0) { aln_pair_result &res = a.results[pair]; res.aln_len = (uint32_t)ts; res.start_x = (uint32_t)wall_clock64(); res.start_y = wave; }
#endif
        pair_done(a, in.lane, pair, plain);
        if (COOP && cp.ctl && in.lane == 0) __hip_atomic_fetch_add(cp.ctl + ALN_COOP_FINISHED, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------- generic kernels, one workgroup per pair (wg_strip)
template <typename SC, int SEM, int R>
__global__ __launch_bounds__(1024) void aln_fill_wgpipe_kernel(WgArgs a)
{
    using O = ScOps<SC>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    PairDesc &d = a.descs[a.pair];
    aln_pair_result &res = a.results[a.pair];
    const int lane = threadIdx.x & 63;
    const uint32_t wave = threadIdx.x >> 6, ns = a.ns, N = d.N, M = d.M;
    // [S][rings][advice][zrow][prod 16][cons 17][flags 4][candidates]
    SC *S = reinterpret_cast<SC *>(smem);
    WgShared<SC> sh;
    sh.S = S;
    unsigned char *p = smem + ((a.rows * a.cols * sizeof(SC) + 15u) & ~15u);
    sh.rings = reinterpret_cast<SC *>(p); p += (size_t)ns * ALN_WG_RING * sizeof(SC);
    const uint32_t adv_bytes = (N + 66u + 15u) & ~15u;
    sh.advice = p; p += adv_bytes;
    sh.zrow = p; p += adv_bytes;
    sh.row1tag = p; p += adv_bytes;
    sh.row1 = reinterpret_cast<SC *>(a.scratch);
    sh.prod = reinterpret_cast<uint32_t *>(p); sh.cons = sh.prod + 20; sh.flags = sh.prod + 44; p += 64 * 4;
    unsigned char *cand = p;                            // per wave: SC value, SC corner, by, bx (32 bytes)
    if (N <= 2048u) sh.row1 = reinterpret_cast<SC *>(cand + ns * 32u);      // (aln_wg_lds_bytes)
    if (d.status != ALN_OK) { skip_invalid(res, d.status, (int)threadIdx.x); return; }
    {   // residue codes outside the matrix: the reference panics (simple/mod.rs:85,198)
        const uint8_t *q = a.seqs + d.q_off, *t = a.seqs + d.t_off;
        uint32_t worst_q = 0, worst_t = 0;
        if (!a.pwm) for (uint32_t i = threadIdx.x; i < N; i += blockDim.x) worst_q = max(worst_q, (uint32_t)q[i]);
        for (uint32_t i = threadIdx.x; i < M; i += blockDim.x) worst_t = max(worst_t, (uint32_t)t[i]);
        if (__syncthreads_or(worst_q >= a.cols || worst_t >= a.rows)) { skip_invalid(res, ALN_ERR_CODE_OUT_OF_RANGE, (int)threadIdx.x); return; }
    }
    const SC *gm = reinterpret_cast<const SC *>(a.matrix);
    for (uint32_t i = threadIdx.x; i < a.rows * a.cols; i += blockDim.x) S[i] = gm[i];
    const SC del = O::from_double(a.del), ext = O::from_double(a.ext);
    const bool hazard = (SEM == ALN_CORE_LOCAL) && (del != ext) && N >= 2;
    for (uint32_t x = threadIdx.x; x < adv_bytes; x += blockDim.x) { sh.advice[x] = 0; sh.zrow[x] = 0; }
    if (a.hmat != nullptr) {                             // borders of the optional H dump (simple/mod.rs:55-70)
        SC *hm = reinterpret_cast<SC *>(a.hmat) + d.h_off;
        for (uint32_t x = threadIdx.x; x <= N; x += blockDim.x) hm[x] = border_top<SC, SEM>(x, N, del);
        for (uint32_t y = threadIdx.x; y <= M; y += blockDim.x) hm[(size_t)y * (N + 1)] = border_left<SC, SEM>(y, M, del);
    }
    if (threadIdx.x < 64) sh.prod[threadIdx.x] = 0;     // prod, cons and flags
    __syncthreads();

    const uint32_t max_passes = a.max_passes ? a.max_passes : 4u;
    uint32_t passes = 0;
    bool converged = false;
    SC bv = O::lowest(), corner = (SC)0;
    uint32_t by = 0, bx = 0;
    for (;;) {
        bv = (SEM == ALN_LEGACY_LOCAL) ? (SC)-1 : O::lowest(); by = 0; bx = 0;
        wg_strip<SC, SEM, R>(a, d, sh, wave, wave + 1 == ns, hazard, lane, bv, by, bx, corner);
        ++passes;
        __syncthreads();
        const bool aborted = sh.flags[0] != 0;
        if (aborted) break;
        if (!hazard) { converged = true; break; }
        int same = 0;
        const bool unchanged = adopt_advice_checked<SC, SEM>(sh.advice, sh.zrow, sh.row1, sh.row1tag, S, a.cols, a.seqs + d.q_off, a.seqs + d.t_off,
                                                             a.pwm != 0, del, ext, N, (int)threadIdx.x, (int)blockDim.x, same);
        const int need = __syncthreads_or(!unchanged), moved = __syncthreads_or(!same);
        converged = !need || !moved;                     // the advice stands, or changing it moves no cell of row 1
        if (converged || passes >= max_passes) break;
        if (threadIdx.x < 64) sh.prod[threadIdx.x] = 0;
        __syncthreads();
    }
    if (!converged) {                                    // strict reference order (exact for every input), one lane
        if (threadIdx.x == 0) {
            Wave<SC> c;
            c.lane = 0; c.N = N; c.M = M; c.q = a.seqs + d.q_off; c.t = a.seqs + d.t_off; c.S = S; c.cols = a.cols; c.del = del; c.ext = ext;
            c.dirw = reinterpret_cast<uint32_t *>(a.dirs + d.dir_off); c.brow = reinterpret_cast<SC *>(a.scratch);
            c.hmat = a.hmat ? reinterpret_cast<SC *>(a.hmat) + d.h_off : nullptr;
            c.store_dirs = a.store_dirs != 0; c.pwm = a.pwm != 0; c.bv = O::lowest(); c.by = 0; c.bx = 0; c.corner = (SC)0;
            serial_fill_impl<SC, SEM>(c);
            d.layout = ALN_LAYOUT_ROWMAJOR;
            write_result<SEM>(res, (double)c.bv, c.by, c.bx, (double)c.corner, N, M, passes | 0x80u, (sizeof(SC) == 4 ? 1u : 0u) | 4u, a.pwm != 0);
        }
        return;
    }
    // end cell: per wave a butterfly with the exact tie rule, then the waves' candidates through LDS
    if (is_local<SEM>()) {
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const SC ov = O::xshfl(bv, m);
            const uint32_t oy = (uint32_t)__shfl_xor((int)by, m), ox = (uint32_t)__shfl_xor((int)bx, m);
            if (ox != 0 && (bx == 0 || better<SC, SEM>(ov, oy, ox, bv, by, bx))) { bv = ov; by = oy; bx = ox; }
        }
    }
    if (lane == 0) {
        double *cd = reinterpret_cast<double *>(cand + wave * 32u);
        uint32_t *cu = reinterpret_cast<uint32_t *>(cand + wave * 32u + 16u);
        cd[0] = (double)bv; cd[1] = (double)corner; cu[0] = by; cu[1] = bx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double best = 0.0, cor = 0.0;
        uint32_t fy = 0, fx = 0;
        for (uint32_t w = 0; w < ns; ++w) {
            const double *cd = reinterpret_cast<const double *>(cand + w * 32u);
            const uint32_t *cu = reinterpret_cast<const uint32_t *>(cand + w * 32u + 16u);
            if (is_local<SEM>() && cu[1] != 0 && (fx == 0 || better<double, SEM>(cd[0], cu[0], cu[1], best, fy, fx))) { best = cd[0]; fy = cu[0]; fx = cu[1]; }
            if (w + 1 == ns) cor = cd[1];
        }
        if (is_local<SEM>() && fx == 0) best = (SEM == ALN_LEGACY_LOCAL) ? -1.0 : -DBL_MAX;
        d.layout = ALN_LAYOUT_UNIFORM | (a.R << 8);
        write_result<SEM>(res, best, fy, fx, cor, N, M, passes, (sizeof(SC) == 4 ? 1u : 0u) | 4u, a.pwm != 0);
    }
}

// ---------------------------------------------------------------- single-pair kernel: one wave per strip
// Grid = number of strips; strip s consumes the granule row strip s-1 publishes 16 columns at a time, so the strips
// form a software pipeline across CUs (lag per strip ~ 64 + 63 steps + one L2 round trip).  Every wave that waits
// polls a bounded number of times and then poisons the run (ctrl[0]) instead of hanging.
// W = 4 (core local, R <= 2, LDS permitting): the four waves of a workgroup own four consecutive strips, one per SIMD,
// and hand their bottom rows over through LDS rings (a hop costs ~100 cycles instead of ~2 us); only every fourth hop
// goes through the granule rows.  LDS: [W-1 rings, 16 KiB each, size-aligned] [S] [qo_pad, shared] [per wave: profile,
// boundary ring of the C++ step].
template <int SEM, int R, int W>
__global__ __launch_bounds__(64 * W) void aln_fill_single_kernel(SingleArgs a)
{
    const bool repair = a.mode == 1;             // the repair run: the first rep_S strips, their leading columns, gated by ctrl[8]
    if (__hip_atomic_load(a.ctrl + (repair ? 8u : 1u + a.pass), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
    if (repair && blockIdx.x * W >= a.rep_S) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr uint32_t RING_BYTES = 4u * ALN_RING;
    unsigned char *base = smem + (W - 1) * RING_BYTES;
    int *S = reinterpret_cast<int *>(base);
    const int *gm = reinterpret_cast<const int *>(a.matrix);
    for (uint32_t i = threadIdx.x; i < a.rows * a.cols; i += blockDim.x) S[i] = gm[i];
    for (uint32_t i = threadIdx.x; i < (W - 1) * ALN_RING; i += blockDim.x) reinterpret_cast<uint32_t *>(smem)[i] = 0;
    const PairDesc &desc = a.descs[a.pair];
    const uint32_t s_bytes = (a.rows * a.cols * 4u + 15u) & ~15u;
    const uint32_t prof_bytes = (a.cols * 64u * R + 15u) & ~15u;
    const uint32_t qo_bytes = ((desc.N + 192u) * 2u + 15u) & ~15u;
    uint16_t *qo_pad = reinterpret_cast<uint16_t *>(base + s_bytes);
    const uint8_t *qseq = a.seqs + desc.q_off;
    // query offsets (q[x] * 64R at index x + 63, zero padded), shared by the waves
    for (uint32_t i = threadIdx.x; i < desc.N + 192u; i += blockDim.x) {
        const uint32_t x = i - 63u;
        qo_pad[i] = (i >= 63u && x < desc.N) ? (uint16_t)((uint32_t)qseq[x] * 64u * R) : (uint16_t)0;
    }
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t strip = blockIdx.x * W + wave;
    if (strip >= a.ns) return;
    if (a.test_drop != 0 && strip + 1 == a.test_drop) return;       // fault injection, see SingleArgs
    FastIn in;
    in.lane = threadIdx.x & 63;
    in.fair = 0; in.ck_last = 512u;
    in.N = desc.N; in.M = desc.M;
    in.q = qseq;
    in.t = a.seqs + desc.t_off;
    in.S = S; in.cols = a.cols;
    in.nd4 = -4 * (int)a.del; in.ne4 = -4 * (int)a.ext;
    in.prof = base + s_bytes + qo_bytes + wave * (prof_bytes + 512u);
    in.lds_scratch = ((uint32_t)(uintptr_t)(base + s_bytes + qo_bytes + W * (prof_bytes + 512u)) + 255u) & ~255u;   // 512 B, 256-aligned
    in.dirw = reinterpret_cast<uint32_t *>(a.dirs + desc.dir_off);
    in.brow_in = nullptr; in.brow_out = nullptr; in.ckpt = nullptr;
    in.advice = a.advice; in.zrow = a.zrow;
    in.hazard = a.hazard != 0;
    in.adv_any = a.hazard != 0 && a.pass != 0;
    in.store_dirs = a.store_dirs != 0;
    in.pwm = false; in.pwm_words = nullptr;
    in.ck_mode = 0; in.last_flip = 0; in.ck_stop = 0;
    const bool last = strip + 1 == a.ns;
    uint32_t *gbase = repair ? a.rgranules : a.granules;
    in.gin = gbase + (uint64_t)(strip ? strip - 1 : 0) * a.gstride;
    in.gout = gbase + (uint64_t)strip * a.gstride;
    if (SEM == ALN_CORE_LOCAL && a.rep_S != 0 && strip < a.rep_S && (repair || a.pass == 0)) {
        in.ckpt = a.ckpt + (size_t)strip * (18 * 64);
        in.ck_stop = a.rep_K + 64u * (a.rep_S - 1u - strip);
        in.ck_mode = repair ? 2 : 1;
    }
    if (repair) { if (strip >= a.rep_S) return; in.adv_any = true; }
    in.ring_in = (W > 1 && wave > 0) ? reinterpret_cast<uint32_t *>(smem + (wave - 1) * RING_BYTES) : nullptr;
    in.ring_out = (W > 1 && wave + 1 < W && !last) ? reinterpret_cast<uint32_t *>(smem + wave * RING_BYTES) : nullptr;
    in.abort_flag = a.ctrl;
    FastOut o;
    o.bv = INT_MIN; o.by = 0; o.bx = 0; o.corner = 0; o.repaired = false; o.brow_bad = false; o.aborted = false; o.ck_slot = 0;
    in.qo_pad = qo_pad;
    in.bring = reinterpret_cast<int *>(in.prof + prof_bytes);
    // (after a repair run: the candidates of the re-run prefix and of the prefix pass 0 had computed, for the merge)
    auto run = [&](auto &fs) {
        o = fs.run(o);
        if (repair) {
            FastOut nw = o, od = o;
            nw.bv = INT_MIN; nw.by = 0; nw.bx = 0; od.bv = INT_MIN; od.by = 0; od.bx = 0;
            fs.fold(nw, 0);
            fs.fold_saved(od, 0);
            reduce_best<SEM>(nw);
            reduce_best<SEM>(od);
            if (in.lane == 0) {
                int32_t *c = a.rcand + 8 * strip;
                c[0] = nw.bv; c[1] = (int32_t)nw.by; c[2] = (int32_t)nw.bx;
                c[3] = od.bv; c[4] = (int32_t)od.by; c[5] = (int32_t)od.bx;
                c[6] = (o.repaired && !o.aborted) ? 1 : 0; c[7] = 0;
            }
        }
    };
    if (strip == 0) {
        if (last) { FastStrip<SEM, R, true, true, true> fs(in, strip); run(fs); }
        else { FastStrip<SEM, R, true, true, false> fs(in, strip); run(fs); }
    } else {
        if (last) { FastStrip<SEM, R, true, false, true> fs(in, strip); run(fs); }
        else { FastStrip<SEM, R, true, false, false> fs(in, strip); run(fs); }
    }
    if (repair) return;
    if (is_local<SEM>()) reduce_best<SEM>(o);
    if (in.lane == 0) {
        int32_t *c = a.cand + 4 * strip;
        c[0] = o.bv; c[1] = (int32_t)o.by; c[2] = (int32_t)o.bx; c[3] = o.corner;
    }
}

// One block of 1024 threads: folds the per-strip candidates, checks the row-1 advice against the bottom row this pass
// produced and either publishes the result or adopts the observed zeros as the new advice, zeroes the granule rows and
// arms the next pass (ctrl[1 + pass + 1]) / the serial fallback (ctrl[15]).  (Zeroing here rather than with a memset
// per pass keeps the launches of the passes that turn out not to be needed down to two early-exit kernels each.)
template <int SEM>
__global__ __launch_bounds__(1024) void aln_single_finalize_kernel(SingleArgs a)
{
    if (__hip_atomic_load(a.ctrl + 1 + a.pass, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
    const uint32_t tid = threadIdx.x;
    const int lane = (int)(tid & 63u);
    PairDesc &desc = a.descs[a.pair];
    aln_pair_result &res = a.results[a.pair];
    const uint32_t N = desc.N, M = desc.M;
    // H[M][x] == 0 <=> the stored tag of cell (M, x) is 3; the last strip recorded the direction words of the lane that
    // owns row M, one per block
    const uint32_t rows_last = M - (a.ns - 1) * 64u * a.R;
    const uint32_t lb = (rows_last - 1) / a.R, rb = (rows_last - 1) % a.R, spb = aln_spb(a.R);
    const uint32_t *zdw = reinterpret_cast<const uint32_t *>(a.zrow);
    auto bottom_zero = [&](uint32_t x) -> uint8_t {                 // x = 1 .. N
        const uint32_t k = x - 1 + lb;
        return ((zdw[k / spb] >> aln_dir_bitpos(k, rb, lb, N, (int)a.R)) & 3u) == 3u ? 1 : 0;
    };
    int mismatch = 0;
    __shared__ uint32_t far_flip;                  // does the advice change beyond the columns a repair run can absorb?
    if (tid == 0) far_flip = 0;
    __syncthreads();
    if (a.hazard) for (uint32_t x = 2 + tid; x <= N; x += blockDim.x)
        if (a.advice[x] != bottom_zero(x - 1)) { mismatch = 1; if (x > a.rep_K / 2u) far_flip = 1; }
    const bool again = __syncthreads_or(mismatch) != 0;
    const bool aborted = a.ctrl[0] != 0;
    if (again && !aborted) {
        for (uint32_t x = 2 + tid; x <= N; x += blockDim.x) a.advice[x] = bottom_zero(x - 1);
        if (a.pass == 0 && a.rep_S != 0 && far_flip == 0 && a.max_passes >= 2) {
            // Only leading columns of row 1 change: arm the repair run instead of a second pass of the whole pipeline.
            // Its granule rows must read "not yet produced".
            uint4 *g = reinterpret_cast<uint4 *>(a.rgranules);
            const uint64_t n16 = (uint64_t)a.rep_S * a.gstride / 4u;
            for (uint64_t i = tid; i < n16; i += blockDim.x) g[i] = make_uint4(0, 0, 0, 0);
            __threadfence();
            __syncthreads();
            if (tid == 0) a.ctrl[8] = 1;
            return;
        }
        if (a.pass + 1 < a.max_passes) {
            uint4 *g = reinterpret_cast<uint4 *>(a.granules);       // the next pass needs "not yet produced" everywhere
            const uint64_t n16 = (uint64_t)a.ns * a.gstride / 4u;   // gstride is a multiple of 64
            for (uint64_t i = tid; i < n16; i += blockDim.x) g[i] = make_uint4(0, 0, 0, 0);
        }
        __threadfence();
        __syncthreads();
        if (tid == 0) {
            if (a.pass + 1 < a.max_passes) a.ctrl[1 + a.pass + 1] = 1;
            else a.ctrl[15] = a.pass + 1;               // not self-consistent within the cap: strict-order kernel
        }
        return;
    }
    if (tid >= 64) return;
    FastOut o;
    o.bv = INT_MIN; o.by = 0; o.bx = 0; o.corner = 0; o.repaired = false; o.brow_bad = false; o.aborted = false; o.ck_slot = 0;
    for (uint32_t s = lane; s < a.ns; s += 64) {
        const int32_t *c = a.cand + 4 * s;
        if (c[2] != 0 && (o.bx == 0 || better_i<SEM>(c[0], (uint32_t)c[1], (uint32_t)c[2], o.bv, o.by, o.bx))) { o.bv = c[0]; o.by = c[1]; o.bx = c[2]; }
    }
    reduce_best<SEM>(o);
    if (lane == 0) {
        const int corner = a.cand[4 * (a.ns - 1) + 3] >> 2;
        desc.layout = ALN_LAYOUT_UNIFORM | (a.R << 8);
        write_result<SEM>(res, (double)(o.bv >> 2), o.by, o.bx, (double)corner, N, M, a.pass + 1, 1u | 2u);
        if (aborted) res.status = ALN_ERR_DEVICE;
    }
}

// After the repair run (gated by ctrl[8]).  The repair is valid iff every re-run strip rejoined its checkpointed lane state,
// the bottom row of the last of them -- what the first strip NOT re-run reads -- came out as pass 0 had it, and no strip's
// end-cell candidate stems from a prefix cell that the re-run no longer produces.  Then the result is published with the
// re-run prefixes' candidates merged in (the directions of those columns have been rewritten in place); otherwise pass 1 of
// the whole pipeline is armed, as the finalize kernel would have done.
template <int SEM>
__global__ __launch_bounds__(1024) void aln_single_repair_finalize_kernel(SingleArgs a)
{
    if (__hip_atomic_load(a.ctrl + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
    const uint32_t tid = threadIdx.x;
    const int lane = (int)(tid & 63u);
    PairDesc &desc = a.descs[a.pair];
    aln_pair_result &res = a.results[a.pair];
    const uint32_t S = a.rep_S;
    int bad = 0;
    if (tid < S) {
        const int32_t *c = a.rcand + 8 * tid, *w = a.cand + 4 * tid;
        if (c[6] == 0) bad = 1;
        // pass 0's winner of this strip came from the old prefix, and the new prefix does not have that cell any more
        const bool old_leads = c[5] != 0 && c[3] == w[0] && c[4] == w[1] && c[5] == w[2];
        const bool same = c[0] == c[3] && c[1] == c[4] && c[2] == c[5];
        if (old_leads && !same) bad = 1;
    }
    // bottom row of strip S - 1, columns the repair run recomputed (its last strip stopped at step rep_K: lane 63 at column rep_K - 63)
    const uint32_t *gn = a.rgranules + (uint64_t)(S - 1) * a.gstride, *go = a.granules + (uint64_t)(S - 1) * a.gstride;
    for (uint32_t x = tid; x + 63u < a.rep_K && x < desc.N; x += blockDim.x) if (gn[x] != go[x]) bad = 1;
    // (test_drop == ~0: fault injection, the repair is declared failed -- pass 1 must then run and give the same answer)
    const bool failed = __syncthreads_or(bad) != 0 || a.ctrl[0] != 0 || a.test_drop == 0xffffffffu;
    if (failed) {
        uint4 *g = reinterpret_cast<uint4 *>(a.granules);           // pass 1 needs "not yet produced" everywhere
        const uint64_t n16 = (uint64_t)a.ns * a.gstride / 4u;
        for (uint64_t i = tid; i < n16; i += blockDim.x) g[i] = make_uint4(0, 0, 0, 0);
        __threadfence();
        __syncthreads();
        if (tid == 0) { a.ctrl[8] = 0; a.ctrl[2] = 1; }
        return;
    }
    if (tid >= 64) return;
    FastOut o;
    o.bv = INT_MIN; o.by = 0; o.bx = 0; o.corner = 0; o.repaired = false; o.brow_bad = false; o.aborted = false; o.ck_slot = 0; o.c_out = 0;
    for (uint32_t s = lane; s < a.ns; s += 64) {
        const int32_t *c = a.cand + 4 * s;
        if (c[2] != 0 && (o.bx == 0 || better_i<SEM>(c[0], (uint32_t)c[1], (uint32_t)c[2], o.bv, o.by, o.bx))) { o.bv = c[0]; o.by = c[1]; o.bx = c[2]; }
        if (s < S) {
            const int32_t *n = a.rcand + 8 * s;
            if (n[2] != 0 && (o.bx == 0 || better_i<SEM>(n[0], (uint32_t)n[1], (uint32_t)n[2], o.bv, o.by, o.bx))) { o.bv = n[0]; o.by = n[1]; o.bx = n[2]; }
        }
    }
    reduce_best<SEM>(o);
    if (lane == 0) {
        const int corner = a.cand[4 * (a.ns - 1) + 3] >> 2;
        desc.layout = ALN_LAYOUT_UNIFORM | (a.R << 8);
        write_result<SEM>(res, (double)(o.bv >> 2), o.by, o.bx, (double)corner, desc.N, desc.M, 1u | 0x100u, 1u | 2u);
        a.ctrl[8] = 0;
    }
}

// Strict reference order for one (large) pair; runs only when ctrl[15] is armed.  scratch holds M + 1 ints.
template <int SEM>
__global__ __launch_bounds__(64) void aln_single_serial_kernel(SingleArgs a)
{
    if (a.ctrl[15] == 0) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *S = reinterpret_cast<int *>(smem);
    const int *gm = reinterpret_cast<const int *>(a.matrix);
    for (uint32_t i = threadIdx.x; i < a.rows * a.cols; i += blockDim.x) S[i] = gm[i];
    __syncthreads();
    if (threadIdx.x != 0) return;
    PairDesc &desc = a.descs[a.pair];
    aln_pair_result &res = a.results[a.pair];
    Wave<int> w;
    w.lane = 0; w.N = desc.N; w.M = desc.M;
    w.q = a.seqs + desc.q_off; w.t = a.seqs + desc.t_off;
    w.S = S; w.cols = a.cols; w.del = (int)a.del; w.ext = (int)a.ext;
    w.dirw = reinterpret_cast<uint32_t *>(a.dirs + desc.dir_off);
    w.brow = reinterpret_cast<int *>(a.granules);
    w.hmat = nullptr; w.store_dirs = a.store_dirs != 0; w.pwm = false;
    serial_fill_impl<int, SEM>(w);
    desc.layout = ALN_LAYOUT_ROWMAJOR;
    write_result<SEM>(res, (double)w.bv, w.by, w.bx, (double)w.corner, desc.N, desc.M, a.ctrl[15] | 0x80u, 1u | 2u);
}

// arms pass 0 and clears the advice / bottom-row bytes.  A pair the validation kernel rejected (a residue code outside the
// matrix) is not armed: every later kernel of the route returns at once, and the summary carries the status.
#if ALN_TU & ALN_PART_SINGLE
extern "C" __global__ void aln_single_init_kernel(SingleArgs a, uint32_t n_bytes)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const int st = a.descs[a.pair].status;
    if (i < n_bytes) { a.advice[i] = 0; a.zrow[i] = 0; }
    if (i < 16) a.ctrl[i] = (i == 1 && st == ALN_OK) ? 1u : 0u;
    if (i == 0 && st != ALN_OK) skip_invalid(a.results[a.pair], st, 0);
}

#endif

#if ALN_TU & ALN_PART_GENERIC
// ---------------------------------------------------------------- residue-code validation
// The reference indexes the matrix with every residue (simple/mod.rs:85,198) and panics on a code outside it; here one wave
// per pair scans both sequences and turns that into the pair's status (ALN_ERR_CODE_OUT_OF_RANGE).  The batch fill kernels do
// this themselves (pair_codes_ok); this kernel runs in front of the single-pair route, whose kernels take the status from
// the descriptor.
extern "C" __global__ __launch_bounds__(256) void aln_validate_kernel(const uint8_t *seqs, PairDesc *descs, uint32_t n_pairs,
                                                                      uint32_t rows, uint32_t cols, uint32_t pwm)
{
    const uint32_t pair = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (pair >= n_pairs) return;
    const uint32_t lane = threadIdx.x & 63u;
    PairDesc &d = descs[pair];
    if (d.status != ALN_OK) return;
    const uint8_t *q = seqs + d.q_off, *t = seqs + d.t_off;
    uint32_t bad = 0;
    if (!pwm) for (uint32_t i = lane; i < d.N; i += 64u) bad |= (q[i] >= cols) ? 1u : 0u;
    for (uint32_t i = lane; i < d.M; i += 64u) bad |= (t[i] >= rows) ? 1u : 0u;
    if (__any(bad != 0) && lane == 0) d.status = ALN_ERR_CODE_OUT_OF_RANGE;
}

#endif

#if ALN_TU & ALN_PART_TB
// ---------------------------------------------------------------- traceback
// Direction of cell (y, x) from the packed store (borders are implicit: simple/mod.rs:55-67).
__device__ __forceinline__ int dir_at(const uint8_t *dirs, const PairDesc &d, bool global, uint32_t y, uint32_t x)
{
    if (y == 0 || x == 0) {
        if (!global || (y == 0 && x == 0)) return D_BEG;
        return y == 0 ? D_LEFT : D_TOP;
    }
    const uint8_t *base = dirs + d.dir_off;
    if (d.layout == ALN_LAYOUT_ROWMAJOR) {
        const uint32_t rowbytes = (d.N + 4) / 4;
        return (base[(size_t)y * rowbytes + (x >> 2)] >> (2 * (x & 3u))) & 3;
    }
    uint32_t strip, i;
    int R;
    uint64_t strip_bytes;
    if ((d.layout & 0xffu) == ALN_LAYOUT_UNIFORM || (d.layout & 0xffu) == ALN_LAYOUT_UBATCH) {
        R = (int)((d.layout >> 8) & 0xffu);
        strip = (y - 1) / (64u * R);
        i = (y - 1) - strip * 64u * R;
        strip_bytes = aln_uniform_strip_bytes(d.N, (uint32_t)R);
    } else {
        strip = (y - 1) / ALN_STRIP_ROWS;
        const uint32_t ns = aln_num_strips(d.M);
        R = (strip + 1 == ns) ? aln_pick_r(d.M - strip * ALN_STRIP_ROWS) : ALN_FULL_R;
        i = (y - 1) - strip * ALN_STRIP_ROWS;
        strip_bytes = aln_strip_bytes(d.N);
    }
    const uint32_t lane = i / R, r = i % R;
    const uint32_t k = (x - 1) + lane;
    const uint32_t spb = aln_spb((uint32_t)R);
    const uint32_t *wbase = reinterpret_cast<const uint32_t *>(base + strip * strip_bytes);
    const uint32_t word = wbase[aln_dir_word_index(k, lane, spb)];
    return aln_tag_to_dir((int)((word >> aln_dir_bitpos(k, r, lane, d.N, R)) & 3u));
}

// Per-strip constants of the packed direction store, re-derived only when the walk leaves the strip.
struct StripView {
    const uint32_t *wbase;
    uint32_t y0;        // rows of the strip are y0+1 .. y0+rows
    uint32_t R;         // rows per lane (1..8; the uniform layout of the single-pair route: a power of two)
    uint32_t lgS;       // log2(steps per direction word)
};
__device__ __forceinline__ StripView strip_view(const uint8_t *dirs, const PairDesc &d, uint32_t y)
{
    StripView v;
    const uint8_t *base = dirs + d.dir_off;
    if ((d.layout & 0xffu) == ALN_LAYOUT_UNIFORM || (d.layout & 0xffu) == ALN_LAYOUT_UBATCH) {
        v.R = (d.layout >> 8) & 0xffu;
        const uint32_t strip = (y - 1) / (64u * v.R);
        v.y0 = strip * 64u * v.R;
        v.wbase = reinterpret_cast<const uint32_t *>(base + strip * aln_uniform_strip_bytes(d.N, v.R));
    } else {
        const uint32_t strip = (y - 1) / ALN_STRIP_ROWS;
        const uint32_t ns = aln_num_strips(d.M);
        v.R = (strip + 1 == ns) ? (uint32_t)aln_pick_r(d.M - strip * ALN_STRIP_ROWS) : (uint32_t)ALN_FULL_R;
        v.y0 = strip * ALN_STRIP_ROWS;
        v.wbase = reinterpret_cast<const uint32_t *>(base + strip * aln_strip_bytes(d.N));
    }
    v.lgS = 31u - (uint32_t)__builtin_clz(aln_spb(v.R));
    return v;
}

// One thread per pair: the reference's pointer chase (simple/mod.rs:99-130, :213-245; legacy :146-176, :232-261),
// including the duplicated seed pair.  Pass 1 is the dependent chain -- one direction word per step, shifts only, the
// 2-bit tags go to a scratch byte string; pass 2 turns the tags into the two aligned code strings, written in final
// (forward) order, with loads whose addresses do not depend on loaded data.
__device__ __forceinline__ void tb_walk_pair(const TraceArgs &a, uint32_t pair)
{
    const PairDesc &d = a.descs[pair];
    aln_pair_result &res = a.results[pair];
    if (res.status != ALN_OK) return;
    if ((d.layout & 0xffu) == ALN_LAYOUT_UNIFORM) return;      // large pairs: aln_tb_single_* kernels
    uint8_t *__restrict__ ops = a.tags + d.tag_off;
    const bool global = (a.semantics == ALN_CORE_GLOBAL || a.semantics == ALN_LEGACY_GLOBAL);
    const bool legacy = (a.semantics == ALN_LEGACY_GLOBAL || a.semantics == ALN_LEGACY_LOCAL);
    const uint32_t ey = res.end_y, ex = res.end_x, N = d.N;
    uint32_t cy = ey, cx = ex;
    if (legacy) { cy -= 1; cx -= 1; }   // legacy starts at the diagonal predecessor (aligner_core.rs:146-147, :232-233)
    uint32_t len = 0;                   // steps taken (the seed pair is added in pass 2)
    if (d.layout == ALN_LAYOUT_ROWMAJOR) {
        for (;;) {
            const int dd = dir_at(a.dirs, d, global, cy, cx);
            if (dd == D_BEG) break;
            ops[len++] = (uint8_t)aln_dir_to_tag(dd);
            cy -= (dd != D_LEFT); cx -= (dd != D_TOP);
        }
    } else {
        // The walk is one dependent chain: ~a quarter of a load per step (a lane's quad -- 4 blocks, 16 B -- covers 64/R
        // consecutive steps of R rows and stays in registers until the path leaves it) and the instructions between two
        // loads, which a lone wave issues at one per ~4.6 cycles.  So the step is kept short: the state is (row within the
        // strip, column - 1), both signed -- one sign test catches "left the strip upwards" and both borders.
        if (cy != 0 && cx != 0) {
            StripView sv = strip_view(a.dirs, d, cy);
            int iy = (int)(cy - 1 - sv.y0), cxm = (int)cx - 1;
            // rows per lane R is any of 1..8: lane = iy / R by multiplication (iy < 512), everything about the steps by shifts
            uint32_t R = sv.R, rinv = (65535u + R) / R, lgS = sv.lgS, qsh = lgS + 2u, bmask = (1u << lgS) - 1u;
            const uint4 *wq = reinterpret_cast<const uint4 *>(sv.wbase);
            const uint32_t Nm1 = N - 1u;
            uint32_t qcur = 0xffffffffu;
            uint4 quad = make_uint4(0, 0, 0, 0);
            uint8_t *p = ops;
            for (;;) {
                if ((iy | cxm) < 0) {                       // above the strip, or a border
                    cy = (uint32_t)(iy + 1) + sv.y0; cx = (uint32_t)(cxm + 1);
                    if (cy == 0 || cx == 0) break;
                    sv = strip_view(a.dirs, d, cy);
                    iy = (int)(cy - 1 - sv.y0);
                    R = sv.R; rinv = (65535u + R) / R; lgS = sv.lgS; qsh = lgS + 2u; bmask = (1u << lgS) - 1u;
                    wq = reinterpret_cast<const uint4 *>(sv.wbase);
                    qcur = 0xffffffffu;
                }
                const uint32_t lane = ((uint32_t)iy * rinv) >> 16, r = (uint32_t)iy - lane * R;
                const uint32_t k = (uint32_t)cxm + lane;
                const uint32_t qi = ((k >> qsh) << 6) + lane;
                if (qi != qcur) { quad = wq[qi]; qcur = qi; }
                // the cell's tag sits at bit 32 - 2 R (m + 1) + 2 r of its word, m = steps of this lane left in the block
                // after step k (aln_dir_bitpos); the word is number (k >> lgS) & 3 of the quad
                const uint32_t lend = lane + Nm1;
                const uint32_t m = min(k | bmask, lend) - k;
                const uint32_t j = (k >> lgS) & 3u;
                const uint64_t half = (j & 2u) ? (((uint64_t)quad.w << 32) | quad.z) : (((uint64_t)quad.y << 32) | quad.x);
                const uint32_t tag = (uint32_t)(half >> (32u * (j & 1u) + 32u - 2u * R * (m + 1u) + 2u * r)) & 3u;
                if (tag == 3u) {                            // Beginning
                    cy = (uint32_t)(iy + 1) + sv.y0; cx = (uint32_t)(cxm + 1);
                    break;
                }
                *p++ = (uint8_t)tag;
                iy -= (tag != 1u); cxm -= (tag != 2u);      // 0 Diagonal, 1 Left, 2 Top
            }
            len = (uint32_t)(p - ops);
        }
        if (global) {                                   // borders: D[0][x] = Left, D[y][0] = Top (simple/mod.rs:59-67)
            while (cx != 0 && cy == 0) { ops[len++] = 1; cx--; }
            while (cy != 0 && cx == 0) { ops[len++] = 2; cy--; }
        }
    }
    res.start_y = cy; res.start_x = cx;
    res.aln_len = len + (a.pwm ? 0u : 1u);              // + the duplicated seed pair (none for the PWM aligner)
}
extern "C" __global__ __launch_bounds__(64) void aln_traceback_kernel(TraceArgs a)
{
    const uint32_t pair = blockIdx.x * blockDim.x + threadIdx.x;
    if (pair >= a.n_pairs) return;
    if (a.walked && a.walked[pair] == a.epoch) return;  // the overlap kernel has been here
    // latency-bound and light on issue slots: ahead of the fill waves (of the next chunk) it shares a SIMD with
    __builtin_amdgcn_s_setprio(3);
    tb_walk_pair(a, pair);
}
// ---------------------------------------------------------------- one WAVE per pair: the walk of a batch of few, long pairs
// aln_traceback_kernel's walk waits for memory once per direction quad (a dependent 16-byte load every 4-8 steps, ~2 us when it comes
// from HBM: 3.7 ms for the 8400-step paths of 256 pairs of 4200 x 4200 -- as long as their fill); with one pair per lane that is
// the price of 64 walks in flight, with 256 pairs it is all there is.  Here a wave walks ONE pair and all its lanes fetch ahead:
// wave lane j keeps, for lane j of the strip the path is in, the three quads around the step at which a purely diagonal path from
// the current cell would enter that lane's rows -- one round of loads per strip (the path climbs through the lanes, 64 R rows), a new
// round when gaps have carried it out of the three quads (8+ steps either way).  The state of the walk is wave-uniform: the step
// runs on scalar registers, the quad it needs comes out of the window with v_readlane, the tags leave 64 at a time.
__device__ __forceinline__ uint32_t uni32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
template <class T>
__device__ __forceinline__ const T *uni_ptr(const T *p)
{
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    return reinterpret_cast<const T *>(((uint64_t)uni32((uint32_t)(v >> 32)) << 32) | uni32((uint32_t)v));
}
__device__ __forceinline__ StripView strip_view_uniform(const uint8_t *dirs, const PairDesc &d, uint32_t y)
{
    StripView v = strip_view(dirs, d, y);                    // (loads of the descriptor: the compiler cannot know they are uniform)
    v.wbase = uni_ptr(v.wbase); v.y0 = uni32(v.y0); v.R = uni32(v.R); v.lgS = uni32(v.lgS);
    return v;
}
__device__ __forceinline__ void tb_walk_pair_wave(const TraceArgs &a, const uint32_t pair)
{
    const PairDesc &d = a.descs[pair];
    aln_pair_result &res = a.results[pair];
    const uint32_t lane_id = threadIdx.x & 63u;
    if (__builtin_amdgcn_readfirstlane(res.status) != ALN_OK) return;
    const uint32_t layout = (uint32_t)__builtin_amdgcn_readfirstlane((int)d.layout);
    if ((layout & 0xffu) == ALN_LAYOUT_UNIFORM) return;      // large pairs of the single-pair route: aln_tb_single_* kernels
    if (layout == ALN_LAYOUT_ROWMAJOR) {                     // strict-order fallback: rare, one lane
        if (lane_id == 0) tb_walk_pair(a, pair);
        return;
    }
    uint8_t *__restrict__ ops = a.tags + d.tag_off;
    const bool global = (a.semantics == ALN_CORE_GLOBAL || a.semantics == ALN_LEGACY_GLOBAL);
    const bool legacy = (a.semantics == ALN_LEGACY_GLOBAL || a.semantics == ALN_LEGACY_LOCAL);
    const uint32_t N = (uint32_t)__builtin_amdgcn_readfirstlane((int)d.N);
    uint32_t cy = (uint32_t)__builtin_amdgcn_readfirstlane((int)res.end_y), cx = (uint32_t)__builtin_amdgcn_readfirstlane((int)res.end_x);
    if (legacy) { cy -= 1; cx -= 1; }
    uint32_t len = 0;
    uint32_t tagv = 0;                                       // lane i: the tag of i steps ago
    if (cy != 0 && cx != 0) {
        StripView sv = strip_view_uniform(a.dirs, d, cy);
        int iy = (int)(cy - 1 - sv.y0), cxm = (int)cx - 1;
        uint32_t R = sv.R, rinv = (65535u + R) / R, lgS = sv.lgS, qsh = lgS + 2u, bmask = (1u << lgS) - 1u;
        const uint4 *wq = reinterpret_cast<const uint4 *>(sv.wbase);
        const uint32_t Nm1 = N - 1u;
        uint32_t nq = ((N + 62u) >> qsh) + 1u;              // quad rows of a strip (steps 0 .. N + 62)
        uint4 wa = make_uint4(0, 0, 0, 0), wb = wa, wc = wa; // quad rows qrow + 1, qrow, qrow - 1 of strip lane `lane_id`
        uint32_t qrow = 0;
        bool have = false;                                   // the window belongs to the current strip
        uint32_t cur_lane = 0xffffffffu, cur_q = 0xffffffffu;
        uint32_t q0 = 0, q1 = 0, q2 = 0, q3 = 0;             // the current quad (uniform)
        for (;;) {
            if ((iy | cxm) < 0) {                            // above the strip, or a border
                cy = (uint32_t)(iy + 1) + sv.y0; cx = (uint32_t)(cxm + 1);
                if (cy == 0 || cx == 0) break;
                sv = strip_view_uniform(a.dirs, d, cy);
                iy = (int)(cy - 1 - sv.y0);
                R = sv.R; rinv = (65535u + R) / R; lgS = sv.lgS; qsh = lgS + 2u; bmask = (1u << lgS) - 1u;
                wq = reinterpret_cast<const uint4 *>(sv.wbase);
                nq = ((N + 62u) >> qsh) + 1u;
                have = false; cur_lane = 0xffffffffu;
            }
            const uint32_t lane = ((uint32_t)iy * rinv) >> 16, r = (uint32_t)iy - lane * R;
            const uint32_t k = (uint32_t)cxm + lane;
            const uint32_t qr = k >> qsh;
            if (lane != cur_lane || qr != cur_q) {           // (uniform) another quad: out of the window
                uint32_t dq = (uint32_t)__builtin_amdgcn_readlane((int)qrow, (int)lane) + 1u - qr;      // 0 / 1 / 2: wa / wb / wc
                if (!have || dq > 2u) {
                    // fetch ahead from here: strip lane j <= lane is entered after t = iy - (j R + R - 1) diagonal steps, at step
                    // k_j = (cxm - t) + j; its R rows take the path down to k_j - (R - 1): the middle quad holds k_j - R / 2
                    const int t = iy - (int)(lane_id * R + R - 1u);
                    const int kj = (cxm - (t > 0 ? t : 0)) + (int)lane_id - (int)(R >> 1);
                    qrow = (uint32_t)(kj > 0 ? kj : 0) >> qsh;
                    if (lane_id == lane) qrow = qr;          // (the lane the path is in: exactly the quad of this step)
                    const bool on = lane_id <= lane;
                    wa = wb = wc = make_uint4(0, 0, 0, 0);
                    if (on && qrow + 1u < nq) wa = wq[((size_t)(qrow + 1u) << 6) + lane_id];
                    if (on && qrow < nq) wb = wq[((size_t)qrow << 6) + lane_id];
                    if (on && qrow >= 1u) wc = wq[((size_t)(qrow - 1u) << 6) + lane_id];
                    have = true;
                    dq = 1u;
                }
                const int li = (int)lane;
                if (dq == 0u) { q0 = __builtin_amdgcn_readlane((int)wa.x, li); q1 = __builtin_amdgcn_readlane((int)wa.y, li); q2 = __builtin_amdgcn_readlane((int)wa.z, li); q3 = __builtin_amdgcn_readlane((int)wa.w, li); }
                else if (dq == 1u) { q0 = __builtin_amdgcn_readlane((int)wb.x, li); q1 = __builtin_amdgcn_readlane((int)wb.y, li); q2 = __builtin_amdgcn_readlane((int)wb.z, li); q3 = __builtin_amdgcn_readlane((int)wb.w, li); }
                else { q0 = __builtin_amdgcn_readlane((int)wc.x, li); q1 = __builtin_amdgcn_readlane((int)wc.y, li); q2 = __builtin_amdgcn_readlane((int)wc.z, li); q3 = __builtin_amdgcn_readlane((int)wc.w, li); }
                cur_lane = lane; cur_q = qr;
                asm volatile("" : "+s"(cur_lane), "+s"(cur_q));      // (scalars: say so)
            }
            const uint32_t lend = lane + Nm1;
            const uint32_t m = min(k | bmask, lend) - k;
            const uint32_t j = (k >> lgS) & 3u;
            const uint64_t half = (j & 2u) ? (((uint64_t)q3 << 32) | q2) : (((uint64_t)q1 << 32) | q0);
            const uint32_t tag = (uint32_t)(half >> (32u * (j & 1u) + 32u - 2u * R * (m + 1u) + 2u * r)) & 3u;
            if (tag == 3u) {                                 // Beginning
                cy = (uint32_t)(iy + 1) + sv.y0; cx = (uint32_t)(cxm + 1);
                break;
            }
            // the tags enter a lane shift register at lane 0 (one DPP move per step; the step counter stays out of vector code, or
            // the compiler moves the whole walk state there): after 64 steps lane i holds the tag of step 63 - i of the group
            tagv = (uint32_t)__builtin_amdgcn_update_dpp((int)tag, (int)tagv, 0x138, 0xf, 0xf, false);
            ++len;
            if ((len & 63u) == 0u) ops[len - 1u - lane_id] = (uint8_t)tagv;
            iy -= (tag != 1u); cxm -= (tag != 2u);           // 0 Diagonal, 1 Left, 2 Top
        }
    }
    if (lane_id < (len & 63u)) ops[len - 1u - lane_id] = (uint8_t)tagv;       // the last, partial group: lane i holds tag len - 1 - i
    if (global) {                                            // borders: D[0][x] = Left, D[y][0] = Top (simple/mod.rs:59-67)
        if (cy == 0 && cx != 0) { for (uint32_t i = lane_id; i < cx; i += 64u) ops[len + i] = 1; len += cx; cx = 0; }
        else if (cx == 0 && cy != 0) { for (uint32_t i = lane_id; i < cy; i += 64u) ops[len + i] = 2; len += cy; cy = 0; }
    }
    if (lane_id == 0) {
        res.start_y = cy; res.start_x = cx;
        res.aln_len = len + (a.pwm ? 0u : 1u);               // + the duplicated seed pair (none for the PWM aligner)
    }
}
extern "C" __global__ __launch_bounds__(256) void aln_traceback_wave_kernel(TraceArgs a)
{
    const uint32_t pair = uni32(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (pair >= a.n_pairs) return;
    if (a.walked && uni32(a.walked[pair]) == a.epoch) return;
    tb_walk_pair_wave(a, pair);
}
// The walks are latency-bound and the fill is issue-bound: this kernel runs BESIDE the fill kernel (second stream; the host
// leaves a few workgroup slots free for it) and walks the pairs in the order the fill finishes them: persistent waves, each
// claims the next 64 entries of the fill's completion queue and waits for them to appear.  A wave that waits too long
// gives up -- nothing depends on this kernel: aln_traceback_kernel runs after the fill and walks whatever is left, so a
// runtime that serializes the two kernels (profilers do) only loses the overlap.
extern "C" __global__ __attribute__((amdgpu_flat_work_group_size(64, 64), amdgpu_num_vgpr(16))) void aln_traceback_overlap_kernel(TraceArgs a)
{
    __builtin_amdgcn_s_setprio(3);       // latency-bound and light on issue slots: ahead of the fill waves it shares a SIMD with
    for (;;) {
        uint32_t base = 0;
        if (threadIdx.x == 0) base = __hip_atomic_fetch_add(a.head, 64u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (base >= a.n_order) return;
        // 64 consecutive entries of the completion queue: they fill within microseconds of each other
        const uint32_t my = base + threadIdx.x;
        uint32_t v = 0;
        const uint64_t t0 = wall_clock64();
        for (;;) {
            if (my < a.n_order && v == 0) v = __hip_atomic_load(a.doneq + my, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (__all(my >= a.n_order || v != 0)) break;
            __builtin_amdgcn_s_sleep(127);
            if (wall_clock64() - t0 > a.wait_ticks) return;
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);         // nothing cached here from before the entries appeared
        if (my < a.n_order) {
            tb_walk_pair(a, v - 1u);
            a.walked[v - 1u] = a.epoch;
        }
    }
}
// one wave that sleeps ~30 us: queued in front of the overlap kernel so that the fill kernel's workgroups are placed first
extern "C" __global__ __launch_bounds__(64) void aln_delay_kernel(uint32_t ticks)
{
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

// ---------------------------------------------------------------- parallel traceback of one large pair
// A window of a strip's packed direction words staged in LDS: quads [q_lo, q_lo + q_n) (a quad = 4 blocks x 64 lanes x
// 4 B = 1 KiB, contiguous in memory: aln_dir_word_index).  A walk moves up and left, i.e. towards smaller wave steps, so
// the window ends at the quad of the largest step the block's walks start from; whatever leaves it falls back to global.
struct StripWindow {
    const uint32_t *wbase;
    const uint32_t *lds;
    uint32_t q_lo, q_n;
};
// cooperative copy (16 B per thread, coalesced) of the window that ends at wave step k_hi; call from every thread
__device__ __forceinline__ StripWindow stage_window(const uint32_t *wbase, uint32_t *lds, uint32_t lds_quads, uint32_t k_hi,
                                                    uint32_t lgR)
{
    StripWindow w;
    const uint32_t sh = 4u - lgR;                                  // log2(steps per block)
    const uint32_t q_hi = k_hi >> (sh + 2);
    w.wbase = wbase; w.lds = lds;
    w.q_n = min(lds_quads, q_hi + 1u);
    w.q_lo = q_hi + 1u - w.q_n;
    const uint4 *src = reinterpret_cast<const uint4 *>(wbase) + (size_t)w.q_lo * 64u;
    uint4 *dst = reinterpret_cast<uint4 *>(lds);
    // blocks of 256 threads, at most 48 quads: 12 x 16 B per thread, all loads in flight before the first LDS write
    const uint32_t n = w.q_n * 64u;
    uint4 v[12];
#pragma unroll
    for (uint32_t j = 0; j < 12; ++j) {
        const uint32_t i = threadIdx.x + j * 256u;
        if (i < n) v[j] = src[i];
    }
#pragma unroll
    for (uint32_t j = 0; j < 12; ++j) {
        const uint32_t i = threadIdx.x + j * 256u;
        if (i < n) dst[i] = v[j];
    }
    __syncthreads();
    return w;
}

// Walks from (cy, cx) while the walk stays inside the strip whose rows are y0+1 .. (uniform-R layout), following the
// same rules as aln_traceback_kernel.  Returns true if the walk ended (Beginning / origin), false if it left the strip
// upwards (then cy == y0).  `ops` (optional) receives the tags.  fetch(kb, lane) returns the direction word of block kb
// of that lane.  The interior loop is branch-lean (one exit test, one load, shifts); the borders of the global
// semantics (D[0][x] = Left, D[y][0] = Top: simple/mod.rs:55-67) are whole runs and are handled after it.
// `cap` (exit maps only): a walk that has not left the strip after that many steps is given up (*gave_up): far from the real
// path the tags are long gap runs, and one thread following a 900-column Left run held up its whole block.
template <class Fetch>
__device__ __forceinline__ bool walk_in_strip(Fetch fetch, uint32_t y0, uint32_t lgR, uint32_t N, bool global,
                                              uint32_t &cy, uint32_t &cx, uint32_t &steps, uint8_t *ops,
                                              uint32_t cap = 0xffffffffu, bool *gave_up = nullptr)
{
    const uint32_t R = 1u << lgR, sh = 4u - lgR;
    uint32_t y = cy, x = cx, n = steps;
    const uint32_t n_stop = cap == 0xffffffffu ? cap : steps + cap;
    bool beginning = false;
    bool go = (y > y0 && x != 0);
    while (go) {                                       // single exit, body predicated: the loop is one exec-mask update
        const uint32_t i = y - 1 - y0;
        const uint32_t lane = i >> lgR, r = i & (R - 1u);
        const uint32_t k = x - 1 + lane, kb = k >> sh;
        const uint32_t word = fetch(kb, lane);
        const uint32_t e = min((kb << sh) + (1u << sh) - 1u, lane + N - 1u);
        const uint32_t tag = (word >> (30u - 2u * (((e - k) << lgR) + (R - 1u - r)))) & 3u;
        beginning = (tag == 3u);
        if (ops && !beginning) ops[n] = (uint8_t)tag;
        n += beginning ? 0u : 1u;
        y -= (tag == 0u || tag == 2u) ? 1u : 0u;       // 0 Diagonal, 1 Left, 2 Top (3 moves nothing)
        x -= (tag <= 1u) ? 1u : 0u;
        go = !beginning && y > y0 && x != 0 && n < n_stop;
    }
    if (gave_up) *gave_up = !beginning && y > y0 && x != 0;
    bool ended = beginning;
    if (!beginning) {
        if (!global) ended = (y == 0 || x == 0);                         // local: every border cell is Beginning
        else if (y == 0) {                                               // top border: Left all the way to the origin
            if (ops) for (uint32_t j = 0; j < x; ++j) ops[n + j] = 1;
            n += x; x = 0; ended = true;
        } else if (x == 0) {                                             // left border: Top until the strip is left
            const uint32_t run = y - y0;
            if (ops) for (uint32_t j = 0; j < run; ++j) ops[n + j] = 2;
            n += run; y = y0; ended = false;
        }
    }
    cy = y; cx = x; steps = n;
    return ended;
}
// LDS quads of the traceback kernels' window: 768 wave steps (48 KiB at most)
__host__ __device__ inline uint32_t tb_window_quads(uint32_t R) { const uint32_t q = 12u * R; return q > 48u ? 48u : q; }

// Exit maps are computed only where the path can plausibly enter a strip: a band of TB_BAND columns around the line of
// slope 1 through the traceback's start cell (insertions push the path left of it, deletions right).  The chain kernel
// uses a map entry when the real entry column falls into the band and walks the strip itself when it does not, so the
// band is a speed choice, never a correctness one.  Strips below the start cell need no map at all.
#define TB_BAND 1024u
__device__ __forceinline__ bool tb_band(uint32_t ey, uint32_t ex, uint32_t yb, uint32_t N, uint32_t &c_lo, uint32_t &c_hi)
{
    if (yb > ey) return false;                                   // the path never enters this strip from below
    const uint32_t dy = ey - yb;
    const uint32_t c = ex > dy ? ex - dy : 0u;                   // column of the slope-1 line on the strip's bottom row
    c_lo = c > TB_BAND / 2u ? c - TB_BAND / 2u : 0u;
    c_hi = min(N, c_lo + TB_BAND - 1u);
    return true;
}

// exit map: thread (x, s) enters strip s on its bottom row at column x; the block's 256 walks share one staged window
extern "C" __global__ __launch_bounds__(256) void aln_tb_single_maps_kernel(TraceSingleArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem_words[];
    const PairDesc &d = a.descs[a.pair];
    if (a.results[a.pair].status != ALN_OK || (d.layout & 0xffu) != ALN_LAYOUT_UNIFORM) return;   // serial fallback: aln_traceback_kernel
    const uint32_t s = blockIdx.y;
    const uint32_t lgR = 31u - (uint32_t)__builtin_clz(a.R), rows = 64u << lgR;
    const uint32_t y0 = s * rows;
    const bool global = (a.semantics == ALN_CORE_GLOBAL || a.semantics == ALN_LEGACY_GLOBAL);
    const bool legacy = (a.semantics == ALN_LEGACY_GLOBAL || a.semantics == ALN_LEGACY_LOCAL);
    const uint32_t *wbase = reinterpret_cast<const uint32_t *>(a.dirs + d.dir_off + s * aln_uniform_strip_bytes(d.N, a.R));
    const uint32_t cy0 = min(d.M, y0 + rows);
    uint32_t c_lo, c_hi;
    if (!tb_band(a.results[a.pair].end_y - (legacy ? 1u : 0u), a.results[a.pair].end_x - (legacy ? 1u : 0u), cy0, d.N, c_lo, c_hi)) return;
    if (c_lo + blockIdx.x * blockDim.x > c_hi) return;           // (uniform over the block)
    const uint32_t x = c_lo + blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t x_hi = min(c_hi, c_lo + blockIdx.x * blockDim.x + blockDim.x - 1u);
    const uint32_t k_hi = (x_hi ? x_hi - 1u : 0u) + ((cy0 - 1u - y0) >> lgR);
    const StripWindow w = stage_window(wbase, smem_words, tb_window_quads(a.R), k_hi, lgR);
    if (x > c_hi) return;
    uint32_t cy = cy0, cx = x, steps = 0;
    const uint32_t q_lo = w.q_lo, q_n = w.q_n;
    // LDS through an address-space-3 pointer: a generic pointer would turn the two loads into one flat_load with a select
    const __attribute__((address_space(3))) uint32_t *lds3 = (const __attribute__((address_space(3))) uint32_t *)smem_words;
    auto fetch = [&](uint32_t kb, uint32_t lane) -> uint32_t {
        const uint32_t q = kb >> 2, rel = q - q_lo;
        uint32_t word = lds3[((min(rel, q_n - 1u) * 64u + lane) << 2) + (kb & 3u)];            // ds_read, always in range
        if (__builtin_expect(rel >= q_n, 0)) word = wbase[(((uint64_t)q * 64u + lane) << 2) + (kb & 3u)];   // left the window (rare)
        return word;
    };
    // a path crosses the strip in rows .. rows + (gap columns) steps; beyond twice the rows + 64 the entry is marked unusable
    // (2) and the chain kernel, should the real path ever come this way, walks the strip itself
    bool gave_up = false;
    const bool stopped = walk_in_strip(fetch, y0, lgR, d.N, global, cy, cx, steps, nullptr, 2u * rows + 64u, &gave_up);
    a.map[(size_t)s * (d.N + 1) + x] = make_uint4(cx, cy, steps, gave_up ? 2u : stopped ? 1u : 0u);
}

// one thread: from the end cell through its own strip, then strip by strip through the maps
extern "C" __global__ void aln_tb_single_chain_kernel(TraceSingleArgs a)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const PairDesc &d = a.descs[a.pair];
    aln_pair_result &res = a.results[a.pair];
    for (uint32_t s = 0; s < a.ns; ++s) a.seg[s] = make_uint4(0, 0, 0, 0);
    if (res.status != ALN_OK || (d.layout & 0xffu) != ALN_LAYOUT_UNIFORM) return;
    const uint32_t lgR = 31u - (uint32_t)__builtin_clz(a.R), rows = 64u << lgR;
    const bool global = (a.semantics == ALN_CORE_GLOBAL || a.semantics == ALN_LEGACY_GLOBAL);
    const bool legacy = (a.semantics == ALN_LEGACY_GLOBAL || a.semantics == ALN_LEGACY_LOCAL);
    uint32_t cy = res.end_y, cx = res.end_x;
    if (legacy) { cy -= 1; cx -= 1; }
    const uint32_t cy_start = cy, cx_start = cx;             // the cell the band of the exit maps is centred on
    uint32_t off = 0;
    bool stopped = (cy == 0 || cx == 0) && (!global || (cy == 0 && cx == 0));
    if (!stopped && cy == 0) {                       // only the top border is left (global)
        a.seg[0] = make_uint4(cy, cx, 0, 1);
        off = cx; cx = 0; stopped = true;
    }
    if (!stopped) {
        uint32_t s = (cy - 1) >> (6 + lgR);
        // first segment: from the end cell, which need not be on the strip's bottom row
        a.seg[s] = make_uint4(cy, cx, 0, 1);
        const uint32_t *wbase = reinterpret_cast<const uint32_t *>(a.dirs + d.dir_off + s * aln_uniform_strip_bytes(d.N, a.R));
        auto fetch = [&](uint32_t kb, uint32_t lane) -> uint32_t { return wbase[(((uint64_t)(kb >> 2) * 64u + lane) << 2) + (kb & 3u)]; };
        uint32_t steps = 0;
        stopped = walk_in_strip(fetch, s * rows, lgR, d.N, global, cy, cx, steps, nullptr);
        off = steps;
        const uint32_t ey = cy_start, ex = cx_start;
        while (!stopped && s > 0) {
            --s;
            a.seg[s] = make_uint4(cy, cx, off, 1);
            uint32_t c_lo, c_hi;
            uint4 m = make_uint4(0, 0, 0, 2);
            if (tb_band(ey, ex, cy, d.N, c_lo, c_hi) && cx >= c_lo && cx <= c_hi) m = a.map[(size_t)s * (d.N + 1) + cx];
            if (m.w != 2u) {
                cx = m.x; cy = m.y; off += m.z; stopped = m.w != 0;
            } else {                                             // outside the band, or an entry the map gave up on: walk this strip here
                const uint32_t *wb = reinterpret_cast<const uint32_t *>(a.dirs + d.dir_off + s * aln_uniform_strip_bytes(d.N, a.R));
                auto fetch2 = [&](uint32_t kb, uint32_t lane) -> uint32_t { return wb[(((uint64_t)(kb >> 2) * 64u + lane) << 2) + (kb & 3u)]; };
                uint32_t st = 0;
                stopped = walk_in_strip(fetch2, s * rows, lgR, d.N, global, cy, cx, st, nullptr);
                off += st;
            }
        }
        if (!stopped && global && cy == 0 && cx != 0) {      // left strip 0 through the top border: D[0][x] = Left
            // (walk_in_strip handles the borders inside strip 0, so this cannot happen; kept as a guard)
            off += cx; cx = 0;
        }
    }
    res.start_y = cy; res.start_x = cx;
    res.aln_len = off + (a.pwm ? 0u : 1u);              // + the duplicated seed pair (none for the PWM aligner)
}

// one block per strip: stages the window its segment starts in, then one thread re-walks the segment and writes the
// tags at their final offset
extern "C" __global__ __launch_bounds__(256) void aln_tb_single_segments_kernel(TraceSingleArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem_words[];
    const uint32_t s = blockIdx.x;
    if (s >= a.ns) return;
    const PairDesc &d = a.descs[a.pair];
    if (a.results[a.pair].status != ALN_OK || (d.layout & 0xffu) != ALN_LAYOUT_UNIFORM) return;
    const uint4 sg = a.seg[s];
    if (sg.w == 0) return;
    const uint32_t lgR = 31u - (uint32_t)__builtin_clz(a.R), rows = 64u << lgR;
    const bool global = (a.semantics == ALN_CORE_GLOBAL || a.semantics == ALN_LEGACY_GLOBAL);
    uint8_t *ops = a.tags + d.tag_off + sg.z;
    const uint32_t *wbase = reinterpret_cast<const uint32_t *>(a.dirs + d.dir_off + s * aln_uniform_strip_bytes(d.N, a.R));
    uint32_t cy = sg.x, cx = sg.y, steps = 0;
    const uint32_t y0 = s * rows;
    const uint32_t k_hi = (cx ? cx - 1u : 0u) + ((cy > y0 ? cy - 1u - y0 : 0u) >> lgR);
    const StripWindow w = stage_window(wbase, smem_words, tb_window_quads(a.R), k_hi, lgR);
    const uint32_t q_lo = w.q_lo, q_n = w.q_n;
    // LDS through an address-space-3 pointer: a generic pointer would turn the two loads into one flat_load with a select
    const __attribute__((address_space(3))) uint32_t *lds3 = (const __attribute__((address_space(3))) uint32_t *)smem_words;
    auto fetch = [&](uint32_t kb, uint32_t lane) -> uint32_t {
        const uint32_t q = kb >> 2, rel = q - q_lo;
        uint32_t word = lds3[((min(rel, q_n - 1u) * 64u + lane) << 2) + (kb & 3u)];            // ds_read, always in range
        if (__builtin_expect(rel >= q_n, 0)) word = wbase[(((uint64_t)q * 64u + lane) << 2) + (kb & 3u)];   // left the window (rare)
        return word;
    };
    if (threadIdx.x == 0) walk_in_strip(fetch, y0, lgR, d.N, global, cy, cx, steps, ops);
}

// Pass 2 for one large pair: one block of 1024 threads (the per-pair wave of aln_traceback_expand_kernel would take
// ~100 us for 20 000 tags).  Same arithmetic: each thread sums its contiguous slice, block-wide exclusive scan, replay.
extern "C" __global__ __launch_bounds__(1024) void aln_tb_single_expand_kernel(TraceArgs a, uint32_t pair)
{
    __shared__ uint32_t wsum_y[16], wsum_x[16];
    const PairDesc &d = a.descs[pair];
    const aln_pair_result &res = a.results[pair];
    if (res.status != ALN_OK || (d.layout & 0xffu) != ALN_LAYOUT_UNIFORM) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint8_t *__restrict__ q = a.seqs + d.q_off;
    const uint8_t *__restrict__ t = a.seqs + d.t_off;
    const uint32_t cap = d.N + d.M + 2;
    uint8_t *__restrict__ qa = a.tb + d.tb_off;
    uint8_t *__restrict__ ta = qa + (a.pwm ? 4ull : 1ull) * cap;     // PWM: 4-byte column numbers instead of query residues (pwm/mod.rs:86-101)
    uint32_t *__restrict__ numbered = reinterpret_cast<uint32_t *>(qa);
    const uint8_t *__restrict__ ops = a.tags + d.tag_off;
    const uint32_t len = res.aln_len - (a.pwm ? 0u : 1u);
    const uint32_t per = (len + 1023u) / 1024u;
    const uint32_t lo = min(len, tid * per), hi = min(len, lo + per);
    uint32_t dy = 0, dx = 0;
    for (uint32_t j = lo; j < hi; ++j) {
        const uint32_t tag = ops[len - 1 - j];
        dx += (tag != 2u); dy += (tag != 1u);
    }
    uint32_t sy = dy, sx = dx;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const uint32_t oy = (uint32_t)__shfl_up((int)sy, m), ox = (uint32_t)__shfl_up((int)sx, m);
        if ((int)lane >= m) { sy += oy; sx += ox; }
    }
    if (lane == 63) { wsum_y[wave] = sy; wsum_x[wave] = sx; }
    __syncthreads();
    uint32_t by = 0, bx = 0;
    for (uint32_t wv = 0; wv < wave; ++wv) { by += wsum_y[wv]; bx += wsum_x[wv]; }
    uint32_t py = res.start_y + by + sy - dy, px = res.start_x + bx + sx - dx;
    for (uint32_t j = lo; j < hi; ++j) {
        const uint32_t tag = ops[len - 1 - j];
        px += (tag != 2u); py += (tag != 1u);
        if (a.pwm) numbered[j] = (tag == 2u) ? 0u : px;
        else qa[j] = (tag == 2u) ? a.blank : q[px - 1];
        ta[j] = (tag == 1u) ? a.blank : t[py - 1];
    }
    if (tid == 0 && !a.pwm) {
        qa[len] = q[res.end_x - 1];                     // the duplicated seed pair
        ta[len] = t[res.end_y - 1];
    }
}

// Pass 2, one wave per pair: turns the tag string into the two aligned code strings in final (forward) order.
// Position j of the output is step (len - 1 - j); its cell is the stop cell plus the moves of the steps after it, i.e.
// a prefix sum over the tags -- each lane sums its contiguous slice, one wave scan, then each lane replays its slice.
// 32 VGPRs (the attribute counts pairs): one wave of this kernel fits beside a full grid of the fill kernel (3 x 160 of a SIMD's
// 512 registers), so the strings of chunk i are written while chunk i+1 fills instead of waiting for a workgroup slot.
extern "C" __global__ __attribute__((amdgpu_flat_work_group_size(256, 256), amdgpu_num_vgpr(16))) void aln_traceback_expand_kernel(TraceArgs a)
{
    const uint32_t pair = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (pair >= a.n_pairs) return;
    constexpr int G = 2;
    const int lane = threadIdx.x & 63;
    const PairDesc &d = a.descs[pair];
    const aln_pair_result &res = a.results[pair];
    if (res.status != ALN_OK) return;
    if ((d.layout & 0xffu) == ALN_LAYOUT_UNIFORM) return;      // large pairs: aln_tb_single_expand_kernel
    const uint8_t *__restrict__ q = a.seqs + d.q_off;
    const uint8_t *__restrict__ t = a.seqs + d.t_off;
    const uint32_t cap = d.N + d.M + 2;
    uint8_t *__restrict__ qa = a.tb + d.tb_off;
    uint8_t *__restrict__ ta = qa + (a.pwm ? 4ull : 1ull) * cap;
    const uint8_t *__restrict__ ops = a.tags + d.tag_off;
    uint32_t *__restrict__ numbered = reinterpret_cast<uint32_t *>(qa);    // PWM: column numbers instead of query residues
    const uint32_t len = res.aln_len - (a.pwm ? 0u : 1u);
    // Position j reads ops[len - 1 - j] and needs the number of moves in x / y among positions 0..j: lane l takes position
    // j0 + l, so tags, residues and both strings move in coalesced lines, the prefix counts come from two ballots, and two
    // such groups are in flight at once -- the loads of a group do not depend on the group before (a lane-contiguous split
    // was one dependent byte load per position: 2 x 35 round trips per pair).
    uint32_t base_y = (uint32_t)__builtin_amdgcn_readfirstlane((int)res.start_y);
    uint32_t base_x = (uint32_t)__builtin_amdgcn_readfirstlane((int)res.start_x);
    for (uint32_t j0 = 0; j0 < len; j0 += 64u * G) {
        uint32_t tag[G], px[G], py[G];
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const uint32_t j = j0 + 64u * u + (uint32_t)lane;
            tag[u] = (j < len) ? (uint32_t)ops[len - 1 - j] : 3u;      // 3: beyond the end, moves nothing
        }
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const bool fx = (tag[u] != 2u) && (tag[u] != 3u), fy = (tag[u] != 1u) && (tag[u] != 3u);
            const uint64_t bx = __ballot(fx), by = __ballot(fy);
            px[u] = base_x + __builtin_amdgcn_mbcnt_hi((uint32_t)(bx >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bx, 0u)) + (fx ? 1u : 0u);
            py[u] = base_y + __builtin_amdgcn_mbcnt_hi((uint32_t)(by >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)by, 0u)) + (fy ? 1u : 0u);
            base_x += (uint32_t)__popcll(bx); base_y += (uint32_t)__popcll(by);
        }
        uint32_t qv[G], tv[G];
#pragma unroll
        for (int u = 0; u < G; ++u) {                                   // the cell this step left: (py, px)
            qv[u] = (tag[u] == 2u || tag[u] == 3u) ? (uint32_t)a.blank : (a.pwm ? px[u] : (uint32_t)q[px[u] - 1]);
            tv[u] = (tag[u] == 1u || tag[u] == 3u) ? (uint32_t)a.blank : (uint32_t)t[py[u] - 1];
        }
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const uint32_t j = j0 + 64u * u + (uint32_t)lane;
            if (j < len) {
                if (a.pwm) numbered[j] = (tag[u] == 2u) ? 0u : px[u];  // pwm/mod.rs:86-101
                else qa[j] = (uint8_t)qv[u];
                ta[j] = (uint8_t)tv[u];
            }
        }
    }
    if (lane == 0 && !a.pwm) {
        qa[len] = q[res.end_x - 1];                     // simple/mod.rs:102-105, :213-216: the duplicated seed
        ta[len] = t[res.end_y - 1];
    }
}

// (M+1)x(N+1) Direction bytes for one pair = AlignmentResult.direction_matrix
extern "C" __global__ void aln_unpack_directions_kernel(const uint8_t *dirs, const PairDesc *descs, uint32_t pair,
                                                        int semantics, uint8_t *out)
{
    const PairDesc d = descs[pair];
    const uint64_t total = (uint64_t)(d.M + 1) * (d.N + 1);
    const bool global = (semantics == ALN_CORE_GLOBAL || semantics == ALN_LEGACY_GLOBAL);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t y = (uint32_t)(i / (d.N + 1)), x = (uint32_t)(i % (d.N + 1));
        out[i] = (uint8_t)dir_at(dirs, d, global, y, x);
    }
}

#endif   // ALN_PART_TB

// ---------------------------------------------------------------- launch helpers used by aln_host.hip
#define ALN_FAST_REST_LAUNCH(COOPV)                                                                                        \
    switch (a->semantics) {                                                                                                \
    case ALN_CORE_GLOBAL: hipLaunchKernelGGL((aln_fill_fast_kernel<ALN_CORE_GLOBAL, false, COOPV>), g, b, lds_bytes, s, *a); break; \
    case ALN_CORE_LOCAL: hipLaunchKernelGGL((aln_fill_fast_kernel<ALN_CORE_LOCAL, true, COOPV>), g, b, lds_bytes, s, *a); break;   /* PWM scoring */ \
    case ALN_LEGACY_GLOBAL: hipLaunchKernelGGL((aln_fill_fast_kernel<ALN_LEGACY_GLOBAL, false, COOPV>), g, b, lds_bytes, s, *a); break; \
    default: hipLaunchKernelGGL((aln_fill_fast_kernel<ALN_LEGACY_LOCAL, false, COOPV>), g, b, lds_bytes, s, *a); break;      \
    }
#if ALN_TU & ALN_PART_FAST_CL
extern "C" void aln_launch_fill_fast_cl(const FillArgs *a, uint32_t grid, uint32_t lds_bytes, hipStream_t s)
{
    hipLaunchKernelGGL((aln_fill_fast_kernel<ALN_CORE_LOCAL, false, true>), dim3(grid), dim3(256), lds_bytes, s, *a);
}
#endif
#if ALN_TU & ALN_PART_FAST_CL_SOLO
extern "C" void aln_launch_fill_fast_cl_solo(const FillArgs *a, uint32_t grid, uint32_t lds_bytes, hipStream_t s)
{
    hipLaunchKernelGGL((aln_fill_fast_kernel<ALN_CORE_LOCAL, false, false>), dim3(grid), dim3(256), lds_bytes, s, *a);
}
#endif
#if ALN_TU & ALN_PART_FAST_REST
extern "C" void aln_launch_fill_fast_rest(const FillArgs *a, uint32_t grid, uint32_t lds_bytes, hipStream_t s)
{
    const dim3 g(grid), b(256);
    ALN_FAST_REST_LAUNCH(true)
}
#endif
#if ALN_TU & ALN_PART_FAST_REST_SOLO
__device__ __forceinline__ bool lane_is_first_cell(uint32_t j, uint32_t k) { return j == 0u && k == 0u; }
// ---------------------------------------------------------------- two short pairs per wave (core global: read pairs, C3)
// One wave per pair leaves a 150 x 150 pair with R = 3 rows per lane: 50 of 64 lanes busy, 49 steps of skew on 150 columns, the
// per-step feed shared by three cells, and a trip to the queue, the descriptors and the code check for every 22 500 cells.  Here
// lanes 0..31 hold one pair and lanes 32..63 another, R = ceil(rows / 32) rows per lane (C3: 5): 30 + 30 lanes busy, 29 steps of
// skew, the feed shared by five cells, half the trips.  What changes against FastStrip::step:
//   * the row above and the query enter at lane 0 AND lane 32: the border value is computed per lane and selected into the lanes
//     whose sub-lane is 0; the query code does not flow down the lanes -- both queries are staged in LDS as profile offsets
//     (zero-padded on both sides) and every lane reads its column's one two steps ahead, its profile word one step ahead;
//   * every lane stores its direction quads into ITS pair's region at sub-lane j, so each region is an ordinary one-strip region of
//     the uniform layout with R rows per lane (ALN_LAYOUT_UBATCH | R << 8) in which lanes 32..63 are never read: the walk kernels
//     do not know about any of this;
//   * the two pairs may differ in both lengths: a lane masks its cells by its own pair's N, stores only the quads its own region has.
// No row-1 hazard in the global semantics, no end-cell tracker: the cell is the six instructions of v_cell.
template <int R>
__device__ __forceinline__ void duo_fill(const FillArgs &a, uint8_t *prof, uint16_t *qo, const uint32_t qo_stride, const int *S, const int lane,
                                         const uint32_t N, const uint32_t M, const uint8_t *q, const uint8_t *t, uint4 *dirq, const uint32_t myquads,
                                         const uint32_t nsteps, const uint32_t maxN, int &corner)
{
    constexpr int SPB = (int)aln_spb(R);
    constexpr int RP = ProfWord<R>::RP;
    using PW = typename ProfWord<R>::T;
    const uint32_t j = (uint32_t)lane & 31u, h = (uint32_t)lane >> 5;
    const int nd4 = -4 * (int)a.del, ne4 = -4 * (int)a.ext;
    // ---- the two queries as profile offsets: entry i of a half is column i - 32 (zero outside the query)
    uint16_t *myqo = qo + h * qo_stride;
    for (uint32_t i = j; i < maxN + 136u; i += 32u) myqo[i] = (i >= 32u && i - 32u < N) ? (uint16_t)((uint32_t)q[i - 32u] * (64u * RP)) : (uint16_t)0;
    // ---- this lane's rows, their profile columns, the left border (simple/mod.rs:64-70)
    int tc[R], Tl[R];
    const uint32_t yb = j * R;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t y = yb + 1 + r;
        tc[r] = (y <= M) ? (int)t[y - 1] * (int)a.cols : 0;
        Tl[r] = (y == M) ? 2 + (int)(M + 1) * nd4 : 2 + (int)y * nd4;
    }
    for (uint32_t c = 0; c < a.cols; ++c) {
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t bte = (uint32_t)(4 * S[tc[r] + c] - 2) & 0xffu;
            if (r < 4) lo |= bte << (8 * r); else hi |= bte << (8 * (r - 4));
        }
        uint8_t *dst = prof + c * (64 * RP) + lane * RP;
        if constexpr (RP == 8) *reinterpret_cast<uint2 *>(dst) = make_uint2(lo, hi);
        else if constexpr (RP == 4) *reinterpret_cast<uint32_t *>(dst) = lo;
        else if constexpr (RP == 2) *reinterpret_cast<uint16_t *>(dst) = (uint16_t)lo;
        else *dst = (uint8_t)lo;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint8_t *prow = prof + lane * RP;
    const uint16_t *qmine = myqo + 32u - j;                 // qmine[c] = the offset of column c (0-based; zero for c < 0 or c >= N)
    int hdiag = (yb == 0) ? 2 : 2 + (int)yb * nd4;          // H[yb][0]
    int bottom = Tl[R - 1];
    uint32_t dw = 0;
    PW pw = *reinterpret_cast<const PW *>(prow + qmine[0]);  // step 0: column -j
    uint32_t qv = qmine[1];                                  // step 1
    const uint32_t nkb = ((nsteps + SPB - 1) / SPB + 3u) & ~3u;
    for (uint32_t kb = 0; kb < nkb; kb += 4) {
        uint4 v = make_uint4(0, 0, 0, 0);
#pragma unroll 1
        for (uint32_t b4 = 0; b4 < 4; ++b4) {
            const uint32_t k0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)((kb + b4) * SPB));
#pragma unroll
            for (int kk = 0; kk < SPB; ++kk) {
                const uint32_t k = k0 + (uint32_t)kk;
                // the row above: H[0][x] = -x del, the corner H[0][N] = -(N + 1) del (simple/mod.rs:59-62), for sub-lane 0; else the lane above
                const int top0 = (k + 1 == N) ? 2 + (int)(N + 1) * nd4 : 2 + (int)(k + 1) * nd4;
                int topIn = shr1_i(top0, bottom);
                topIn = (j == 0) ? top0 : topIn;
                const PW pwc = pw;
                pw = *reinterpret_cast<const PW *>(prow + qv);               // step k + 1
                qv = qmine[k + 2];                                            // step k + 2
                const uint32_t xm1 = k - j;
                if (xm1 < N) {
                    int top = topIn, diag = hdiag;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int negp = (r == 0 && lane_is_first_cell(j, k)) ? nd4 : ne4;     // del for the first visited cell only (simple/mod.rs:72,88-92)
                        int key, nt;
                        const uint32_t w32 = prof_word<R>(pwc, r);
                        switch (r & 3) {
                        case 0: v_cell<0>(top, Tl[r], negp, diag, w32, key, nt); break;
                        case 1: v_cell<1>(top, Tl[r], negp, diag, w32, key, nt); break;
                        case 2: v_cell<2>(top, Tl[r], negp, diag, w32, key, nt); break;
                        default: v_cell<3>(top, Tl[r], negp, diag, w32, key, nt); break;
                        }
                        dw = __builtin_amdgcn_alignbit((uint32_t)key, dw, 2);
                        diag = Tl[r];
                        Tl[r] = nt;
                        top = nt;
                    }
                    hdiag = topIn;
                    bottom = Tl[R - 1];
                }
            }
            if (b4 == 0) v.x = dw;
            else if (b4 == 1) v.y = dw;
            else if (b4 == 2) v.z = dw;
            else v.w = dw;
        }
        if (a.store_dirs && (kb >> 2) < myquads) dirq[(size_t)(kb >> 2) * 64] = v;
    }
    // H[M][N] lives in the lane that owns row M
    const uint32_t rb = (M - 1u) % R;
    int hb = Tl[0];
#pragma unroll
    for (int r = 1; r < R; ++r) if ((uint32_t)r == rb) hb = Tl[r];
    corner = hb;
}

extern "C" __global__ __attribute__((amdgpu_flat_work_group_size(256, 256), amdgpu_waves_per_eu(3, 3)))
void aln_fill_duo_kernel(FillArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *S = reinterpret_cast<int *>(smem);
    const int *gm = reinterpret_cast<const int *>(a.matrix);
    for (uint32_t i = threadIdx.x; i < a.rows * a.cols; i += blockDim.x) S[i] = gm[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t wv = threadIdx.x >> 6;
    // per wave: the profile, then the two staged queries (FillArgs::duo_qo entries each)
    uint8_t *wbase = smem + ((a.rows * a.cols * 4u + 15u) & ~15u) + wv * (a.prof_stride + 4u * a.duo_qo);
    uint8_t *prof = wbase;
    uint16_t *qo = reinterpret_cast<uint16_t *>(wbase + a.prof_stride);
    const uint32_t h = (uint32_t)lane >> 5;
    uint32_t run_left = 0, run_pos = 0;
    const uint32_t n_items = (a.n_pairs + 1u) / 2u, claim = a.claim ? a.claim : 1u;
    for (;;) {
        if (run_left == 0u) {
            unsigned long long v = 0;
            if (lane == 0) v = __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(a.counter + 4), (unsigned long long)claim, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t f = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
            if (f >= n_items) break;
            run_pos = f; run_left = min(claim, n_items - f);
        }
        const uint32_t item = run_pos++;
        --run_left;
        // this lane's pair: queue positions 2 item (lanes 0..31) and 2 item + 1 (lanes 32..63; none for the last item of an odd batch)
        const uint32_t qpos = 2u * item + h;
        const bool have = qpos < a.n_pairs;
        const uint32_t pair = have ? a.order[qpos] : 0u;
        const PairDesc &d = a.descs[pair];
        int st = have ? d.status : ALN_ERR_INVALID_ARGUMENT;
        uint32_t N = have ? d.N : 0u, M = have ? d.M : 0u;
        const uint8_t *q = a.seqs + d.q_off, *t = a.seqs + d.t_off;
        if (st == ALN_OK) {                                  // residue codes outside the matrix: the reference panics (simple/mod.rs:85)
            uint32_t worst_q = 0, worst_t = 0;
            for (uint32_t i = (uint32_t)lane & 31u; i < N; i += 32u) worst_q = max(worst_q, (uint32_t)q[i]);
            for (uint32_t i = (uint32_t)lane & 31u; i < M; i += 32u) worst_t = max(worst_t, (uint32_t)t[i]);
            const uint64_t badm = __ballot(worst_q >= a.cols || worst_t >= a.rows);
            if ((badm >> (32u * h)) & 0xffffffffull) st = ALN_ERR_CODE_OUT_OF_RANGE;
        }
        const bool ok = st == ALN_OK;
        if (!ok) { N = 0; M = 0; }
        // rows per lane for both: the larger of the two pairs' needs
        const uint32_t Rl = ok ? (M + 31u) / 32u : 1u;
        uint32_t R = max((uint32_t)__builtin_amdgcn_readlane((int)Rl, 0), (uint32_t)__builtin_amdgcn_readlane((int)Rl, 32));
        const uint32_t L = ok ? (M + R - 1u) / R : 0u;
        const uint32_t st_l = ok ? N + L - 1u : 0u;
        const uint32_t nsteps = max((uint32_t)__builtin_amdgcn_readlane((int)st_l, 0), (uint32_t)__builtin_amdgcn_readlane((int)st_l, 32));
        const uint32_t maxN = max((uint32_t)__builtin_amdgcn_readlane((int)N, 0), (uint32_t)__builtin_amdgcn_readlane((int)N, 32));
        uint4 *dirq = reinterpret_cast<uint4 *>(a.dirs + d.dir_off) + ((uint32_t)lane & 31u);
        const uint32_t myquads = ok ? aln_strip_blocks(N + 63u, aln_spb(R)) / 4u : 0u;
        int corner = 0;
        if (nsteps != 0u) {
            switch (R) {
            case 1: duo_fill<1>(a, prof, qo, a.duo_qo, S, lane, N, M, q, t, dirq, myquads, nsteps, maxN, corner); break;
            case 2: duo_fill<2>(a, prof, qo, a.duo_qo, S, lane, N, M, q, t, dirq, myquads, nsteps, maxN, corner); break;
            case 3: duo_fill<3>(a, prof, qo, a.duo_qo, S, lane, N, M, q, t, dirq, myquads, nsteps, maxN, corner); break;
            case 4: duo_fill<4>(a, prof, qo, a.duo_qo, S, lane, N, M, q, t, dirq, myquads, nsteps, maxN, corner); break;
            case 5: duo_fill<5>(a, prof, qo, a.duo_qo, S, lane, N, M, q, t, dirq, myquads, nsteps, maxN, corner); break;
            case 6: duo_fill<6>(a, prof, qo, a.duo_qo, S, lane, N, M, q, t, dirq, myquads, nsteps, maxN, corner); break;
            case 7: duo_fill<7>(a, prof, qo, a.duo_qo, S, lane, N, M, q, t, dirq, myquads, nsteps, maxN, corner); break;
            default: duo_fill<8>(a, prof, qo, a.duo_qo, S, lane, N, M, q, t, dirq, myquads, nsteps, maxN, corner); break;
            }
        }
        // the summaries: lane (M - 1) / R of each half holds H[M][N]
        const uint32_t lb = ok ? (M - 1u) / R : 0u;
        if (have && ((uint32_t)lane & 31u) == lb) {
            aln_pair_result &res = a.results[pair];
            if (!ok) skip_invalid(res, st, 0);
            else {
                a.descs[pair].layout = ALN_LAYOUT_UBATCH | (R << 8);
                write_result<ALN_CORE_GLOBAL>(res, 0.0, 0, 0, (double)(corner >> 2), N, M, 1u, 1u);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");          // the LDS of this item is reused by the next
        __builtin_amdgcn_wave_barrier();
    }
}
extern "C" void aln_launch_fill_duo(const FillArgs *a, uint32_t grid, uint32_t lds_bytes, hipStream_t s)
{
    hipLaunchKernelGGL(aln_fill_duo_kernel, dim3(grid), dim3(256), lds_bytes, s, *a);
}
extern "C" void aln_launch_fill_fast_rest_solo(const FillArgs *a, uint32_t grid, uint32_t lds_bytes, hipStream_t s)
{
    const dim3 g(grid), b(256);
    ALN_FAST_REST_LAUNCH(false)
}
#endif
#undef ALN_FAST_REST_LAUNCH
// ---------------------------------------------------------------- code-object warm-up (aln_create)
// Every translation unit is a code object of its own that the HIP runtime loads onto a device when the first kernel out of it is
// launched (milliseconds each: a first call used to pay for them).  aln_create asks for the attributes of one kernel per unit,
// which loads the unit without launching anything.
#define ALN_WARM(fn, kernel)                                                                    \
    extern "C" int fn(void)                                                                     \
    {                                                                                           \
        hipFuncAttributes attr;                                                                 \
        return (int)hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(kernel));        \
    }
#if ALN_TU & ALN_PART_FAST_CL
ALN_WARM(aln_warm_fast_cl, (&aln_fill_fast_kernel<ALN_CORE_LOCAL, false, true>))
#endif
#if ALN_TU & ALN_PART_FAST_CL_SOLO
ALN_WARM(aln_warm_fast_cl_solo, (&aln_fill_fast_kernel<ALN_CORE_LOCAL, false, false>))
#endif
#if ALN_TU & ALN_PART_FAST_REST
ALN_WARM(aln_warm_fast_rest, (&aln_fill_fast_kernel<ALN_CORE_GLOBAL, false, true>))
#endif
#if ALN_TU & ALN_PART_FAST_REST_SOLO
ALN_WARM(aln_warm_fast_rest_solo, (&aln_fill_fast_kernel<ALN_CORE_GLOBAL, false, false>))
#endif
#if ALN_TU & ALN_PART_GENERIC
ALN_WARM(aln_warm_generic, (&aln_validate_kernel))
#endif
#if ALN_TU & ALN_PART_SINGLE
ALN_WARM(aln_warm_single, (&aln_single_init_kernel))
#endif
#if ALN_TU & ALN_PART_TB
ALN_WARM(aln_warm_tb, (&aln_traceback_expand_kernel))
#endif
#undef ALN_WARM
#if ALN_TU & ALN_PART_GENERIC
extern "C" void aln_launch_fill_fast_cl(const FillArgs *a, uint32_t grid, uint32_t lds_bytes, hipStream_t s);
extern "C" void aln_launch_fill_fast_rest(const FillArgs *a, uint32_t grid, uint32_t lds_bytes, hipStream_t s);
extern "C" void aln_launch_fill_fast_cl_solo(const FillArgs *a, uint32_t grid, uint32_t lds_bytes, hipStream_t s);
extern "C" void aln_launch_fill_fast_rest_solo(const FillArgs *a, uint32_t grid, uint32_t lds_bytes, hipStream_t s);
extern "C" void aln_launch_fill_duo(const FillArgs *a, uint32_t grid, uint32_t lds_bytes, hipStream_t s);
extern "C" void aln_launch_fill(const FillArgs *a, int is_int, int fast, uint32_t grid, uint32_t lds_bytes, hipStream_t s)
{
    const dim3 g(grid), b(256);
#define ALN_LAUNCH(SC, SEM) hipLaunchKernelGGL((aln_fill_kernel<SC, SEM>), g, b, lds_bytes, s, *a)
    if (is_int && fast && a->duo_qo) aln_launch_fill_duo(a, grid, lds_bytes, s);      // two short pairs per wave (core global)
    else if (is_int && fast) {
        const bool cl = a->semantics == ALN_CORE_LOCAL && !a->pwm;
        if (a->coop) { if (cl) aln_launch_fill_fast_cl(a, grid, lds_bytes, s); else aln_launch_fill_fast_rest(a, grid, lds_bytes, s); }
        else { if (cl) aln_launch_fill_fast_cl_solo(a, grid, lds_bytes, s); else aln_launch_fill_fast_rest_solo(a, grid, lds_bytes, s); }
    } else if (is_int) {
        switch (a->semantics) {
        case ALN_CORE_GLOBAL: ALN_LAUNCH(int, ALN_CORE_GLOBAL); break;
        case ALN_CORE_LOCAL: ALN_LAUNCH(int, ALN_CORE_LOCAL); break;
        case ALN_LEGACY_GLOBAL: ALN_LAUNCH(int, ALN_LEGACY_GLOBAL); break;
        default: ALN_LAUNCH(int, ALN_LEGACY_LOCAL); break;
        }
    } else {
        const bool lean = a->hmat == nullptr && !a->f64_old;
        if (lean && a->semantics == ALN_CORE_GLOBAL) hipLaunchKernelGGL((aln_fill_f64_kernel<ALN_CORE_GLOBAL>), g, b, lds_bytes, s, *a);
        else if (lean) hipLaunchKernelGGL((aln_fill_f64_kernel<ALN_CORE_LOCAL>), g, b, lds_bytes, s, *a);
        else if (a->semantics == ALN_CORE_GLOBAL) ALN_LAUNCH(double, ALN_CORE_GLOBAL);
        else ALN_LAUNCH(double, ALN_CORE_LOCAL);
    }
#undef ALN_LAUNCH
}
#endif
#if ALN_TU & ALN_PART_SINGLE
// LDS bytes of the fill kernel for W waves per workgroup (rings + S + query offsets + per-wave profile and boundary ring)
extern "C" uint32_t aln_single_lds_bytes(uint32_t rows, uint32_t cols, uint32_t R, uint32_t N, uint32_t W)
{
    const uint32_t s_bytes = (rows * cols * 4u + 15u) & ~15u, prof_bytes = (cols * 64u * R + 15u) & ~15u;
    const uint32_t qo_bytes = ((N + 192u) * 2u + 15u) & ~15u;
    return (W - 1u) * 4u * ALN_RING + s_bytes + qo_bytes + W * (prof_bytes + 512u) + 768u;      // + scratch words (aligned up)
}
// four strips per workgroup where the hand-written steady state exists and the LDS fits
extern "C" uint32_t aln_single_waves(int semantics, uint32_t R, uint32_t rows, uint32_t cols, uint32_t N, uint32_t ns)
{
    if (getenv("ALN_SINGLE_W1")) return 1;
    if ((semantics != ALN_CORE_LOCAL && semantics != ALN_CORE_GLOBAL) || R > 2 || ns < 2) return 1;
    return aln_single_lds_bytes(rows, cols, R, N, 4) <= 150u * 1024u ? 4u : 1u;
}
extern "C" void aln_launch_single(const SingleArgs *a, uint32_t N, int with_serial, hipStream_t s)
{
    const uint32_t W = aln_single_waves(a->semantics, a->R, a->rows, a->cols, N, a->ns);
    const uint32_t lds_bytes = aln_single_lds_bytes(a->rows, a->cols, a->R, N, W);
    const dim3 g((a->ns + W - 1) / W), b(64 * W);
    const uint32_t serial_lds = (a->rows * a->cols * 4u + 15u) & ~15u;
#define ALN_SINGLE_LAUNCH(SEM, RR, WW)                                                                         \
    do {                                                                                                       \
        auto kern = aln_fill_single_kernel<SEM, RR, WW>;                                                       \
        if (lds_bytes > 64u * 1024u)                                                                           \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
        hipLaunchKernelGGL(kern, g, b, lds_bytes, s, *a);                                                      \
    } while (0)
#define ALN_SINGLE(SEM)                                                                                        \
    do {                                                                                                       \
        if (a->R == 1) ALN_SINGLE_LAUNCH(SEM, 1, 1);                                                           \
        else if (a->R == 2) ALN_SINGLE_LAUNCH(SEM, 2, 1);                                                      \
        else if (a->R == 4) ALN_SINGLE_LAUNCH(SEM, 4, 1);                                                      \
        else ALN_SINGLE_LAUNCH(SEM, 8, 1);                                                                     \
        hipLaunchKernelGGL((aln_single_finalize_kernel<SEM>), dim3(1), dim3(1024), 0, s, *a);                    \
        if (with_serial) hipLaunchKernelGGL((aln_single_serial_kernel<SEM>), dim3(1), dim3(64), serial_lds, s, *a); \
    } while (0)
#define ALN_SINGLE4(SEM)                                                                                       \
    do {                                                                                                       \
        if (a->R == 1) ALN_SINGLE_LAUNCH(SEM, 1, 4);                                                           \
        else ALN_SINGLE_LAUNCH(SEM, 2, 4);                                                                     \
        hipLaunchKernelGGL((aln_single_finalize_kernel<SEM>), dim3(1), dim3(1024), 0, s, *a);                  \
        if (with_serial) hipLaunchKernelGGL((aln_single_serial_kernel<SEM>), dim3(1), dim3(64), serial_lds, s, *a); \
    } while (0)
    switch (a->semantics) {
    case ALN_CORE_GLOBAL:
        if (W == 4) ALN_SINGLE4(ALN_CORE_GLOBAL); else ALN_SINGLE(ALN_CORE_GLOBAL);
        break;
    case ALN_CORE_LOCAL:
        if (W == 4) ALN_SINGLE4(ALN_CORE_LOCAL); else ALN_SINGLE(ALN_CORE_LOCAL);
        break;
    case ALN_LEGACY_GLOBAL: ALN_SINGLE(ALN_LEGACY_GLOBAL); break;
    default: ALN_SINGLE(ALN_LEGACY_LOCAL); break;
    }
#undef ALN_SINGLE4
#undef ALN_SINGLE
#undef ALN_SINGLE_LAUNCH
}
#endif
#if ALN_TU & ALN_PART_GENERIC
template <typename SC, int SEM>
static void launch_wgpipe_r(const WgArgs *a, uint32_t lds, hipStream_t s)
{
    const dim3 g(1), b(64 * a->ns);
    if (a->R == 1) hipLaunchKernelGGL((aln_fill_wgpipe_kernel<SC, SEM, 1>), g, b, lds, s, *a);
    else if (a->R == 2) hipLaunchKernelGGL((aln_fill_wgpipe_kernel<SC, SEM, 2>), g, b, lds, s, *a);
    else hipLaunchKernelGGL((aln_fill_wgpipe_kernel<SC, SEM, 4>), g, b, lds, s, *a);
}
template <typename SC>
static void launch_wgpipe_sem(const WgArgs *a, uint32_t lds, hipStream_t s)
{
    switch (a->semantics) {
    case ALN_CORE_GLOBAL: launch_wgpipe_r<SC, ALN_CORE_GLOBAL>(a, lds, s); break;
    case ALN_CORE_LOCAL: launch_wgpipe_r<SC, ALN_CORE_LOCAL>(a, lds, s); break;
    case ALN_LEGACY_GLOBAL: launch_wgpipe_r<SC, ALN_LEGACY_GLOBAL>(a, lds, s); break;
    default: launch_wgpipe_r<SC, ALN_LEGACY_LOCAL>(a, lds, s); break;
    }
}
extern "C" void aln_launch_wgpipe(const WgArgs *a, int is_int, uint32_t lds_bytes, hipStream_t s)
{
    if (is_int) launch_wgpipe_sem<int>(a, lds_bytes, s);
    else launch_wgpipe_sem<double>(a, lds_bytes, s);
}
extern "C" void aln_launch_validate(const uint8_t *seqs, PairDesc *descs, uint32_t n_pairs, uint32_t rows, uint32_t cols, int pwm,
                                    hipStream_t s)
{
    if (n_pairs) hipLaunchKernelGGL(aln_validate_kernel, dim3((n_pairs + 3) / 4), dim3(256), 0, s, seqs, descs, n_pairs, rows, cols, pwm ? 1u : 0u);
}
#endif
#if ALN_TU & ALN_PART_SINGLE
// the repair run + its finalize (both exit at once unless pass 0's finalize armed ctrl[8]); core local only
extern "C" void aln_launch_single_repair(const SingleArgs *a0, uint32_t N, hipStream_t s)
{
    if (a0->rep_S == 0 || a0->semantics != ALN_CORE_LOCAL) return;
    SingleArgs a = *a0;
    a.mode = 1;
    const uint32_t W = aln_single_waves(a.semantics, a.R, a.rows, a.cols, N, a.ns);
    const uint32_t lds_bytes = aln_single_lds_bytes(a.rows, a.cols, a.R, N, W);
    const dim3 g((a.rep_S + W - 1) / W), b(64 * W);
#define ALN_REPAIR_LAUNCH(RR, WW)                                                                              \
    do {                                                                                                       \
        auto kern = aln_fill_single_kernel<ALN_CORE_LOCAL, RR, WW>;                                            \
        if (lds_bytes > 64u * 1024u)                                                                           \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
        hipLaunchKernelGGL(kern, g, b, lds_bytes, s, a);                                                       \
    } while (0)
    if (W == 4) { if (a.R == 1) ALN_REPAIR_LAUNCH(1, 4); else ALN_REPAIR_LAUNCH(2, 4); }
    else if (a.R == 1) ALN_REPAIR_LAUNCH(1, 1);
    else if (a.R == 2) ALN_REPAIR_LAUNCH(2, 1);
    else if (a.R == 4) ALN_REPAIR_LAUNCH(4, 1);
    else ALN_REPAIR_LAUNCH(8, 1);
#undef ALN_REPAIR_LAUNCH
    hipLaunchKernelGGL((aln_single_repair_finalize_kernel<ALN_CORE_LOCAL>), dim3(1), dim3(1024), 0, s, a);
}
extern "C" void aln_launch_single_init(const SingleArgs *a, uint32_t n_bytes, hipStream_t s)
{
    hipLaunchKernelGGL(aln_single_init_kernel, dim3((n_bytes + 255) / 256), dim3(256), 0, s, *a, n_bytes);
}
#endif
#if ALN_TU & ALN_PART_TB
extern "C" void aln_launch_traceback(const TraceArgs *a, hipStream_t s)
{
    const uint32_t grid = (a->n_pairs + 63) / 64;
    hipLaunchKernelGGL(aln_traceback_kernel, dim3(grid), dim3(64), 0, s, *a);
}
// dyadic schemes filled by the integer kernels (aln_host.hip, call_init): both scores of every summary times 2^-k, exactly
extern "C" __global__ __launch_bounds__(256) void aln_scale_results_kernel(aln_pair_result *results, uint32_t n, double factor)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && results[i].status == ALN_OK) { results[i].f *= factor; results[i].score *= factor; }
}
extern "C" void aln_launch_scale_results(aln_pair_result *results, uint32_t n, double factor, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(aln_scale_results_kernel, dim3((n + 255) / 256), dim3(256), 0, s, results, n, factor);
}
extern "C" void aln_launch_traceback_wave(const TraceArgs *a, hipStream_t s)
{
    hipLaunchKernelGGL(aln_traceback_wave_kernel, dim3((a->n_pairs + 3) / 4), dim3(256), 0, s, *a);
}
extern "C" void aln_launch_traceback_overlap(const TraceArgs *a, uint32_t waves, hipStream_t s)
{
    hipLaunchKernelGGL(aln_delay_kernel, dim3(1), dim3(64), 0, s, 3000u);
    hipLaunchKernelGGL(aln_traceback_overlap_kernel, dim3(waves), dim3(64), 0, s, *a);
}
extern "C" void aln_launch_traceback_expand(const TraceArgs *a, hipStream_t s)
{
    hipLaunchKernelGGL(aln_traceback_expand_kernel, dim3((a->n_pairs + 3) / 4), dim3(256), 0, s, *a);
}
extern "C" void aln_launch_traceback_single(const TraceSingleArgs *a, uint32_t N, hipStream_t s)
{
    const uint32_t lds = tb_window_quads(a->R) * 1024u;
    (void)N;
    hipLaunchKernelGGL(aln_tb_single_maps_kernel, dim3(TB_BAND / 256u, a->ns), dim3(256), lds, s, *a);
    hipLaunchKernelGGL(aln_tb_single_chain_kernel, dim3(1), dim3(64), 0, s, *a);
    hipLaunchKernelGGL(aln_tb_single_segments_kernel, dim3(a->ns), dim3(256), lds, s, *a);
}
extern "C" void aln_launch_traceback_expand_single(const TraceArgs *a, uint32_t pair, hipStream_t s)
{
    hipLaunchKernelGGL(aln_tb_single_expand_kernel, dim3(1), dim3(1024), 0, s, *a, pair);
}
extern "C" void aln_launch_unpack(const uint8_t *dirs, const PairDesc *descs, uint32_t pair, int semantics, uint8_t *out,
                                  uint64_t cells, hipStream_t s)
{
    uint32_t grid = (uint32_t)((cells + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(aln_unpack_directions_kernel, dim3(grid), dim3(256), 0, s, dirs, descs, pair, semantics, out);
}
#endif
