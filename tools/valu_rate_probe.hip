// valu_rate_probe.hip -- how many cycles does one wave64 integer VALU instruction occupy a gfx950 SIMD for?
//
// The batch fill kernel is VALU-issue-bound, so its roof is  (SIMDs x clock) / (cycles per wave-instruction).  The local
// guide's constants table gives 2 cycles for v_fma_f32 with several waves per SIMD (SIMD-32, 64 lanes in two passes) and 4
// for one wave alone; this probe measures the instructions the fill kernel is actually made of, at 1/2/3/4/8 waves per SIMD.
//
// Each kernel runs ITER x 64 instructions of ONE kind over 8 independent accumulator chains (so dependent-issue latency
// never limits), one workgroup of 256 threads = one wave per SIMD, `w` workgroups per CU.  Reported per kind and w:
//   cyc/inst/SIMD = elapsed shader cycles of a wave (s_memtime) / (instructions per wave x w)
//   and the same from wall time (HIP events) x the clock implied by s_memtime, as a cross-check.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/valu_rate_probe tools/valu_rate_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHK(x)                                                                                     \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
    } while (0)

#define R8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define R64(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP)

// operands: %0..%7 accumulators, %8 / %9 loop-invariant VGPR sources, %10 an SGPR source
#define OPERANDS                                                                                               \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                           \
    : "v"(s0), "v"(s1), "s"(ss)                                                                                \
    : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7"

#define PROBE_T(NAME, OP, PER, TY)                                                                                  \
    extern "C" __global__ __launch_bounds__(256) void probe_##NAME(uint64_t *out, uint32_t iters, uint32_t seed) \
    {                                                                                                          \
        TY a0 = (TY)(seed + threadIdx.x), a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u,  \
           a6 = a0 * 17u, a7 = a0 * 19u;                                                                       \
        const TY s0 = (TY)(seed | 1u), s1 = (TY)(seed * 9u + threadIdx.x);                                     \
        const uint32_t ss = (uint32_t)__builtin_amdgcn_readfirstlane((int)(seed + 5u));                        \
        __syncthreads();                                                                                       \
        const uint64_t t0 = __builtin_readcyclecounter();                                                      \
        for (uint32_t i = 0; i < iters; ++i) asm volatile(R64(OP) "1:\n\t" OPERANDS);                                   \
        const uint64_t t1 = __builtin_readcyclecounter();                                                      \
        const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;                                    \
        if ((threadIdx.x & 63u) == 0) out[wave] = t1 - t0;                                                     \
        if ((a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7) == (TY)0x12345u) out[0] = 0;                               \
    }                                                                                                          \
    static const int per_##NAME = PER;

#define S(x) #x
#define OP_ADD(i) "v_add_u32 %" S(i) ", %" S(i) ", %8\n\t"
#define OP_SUB(i) "v_sub_u32 %" S(i) ", %" S(i) ", %8\n\t"
#define OP_MAX(i) "v_max_i32 %" S(i) ", %" S(i) ", %8\n\t"
#define OP_MAXU(i) "v_max_u32 %" S(i) ", %" S(i) ", 3\n\t"
#define OP_MAX3(i) "v_max3_i32 %" S(i) ", %" S(i) ", %8, %9\n\t"
#define OP_ADD3(i) "v_add3_u32 %" S(i) ", %" S(i) ", %8, -1\n\t"
#define OP_ANDOR(i) "v_and_or_b32 %" S(i) ", %" S(i) ", -4, 2\n\t"
#define OP_ALIGNBIT(i) "v_alignbit_b32 %" S(i) ", %8, %" S(i) ", 2\n\t"
#define OP_LSHLADD(i) "v_lshl_add_u32 %" S(i) ", %" S(i) ", 11, %10\n\t"
#define OP_SDWA(i) "v_add_u32_sdwa %" S(i) ", %" S(i) ", sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"
#define OP_DPP(i) "v_mov_b32_dpp %" S(i) ", %" S(i) " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define OP_DPPROR(i) "v_mov_b32_dpp %" S(i) ", %" S(i) " row_ror:3 row_mask:0xf bank_mask:0xf\n\t"
#define OP_CMPCND(i) "v_cmp_eq_u32 vcc, %" S(i) ", %8\n\tv_cndmask_b32 %" S(i) ", %8, %9, vcc\n\t"
#define OP_CNDMASK(i) "v_cndmask_b32 %" S(i) ", %" S(i) ", %9, vcc\n\t"
#define OP_CMP(i) "v_cmp_eq_u32 vcc, %" S(i) ", %8\n\t"
#define OP_CMPS(i) "v_cmp_eq_u32 s[20:21], %" S(i) ", %8\n\t"
#define OP_PKADD(i) "v_pk_add_i16 %" S(i) ", %" S(i) ", %8\n\t"
#define OP_PKMAX(i) "v_pk_max_i16 %" S(i) ", %" S(i) ", %8\n\t"
#define OP_PKADDU(i) "v_pk_add_u16 %" S(i) ", %" S(i) ", %8\n\t"
#define OP_PKSUBU(i) "v_pk_sub_u16 %" S(i) ", %" S(i) ", %8 clamp\n\t"
#define OP_FMA(i) "v_fma_f32 %" S(i) ", %" S(i) ", %8, %9\n\t"
#define OP_PKFMA(i) "v_add_f32 %" S(i) ", %" S(i) ", %8\n\t"
#define OP_READLANE(i) "v_readlane_b32 s2" S(i) ", %" S(i) ", 5\n\t"
#define OP_WRITELANE(i) "v_writelane_b32 %" S(i) ", %10, 0\n\t"
#define OP_PERM(i) "v_perm_b32 %" S(i) ", %" S(i) ", %8, %9\n\t"
#define OP_XOR(i) "v_xor_b32 %" S(i) ", %" S(i) ", %8\n\t"
#define OP_MOV(i) "v_mov_b32 %" S(i) ", %8\n\t"
#define OP_LSHR(i) "v_lshrrev_b32 %" S(i) ", 2, %" S(i) "\n\t"
#define OP_BFE(i) "v_bfe_u32 %" S(i) ", %" S(i) ", 2, 7\n\t"
#define OP_MAD(i) "v_mad_u32_u24 %" S(i) ", %" S(i) ", %8, %9\n\t"
#define OP_MINMAX(i) "v_min_i32 %" S(i) ", %" S(i) ", %8\n\tv_max_i32 %" S(i) ", %" S(i) ", %9\n\t"
#define OP_MAXF(i) "v_max_f32 %" S(i) ", %" S(i) ", %8\n\t"
#define OP_MINF(i) "v_min_f32 %" S(i) ", %" S(i) ", %8\n\t"
#define OP_MAX3F(i) "v_max3_f32 %" S(i) ", %" S(i) ", %8, %9\n\t"
#define OP_MED3F(i) "v_med3_f32 %" S(i) ", %" S(i) ", %8, %9\n\t"
#define OP_AND(i) "v_and_b32 %" S(i) ", %" S(i) ", %8\n\t"
#define OP_OR(i) "v_or_b32 %" S(i) ", %" S(i) ", %8\n\t"
#define OP_ANDIMM(i) "v_and_b32 %" S(i) ", -4, %" S(i) "\n\t"
#define OP_LSHL(i) "v_lshlrev_b32 %" S(i) ", 2, %" S(i) "\n\t"
#define OP_ASHR(i) "v_ashrrev_i32 %" S(i) ", 2, %" S(i) "\n\t"
#define OP_CNDVCC(i) "v_cndmask_b32 %" S(i) ", %8, %" S(i) ", vcc\n\t"
#define OP_CNDSG(i) "v_cndmask_b32 %" S(i) ", %8, %" S(i) ", s[22:23]\n\t"
#define OP_CMPF(i) "v_cmp_gt_f32 vcc, %" S(i) ", %8\n\t"
#define OP_CMPI(i) "v_cmp_gt_i32 vcc, %" S(i) ", %8\n\t"
#define OP_SUBABSCLAMP(i) "v_sub_f32 %" S(i) ", 1.0, |%" S(i) "| clamp\n\t"
#define OP_MULF(i) "v_mul_f32 %" S(i) ", %" S(i) ", %8\n\t"
#define OP_CVTUB1(i) "v_cvt_f32_ubyte1 %" S(i) ", %" S(i) "\n\t"
#define OP_CVTFI(i) "v_cvt_f32_i32 %" S(i) ", %" S(i) "\n\t"
#define OP_CVTIF(i) "v_cvt_i32_f32 %" S(i) ", %" S(i) "\n\t"
#define OP_FMAMIX(i) "v_fma_mix_f32 %" S(i) ", %" S(i) ", 1.0, %8 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
#define OP_PKADDF(i) "v_pk_add_f32 %" S(i) ", %" S(i) ", %8\n\t"
#define OP_PKFMAF(i) "v_pk_fma_f32 %" S(i) ", %" S(i) ", %8, %9\n\t"
#define OP_PKMOV(i) "v_pk_mov_b32 %" S(i) ", %" S(i) ", %8\n\t"
#define OP_MAXF64(i) "v_max_f64 %" S(i) ", %" S(i) ", %8\n\t"
#define OP_ADDF64(i) "v_add_f64 %" S(i) ", %" S(i) ", %8\n\t"
#define OP_PKMAXF16(i) "v_pk_max_f16 %" S(i) ", %" S(i) ", %8\n\t"
#define OP_MAXF16(i) "v_max_f16 %" S(i) ", %" S(i) ", %8\n\t"
#define OP_MAXI16(i) "v_max_i16 %" S(i) ", %" S(i) ", %8\n\t"
#define OP_ADDU16(i) "v_add_u16 %" S(i) ", %" S(i) ", %8\n\t"
#define OP_ADDCO(i) "v_add_co_u32 %" S(i) ", vcc, %" S(i) ", %8\n\t"
#define OP_BFI(i) "v_bfi_b32 %" S(i) ", %8, %" S(i) ", %9\n\t"
#define OP_OR3(i) "v_or3_b32 %" S(i) ", %" S(i) ", %8, %9\n\t"
#define OP_LSHLOR(i) "v_lshl_or_b32 %" S(i) ", %" S(i) ", 2, %8\n\t"
#define OP_XAD(i) "v_xad_u32 %" S(i) ", %" S(i) ", %8, %9\n\t"
#define OP_MIN3(i) "v_min3_i32 %" S(i) ", %" S(i) ", %8, %9\n\t"
#define OP_MULLO(i) "v_mul_lo_u32 %" S(i) ", %" S(i) ", %8\n\t"
#define OP_MULU24(i) "v_mul_u32_u24 %" S(i) ", %" S(i) ", %8\n\t"
#define OP_FLOOR(i) "v_floor_f32 %" S(i) ", %" S(i) "\n\t"
#define OP_ADDF_DPP(i) "v_add_f32_dpp %" S(i) ", %" S(i) ", %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define OP_ADDU_DPP(i) "v_add_u32_dpp %" S(i) ", %" S(i) ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define OP_MOVSDWA(i) "v_mov_b32_sdwa %" S(i) ", sext(%" S(i) ") dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n\t"
#define OP_ACCW(i) "v_accvgpr_write_b32 a" S(i) ", %" S(i) "\n\t"
#define OP_ACCR(i) "v_accvgpr_read_b32 %" S(i) ", a" S(i) "\n\t"
#define OP_MAXF_NEG(i) "v_max_f32 %" S(i) ", -%" S(i) ", %8\n\t"
#define OP_FMA_SGPR(i) "v_fma_f32 %" S(i) ", %" S(i) ", %10, %8\n\t"
#define OP_FMAC(i) "v_fmac_f32 %" S(i) ", %8, %9\n\t"
#define OP_ADDIMM(i) "v_add_u32 %" S(i) ", -1, %" S(i) "\n\t"
#define OP_ADDLIT(i) "v_add_u32 %" S(i) ", 0x12345, %" S(i) "\n\t"
#define OP_ADDSGPR(i) "v_add_u32 %" S(i) ", %10, %" S(i) "\n\t"
// mixes: does a scalar instruction of the same wave's stream cost the SIMD anything?
#define OP_MAX3_SADD(i) "v_max3_i32 %" S(i) ", %" S(i) ", %8, %9\n\ts_add_u32 s2" S(i) ", s2" S(i) ", 1\n\t"
#define OP_MAX3_SNOP(i) "v_max3_i32 %" S(i) ", %" S(i) ", %8, %9\n\ts_nop 0\n\t"
#define OP_MAX3_SCMP_BR(i) "v_max3_i32 %" S(i) ", %" S(i) ", %8, %9\n\ts_cmp_eq_u32 s2" S(i) ", 77\n\ts_cbranch_scc1 1f\n\t"
#define OP_MAX3_2SALU(i) "v_max3_i32 %" S(i) ", %" S(i) ", %8, %9\n\ts_add_u32 s2" S(i) ", s2" S(i) ", 1\n\ts_xor_b32 s2" S(i) ", s2" S(i) ", 5\n\t"
#define OP_ADD_SADD(i) "v_add_u32 %" S(i) ", %" S(i) ", %8\n\ts_add_u32 s2" S(i) ", s2" S(i) ", 1\n\t"
#define OP_MAX3_ADD(i) "v_max3_i32 %" S(i) ", %" S(i) ", %8, %9\n\tv_add_u32 %" S(i) ", %" S(i) ", %8\n\t"
#define OP_MAX3_DEP(i) "v_max3_i32 %0, %0, %8, %9\n\t"
#define OP_ADD_DEP(i) "v_add_u32 %0, %0, %8\n\t"
#define OP_MAX3_DEP2(i) "v_max3_i32 %0, %0, %8, %9\n\tv_max3_i32 %1, %1, %8, %9\n\t"
// VALU instructions under an EMPTY exec mask: does the SIMD skip them?  (8 v_max3 with exec = 0 per 1 with exec = all)
#define OP_EXEC0(i) "s_mov_b64 s[22:23], exec\n\ts_mov_b64 exec, 0\n\tv_max3_i32 %" S(i) ", %" S(i) ", %8, %9\n\tv_max3_i32 %" S(i) ", %" S(i) ", %8, %9\n\tv_max3_i32 %" S(i) ", %" S(i) ", %8, %9\n\tv_max3_i32 %" S(i) ", %" S(i) ", %8, %9\n\ts_mov_b64 exec, s[22:23]\n\tv_max3_i32 %" S(i) ", %" S(i) ", %8, %9\n\t"
#define OP_EXECONE(i) "s_mov_b64 s[22:23], exec\n\ts_mov_b64 exec, 1\n\tv_max3_i32 %" S(i) ", %" S(i) ", %8, %9\n\tv_max3_i32 %" S(i) ", %" S(i) ", %8, %9\n\tv_max3_i32 %" S(i) ", %" S(i) ", %8, %9\n\tv_max3_i32 %" S(i) ", %" S(i) ", %8, %9\n\ts_mov_b64 exec, s[22:23]\n\tv_max3_i32 %" S(i) ", %" S(i) ", %8, %9\n\t"
#define OP_SAVEEXEC_ONLY(i) "s_mov_b64 s[22:23], exec\n\ts_mov_b64 exec, 0\n\ts_mov_b64 exec, s[22:23]\n\tv_max3_i32 %" S(i) ", %" S(i) ", %8, %9\n\t"
// the fill kernel's core-local cell (aln_fast.h): cmp + cndmask + sdwa add + add + add3 + max3 + and_or + max_u32 +
// alignbit + lshl_add + max  = 11 VALU on a chain of its own
#define OP_CELL(i)                                                                                             \
    "v_cmp_eq_u32 vcc, %" S(i) ", %9\n\t"                                                                      \
    "v_cndmask_b32 %" S(i) ", %8, %9, vcc\n\t"                                                                 \
    "v_add_u32_sdwa %" S(i) ", %" S(i) ", sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t" \
    "v_add_u32 %" S(i) ", %" S(i) ", %8\n\t"                                                                   \
    "v_add3_u32 %" S(i) ", %" S(i) ", %8, -1\n\t"                                                              \
    "v_max3_i32 %" S(i) ", %" S(i) ", %8, %9\n\t"                                                              \
    "v_and_or_b32 %" S(i) ", %" S(i) ", -4, 2\n\t"                                                             \
    "v_max_u32 %" S(i) ", %" S(i) ", 3\n\t"                                                                    \
    "v_alignbit_b32 %" S(i) ", %8, %" S(i) ", 2\n\t"                                                           \
    "v_lshl_add_u32 %" S(i) ", %" S(i) ", 11, %10\n\t"                                                         \
    "v_max_i32 %" S(i) ", %" S(i) ", %9\n\t"

// The same cell for TWO cells in packed 16-bit halves (prototype of a v_pk_*_i16 fill: T = 4H + 2 in 16 bits, two independent
// cells per register): profile bytes into both halves (2 SDWA adds), top/left keys (2 pk_add), 2 pk_max, the two carried forms
// (nt and nt - 1: 2 and_or), Beginning tag (pk_max_u16), next row's penalty (pk_min_u16 + pk_mad_i16), direction bits (and +
// lshl_or), one end-cell key per half (lshl_or, and_or) + their max = 16 VALU + 2 max for 2 cells
#define OP_PKCELL(i)                                                                                           \
    "v_add_u16_sdwa %" S(i) ", %" S(i) ", sext(%8) dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:BYTE_1\n\t" \
    "v_add_u16_sdwa %" S(i) ", %" S(i) ", sext(%9) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_2\n\t" \
    "v_pk_add_i16 %" S(i) ", %" S(i) ", %8\n\t"                                                                \
    "v_pk_add_i16 %" S(i) ", %" S(i) ", %9\n\t"                                                                \
    "v_pk_max_i16 %" S(i) ", %" S(i) ", %8\n\t"                                                                \
    "v_pk_max_i16 %" S(i) ", %" S(i) ", %9\n\t"                                                                \
    "v_and_or_b32 %" S(i) ", %" S(i) ", %8, %9\n\t"                                                            \
    "v_and_or_b32 %" S(i) ", %" S(i) ", %9, %8\n\t"                                                            \
    "v_pk_max_u16 %" S(i) ", %" S(i) ", %8\n\t"                                                                \
    "v_pk_min_u16 %" S(i) ", %" S(i) ", %9\n\t"                                                                \
    "v_pk_mad_i16 %" S(i) ", %" S(i) ", %8, %9\n\t"                                                            \
    "v_and_b32 %" S(i) ", %" S(i) ", %8\n\t"                                                                   \
    "v_lshl_or_b32 %" S(i) ", %" S(i) ", 2, %9\n\t"                                                            \
    "v_lshl_or_b32 %" S(i) ", %" S(i) ", 16, %10\n\t"                                                          \
    "v_and_or_b32 %" S(i) ", %" S(i) ", %8, %10\n\t"                                                           \
    "v_max_i32 %" S(i) ", %" S(i) ", %9\n\t"                                                                   \
    "v_max_i32 %" S(i) ", %" S(i) ", %8\n\t"

#define PROBE(NAME, OP, PER) PROBE_T(NAME, OP, PER, uint32_t)
PROBE(add_u32, OP_ADD, 1)
PROBE(sub_u32, OP_SUB, 1)
PROBE(max_i32, OP_MAX, 1)
PROBE(max_u32_imm, OP_MAXU, 1)
PROBE(max3_i32, OP_MAX3, 1)
PROBE(add3_u32, OP_ADD3, 1)
PROBE(and_or_b32, OP_ANDOR, 1)
PROBE(alignbit_b32, OP_ALIGNBIT, 1)
PROBE(lshl_add_u32_sgpr, OP_LSHLADD, 1)
PROBE(add_u32_sdwa, OP_SDWA, 1)
PROBE(mov_dpp_wave_shr, OP_DPP, 1)
PROBE(mov_dpp_row_ror, OP_DPPROR, 1)
PROBE(cmp_vcc_plus_cndmask, OP_CMPCND, 2)
PROBE(cndmask_b32, OP_CNDMASK, 1)
PROBE(cmp_eq_vcc, OP_CMP, 1)
PROBE(cmp_eq_sgpr, OP_CMPS, 1)
PROBE(pk_add_i16, OP_PKADD, 1)
PROBE(pk_max_i16, OP_PKMAX, 1)
PROBE(pk_add_u16, OP_PKADDU, 1)
PROBE(pk_sub_u16_clamp, OP_PKSUBU, 1)
PROBE(fma_f32, OP_FMA, 1)
PROBE(add_f32, OP_PKFMA, 1)
PROBE(readlane, OP_READLANE, 1)
PROBE(writelane, OP_WRITELANE, 1)
PROBE(perm_b32, OP_PERM, 1)
PROBE(xor_b32, OP_XOR, 1)
PROBE(mov_b32, OP_MOV, 1)
PROBE(lshrrev_b32, OP_LSHR, 1)
PROBE(bfe_u32, OP_BFE, 1)
PROBE(mad_u32_u24, OP_MAD, 1)
PROBE(min_max_i32, OP_MINMAX, 2)
PROBE(fill_cell_11, OP_CELL, 11)
PROBE(pk16_two_cells_17, OP_PKCELL, 17)
PROBE(mix_max3_plus_sadd, OP_MAX3_SADD, 1)
PROBE(mix_max3_plus_snop, OP_MAX3_SNOP, 1)
PROBE(mix_max3_plus_scmp_branch, OP_MAX3_SCMP_BR, 1)
PROBE(mix_max3_plus_2salu, OP_MAX3_2SALU, 1)
PROBE(mix_add_plus_sadd, OP_ADD_SADD, 1)
PROBE(mix_max3_plus_vadd, OP_MAX3_ADD, 2)
PROBE(dep_chain_max3, OP_MAX3_DEP, 1)
PROBE(dep_chain_add, OP_ADD_DEP, 1)
PROBE(dep_2chains_max3, OP_MAX3_DEP2, 2)
PROBE(exec0_4max3_per_1, OP_EXEC0, 5)
PROBE(exec1lane_4max3_per_1, OP_EXECONE, 5)
PROBE(exec_toggle_plus_max3, OP_SAVEEXEC_ONLY, 1)
PROBE(max_f32, OP_MAXF, 1)
PROBE(min_f32, OP_MINF, 1)
PROBE(max3_f32, OP_MAX3F, 1)
PROBE(med3_f32, OP_MED3F, 1)
PROBE(max_f32_negmod, OP_MAXF_NEG, 1)
PROBE(and_b32, OP_AND, 1)
PROBE(or_b32, OP_OR, 1)
PROBE(and_b32_imm, OP_ANDIMM, 1)
PROBE(lshlrev_b32, OP_LSHL, 1)
PROBE(ashrrev_i32, OP_ASHR, 1)
PROBE(cndmask_vcc_src, OP_CNDVCC, 1)
PROBE(cndmask_sgpr_src, OP_CNDSG, 1)
PROBE(cmp_gt_f32, OP_CMPF, 1)
PROBE(cmp_gt_i32, OP_CMPI, 1)
PROBE(sub_f32_abs_clamp, OP_SUBABSCLAMP, 1)
PROBE(mul_f32, OP_MULF, 1)
PROBE(cvt_f32_ubyte1, OP_CVTUB1, 1)
PROBE(cvt_f32_i32, OP_CVTFI, 1)
PROBE(cvt_i32_f32, OP_CVTIF, 1)
PROBE(fma_mix_f32, OP_FMAMIX, 1)
PROBE_T(pk_add_f32, OP_PKADDF, 1, uint64_t)
PROBE_T(pk_fma_f32, OP_PKFMAF, 1, uint64_t)
PROBE_T(pk_mov_b32, OP_PKMOV, 1, uint64_t)
PROBE_T(max_f64, OP_MAXF64, 1, uint64_t)
PROBE_T(add_f64, OP_ADDF64, 1, uint64_t)
PROBE(pk_max_f16, OP_PKMAXF16, 1)
PROBE(max_f16, OP_MAXF16, 1)
PROBE(max_i16, OP_MAXI16, 1)
PROBE(add_u16, OP_ADDU16, 1)
PROBE(add_co_u32, OP_ADDCO, 1)
PROBE(bfi_b32, OP_BFI, 1)
PROBE(or3_b32, OP_OR3, 1)
PROBE(lshl_or_b32, OP_LSHLOR, 1)
PROBE(xad_u32, OP_XAD, 1)
PROBE(min3_i32, OP_MIN3, 1)
PROBE(mul_lo_u32, OP_MULLO, 1)
PROBE(mul_u32_u24, OP_MULU24, 1)
PROBE(floor_f32, OP_FLOOR, 1)
PROBE(add_f32_dpp, OP_ADDF_DPP, 1)
PROBE(add_u32_dpp_row_shr, OP_ADDU_DPP, 1)
PROBE(mov_b32_sdwa, OP_MOVSDWA, 1)
PROBE(accvgpr_write, OP_ACCW, 1)
PROBE(accvgpr_read, OP_ACCR, 1)
PROBE(fma_f32_sgpr, OP_FMA_SGPR, 1)
PROBE(fmac_f32, OP_FMAC, 1)
PROBE(add_u32_inline_imm, OP_ADDIMM, 1)
PROBE(add_u32_literal, OP_ADDLIT, 1)
PROBE(add_u32_sgpr, OP_ADDSGPR, 1)

struct Entry {
    const char *name;
    void (*fn)(uint64_t *, uint32_t, uint32_t);
    int per;
};
#define E(NAME) {#NAME, probe_##NAME, per_##NAME}
static const Entry entries[] = {
    E(add_u32), E(sub_u32), E(max_i32), E(max_u32_imm), E(max3_i32), E(add3_u32), E(and_or_b32), E(alignbit_b32),
    E(lshl_add_u32_sgpr), E(add_u32_sdwa), E(mov_dpp_wave_shr), E(mov_dpp_row_ror), E(cmp_vcc_plus_cndmask),
    E(cndmask_b32), E(cmp_eq_vcc), E(cmp_eq_sgpr), E(pk_add_i16), E(pk_max_i16), E(pk_add_u16), E(pk_sub_u16_clamp),
    E(fma_f32), E(add_f32), E(readlane), E(writelane), E(perm_b32), E(xor_b32), E(mov_b32), E(lshrrev_b32), E(bfe_u32),
    E(mad_u32_u24), E(min_max_i32), E(fill_cell_11), E(pk16_two_cells_17),
    E(exec0_4max3_per_1), E(exec1lane_4max3_per_1), E(exec_toggle_plus_max3),
    E(mix_max3_plus_sadd), E(mix_max3_plus_snop), E(mix_max3_plus_scmp_branch), E(mix_max3_plus_2salu), E(mix_add_plus_sadd), E(mix_max3_plus_vadd), E(dep_chain_max3), E(dep_chain_add), E(dep_2chains_max3),
    E(max_f32), E(min_f32), E(max3_f32), E(med3_f32), E(max_f32_negmod), E(and_b32), E(or_b32), E(and_b32_imm), E(lshlrev_b32), E(ashrrev_i32), E(cndmask_vcc_src), E(cndmask_sgpr_src), E(cmp_gt_f32), E(cmp_gt_i32), E(sub_f32_abs_clamp), E(mul_f32), E(cvt_f32_ubyte1), E(cvt_f32_i32), E(cvt_i32_f32), E(fma_mix_f32), E(pk_add_f32), E(pk_fma_f32), E(pk_mov_b32), E(max_f64), E(add_f64), E(pk_max_f16), E(max_f16), E(max_i16), E(add_u16), E(add_co_u32), E(bfi_b32), E(or3_b32), E(lshl_or_b32), E(xad_u32), E(min3_i32), E(mul_lo_u32), E(mul_u32_u24), E(floor_f32), E(add_f32_dpp), E(add_u32_dpp_row_shr), E(mov_b32_sdwa), E(accvgpr_write), E(accvgpr_read), E(fma_f32_sgpr), E(fmac_f32), E(add_u32_inline_imm), E(add_u32_literal), E(add_u32_sgpr),
};

int main(int argc, char **argv)
{
    uint32_t iters = argc > 1 ? (uint32_t)atoi(argv[1]) : 4000u;
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("# device %s (%s), %d CUs, clockRate %.0f MHz; %u iterations x 64 instructions per wave, 8 independent chains\n",
           prop.name, prop.gcnArchName, cus, prop.clockRate / 1e3, iters);
    printf("# cyc/inst/SIMD: shader cycles (s_memtime) a wave took / (its instructions x waves per SIMD); wall: the same from\n"
           "# HIP-event time x 2.4 GHz nominal.  One workgroup = 4 waves = one wave per SIMD; w workgroups per CU.\n");
    const int ws[] = {1, 2, 3, 4, 8};
    uint64_t *d_out;
    CHK(hipMalloc(&d_out, sizeof(uint64_t) * cus * 8 * 4));
    std::vector<uint64_t> h(cus * 8 * 4);
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    printf("%-24s", "instruction");
    for (int w : ws) printf("   w=%d cyc   (wall)", w);
    printf("\n");
    for (const Entry &en : entries) {
        printf("%-24s", en.name);
        for (int w : ws) {
            const uint32_t grid = (uint32_t)(cus * w);
            hipLaunchKernelGGL(en.fn, dim3(grid), dim3(256), 0, 0, d_out, 16u, 1u);           // warm-up
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(en.fn, dim3(grid), dim3(256), 0, 0, d_out, iters, 1u);
            CHK(hipEventRecord(e1, 0));
            CHK(hipEventSynchronize(e1));
            float ms = 0;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            CHK(hipMemcpy(h.data(), d_out, sizeof(uint64_t) * grid * 4, hipMemcpyDeviceToHost));
            double sum = 0;
            for (uint32_t i = 0; i < grid * 4; ++i) sum += (double)h[i];
            const double insts = (double)iters * 64.0 * en.per;
            const double cyc = sum / (grid * 4) / (insts * w);
            const double wall = (double)ms * 1e-3 * 2.4e9 / (insts * w);
            printf("   %8.3f (%6.3f)", cyc, wall);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
