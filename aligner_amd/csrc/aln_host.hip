// aln_host.hip -- host side of the C ABI declared in include/aligner_hip.h.
//
// Batch packing (validation, LPT ordering, direction-region layout), HBM staging, kernel launches on a HIP
// stream, result fetch.  One context per process per GPU; multi-GPU runs are one process per GPU (the Python
// driver shards pairs across ranks and gathers the 48-byte summaries with RCCL through torch.distributed).
// There is NO CPU implementation of the DP here: if the device or the kernels are unavailable every entry point
// fails with ALN_ERR_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <numeric>
#include <string>
#include <vector>

#include "aln_device.h"

#define ALN_TIMING_SLOTS 256u

extern "C" void aln_launch_fill(const FillArgs *a, int is_int, int fast, uint32_t grid, uint32_t lds_bytes, hipStream_t s);
extern "C" void aln_launch_traceback(const TraceArgs *a, hipStream_t s);
extern "C" void aln_launch_traceback_overlap(const TraceArgs *a, uint32_t waves, hipStream_t s);
extern "C" void aln_launch_traceback_expand(const TraceArgs *a, hipStream_t s);
extern "C" void aln_launch_traceback_single(const TraceSingleArgs *a, uint32_t N, hipStream_t s);
extern "C" void aln_launch_traceback_expand_single(const TraceArgs *a, uint32_t pair, hipStream_t s);
extern "C" void aln_launch_single(const SingleArgs *a, uint32_t N, int with_serial, hipStream_t s);
extern "C" void aln_launch_single_init(const SingleArgs *a, uint32_t n_bytes, hipStream_t s);
extern "C" void aln_launch_unpack(const uint8_t *dirs, const PairDesc *descs, uint32_t pair, int semantics, uint8_t *out,
                                  uint64_t cells, hipStream_t s);

static thread_local std::string g_err;

struct aln_ctx {
    int device = 0;
    int cus = 0;
    size_t hbm = 0;
    char name[128] = {0};
    hipStream_t stream = nullptr;
    std::mutex mu;
};

// bytes of the single-pair kernel's advice array and of its bottom-row record (one direction dword per block of the last
// strip, at most (N + 63) / 2 + 4 blocks at R = 8), equal sizes, 256-aligned
static inline uint64_t single_advice_bytes(uint64_t N)
{
    const uint64_t a = N + 128, z = 4 * ((N + 63) / 2 + 8);
    return ((a > z ? a : z) + 255) & ~255ull;
}

struct aln_batch {
    aln_ctx *ctx = nullptr;
    aln_params params{};
    size_t n = 0;
    bool is_int = true;
    bool pwm = false;         // ALN_PWM_LOCAL: kernels run as CORE_LOCAL with position-specific scoring
    uint32_t *d_pwm_words = nullptr;
    bool store_dirs = true;   // false: score-only batch (outputs has neither TRACEBACK nor DIRECTIONS)
    bool fast = false;        // integer kernels with the LDS query profile + packed max3 keys
    uint32_t prof_stride = 0;
    uint64_t cells = 0;
    uint64_t dir_bytes = 0;
    uint64_t tb_bytes = 0;
    uint32_t max_len = 0;
    uint32_t grid = 0;
    uint64_t scratch_stride = 0;
    uint32_t zrow_bytes = 0;
    uint32_t lds_bytes = 0;
    std::vector<PairDesc> descs;
    // device
    uint8_t *d_seqs = nullptr;
    PairDesc *d_descs = nullptr;
    uint32_t *d_order = nullptr;
    uint32_t *d_counter = nullptr;
    // overlapped traceback (aln_batch_run): second stream, fork/join events, per-pair "walked in run #epoch" marks
    bool overlap = false;
    uint32_t tb_waves = 0, epoch = 0;
    size_t counter_bytes = 256;
    uint32_t *d_walked = nullptr;
    hipStream_t tb_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    uint8_t *d_dirs = nullptr;
    aln_pair_result *d_results = nullptr;
    uint8_t *d_tb = nullptr;
    uint8_t *d_scratch = nullptr;
    void *d_matrix = nullptr;
    void *d_hmat = nullptr;
    uint64_t hmat_elems = 0;
    // pairs routed to the single-pair (one wave per strip) kernel, processed one after another
    std::vector<uint32_t> single_pairs;
    std::vector<uint32_t> single_r;
    size_t n_small = 0;
    uint64_t max_cells = 0;
    uint32_t *d_granules = nullptr;
    uint64_t granule_bytes = 0;
    uint8_t *d_advice1 = nullptr;
    int32_t *d_cand = nullptr;
    uint32_t *d_ctrl = nullptr;
    uint32_t single_max_n = 0;
    uint4 *d_tbmap = nullptr;   // parallel traceback of large pairs: exit maps + per-strip segments
    uint64_t tbmap_entries = 0;
    hipStream_t last_stream = nullptr;
    // timing ring: one event triple per run (fill start, fill end, traceback end), recorded on the launch stream
    bool timing = false;
    std::vector<hipEvent_t> ev;
    uint32_t ev_runs = 0;
    uint32_t fill_launches = 0;
};

static int fail(hipError_t e, const char *what)
{
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    g_err = buf;
    return e == hipErrorOutOfMemory ? ALN_ERR_OOM : ALN_ERR_DEVICE;
}
#define HIPCHK(call)                                         \
    do {                                                     \
        hipError_t e_ = (call);                              \
        if (e_ != hipSuccess) return fail(e_, #call);        \
    } while (0)

extern "C" const char *aln_last_error(void) { return g_err.c_str(); }
extern "C" int aln_abi_version(void) { return ALN_ABI_VERSION; }

extern "C" aln_ctx *aln_create(int device_id, int *status)
{
    int st = ALN_OK;
    aln_ctx *c = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_err = e != hipSuccess ? std::string("hipGetDeviceCount: ") + hipGetErrorString(e) : "no HIP device visible";
        st = ALN_ERR_DEVICE;
    } else if (device_id < 0 || device_id >= ndev) {
        g_err = "device id out of range";
        st = ALN_ERR_INVALID_ARGUMENT;
    } else {
        hipDeviceProp_t prop;
        if ((e = hipSetDevice(device_id)) != hipSuccess || (e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess) {
            st = fail(e, "hipSetDevice/hipGetDeviceProperties");
        } else {
            c = new aln_ctx();
            c->device = device_id;
            c->cus = prop.multiProcessorCount;
            c->hbm = prop.totalGlobalMem;
            snprintf(c->name, sizeof c->name, "%s (%s)", prop.name, prop.gcnArchName);
            if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
                st = fail(e, "hipStreamCreate");
                delete c;
                c = nullptr;
            }
        }
    }
    if (status) *status = st;
    return c;
}

extern "C" void aln_destroy(aln_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int aln_device_info(aln_ctx *ctx, int *cus, size_t *hbm, char *name, size_t cap)
{
    if (!ctx) return ALN_ERR_INVALID_ARGUMENT;
    if (cus) *cus = ctx->cus;
    if (hbm) *hbm = ctx->hbm;
    if (name && cap) { strncpy(name, ctx->name, cap - 1); name[cap - 1] = 0; }
    return ALN_OK;
}

static bool integral(double v) { return std::isfinite(v) && v == std::floor(v) && std::fabs(v) < 1e9; }

static void batch_free(aln_batch *b)
{
    if (!b) return;
    (void)hipSetDevice(b->ctx->device);
    void *ptrs[] = {b->d_seqs, b->d_descs, b->d_order, b->d_counter, b->d_dirs, b->d_results, b->d_tb, b->d_scratch,
                    b->d_matrix, b->d_hmat, b->d_granules, b->d_advice1, b->d_cand, b->d_ctrl, b->d_tbmap, b->d_pwm_words};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (auto &e : b->ev) if (e) (void)hipEventDestroy(e);
    if (b->d_walked) (void)hipFree(b->d_walked);
    if (b->ev_fork) (void)hipEventDestroy(b->ev_fork);
    if (b->ev_join) (void)hipEventDestroy(b->ev_join);
    if (b->tb_stream) (void)hipStreamDestroy(b->tb_stream);
    delete b;
}

extern "C" void aln_batch_destroy(aln_batch *b) { batch_free(b); }

static int batch_build(aln_ctx *ctx, const aln_params *p, const uint8_t *seqs, const uint64_t *q_off,
                       const uint64_t *q_len, const uint64_t *t_off, const uint64_t *t_len, size_t n, bool want_h,
                       aln_batch **out)
{
    *out = nullptr;
    if (!ctx || !p || (n && (!seqs || !q_off || !q_len || !t_off || !t_len))) { g_err = "null argument"; return ALN_ERR_INVALID_ARGUMENT; }
    if (p->semantics < ALN_CORE_GLOBAL || p->semantics > ALN_PWM_LOCAL) { g_err = "bad semantics"; return ALN_ERR_INVALID_ARGUMENT; }
    const bool pwm = p->semantics == ALN_PWM_LOCAL;
    const bool core = p->semantics == ALN_CORE_GLOBAL || p->semantics == ALN_CORE_LOCAL || pwm;
    // simple/mod.rs:49-51 / :175-177
    if (core && p->heuristics_present) return ALN_ERR_UNNECESSARY_ARGUMENT;
    if (!p->matrix || p->rows == 0 || p->cols == 0) { g_err = "matrix missing"; return ALN_ERR_INVALID_ARGUMENT; }
    if (pwm && p->rows != 4) return ALN_ERR_MATRIX_SHAPE;                      // pwm/mod.rs:40-42
    // the matrix lives in LDS: 32 KiB next to the query profiles; a position-weight matrix (no profiles in LDS) may be
    // 4 x 2000 wide (62.5 KiB as f64)
    if ((uint64_t)p->rows * p->cols > (pwm ? 8000u : 4096u)) {
        g_err = pwm ? "position-weight matrix larger than 8000 entries (4 x 2000)" : "substitution matrix larger than 4096 entries";
        return ALN_ERR_UNSUPPORTED;
    }
    if (n > 0xFFFFFFF0ull) { g_err = "too many pairs"; return ALN_ERR_UNSUPPORTED; }
    HIPCHK(hipSetDevice(ctx->device));

    aln_batch *b = new aln_batch();
    b->ctx = ctx;
    b->params = *p;
    b->params.matrix = nullptr;
    b->n = n;
    b->pwm = pwm;
    if (pwm) b->params.semantics = ALN_CORE_LOCAL;   // same recurrence and tie rules; only the score lookup differs

    // ---- compact the matrix, pick the arithmetic
    const uint32_t rows = p->rows, cols = p->cols;
    const int64_t rs = p->row_stride ? p->row_stride : (int64_t)cols;
    std::vector<double> md((size_t)rows * cols);
    double maxabs = std::max(std::fabs(p->del), std::fabs(p->ext));
    bool all_int = integral(p->del) && (core ? integral(p->ext) : true);
    for (uint32_t r = 0; r < rows; ++r)
        for (uint32_t c = 0; c < cols; ++c) {
            double v = p->matrix[(int64_t)r * rs + c];
            md[(size_t)r * cols + c] = v;
            all_int = all_int && integral(v);
            maxabs = std::max(maxabs, std::fabs(v));
        }
    if (!core && !all_int) { g_err = "legacy semantics are i32: del and matrix must be integral"; batch_free(b); return ALN_ERR_INVALID_ARGUMENT; }
    if (!core && p->force_f64) { g_err = "legacy semantics have no f64 form"; batch_free(b); return ALN_ERR_UNSUPPORTED; }

    // ---- per-pair validation
    b->descs.resize(n);
    uint64_t cells = 0;
    uint32_t max_len = 1;
    uint64_t max_span = 0;
    for (size_t i = 0; i < n; ++i) {
        PairDesc &d = b->descs[i];
        memset(&d, 0, sizeof d);
        d.q_off = q_off[i];
        d.t_off = t_off[i];
        if (q_len[i] > 0x7FFFFFF0ull || t_len[i] > 0x7FFFFFF0ull) { g_err = "sequence too long"; batch_free(b); return ALN_ERR_UNSUPPORTED; }
        d.N = pwm ? cols : (uint32_t)q_len[i];                                 // PWM: the columns are the PWM positions
        d.M = (uint32_t)t_len[i];
        d.status = ALN_OK;
        if (d.N == 0 || d.M == 0) d.status = ALN_ERR_EMPTY_SEQUENCE;          // reference panics (PWM: treated alike)
        else {
            const uint8_t *q = seqs + d.q_off, *t = seqs + d.t_off;
            if (!pwm) for (uint32_t k = 0; k < d.N && d.status == ALN_OK; ++k) if (q[k] >= cols) d.status = ALN_ERR_CODE_OUT_OF_RANGE;
            for (uint32_t k = 0; k < d.M && d.status == ALN_OK; ++k) if (t[k] >= rows) d.status = ALN_ERR_CODE_OUT_OF_RANGE;
        }
        if (d.status != ALN_OK) continue;
        cells += (uint64_t)d.N * d.M;
        max_len = std::max(max_len, std::max(d.N, d.M));
        max_span = std::max(max_span, (uint64_t)d.N + d.M + 2);
    }
    b->cells = cells;
    b->max_len = max_len;
    // integer kernels are exact iff every value is integral and |H| cannot leave i32
    b->is_int = all_int && !p->force_f64 && maxabs * (double)max_span < 1073741824.0;
    if (!core && !b->is_int) { g_err = "legacy scores overflow i32 for these lengths"; batch_free(b); return ALN_ERR_UNSUPPORTED; }
    // fast integer kernels: keys are 4*H + tag in i32 and the profile holds 4*s - 1 as int8
    double smin = 0, smax = 0;
    for (double v : md) { smin = std::min(smin, v); smax = std::max(smax, v); }
    b->fast = b->is_int && !want_h && !p->force_serial && !p->force_generic && (pwm || cols <= 64) && smin >= -31.0 && smax <= 32.0 &&
              maxabs * (double)max_span < 268435456.0;

    {
        const uint32_t outs = p->outputs ? p->outputs : (ALN_OUT_SCORE | ALN_OUT_TRACEBACK);
        b->store_dirs = (outs & (ALN_OUT_TRACEBACK | ALN_OUT_DIRECTIONS)) != 0;
    }
    // ---- routing + HBM layout.  A pair goes to the single-pair kernel (one wave per strip, strips pipelined across
    // CUs) when it is large, or when the batch is too small to fill the chip with one wave per pair.
    uint64_t dir_total = 0, tb_total = 0, hm_total = 0;
    const char *env_r = getenv("ALN_SINGLE_R");
    const char *env_off = getenv("ALN_NO_SINGLE");
    for (size_t i = 0; i < n; ++i) {
        PairDesc &d = b->descs[i];
        if (d.status != ALN_OK) continue;
        const uint64_t pc = (uint64_t)d.N * d.M;
        bool single = b->fast && !pwm && !env_off && d.N >= 64 && d.M >= 128 && (pc >= (1ull << 24) || (n <= 16 && pc >= (1ull << 18)));
        uint64_t dbytes = aln_dir_bytes(d.N, d.M);
        if (single) {
            uint32_t R = env_r ? (uint32_t)atoi(env_r) : (d.M > 4096 ? 2u : 1u);
            if (R != 1 && R != 2 && R != 4 && R != 8) R = 2;
            while ((d.M + 64 * R - 1) / (64 * R) > 4096 && R < 8) R *= 2;      // keep every strip's wave resident
            if ((d.M + 64 * R - 1) / (64 * R) > 4096) single = false;
            if ((uint64_t)rows * cols * 4 + (uint64_t)cols * 64 * R + 2ull * (d.N + 192) + 1024 > 65536) single = false;   // LDS budget
            if (single) {
                const uint32_t ns = (d.M + 64 * R - 1) / (64 * R), spb = 16 / R;
                dbytes = std::max<uint64_t>(dbytes, (uint64_t)ns * aln_uniform_strip_bytes(d.N, R));
                (void)spb;
                b->single_pairs.push_back((uint32_t)i);
                b->single_r.push_back(R);
                const uint64_t gstride = ((uint64_t)d.N + 64 + 63) & ~63ull;
                b->granule_bytes = std::max<uint64_t>(b->granule_bytes, std::max<uint64_t>((uint64_t)ns * gstride * 4, 4ull * (d.M + 2)));
                b->single_max_n = std::max(b->single_max_n, std::max(d.N, ns));
                b->tbmap_entries = std::max<uint64_t>(b->tbmap_entries, (uint64_t)ns * (d.N + 1) + ns + 64);
            }
        }
        d.dir_off = dir_total;
        if (b->store_dirs) {
            dir_total += dbytes;
            d.tb_off = tb_total;
            // aligned query, aligned target, traceback tag scratch (PWM: u32 column numbers, residues, tags)
            tb_total += ((pwm ? 6ull : 3ull) * ((uint64_t)d.N + d.M + 2) + 3) & ~3ull;
        }   // aligned query, aligned target, traceback tag scratch
        if (want_h) { d.h_off = hm_total; hm_total += (uint64_t)(d.N + 1) * (d.M + 1); }
    }
    b->dir_bytes = dir_total;
    b->tb_bytes = tb_total;
    b->hmat_elems = hm_total;

    // ---- LPT order: largest pairs first into the device work queue
    std::vector<uint32_t> order;
    order.reserve(n);
    {
        std::vector<char> is_single(n, 0);
        for (uint32_t i : b->single_pairs) is_single[i] = 1;
        for (size_t i = 0; i < n; ++i) if (!is_single[i]) order.push_back((uint32_t)i);
    }
    b->n_small = order.size();
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t c) {
        return (uint64_t)b->descs[a].N * b->descs[a].M > (uint64_t)b->descs[c].N * b->descs[c].M;
    });

    b->max_cells = order.empty() ? 0 : (uint64_t)b->descs[order[0]].N * b->descs[order[0]].M;

    // ---- grid: persistent waves, 4 per workgroup
    const uint32_t wg_needed = (uint32_t)((b->n_small + 3) / 4);
    b->grid = std::max(1u, std::min(wg_needed, (uint32_t)ctx->cus * 4u));
    // Overlapped traceback (aln_batch_run): the walk kernel runs beside the fill.  The fast fill kernel is built for 160 VGPRs,
    // three workgroups per CU, so that every SIMD keeps 32 registers free: exactly one wave of the walk kernel (32 VGPRs, no
    // LDS).  ALN_TB_OVERLAP: unset = one walk wave per SIMD beside a full fill grid; 0 = off; n > 0 = the earlier scheme (the
    // fill grid stops n workgroups short of residency and the walk waves crowd onto those CUs, 20 per slot).
    {
        const uint32_t outs = p->outputs ? p->outputs : (ALN_OUT_SCORE | ALN_OUT_TRACEBACK);
        const uint32_t resident = (uint32_t)ctx->cus * 3u;
        const char *e = getenv("ALN_TB_OVERLAP");
        const uint32_t reserve = e ? (uint32_t)atoi(e) : 0u;
        const bool off = e && reserve == 0;
        b->overlap = !off && b->fast && b->is_int && (outs & ALN_OUT_TRACEBACK) && reserve < resident &&
                     b->n_small >= 4096 && wg_needed >= resident;
        if (b->overlap) {
            b->grid = resident - reserve;
            b->tb_waves = reserve ? reserve * 20u : (uint32_t)ctx->cus * 4u;
            b->counter_bytes = 256 + 4ull * b->n_small;
        }
    }
    const uint64_t sc_size = b->is_int ? 4 : 8;
    const uint64_t brow_bytes = (((uint64_t)max_len + 66) * sc_size + 63) & ~63ull;
    const uint64_t adv_bytes = ((uint64_t)max_len + 66 + 63) & ~63ull;
    // + 4 checkpoints of strip 0's lane state (26 ints x 64 lanes) and a copy of strip 0's bottom row (fast path)
    const uint64_t ck_bytes = b->fast ? ((uint64_t)ALN_CK_SLOTS * 18 * 64 * 4 + (((uint64_t)max_len + 66) * 4 + 63 & ~63ull)) : 0;
    // bottom-row record: one byte per column (generic kernels) or one direction dword per block of the last strip (fast
    // path: at most (max_len + 63) / 2 + 4 blocks)
    const uint64_t zrow_bytes = std::max<uint64_t>(adv_bytes, (4ull * (((uint64_t)max_len + 63) / 2 + 8) + 63) & ~63ull);
    b->zrow_bytes = (uint32_t)zrow_bytes;
    b->scratch_stride = brow_bytes + adv_bytes + zrow_bytes + ck_bytes;
    b->lds_bytes = (uint32_t)(((uint64_t)rows * cols * sc_size + 15) & ~15ull);
    if (b->fast && !pwm) { b->prof_stride = cols * 64u * ALN_FULL_R; b->lds_bytes += 4u * b->prof_stride; }

    // ---- device allocations + H2D
    uint64_t seq_bytes = 0;
    for (size_t i = 0; i < n; ++i) {
        if (!pwm) seq_bytes = std::max(seq_bytes, q_off[i] + q_len[i]);
        seq_bytes = std::max(seq_bytes, t_off[i] + t_len[i]);
    }
    auto dmalloc = [&](void **ptr, uint64_t bytes) { return hipMalloc(ptr, std::max<uint64_t>(bytes, 256)); };
    hipError_t e;
#define BCHK(call) if ((e = (call)) != hipSuccess) { int st_ = fail(e, #call); batch_free(b); return st_; }
    BCHK(dmalloc((void **)&b->d_seqs, seq_bytes + 64));
    BCHK(dmalloc((void **)&b->d_descs, n * sizeof(PairDesc)));
    BCHK(dmalloc((void **)&b->d_order, n * sizeof(uint32_t)));
    BCHK(dmalloc((void **)&b->d_counter, b->counter_bytes));
    if (b->overlap) {
        BCHK(dmalloc((void **)&b->d_walked, 4ull * n));
        BCHK(hipMemset(b->d_walked, 0, 4ull * n));
        BCHK(hipStreamCreateWithFlags(&b->tb_stream, hipStreamNonBlocking));
        BCHK(hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming));
        BCHK(hipEventCreateWithFlags(&b->ev_join, hipEventDisableTiming));
    }
    BCHK(dmalloc((void **)&b->d_dirs, dir_total));
    BCHK(dmalloc((void **)&b->d_results, n * sizeof(aln_pair_result)));
    BCHK(dmalloc((void **)&b->d_tb, tb_total));
    BCHK(dmalloc((void **)&b->d_scratch, (uint64_t)b->grid * 4 * b->scratch_stride));
    BCHK(dmalloc((void **)&b->d_matrix, (uint64_t)rows * cols * sc_size));
    if (want_h) BCHK(dmalloc((void **)&b->d_hmat, hm_total * sc_size));
    if (!b->single_pairs.empty()) {
        BCHK(dmalloc((void **)&b->d_granules, b->granule_bytes));
        BCHK(dmalloc((void **)&b->d_advice1, 2ull * single_advice_bytes(b->single_max_n)));
        BCHK(dmalloc((void **)&b->d_cand, 16ull * (b->single_max_n + 64)));
        BCHK(dmalloc((void **)&b->d_ctrl, 256));
        BCHK(dmalloc((void **)&b->d_tbmap, b->tbmap_entries * 16));
    }
    if (n) {
        BCHK(hipMemcpy(b->d_seqs, seqs, seq_bytes, hipMemcpyHostToDevice));
        BCHK(hipMemcpy(b->d_descs, b->descs.data(), n * sizeof(PairDesc), hipMemcpyHostToDevice));
        if (!order.empty()) BCHK(hipMemcpy(b->d_order, order.data(), order.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    if (b->is_int) {
        std::vector<int32_t> mi(md.size());
        for (size_t i = 0; i < md.size(); ++i) mi[i] = (int32_t)md[i];
        BCHK(hipMemcpy(b->d_matrix, mi.data(), mi.size() * 4, hipMemcpyHostToDevice));
    } else {
        BCHK(hipMemcpy(b->d_matrix, md.data(), md.size() * 8, hipMemcpyHostToDevice));
    }
    if (pwm && b->fast) {
        std::vector<uint32_t> words(cols);
        for (uint32_t c = 0; c < cols; ++c) {
            uint32_t wv = 0;
            for (uint32_t r = 0; r < 4; ++r) wv |= ((uint32_t)(int32_t)(4 * (int32_t)md[(size_t)r * cols + c] - 2) & 0xffu) << (8 * r);
            words[c] = wv;
        }
        BCHK(dmalloc((void **)&b->d_pwm_words, (uint64_t)cols * 4));
        BCHK(hipMemcpy(b->d_pwm_words, words.data(), (size_t)cols * 4, hipMemcpyHostToDevice));
    }
    BCHK(hipMemset(b->d_results, 0, std::max<uint64_t>(n * sizeof(aln_pair_result), 1)));
#undef BCHK
    *out = b;
    return ALN_OK;
}

extern "C" aln_batch *aln_batch_create(aln_ctx *ctx, const aln_params *params, const uint8_t *seqs,
                                       const uint64_t *q_off, const uint64_t *q_len, const uint64_t *t_off,
                                       const uint64_t *t_len, size_t n_pairs, int *status)
{
    aln_batch *b = nullptr;
    int st = batch_build(ctx, params, seqs, q_off, q_len, t_off, t_len, n_pairs, false, &b);
    if (status) *status = st;
    return b;
}

extern "C" void aln_batch_enable_timing(aln_batch *b, int on)
{
    if (!b) return;
    b->timing = on != 0;
    b->ev_runs = 0;
    if (b->timing && b->ev.empty()) {
        (void)hipSetDevice(b->ctx->device);
        b->ev.resize(3 * ALN_TIMING_SLOTS, nullptr);
        for (auto &e : b->ev) (void)hipEventCreate(&e);
    }
}

extern "C" int aln_batch_run(aln_batch *b, void *stream)
{
    if (!b) return ALN_ERR_INVALID_ARGUMENT;
    HIPCHK(hipSetDevice(b->ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : b->ctx->stream;
    b->last_stream = s;
    if (b->n == 0) return ALN_OK;
    HIPCHK(hipMemsetAsync(b->d_counter, 0, b->counter_bytes, s));
    FillArgs fa{};
    fa.seqs = b->d_seqs; fa.descs = b->d_descs; fa.order = b->d_order; fa.n_pairs = (uint32_t)b->n_small;
    fa.counter = b->d_counter; fa.dirs = b->d_dirs; fa.results = b->d_results;
    fa.scratch = b->d_scratch; fa.scratch_stride = b->scratch_stride; fa.max_len = b->max_len; fa.zrow_bytes = b->zrow_bytes;
    fa.matrix = b->d_matrix; fa.rows = b->params.rows; fa.cols = b->params.cols; fa.prof_stride = b->prof_stride;
    fa.del = b->params.del; fa.ext = b->params.ext; fa.semantics = b->params.semantics;
    fa.max_passes = b->params.max_passes; fa.force_serial = b->params.force_serial;
    fa.no_repair = getenv("ALN_NO_REPAIR") ? 1u : 0u;
    fa.max_cells = b->max_cells;
    fa.store_dirs = b->store_dirs ? 1u : 0u;
    fa.pwm = b->pwm ? 1u : 0u;
    fa.pwm_words = b->d_pwm_words;
    fa.hmat = b->d_hmat; fa.blank = b->params.blank_code;
    hipEvent_t *ev = b->timing ? &b->ev[3 * (b->ev_runs % ALN_TIMING_SLOTS)] : nullptr;
    if (ev) HIPCHK(hipEventRecord(ev[0], s));
    b->fill_launches = 0;
    const uint32_t outs = b->params.outputs ? b->params.outputs : (ALN_OUT_SCORE | ALN_OUT_TRACEBACK);
    TraceArgs ta{};
    ta.seqs = b->d_seqs; ta.descs = b->d_descs; ta.n_pairs = (uint32_t)b->n; ta.dirs = b->d_dirs;
    ta.results = b->d_results; ta.tb = b->d_tb; ta.semantics = b->params.semantics; ta.blank = b->params.blank_code; ta.pwm = b->pwm ? 1 : 0;
    const bool overlap = b->overlap && b->store_dirs && (outs & ALN_OUT_TRACEBACK);
    if (overlap) {
        fa.doneq = b->d_counter + 64;
        ta.walked = b->d_walked; ta.epoch = ++b->epoch; ta.n_order = (uint32_t)b->n_small;
        ta.doneq = b->d_counter + 64; ta.head = b->d_counter + 2; ta.wait_ticks = 50000000ull;          // 0.5 s
        if (const char *w = getenv("ALN_TB_WAIT_US")) ta.wait_ticks = 100ull * strtoull(w, nullptr, 10);   // testing: 0 = give up at once
        HIPCHK(hipEventRecord(b->ev_fork, s));                               // after the memset, before the fill
    }
    if (b->n_small) {
        aln_launch_fill(&fa, b->is_int ? 1 : 0, b->fast ? 1 : 0, b->grid, b->lds_bytes, s);
        HIPCHK(hipGetLastError());
        b->fill_launches = 1;
    }
    if (overlap) {                                                           // submitted after the fill, runs beside it
        HIPCHK(hipStreamWaitEvent(b->tb_stream, b->ev_fork, 0));
        aln_launch_traceback_overlap(&ta, b->tb_waves, b->tb_stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(b->ev_join, b->tb_stream));
    }
    for (size_t j = 0; j < b->single_pairs.size(); ++j) {
        const PairDesc &d = b->descs[b->single_pairs[j]];
        SingleArgs sa{};
        sa.seqs = b->d_seqs; sa.descs = b->d_descs; sa.pair = b->single_pairs[j]; sa.dirs = b->d_dirs;
        sa.results = b->d_results; sa.granules = b->d_granules;
        sa.gstride = ((uint64_t)d.N + 64 + 63) & ~63ull;
        sa.advice = b->d_advice1; sa.zrow = b->d_advice1 + single_advice_bytes(b->single_max_n);
        sa.cand = b->d_cand; sa.ctrl = b->d_ctrl; sa.matrix = b->d_matrix;
        sa.rows = b->params.rows; sa.cols = b->params.cols; sa.del = b->params.del; sa.ext = b->params.ext;
        sa.semantics = b->params.semantics; sa.R = b->single_r[j];
        sa.ns = (d.M + 64 * sa.R - 1) / (64 * sa.R);
        sa.hazard = (sa.semantics == ALN_CORE_LOCAL && sa.del != sa.ext && d.N >= 2) ? 1u : 0u;
        sa.store_dirs = b->store_dirs ? 1u : 0u;
        { const char *td = getenv("ALN_TEST_DROP_STRIP"); sa.test_drop = td ? (uint32_t)atoi(td) : 0u; }
        sa.max_passes = sa.hazard ? std::min<uint32_t>(b->params.max_passes ? b->params.max_passes : 4u, 12u) : 1u;
        aln_launch_single_init(&sa, (uint32_t)single_advice_bytes(d.N), s);
        // the granule rows must read "not yet produced" before a pass: one memset here, later passes are zeroed by the
        // finalize kernel that arms them
        HIPCHK(hipMemsetAsync(b->d_granules, 0, (size_t)sa.ns * sa.gstride * 4, s));
        for (uint32_t pass = 0; pass < sa.max_passes; ++pass) {
            sa.pass = pass;
            aln_launch_single(&sa, d.N, pass + 1 == sa.max_passes ? 1 : 0, s);
            b->fill_launches++;
        }
        HIPCHK(hipGetLastError());
    }
    if (ev) HIPCHK(hipEventRecord(ev[1], s));
    if ((outs & ALN_OUT_TRACEBACK) && b->store_dirs) {
        if (overlap) HIPCHK(hipStreamWaitEvent(s, b->ev_join, 0));
        aln_launch_traceback(&ta, s);       // every pair except those in the uniform-R layout (handled below)
        for (size_t j = 0; j < b->single_pairs.size(); ++j) {
            const PairDesc &d = b->descs[b->single_pairs[j]];
            TraceSingleArgs tsa{};
            tsa.seqs = b->d_seqs; tsa.descs = b->d_descs; tsa.pair = b->single_pairs[j]; tsa.dirs = b->d_dirs;
            tsa.results = b->d_results; tsa.tb = b->d_tb; tsa.semantics = b->params.semantics;
            tsa.R = b->single_r[j]; tsa.ns = (d.M + 64 * tsa.R - 1) / (64 * tsa.R);
            tsa.map = b->d_tbmap; tsa.seg = b->d_tbmap + (uint64_t)tsa.ns * (d.N + 1);
            aln_launch_traceback_single(&tsa, d.N, s);
            aln_launch_traceback_expand_single(&ta, tsa.pair, s);
        }
        aln_launch_traceback_expand(&ta, s);
        HIPCHK(hipGetLastError());
    }
    if (ev) { HIPCHK(hipEventRecord(ev[2], s)); b->ev_runs++; }
    return ALN_OK;
}

extern "C" int aln_batch_sync(aln_batch *b)
{
    if (!b) return ALN_ERR_INVALID_ARGUMENT;
    HIPCHK(hipSetDevice(b->ctx->device));
    HIPCHK(hipStreamSynchronize(b->last_stream ? b->last_stream : b->ctx->stream));
    return ALN_OK;
}

extern "C" int aln_batch_timing(aln_batch *b, double *fill_ms, double *tb_ms, uint32_t *fill_launches)
{
    if (!b || !b->timing || b->ev.empty() || b->ev_runs == 0) return ALN_ERR_INVALID_ARGUMENT;
    // mean over the runs recorded since aln_batch_enable_timing (at most the last ALN_TIMING_SLOTS)
    const uint32_t runs = std::min<uint32_t>(b->ev_runs, ALN_TIMING_SLOTS);
    double fs = 0, ts = 0;
    for (uint32_t r = 0; r < runs; ++r) {
        hipEvent_t *ev = &b->ev[3 * r];
        float f = 0, t = 0;
        HIPCHK(hipEventSynchronize(ev[2]));
        HIPCHK(hipEventElapsedTime(&f, ev[0], ev[1]));
        HIPCHK(hipEventElapsedTime(&t, ev[1], ev[2]));
        fs += f; ts += t;
    }
    if (fill_ms) *fill_ms = fs / runs;
    if (tb_ms) *tb_ms = ts / runs;
    if (fill_launches) *fill_launches = b->fill_launches;
    return ALN_OK;
}

extern "C" int aln_batch_fetch(aln_batch *b, aln_pair_result *results, uint8_t *tb_buf, const uint64_t *tb_off)
{
    if (!b || !results) return ALN_ERR_INVALID_ARGUMENT;
    int st = aln_batch_sync(b);
    if (st != ALN_OK) return st;
    if (b->n == 0) return ALN_OK;
    HIPCHK(hipMemcpy(results, b->d_results, b->n * sizeof(aln_pair_result), hipMemcpyDeviceToHost));
    if (tb_buf && tb_off && b->tb_bytes) {
        std::vector<uint8_t> h(b->tb_bytes);
        HIPCHK(hipMemcpy(h.data(), b->d_tb, b->tb_bytes, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < b->n; ++i) {
            const PairDesc &d = b->descs[i];
            if (results[i].status != ALN_OK) continue;
            const uint64_t cap = (uint64_t)d.N + d.M + 2;
            if (b->pwm) {   // u32 column numbers, then the residue string
                memcpy(tb_buf + tb_off[i], h.data() + d.tb_off, 4ull * results[i].aln_len);
                memcpy(tb_buf + tb_off[i] + 4 * cap, h.data() + d.tb_off + 4 * cap, results[i].aln_len);
            } else {
                memcpy(tb_buf + tb_off[i], h.data() + d.tb_off, results[i].aln_len);
                memcpy(tb_buf + tb_off[i] + cap, h.data() + d.tb_off + cap, results[i].aln_len);
            }
        }
    }
    return ALN_OK;
}

extern "C" uint64_t aln_batch_cells(const aln_batch *b) { return b ? b->cells : 0; }
extern "C" size_t aln_batch_size(const aln_batch *b) { return b ? b->n : 0; }
extern "C" void *aln_batch_results_device(aln_batch *b) { return b ? b->d_results : nullptr; }
extern "C" uint64_t aln_batch_direction_bytes(const aln_batch *b)
{
    if (!b) return 0;
    // bytes the fill kernel actually stores: per strip, ceil(steps / SPB) blocks of 256 B
    uint64_t total = 0;
    for (const PairDesc &d : b->descs) {
        if (d.status != ALN_OK) continue;
        const uint32_t ns = aln_num_strips(d.M);
        for (uint32_t s = 0; s < ns; ++s) {
            const bool last = s + 1 == ns;
            const uint32_t rem = d.M - s * ALN_STRIP_ROWS;
            const int R = last ? aln_pick_r(rem) : ALN_FULL_R;
            const uint32_t rows = std::min<uint32_t>(rem, 64u * R), L = (rows + R - 1) / R, spb = 16 / R;
            total += (uint64_t)aln_strip_blocks(d.N + L - 1, spb) * 256u;
        }
    }
    return total;
}

extern "C" int aln_align_batch(aln_ctx *ctx, const aln_params *params, const uint8_t *seqs, const uint64_t *q_off,
                               const uint64_t *q_len, const uint64_t *t_off, const uint64_t *t_len, size_t n_pairs,
                               aln_pair_result *results, uint8_t *tb_buf, const uint64_t *tb_off)
{
    aln_batch *b = nullptr;
    int st = batch_build(ctx, params, seqs, q_off, q_len, t_off, t_len, n_pairs, false, &b);
    if (st != ALN_OK) return st;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        st = aln_batch_run(b, nullptr);
        if (st == ALN_OK) st = aln_batch_fetch(b, results, tb_buf, tb_off);
    }
    batch_free(b);
    return st;
}

extern "C" int aln_align_pair(aln_ctx *ctx, const aln_params *params, const uint8_t *query, size_t N,
                              const uint8_t *target, size_t M, aln_pair_result *out, uint8_t *q_aln, uint8_t *t_aln,
                              uint8_t *directions, double *h_matrix)
{
    if (!ctx || !params || !out) { g_err = "null argument"; return ALN_ERR_INVALID_ARGUMENT; }
    const bool pwm = params->semantics == ALN_PWM_LOCAL;
    if (pwm) { N = params->cols; query = nullptr; }          // the columns are the PWM positions (pwm/mod.rs:44-46)
    const size_t nq = pwm ? 0 : N;
    std::vector<uint8_t> seqs(nq + M + 1);
    if (nq) memcpy(seqs.data(), query, nq);
    if (M) memcpy(seqs.data() + nq, target, M);
    const uint64_t qo = 0, ql = N, to = nq, tl = M;
    aln_params p = *params;
    p.outputs = (params->outputs ? params->outputs : (ALN_OUT_SCORE | ALN_OUT_TRACEBACK));
    if (q_aln && t_aln) p.outputs |= ALN_OUT_TRACEBACK;
    if (directions) p.outputs |= ALN_OUT_DIRECTIONS;
    aln_batch *b = nullptr;
    int st = batch_build(ctx, &p, seqs.data(), &qo, &ql, &to, &tl, 1, h_matrix != nullptr, &b);
    if (st != ALN_OK) { memset(out, 0, sizeof *out); out->status = st; return st; }
    std::lock_guard<std::mutex> lk(ctx->mu);
    st = aln_batch_run(b, nullptr);
    const size_t cap = N + M + 2;
    std::vector<uint8_t> tb((pwm ? 5 : 2) * cap + 8);
    const uint64_t tbo = 0;
    if (st == ALN_OK) st = aln_batch_fetch(b, out, tb.data(), &tbo);
    if (st == ALN_OK && out->status == ALN_OK) {
        if (q_aln && t_aln) {
            if (pwm) {
                memcpy(q_aln, tb.data(), 4ull * out->aln_len);          // uint32 PWM column numbers
                memcpy(t_aln, tb.data() + 4 * cap, out->aln_len);
            } else {
                memcpy(q_aln, tb.data(), out->aln_len);
                memcpy(t_aln, tb.data() + cap, out->aln_len);
            }
        }
        const uint64_t cells = (uint64_t)(N + 1) * (M + 1);
        if (directions) {
            uint8_t *d_out = nullptr;
            hipError_t e = hipMalloc((void **)&d_out, cells);
            if (e != hipSuccess) { st = fail(e, "hipMalloc(directions)"); }
            else {
                aln_launch_unpack(b->d_dirs, b->d_descs, 0, b->params.semantics, d_out, cells, b->ctx->stream);
                e = hipStreamSynchronize(b->ctx->stream);
                if (e == hipSuccess) e = hipMemcpy(directions, d_out, cells, hipMemcpyDeviceToHost);
                if (e != hipSuccess) st = fail(e, "unpack directions");
                (void)hipFree(d_out);
            }
        }
        if (h_matrix && st == ALN_OK) {
            if (b->is_int) {
                std::vector<int32_t> hi(cells);
                hipError_t e = hipMemcpy(hi.data(), b->d_hmat, cells * 4, hipMemcpyDeviceToHost);
                if (e != hipSuccess) st = fail(e, "fetch H");
                else for (uint64_t i = 0; i < cells; ++i) h_matrix[i] = (double)hi[i];
            } else {
                hipError_t e = hipMemcpy(h_matrix, b->d_hmat, cells * 8, hipMemcpyDeviceToHost);
                if (e != hipSuccess) st = fail(e, "fetch H");
            }
        }
    }
    batch_free(b);
    return st != ALN_OK ? st : out->status;
}
