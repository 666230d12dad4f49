// aln_host.hip -- host side of the C ABI declared in include/aligner_hip.h.
//
// The reference's callers own host buffers (Vec<T> per sequence, statistics/mod.rs:255-286; simple/mod.rs:35-40) and get
// owned results back, so the entry points take HOST pointers and everything between them and the kernels lives here:
//
//   plan      per chunk of pairs: validation of the lengths, routing (batch kernel / single-pair kernel), LPT order of the
//             device work queue, layout of the direction, string and tag regions            (chunk_plan)
//   pool      every context owns a few SLOTS: device buffers (grow-only), a stream, pinned staging for the small tables --
//             nothing is hipMalloc'ed per call once the pool is warm                         (Slot, slot_ensure)
//   pipeline  aln_align_batch cuts the batch into chunks of ~1e10 cells in the CALLER's pair order, so that a chunk's
//             residues, summaries and aligned strings are contiguous spans of the caller's buffers: one H2D and two D2H
//             copies per chunk, straight from / into the caller's memory (no bounce, no per-pair memcpy; the runtime moves
//             pageable memory at the link rate, profiles/r02_host_link.txt).  Chunk i+1 is staged and chunk i-1 is fetched
//             (by a second host thread) while chunk i fills; chunks run on different slots = different streams, so the
//             tail of one fill, the traceback behind it and the head of the next fill overlap on the device, and the
//             directions of a chunk only live until its traceback is done (4 x ~2.5 GB instead of 35 GB for C5).
//   staged    aln_batch_* keeps ONE chunk = the whole batch resident in a private slot (inputs in HBM before the timed
//             region: what bench.py's headline measures) -- same plan, same launches.
//
// One context per process per GPU; multi-GPU runs are one process per GPU (the Python driver shards pairs across ranks and
// gathers the 48-byte summaries with RCCL through torch.distributed).  There is NO CPU implementation of the DP here: if the
// device or the kernels are unavailable every entry point fails with ALN_ERR_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <numeric>
#include <string>
#include <chrono>
#include <thread>
#include <vector>

#include "aln_device.h"

#define ALN_TIMING_SLOTS 256u
// HIP multiplexes streams onto 4 hardware queues by default (GPU_MAX_HW_QUEUES): with more slots than that two chunks share a
// queue and wait for each other (measured: 6 slots 67 ms, 4 slots 57 ms for the C5 batch)
#define ALN_POOL_SLOTS 4

extern "C" void aln_launch_fill(const FillArgs *a, int is_int, int fast, uint32_t grid, uint32_t lds_bytes, hipStream_t s);
extern "C" void aln_launch_validate(const uint8_t *seqs, PairDesc *descs, uint32_t n_pairs, uint32_t rows, uint32_t cols, int pwm,
                                    hipStream_t s);
extern "C" void aln_launch_traceback(const TraceArgs *a, hipStream_t s);
extern "C" void aln_launch_traceback_wave(const TraceArgs *a, hipStream_t s);
extern "C" void aln_launch_scale_results(aln_pair_result *results, uint32_t n, double factor, hipStream_t s);
extern "C" void aln_launch_traceback_overlap(const TraceArgs *a, uint32_t waves, hipStream_t s);
extern "C" void aln_launch_traceback_expand(const TraceArgs *a, hipStream_t s);
extern "C" void aln_launch_traceback_single(const TraceSingleArgs *a, uint32_t N, hipStream_t s);
extern "C" void aln_launch_traceback_expand_single(const TraceArgs *a, uint32_t pair, hipStream_t s);
extern "C" void aln_launch_single(const SingleArgs *a, uint32_t N, int with_serial, hipStream_t s);
extern "C" void aln_launch_wgpipe(const WgArgs *a, int is_int, uint32_t lds_bytes, hipStream_t s);
extern "C" void aln_launch_single_init(const SingleArgs *a, uint32_t n_bytes, hipStream_t s);
extern "C" void aln_launch_single_repair(const SingleArgs *a, uint32_t N, hipStream_t s);
extern "C" uint32_t aln_single_lds_bytes(uint32_t rows, uint32_t cols, uint32_t R, uint32_t N, uint32_t W);
extern "C" void aln_launch_unpack(const uint8_t *dirs, const PairDesc *descs, uint32_t pair, int semantics, uint8_t *out,
                                  uint64_t cells, hipStream_t s);

static thread_local std::string g_err;

// HIP multiplexes the streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4).  A pipelined batch call keeps
// four streams busy at once; when other streams of the process (a framework's, a staged batch's) are mapped onto the same
// queues, two chunks end up in one queue and wait for each other (C5: 52 ms -> 60-63 ms; eight queues remove that,
// profiles/r02_e2e_chunking.txt).  The library does NOT touch the environment (it did in r02: a constructor that called setenv in
// someone else's process): a host that runs other streams beside batch calls exports GPU_MAX_HW_QUEUES=8 itself before the HIP
// runtime initialises -- bench.py and the test harness do ("Environment").

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

// ---------------------------------------------------------------- grow-only buffers
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};
static int dev_ensure(DevBuf &b, size_t bytes, bool slack)
{
    bytes = std::max<size_t>(bytes, 256);
    if (b.cap >= bytes) return ALN_OK;
    if (b.p) { (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }
    size_t want = slack ? bytes + bytes / 8 : bytes;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess && want != bytes) { want = bytes; e = hipMalloc(&b.p, want); }
    if (e != hipSuccess) { b.p = nullptr; return fail(e, "hipMalloc"); }
    b.cap = want;
    return ALN_OK;
}
static void dev_free(DevBuf &b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr; b.cap = 0;
}
struct PinBuf {
    void *p = nullptr;
    size_t cap = 0;
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};
static int pin_ensure(PinBuf &b, size_t bytes)
{
    bytes = std::max<size_t>(bytes, 4096);
    if (b.cap >= bytes) return ALN_OK;
    if (b.p) { (void)hipHostFree(b.p); b.p = nullptr; b.cap = 0; }
    const size_t want = bytes + bytes / 4;
    hipError_t e = hipHostMalloc(&b.p, want, hipHostMallocDefault);
    if (e != hipSuccess) { b.p = nullptr; return fail(e, "hipHostMalloc"); }
    b.cap = want;
    return ALN_OK;
}
static void pin_free(PinBuf &b)
{
    if (b.p) (void)hipHostFree(b.p);
    b.p = nullptr; b.cap = 0;
}

// ---------------------------------------------------------------- a slot: everything one chunk needs on the device
struct Slot {
    bool pooled = true;       // pool slots keep some slack when they grow; a staged batch's private slot is sized exactly
    DevBuf seqs, descs, order, counter, walked, dirs, results, tb, tags, scratch, matrix, pwm_words, hmat;
    DevBuf granules, advice1, cand, ctrl, tbmap, unpack, repair, coop;
    PinBuf h_meta;            // descs + order + matrix + pwm words (small, truly asynchronous H2D)
    PinBuf h_in, h_out;       // fallback staging: sequences gathered from scattered offsets / strings for a foreign tb layout
    hipStream_t stream = nullptr;
    hipStream_t tb_stream = nullptr;                 // in-kernel overlapped traceback (staged batches)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_done = nullptr, ev_fill = nullptr;
    uint32_t epoch = 0;
    bool walked_clean = false;
    // fast batch kernels: the boundary rows in `scratch` hold tagged granules (aln_coop_tag); the tag's salt counts this slot's
    // launches, and the rows are cleared whenever the buffer is new or the salt's 10 bits wrap
    uint32_t salt = 0;
    bool scratch_clean = false;
};

// one GPU of a context: its properties and its slot pool
struct DevCtx {
    int device = 0;
    int cus = 0;
    size_t hbm = 0;
    char name[128] = {0};
    std::mutex mu;                                   // guards the pool
    std::condition_variable cv;
    Slot *slots[ALN_POOL_SLOTS] = {nullptr};
    bool busy[ALN_POOL_SLOTS] = {false};
};

// A context spans one or more GPUs of this process (aln_create: one; aln_create_multi: a list).  Single calls go to the
// devices in turn; a batch call is cut into chunks that the devices take from a common queue.
struct aln_ctx {
    std::vector<DevCtx *> devs;
    std::atomic<uint32_t> turn{0};
    DevCtx *next_device() { return devs[turn.fetch_add(1, std::memory_order_relaxed) % devs.size()]; }
};

static int slot_init(Slot &s)
{
    if (!s.stream) HIPCHK(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    if (!s.ev_done) HIPCHK(hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming));
    if (!s.ev_fill) HIPCHK(hipEventCreateWithFlags(&s.ev_fill, hipEventDisableTiming));
    return ALN_OK;
}
static void slot_destroy(Slot *s)
{
    if (!s) return;
    DevBuf *d[] = {&s->seqs, &s->descs, &s->order, &s->counter, &s->walked, &s->dirs, &s->results, &s->tb, &s->tags, &s->scratch,
                   &s->matrix, &s->pwm_words, &s->hmat, &s->granules, &s->advice1, &s->cand, &s->ctrl, &s->tbmap, &s->unpack, &s->repair, &s->coop};
    for (DevBuf *b : d) dev_free(*b);
    pin_free(s->h_meta); pin_free(s->h_in); pin_free(s->h_out);
    if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
    if (s->ev_join) (void)hipEventDestroy(s->ev_join);
    if (s->ev_done) (void)hipEventDestroy(s->ev_done);
    if (s->ev_fill) (void)hipEventDestroy(s->ev_fill);
    if (s->tb_stream) (void)hipStreamDestroy(s->tb_stream);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

// takes between 1 and `want` free pool slots (blocks while none is free); concurrent callers share the pool
static int pool_lease(DevCtx *ctx, int want, Slot **out)
{
    std::unique_lock<std::mutex> lk(ctx->mu);
    int got = 0;
    for (;;) {
        for (int i = 0; i < ALN_POOL_SLOTS && got < want; ++i)
            if (!ctx->busy[i]) {
                if (!ctx->slots[i]) ctx->slots[i] = new Slot();
                ctx->busy[i] = true;
                out[got++] = ctx->slots[i];
            }
        if (got) return got;
        ctx->cv.wait(lk);
    }
}
static void pool_release(DevCtx *ctx, Slot **slots, int n)
{
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        for (int k = 0; k < n; ++k)
            for (int i = 0; i < ALN_POOL_SLOTS; ++i)
                if (ctx->slots[i] == slots[k]) ctx->busy[i] = false;
    }
    ctx->cv.notify_all();
}

extern "C" const char *aln_last_error(void) { return g_err.c_str(); }
extern "C" int aln_abi_version(void) { return ALN_ABI_VERSION; }

// aln_create's warm-up: what a first call used to pay for (26-31 ms for a first 1000 x 1000 pair against 0.4 ms for the second;
// aligner-cli aligns ONE pair per process, aligner-cli/main.rs:41-53).  Per device: the code objects are loaded (one per kernel
// translation unit, aln_warm_*), and one synthetic 1000 x 1000 pair goes through aln_align_pair -- pool slot 0 with its stream,
// events and staging, the buffers of the single-pair route at the size of the usual pair, the first launch of every kernel on
// that route.  ALN_NO_WARMUP=1 leaves all of it to the first call as before.  A failure here is not an error of aln_create: the
// first real call meets the same condition and reports it.
extern "C" int aln_warm_fast_cl(void);
extern "C" int aln_warm_fast_cl_solo(void);
extern "C" int aln_warm_fast_rest(void);
extern "C" int aln_warm_fast_rest_solo(void);
extern "C" int aln_warm_generic(void);
extern "C" int aln_warm_single(void);
extern "C" int aln_warm_tb(void);
static void warm_context(aln_ctx *c)
{
    if (getenv("ALN_NO_WARMUP")) return;
    const uint32_t L = 1000;
    std::vector<uint8_t> q(L), t(L);
    uint32_t x = 12345u;
    for (uint32_t i = 0; i < L; ++i) {
        x = x * 1664525u + 1013904223u; q[i] = (uint8_t)((x >> 16) % 20u);
        x = x * 1664525u + 1013904223u; t[i] = (i % 3u) ? q[i] : (uint8_t)((x >> 16) % 20u);
    }
    std::vector<double> mat(20 * 20);
    for (int i = 0; i < 20; ++i) for (int j = 0; j < 20; ++j) mat[i * 20 + j] = i == j ? 5.0 : -2.0;
    aln_params p;
    memset(&p, 0, sizeof p);
    p.semantics = ALN_CORE_LOCAL; p.del = 11.0; p.ext = 2.0;
    p.matrix = mat.data(); p.rows = 20; p.cols = 20; p.row_stride = 20;
    p.outputs = ALN_OUT_SCORE | ALN_OUT_TRACEBACK;
    std::vector<uint8_t> qa(2 * L + 2), ta(2 * L + 2);
    for (size_t d = 0; d < c->devs.size(); ++d) {
        if (hipSetDevice(c->devs[d]->device) != hipSuccess) continue;
        (void)aln_warm_single(); (void)aln_warm_tb(); (void)aln_warm_generic();
        (void)aln_warm_fast_cl(); (void)aln_warm_fast_cl_solo(); (void)aln_warm_fast_rest(); (void)aln_warm_fast_rest_solo();
        aln_pair_result r;
        (void)aln_align_pair(c, &p, q.data(), L, t.data(), L, &r, qa.data(), ta.data(), nullptr, nullptr);   // devices in turn
    }
    c->turn.store(0);
    (void)hipGetLastError();
    g_err.clear();
}

extern "C" aln_ctx *aln_create_multi(int n_devices, const int *device_ids, int *status)
{
    int st = ALN_OK;
    aln_ctx *c = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_err = e != hipSuccess ? std::string("hipGetDeviceCount: ") + hipGetErrorString(e) : "no HIP device visible";
        st = ALN_ERR_DEVICE;
    } else if (n_devices < 0 || n_devices > 64 || (n_devices > 0 && !device_ids)) {
        g_err = "bad device list";
        st = ALN_ERR_INVALID_ARGUMENT;
    } else {
        std::vector<int> ids;
        if (n_devices == 0) for (int i = 0; i < ndev; ++i) ids.push_back(i);          // every visible device
        else ids.assign(device_ids, device_ids + n_devices);
        c = new aln_ctx();
        for (int id : ids) {
            if (id < 0 || id >= ndev) { g_err = "device id out of range"; st = ALN_ERR_INVALID_ARGUMENT; break; }
            hipDeviceProp_t prop;
            if ((e = hipSetDevice(id)) != hipSuccess || (e = hipGetDeviceProperties(&prop, id)) != hipSuccess) {
                st = fail(e, "hipSetDevice/hipGetDeviceProperties");
                break;
            }
            DevCtx *d = new DevCtx();
            d->device = id;
            d->cus = prop.multiProcessorCount;
            d->hbm = prop.totalGlobalMem;
            snprintf(d->name, sizeof d->name, "%s (%s)", prop.name, prop.gcnArchName);
            c->devs.push_back(d);
        }
        if (st != ALN_OK) {
            for (DevCtx *d : c->devs) delete d;
            delete c;
            c = nullptr;
        }
    }
    if (c) warm_context(c);
    if (status) *status = st;
    return c;
}

extern "C" aln_ctx *aln_create(int device_id, int *status) { return aln_create_multi(1, &device_id, status); }

extern "C" void aln_destroy(aln_ctx *ctx)
{
    if (!ctx) return;
    for (DevCtx *d : ctx->devs) {
        (void)hipSetDevice(d->device);
        for (Slot *s : d->slots) slot_destroy(s);
        delete d;
    }
    delete ctx;
}

extern "C" int aln_device_count(const aln_ctx *ctx) { return ctx ? (int)ctx->devs.size() : 0; }

extern "C" int aln_device_info(aln_ctx *ctx, int *cus, size_t *hbm, char *name, size_t cap)
{
    if (!ctx) return ALN_ERR_INVALID_ARGUMENT;
    const DevCtx *d = ctx->devs[0];
    if (cus) *cus = d->cus;
    if (hbm) *hbm = d->hbm;
    if (name && cap) { strncpy(name, d->name, cap - 1); name[cap - 1] = 0; }
    return ALN_OK;
}

// ---------------------------------------------------------------- per call: arguments of perform_alignment, analysed once
struct Call {
    aln_params p{};
    bool pwm = false, core = false;
    uint32_t rows = 0, cols = 0;
    std::vector<double> md;           // the matrix, compact row-major
    bool all_int = false;
    double maxabs = 0, smin = 0, smax = 0;
    uint32_t outs = 0;
    bool store_dirs = true, want_tb = true, want_h = false;
    bool is_int = true, fast = false; // decided from the whole batch (the longest pair), the same for every chunk
    double unscale = 1.0;             // dyadic schemes on the integer kernels: scores come back multiplied by this (2^-k), see call_init
    int semantics = 0;                // what the kernels run (PWM runs as CORE_LOCAL with position-specific scoring)
};

static bool integral(double v) { return std::isfinite(v) && v == std::floor(v) && std::fabs(v) < 1e9; }

// bytes of the single-pair kernel's advice array and of its bottom-row record (one direction dword per block of the last
// strip, at most (N + 63) / 2 + 4 blocks at R = 8), equal sizes, 256-aligned
static inline uint64_t single_advice_bytes(uint64_t N)
{
    const uint64_t a = N + 128, z = 4 * ((N + 63) / 2 + 8);
    return ((a > z ? a : z) + 255) & ~255ull;
}

static int call_init(Call &c, const aln_params *p, const uint64_t *q_len, const uint64_t *t_len, size_t n, bool want_h)
{
    if (!p) { g_err = "null argument"; return ALN_ERR_INVALID_ARGUMENT; }
    if (p->semantics < ALN_CORE_GLOBAL || p->semantics > ALN_PWM_LOCAL) { g_err = "bad semantics"; return ALN_ERR_INVALID_ARGUMENT; }
    c.p = *p;
    c.pwm = p->semantics == ALN_PWM_LOCAL;
    c.core = p->semantics == ALN_CORE_GLOBAL || p->semantics == ALN_CORE_LOCAL || c.pwm;
    c.semantics = c.pwm ? ALN_CORE_LOCAL : p->semantics;       // same recurrence and tie rules; only the score lookup differs
    // simple/mod.rs:49-51 / :175-177
    if (c.core && p->heuristics_present) return ALN_ERR_UNNECESSARY_ARGUMENT;
    if (!p->matrix || p->rows == 0 || p->cols == 0) { g_err = "matrix missing"; return ALN_ERR_INVALID_ARGUMENT; }
    if (c.pwm && p->rows != 4) return ALN_ERR_MATRIX_SHAPE;                    // pwm/mod.rs:40-42
    // the matrix lives in LDS: 32 KiB next to the query profiles; a position-weight matrix (no profiles in LDS) may be
    // 4 x 2000 wide (62.5 KiB as f64)
    if ((uint64_t)p->rows * p->cols > (c.pwm ? 8000u : 4096u)) {
        g_err = c.pwm ? "position-weight matrix larger than 8000 entries (4 x 2000)" : "substitution matrix larger than 4096 entries";
        return ALN_ERR_UNSUPPORTED;
    }
    if (n > 0xFFFFFFF0ull) { g_err = "too many pairs"; return ALN_ERR_UNSUPPORTED; }
    c.rows = p->rows; c.cols = p->cols;
    const int64_t rs = p->row_stride ? p->row_stride : (int64_t)c.cols;
    c.md.resize((size_t)c.rows * c.cols);
    c.maxabs = std::max(std::fabs(p->del), std::fabs(p->ext));
    c.all_int = integral(p->del) && (c.core ? integral(p->ext) : true);
    c.smin = 0; c.smax = 0;
    for (uint32_t r = 0; r < c.rows; ++r)
        for (uint32_t k = 0; k < c.cols; ++k) {
            const double v = p->matrix[(int64_t)r * rs + k];
            c.md[(size_t)r * c.cols + k] = v;
            c.all_int = c.all_int && integral(v);
            c.maxabs = std::max(c.maxabs, std::fabs(v));
            c.smin = std::min(c.smin, v); c.smax = std::max(c.smax, v);
        }
    if (!c.core && !c.all_int) { g_err = "legacy semantics are i32: del and matrix must be integral"; return ALN_ERR_INVALID_ARGUMENT; }
    if (!c.core && p->force_f64) { g_err = "legacy semantics have no f64 form"; return ALN_ERR_UNSUPPORTED; }
    c.outs = p->outputs ? p->outputs : (ALN_OUT_SCORE | ALN_OUT_TRACEBACK);
    c.store_dirs = (c.outs & (ALN_OUT_TRACEBACK | ALN_OUT_DIRECTIONS)) != 0;
    c.want_tb = (c.outs & ALN_OUT_TRACEBACK) != 0;
    c.want_h = want_h;
    uint64_t max_span = 0;
    for (size_t i = 0; i < n; ++i) {
        if (q_len[i] > 0x7FFFFFF0ull || t_len[i] > 0x7FFFFFF0ull) { g_err = "sequence too long"; return ALN_ERR_UNSUPPORTED; }
        const uint64_t N = c.pwm ? c.cols : q_len[i], M = t_len[i];
        if (N && M) max_span = std::max(max_span, N + M + 2);
    }
    // Dyadic schemes: when every penalty and score is a multiple of 2^-k (BLOSUM62 in half-bits, del 11.5 ...), the scheme times 2^k
    // is an integer scheme with the same maxima, the same ties and the same zeros -- in the reference's f64 every H is an exact
    // multiple of 2^-k, two candidates differ by 0 or by >= 2^-k > f64::EPSILON, and H == 0 means the same -- so the integer kernels
    // fill it (three times the f64 kernels' rate) and the two scores of every summary are scaled back, exactly, by one small kernel
    // behind the traceback.  Not with the H output (the dump is what the kernels computed), not when f64 is asked for.
    // ALN_NO_DYADIC=1: off.
    if (c.core && !c.all_int && !p->force_f64 && !want_h && !getenv("ALN_NO_DYADIC")) {
        for (int kk = 1; kk <= 8; ++kk) {
            const double sc = (double)(1 << kk);
            bool ok = integral(p->del * sc) && integral(p->ext * sc);
            for (size_t i = 0; ok && i < c.md.size(); ++i) ok = integral(c.md[i] * sc);
            if (!ok) continue;
            if (c.maxabs * sc * (double)max_span < 1073741824.0) {
                for (double &v : c.md) v *= sc;
                c.p.del *= sc; c.p.ext *= sc;
                c.maxabs *= sc; c.smin *= sc; c.smax *= sc;
                c.all_int = true;
                c.unscale = 1.0 / sc;
            }
            break;
        }
    }
    // integer kernels are exact iff every value is integral and |H| cannot leave i32
    c.is_int = c.all_int && !p->force_f64 && c.maxabs * (double)max_span < 1073741824.0;
    if (!c.core && !c.is_int) { g_err = "legacy scores overflow i32 for these lengths"; return ALN_ERR_UNSUPPORTED; }
    // fast integer kernels: keys are 4*H + tag in i32, the profile holds 4*s - 2 as int8, and S + four waves' profiles
    // (cols x 512 B each) have to fit the 64 KiB of LDS a workgroup gets without an opt-in
    const uint64_t fast_lds = (((uint64_t)c.rows * c.cols * 4 + 15) & ~15ull) + (c.pwm ? 0ull : 4ull * c.cols * 64u * ALN_FULL_R);
    c.fast = c.is_int && !want_h && !p->force_serial && !p->force_generic && fast_lds <= 65536 && c.smin >= -31.0 && c.smax <= 32.0 &&
             c.maxabs * (double)max_span < 268435456.0;
    return ALN_OK;
}

// ---------------------------------------------------------------- per chunk: descriptors, routing, layout
struct Chunk {
    size_t first = 0, n = 0;          // pairs [first, first + n) of the call
    std::vector<PairDesc> descs;
    std::vector<uint32_t> order;      // LPT order of the device work queue (pairs of the batch kernel)
    std::vector<uint32_t> single_pairs, single_r;   // pairs routed to the single-pair (one wave per strip) kernel
    std::vector<uint32_t> wg_pairs, wg_r;           // generic kernels: pairs filled by one workgroup each (aln_fill_wgpipe_kernel)
    size_t n_small = 0;
    uint64_t cells = 0, max_cells = 0, dir_bytes = 0, tb_bytes = 0, tag_bytes = 0, hmat_elems = 0;
    uint32_t max_len = 1, grid = 1, zrow_bytes = 0, lds_bytes = 0, prof_stride = 0, tb_waves = 0, single_max_n = 0, cascade_rows = 1;
    uint32_t duo_qo = 0;               // != 0: two short pairs per wave (aln_fill_duo_kernel): u16 entries of one staged query
    uint64_t scratch_stride = 0, granule_bytes = 0, tbmap_entries = 0, granule_stride_max = 0;
    size_t counter_bytes = 256;
    bool overlap = false;             // walk waves beside the fill (in-kernel overlapped traceback)
    // cooperative passes of the fast batch kernel (CoopRec, aln_device.h): hint ring capacities, bytes of the control block
    bool coop = false, coop_linger = false;
    uint32_t coop_tail = 0;
    uint64_t coop_bytes = 0;
    // sequences: either one contiguous span of the caller's buffer, or gathered pair by pair into pinned staging
    bool seq_direct = true;
    uint64_t seq_lo = 0, seq_span = 0;
    // a plan that is planned again keeps its arrays (6.4 MB of descriptors for 100 000 pairs: allocating and faulting them in afresh
    // was 1 of the 2 ms the plan of a 100 000-window call took)
    void reset()
    {
        std::vector<PairDesc> d = std::move(descs);
        std::vector<uint32_t> o = std::move(order), sp = std::move(single_pairs), sr = std::move(single_r), wp = std::move(wg_pairs), wr = std::move(wg_r);
        *this = Chunk();
        d.clear(); o.clear(); sp.clear(); sr.clear(); wp.clear(); wr.clear();
        descs = std::move(d); order = std::move(o); single_pairs = std::move(sp); single_r = std::move(sr); wg_pairs = std::move(wp); wg_r = std::move(wr);
    }
};

// allow_overlap: the chunk has the device to itself (a single-chunk call, a staged batch).  many_chunks: one of more than four
// chunks of a pipelined call.
static int chunk_plan(const DevCtx *ctx, const Call &c, const uint64_t *q_off, const uint64_t *q_len, const uint64_t *t_off,
                      const uint64_t *t_len, size_t first, size_t n, bool allow_overlap, Chunk &k, bool many_chunks = false)
{
    k.first = first; k.n = n;
    k.descs.assign(n, PairDesc{});
    const bool pwm = c.pwm;
    const uint32_t rows = c.rows, cols = c.cols;
    const char *env_r = getenv("ALN_SINGLE_R");
    const char *env_off = getenv("ALN_NO_SINGLE");
    uint64_t seq_lo = ~0ull, seq_hi = 0, seq_sum = 0;
    for (size_t i = 0; i < n; ++i) {
        const size_t g = first + i;
        PairDesc &d = k.descs[i];
        d.N = pwm ? cols : (uint32_t)q_len[g];                                 // PWM: the columns are the PWM positions
        d.M = (uint32_t)t_len[g];
        d.status = ALN_OK;
        if (d.N == 0 || d.M == 0) d.status = (pwm && d.N != 0) ? ALN_PRE_EMPTY_OK : ALN_ERR_EMPTY_SEQUENCE;   // the reference panics
        if (!pwm && q_len[g]) { seq_lo = std::min(seq_lo, q_off[g]); seq_hi = std::max(seq_hi, q_off[g] + q_len[g]); seq_sum += q_len[g]; }
        if (t_len[g]) { seq_lo = std::min(seq_lo, t_off[g]); seq_hi = std::max(seq_hi, t_off[g] + t_len[g]); seq_sum += t_len[g]; }
    }
    if (seq_lo == ~0ull) { seq_lo = 0; seq_hi = 0; }
    // the chunk's residues are one span of the caller's buffer unless the offsets are scattered far beyond what the chunk uses
    k.seq_direct = (seq_hi - seq_lo) <= 2 * seq_sum + 65536;
    k.seq_lo = seq_lo;
    k.seq_span = k.seq_direct ? seq_hi - seq_lo : seq_sum;
    uint64_t gpos = 0;
    for (size_t i = 0; i < n; ++i) {
        const size_t g = first + i;
        PairDesc &d = k.descs[i];
        if (k.seq_direct) { d.q_off = pwm ? 0 : q_off[g] - (q_len[g] ? seq_lo : q_off[g]); d.t_off = t_len[g] ? t_off[g] - seq_lo : 0; }
        else { d.q_off = gpos; gpos += pwm ? 0 : q_len[g]; d.t_off = gpos; gpos += t_len[g]; }
    }

    // ---- routing + HBM layout.  A pair goes to the single-pair kernel (one wave per strip, strips pipelined across
    // CUs) when it is large, or when the chunk is too small to fill the chip with one wave per pair -- unless the chunk holds
    // so many such pairs that the batch kernel, whose waves share the strips of a pair (cooperative passes), is through with all
    // of them sooner than the single-pair route, which takes them one after the other.  Estimates (measured rates): the
    // single-pair route fills at ~60 GCUPS plus ~0.2 ms of launches and traceback per pair; in the batch kernel a wave fills at
    // ~0.85 GCUPS with the chip full (2.6 TCUPS at most), and the pipeline of a pair's strips takes (N + 63 + 170 (strips - 1))
    // steps of ~0.55 us.
    const bool coop_on = c.fast && !getenv("ALN_NO_COOP");
    bool big_to_single = true;
    if (coop_on && !pwm && !env_off) {
        double t_single = 0, cells_all = 0, cells_small = 0, strips_all = 0, strips_small = 0, lat_all = 0, lat_small = 0;
        size_t n_big = 0;
        for (size_t i = 0; i < n; ++i) {
            const PairDesc &d = k.descs[i];
            if (d.status != ALN_OK) continue;
            const double pc = (double)d.N * d.M, st = (double)aln_num_strips(d.M);
            const double lat = ((double)d.N + 63.0 + 170.0 * (st - 1.0)) * 0.55e-6;
            const bool big = d.N >= 64 && d.M >= 128 && (pc >= (double)(1ull << 24) || (n <= 16 && pc >= (double)(1ull << 18)));
            cells_all += pc; strips_all += st; lat_all = std::max(lat_all, lat);
            if (big) { t_single += pc / 60e9 + 0.2e-3; ++n_big; }
            else { cells_small += pc; strips_small += st; lat_small = std::max(lat_small, lat); }
        }
        const double waves = (double)ctx->cus * 12.0;
        auto t_batch = [&](double cells, double strips, double lat) {
            if (cells <= 0) return 0.0;
            return std::max(std::max(cells / 2.6e12, lat), cells / (std::min(strips, waves) * 0.85e9));
        };
        if (n_big) big_to_single = t_single + t_batch(cells_small, strips_small, lat_small) <= t_batch(cells_all, strips_all, lat_all);
        if (const char *e = getenv("ALN_BIG_TO_SINGLE")) big_to_single = atoi(e) != 0;
    }
    uint64_t dir_total = 0, tb_total = 0, tag_total = 0, hm_total = 0, cells = 0;
    uint32_t max_len = 1;
    for (size_t i = 0; i < n; ++i) {
        PairDesc &d = k.descs[i];
        // strings: cumulative 2 * cap per pair (PWM: 5 * cap, 4-aligned) -- the layout aln_align_batch documents for tb_buf
        const uint64_t cap = (uint64_t)d.N + d.M + 2;
        if (c.store_dirs) {
            d.tb_off = tb_total;
            tb_total += pwm ? ((5ull * cap + 3) & ~3ull) : 2ull * cap;
            d.tag_off = tag_total;
            tag_total += (cap + 3) & ~3ull;
        }
        if (d.status != ALN_OK) continue;
        const uint64_t pc = (uint64_t)d.N * d.M;
        cells += pc;
        max_len = std::max(max_len, std::max(d.N, d.M));
        bool single = c.fast && !pwm && !env_off && d.N >= 64 && d.M >= 128 && (pc >= (1ull << 24) || (n <= 16 && pc >= (1ull << 18))) &&
                      (big_to_single || aln_num_strips(d.M) > ALN_COOP_MAX_NS);
        uint64_t dbytes = aln_dir_bytes(d.N, d.M);
        if (single) {
            // rows per lane: two above ~2500 rows (measured, fill + traceback: 3000 x 3000 0.59 ms against 0.62 with one, 4000 x 4000
            // 0.72 against 0.76; below, the traceback's longer strips cost more than the fill gains)
            uint32_t R = env_r ? (uint32_t)atoi(env_r) : (d.M > 2560 ? 2u : 1u);
            if (R != 1 && R != 2 && R != 4 && R != 8) R = 2;
            while ((d.M + 64 * R - 1) / (64 * R) > 4096 && R < 8) R *= 2;      // keep every strip's wave resident
            if ((d.M + 64 * R - 1) / (64 * R) > 4096) single = false;
            // LDS: the whole query's profile offsets are staged (2 B per column) beside S and the waves' profiles; a workgroup may
            // opt in to all 160 KiB of a CU (one wave per workgroup beyond ~43 000 columns, four below: aln_single_waves), which
            // carries the route to ~77 000 columns.  Longer pairs take the batch kernel (one wave, slow but exact).
            if (d.N > 100000u || aln_single_lds_bytes(rows, cols, R, d.N, 1) > 159u * 1024u) single = false;
            if (single) {
                const uint32_t ns = (d.M + 64 * R - 1) / (64 * R);
                dbytes = std::max<uint64_t>(dbytes, (uint64_t)ns * aln_uniform_strip_bytes(d.N, R));
                k.single_pairs.push_back((uint32_t)i);
                k.single_r.push_back(R);
                const uint64_t gstride = ((uint64_t)d.N + 64 + 63) & ~63ull;
                k.granule_stride_max = std::max(k.granule_stride_max, gstride);
                k.granule_bytes = std::max<uint64_t>(k.granule_bytes, std::max<uint64_t>((uint64_t)ns * gstride * 4, 4ull * (d.M + 2)));
                k.single_max_n = std::max(k.single_max_n, std::max(d.N, ns));
                k.tbmap_entries = std::max<uint64_t>(k.tbmap_entries, (uint64_t)ns * (d.N + 1) + ns + 64);
            }
        }
        // Generic kernels (real-valued matrix, or an integer one outside the fast path's limits): a chunk of at most four pairs (their launches run one
        // after the other; a larger chunk is better off with one wave per pair, all at once) gives each pair a whole workgroup, one wave per 64 R-row strip (<= 16 strips), instead of one wave -- the call HeuristicAligner makes
        // once per iteration (heuristic/mod.rs:58-77).  Also with the H dump (AlignmentResult.alignment_matrix) and for PWM scoring (HeuristicPWMAligner).
        if (!single && (!c.fast || c.want_h) && !c.p.force_serial && !getenv("ALN_NO_WGPIPE") && n <= 4 && pc >= (1ull << 14) &&
            d.N >= 16 && d.N <= 8192 && d.M >= 65 && d.M <= 2048) {
            // rows per lane: about eight strips = two waves per SIMD of the one CU (measured, 1000 x 1000 f64: R = 1 1.42 ms,
            // R = 2 1.26, R = 4 1.28; 330 x 300: 0.43 / 0.43 / 0.50)
            uint32_t R = d.M > 1024 ? 4u : d.M > 512 ? 2u : 1u;
            if (const char *e = getenv("ALN_WG_R")) { const uint32_t v = (uint32_t)atoi(e); if ((v == 1 || v == 2 || v == 4) && (d.M + 64 * v - 1) / (64 * v) <= 16) R = v; }
            const uint32_t ns = (d.M + 64 * R - 1) / (64 * R);
            if (aln_wg_lds_bytes(c.rows, c.cols, c.is_int ? 4u : 8u, ns, d.N) <= 64u * 1024u) {
                dbytes = std::max<uint64_t>(dbytes, (uint64_t)ns * aln_uniform_strip_bytes(d.N, R));
                k.wg_pairs.push_back((uint32_t)i);
                k.wg_r.push_back(R);
                k.tbmap_entries = std::max<uint64_t>(k.tbmap_entries, (uint64_t)ns * (d.N + 1) + ns + 64);
            }
        }
        d.dir_off = dir_total;
        if (c.store_dirs) dir_total += dbytes;
        if (c.want_h) { d.h_off = hm_total; hm_total += (uint64_t)(d.N + 1) * (d.M + 1); }
    }
    k.cells = cells; k.max_len = max_len;
    k.dir_bytes = dir_total; k.tb_bytes = tb_total; k.tag_bytes = tag_total; k.hmat_elems = hm_total;

    // ---- LPT order: longest pairs first into the device work queue.  "Longest" = the time one wave needs, not the cells: a strip of
    // R rows per lane takes N + lanes - 1 steps of about 12 + 10.5 R instructions, so a 1900 x 270 pair (one strip of 5 rows per
    // lane) keeps its wave as long as a 1000 x 1000 pair with twice the cells -- ordered by cells it was taken when the queue was
    // almost dry and ended the 8-way shard's fill 0.9 ms after everybody else (tools/tail_timeline.py).
    auto pair_cost = [](const PairDesc &d) -> uint64_t {
        if (d.status != ALN_OK) return 0;
        uint64_t cost = 0;
        const uint32_t ns = aln_num_strips(d.M);
        for (uint32_t st = 0; st < ns; ++st) {
            const uint32_t rows = std::min<uint32_t>(d.M - st * ALN_STRIP_ROWS, ALN_STRIP_ROWS);
            const uint32_t R = st + 1 == ns ? (uint32_t)aln_pick_r(rows) : (uint32_t)ALN_FULL_R, L = (rows + R - 1) / R;
            cost += (uint64_t)(d.N + L - 1) * (24u + 21u * R);
            if (cost >= (1ull << 40)) return (1ull << 40) - 1;
        }
        return cost;
    };
    k.order.clear();
    k.order.reserve(n);
    {
        std::vector<char> is_single(n, 0);
        for (uint32_t i : k.single_pairs) is_single[i] = 1;
        for (uint32_t i : k.wg_pairs) is_single[i] = 1;
        for (size_t i = 0; i < n; ++i) if (!is_single[i]) k.order.push_back((uint32_t)i);
    }
    k.n_small = k.order.size();
    {
        // sort keys packed in one u64: cells descending, then index ascending (a stable order without a stable_sort)
        std::vector<uint64_t> key(k.n_small);
        bool packable = n < (1u << 24);
        for (size_t j = 0; j < k.n_small && packable; ++j) {
            const uint64_t pc = pair_cost(k.descs[k.order[j]]);
            key[j] = ((~pc & ((1ull << 40) - 1)) << 24) | k.order[j];
        }
        if (packable) {
            // (equal pairs -- PWM windows, read pairs -- arrive sorted: the test costs one pass, the sort was 1 of the 1.7 ms the
            // plan of 100 000 windows took)
            if (!std::is_sorted(key.begin(), key.end())) {
                // LSD radix sort on the bits of the cost that vary, 11 at a time (stable: equal costs keep the index order the keys
                // were built in); std::sort was 0.35 of the 0.7 ms the plan of 12 500 pairs took
                uint64_t lo = ~0ull, hi = 0;
                for (uint64_t v : key) { lo = std::min(lo, v >> 24); hi = std::max(hi, v >> 24); }
                const uint64_t range = hi - lo;                      // sort by (v >> 24) - lo
                std::vector<uint64_t> tmp(key.size());
                uint64_t *src = key.data(), *dst = tmp.data();
                for (uint32_t shift = 0; shift < 40 && (range >> shift) != 0; shift += 11) {
                    uint32_t cnt[2049] = {0};
                    for (size_t j = 0; j < key.size(); ++j) ++cnt[((((src[j] >> 24) - lo) >> shift) & 2047u) + 1u];
                    for (uint32_t d = 0; d < 2048; ++d) cnt[d + 1] += cnt[d];
                    for (size_t j = 0; j < key.size(); ++j) dst[cnt[(((src[j] >> 24) - lo) >> shift) & 2047u]++] = src[j];
                    std::swap(src, dst);
                }
                if (src != key.data()) memcpy(key.data(), src, key.size() * sizeof(uint64_t));
            }
            for (size_t j = 0; j < k.n_small; ++j) k.order[j] = (uint32_t)(key[j] & 0xffffffu);
        } else {
            std::stable_sort(k.order.begin(), k.order.end(), [&](uint32_t a, uint32_t b) { return pair_cost(k.descs[a]) > pair_cost(k.descs[b]); });
        }
    }
    k.max_cells = 0;
    for (uint32_t i : k.order) if (k.descs[i].status == ALN_OK) k.max_cells = std::max(k.max_cells, (uint64_t)k.descs[i].N * k.descs[i].M);

    // ---- grid: persistent waves, 4 per workgroup
    const uint32_t wg_needed = (uint32_t)((k.n_small + 3) / 4);
    k.grid = std::max(1u, std::min(wg_needed, (uint32_t)ctx->cus * 4u));
    // Overlapped traceback: the walk kernel runs beside the fill.  The fast fill kernel is built for 160 VGPRs,
    // three workgroups per CU, so that every SIMD keeps 32 registers free: exactly one wave of the walk kernel (32 VGPRs, no
    // LDS).  ALN_TB_OVERLAP: unset = one walk wave per SIMD beside a full fill grid; 0 = off; n > 0 = the earlier scheme (the
    // fill grid stops n workgroups short of residency and the walk waves crowd onto those CUs, 20 per slot).
    {
        const uint32_t resident = (uint32_t)ctx->cus * 3u;
        const char *e = getenv("ALN_TB_OVERLAP");
        const uint32_t reserve = e ? (uint32_t)atoi(e) : 0u;
        const bool off = e && reserve == 0;
        // (only where the fill is long enough to hide anything behind: the walk waves, their 30 us head start and the
        // write-through stores cost a batch of 10 000 read pairs -- 0.3 ms of fill -- 40 % of its time)
        k.overlap = allow_overlap && !off && c.fast && c.is_int && c.want_tb && c.store_dirs && reserve < resident &&
                    k.n_small >= 4096 && wg_needed >= resident && (k.cells >= 2000000000ull || getenv("ALN_TB_OVERLAP_ANY"));
        k.counter_bytes = 256;
        if (k.overlap) {
            k.grid = resident - reserve;
            k.tb_waves = reserve ? reserve * 20u : (uint32_t)ctx->cus * 4u;
            k.counter_bytes = 256 + 4ull * k.n_small;
        }
    }
    // Cooperative passes: waves without a pair of their own take strips of other waves' pairs, so a batch with fewer pairs than
    // resident waves still gets as many waves as it has strips
    // (nothing to share in a batch whose pairs all have one strip -- read pairs, PWM windows, the p-value batch: the kernel then runs
    // exactly as without the machinery)
    uint64_t strips = 0, multi = 0;
    for (size_t j = 0; j < k.n_small; ++j) {
        const PairDesc &d = k.descs[k.order[j]];
        // (a pair has something to share when its first pass has several strips, or when it may be re-filled -- core local with
        // del != ext -- and is large enough for that re-fill to matter: a 330 x 300 window is through in 40 us either way)
        if (d.status == ALN_OK) {
            strips += aln_num_strips(d.M);
            multi += (d.M > ALN_STRIP_ROWS || (c.semantics == ALN_CORE_LOCAL && c.p.del != c.p.ext && d.M > 64u && (uint64_t)d.N * d.M >= (1u << 18))) ? 1u : 0u;
        }
    }
    // (one of many chunks of a pipelined call: its tail hides behind the chunks after it, and sharing re-fills only cost -- measured on
    // C5 through aln_align_batch, eight chunks: 51.2 ms without, 52.0 with; three chunks of the 12 500-pair shard: 10.3 without, 9.3 with)
    k.coop = coop_on && k.n_small != 0 && multi != 0 && !many_chunks;
    if (k.coop) {
        const uint32_t resident = (uint32_t)ctx->cus * 3u;
        // (allow_overlap == false: a chunk of a pipelined call -- the chunks before and after it share the chip with this one, so its
        // grid stays at one wave per pair, nobody lingers, and only re-fills are shared)
        const bool alone = allow_overlap;
        if (!k.overlap && alone) k.grid = std::max(k.grid, (uint32_t)std::min<uint64_t>(resident, (strips + 3) / 4));
        // First passes give strips away only in batches of fewer than two pairs per resident wave -- there the pairs that are still
        // running when the queue is dry are large, and idle waves are what fills the chip.  In a larger batch the last pairs are the
        // shortest ones (measured on the 12 500-pair shard of C5: opening the first passes of its last 6000 pairs cost 0.15 ms of 6.8
        // and shortened nothing); re-fills are always shared.  ALN_COOP_TAIL overrides (pairs from the end whose first pass is opened).
        uint64_t tail = (alone && k.n_small < 2ull * resident * 4u) ? k.n_small : 0;
        if (const char *e = getenv("ALN_COOP_TAIL")) tail = strtoull(e, nullptr, 10);
        k.coop_tail = (uint32_t)(k.n_small > tail ? k.n_small - tail : 0);
        // waves that find the queue dry stay around for strips only when they are what fills the chip: a batch with fewer pairs than
        // resident waves (measured on the 12 500-pair shard: staying costs the waves that still work 0.2 ms of 7)
        k.coop_linger = alone && k.n_small < (uint64_t)resident * 4u;
        // Chains of strips (every first pass open, two or more strips per pair on average, no more pairs than resident waves) run
        // with TWO waves per SIMD: a chain is as fast as its slowest strip, and the third wave of a SIMD is the one the arbiter
        // leaves out (256 / 512 / 1024 pairs of 4200 x 4200, score only: 4.97 / 8.99 / 11.8 -> 4.47 / 7.82 / 10.6 ms; 3072 pairs
        // 23.9 -> 22.8; 5000 pairs -- a wave per pair -- 32.8 -> 36.1: not there).  ALN_CHAIN_WGS=0: off.
        if (alone && !k.overlap && tail >= k.n_small && k.n_small <= (uint64_t)resident * 4u && strips >= 2 * k.n_small && !(getenv("ALN_CHAIN_WGS") && atoi(getenv("ALN_CHAIN_WGS")) == 0))
            k.grid = std::min(k.grid, (uint32_t)ctx->cus * 2u);
    }
    if (const char *e = getenv("ALN_FILL_WGS")) k.grid = std::max(1u, std::min(k.grid, (uint32_t)atoi(e)));   // experiments: fewer resident fill waves
    if (k.coop) k.coop_bytes = 4ull * (ALN_COOP_CTL_WORDS + (((uint64_t)k.grid * 4 + 63) & ~63ull)) + (uint64_t)k.grid * 4 * sizeof(CoopRec);
    const uint64_t sc_size = (c.is_int && !c.fast) ? 4 : 8;          // fast kernels: 8-byte granules {T value, tag}
    const uint64_t brow_bytes = (((uint64_t)max_len + 66) * sc_size + 63) & ~63ull;
    const uint64_t adv_bytes = ((uint64_t)max_len + 66 + 63) & ~63ull;
    // fast path, core local with del != ext: strip 0's bottom row keeps a row of its own and strip 0 checkpoints its lane state
    // (18 ints x 64 lanes per checkpoint): the localized repair of the row-1 hazard, do_pair_fast
    // (fast kernels: a strip never writes the row it reads -- two rows; hazard pairs: strip 0's bottom row keeps one more to itself)
    k.cascade_rows = (c.fast && c.semantics == ALN_CORE_LOCAL && c.p.del != c.p.ext) ? ALN_CASCADE_ROWS : (c.fast ? 2u : 1u);
    const uint64_t ck_bytes = c.fast ? (uint64_t)ALN_CK_SLOTS * 18 * 64 * 4 : 0;
    // bottom-row record: one byte per column (generic kernels) or one direction dword per block of the last strip (fast
    // path: at most (max_len + 63) / 2 + 4 blocks)
    const uint64_t zrow_bytes = std::max<uint64_t>(adv_bytes, ((c.fast ? 8ull : 4ull) * (((uint64_t)max_len + 63) / 2 + 8) + 63) & ~63ull);
    k.zrow_bytes = (uint32_t)zrow_bytes;
    // generic kernels: row 1 as the pass computed it (values + direction tags), for adopt_advice_checked
    const uint64_t row1_bytes = c.fast ? 0 : brow_bytes + adv_bytes;
    k.scratch_stride = (uint64_t)k.cascade_rows * brow_bytes + adv_bytes + zrow_bytes + ck_bytes + row1_bytes;
    k.lds_bytes = (uint32_t)(((uint64_t)rows * cols * (c.is_int ? 4 : 8) + 15) & ~15ull);
    k.prof_stride = 0;
    if (c.fast && !pwm) { k.prof_stride = cols * 64u * ALN_FULL_R; k.lds_bytes += 4u * k.prof_stride; }
    // Two short pairs per wave (core global, read pairs: aln_fill_duo_kernel): every pair of the queue at most 256 rows and 1024
    // columns, more pairs than resident waves (with fewer, a wave per pair is through sooner), nothing shared, no walk waves beside
    // the fill, and the staged queries fit beside the profiles with three workgroups per CU.  ALN_NO_DUO=1: off.
    k.duo_qo = 0;
    if (c.fast && !pwm && c.semantics == ALN_CORE_GLOBAL && !k.coop && !k.overlap && !getenv("ALN_NO_DUO") &&
        k.n_small > (uint64_t)ctx->cus * 12u) {
        uint32_t max_rows = 0, max_cols = 0;
        for (size_t j = 0; j < k.n_small; ++j) { const PairDesc &d = k.descs[k.order[j]]; max_rows = std::max(max_rows, d.M); max_cols = std::max(max_cols, d.N); }
        const uint32_t qo = (max_cols + 136u + 7u) & ~7u;
        // (a wave's profile: R = ceil(rows / 32) bytes per lane and code, rounded up to 1 / 2 / 4 / 8 -- 20-letter alphabets fit with
        // up to 128 rows)
        const uint32_t rmax = (max_rows + 31u) / 32u, rp = rmax > 4u ? 8u : rmax > 2u ? 4u : std::max(rmax, 1u);
        const uint32_t prof = cols * 64u * rp;
        const uint32_t lds = (uint32_t)(((uint64_t)rows * cols * 4 + 15) & ~15ull) + 4u * (prof + 4u * qo);
        if (max_rows <= 256u && max_cols <= 1024u && lds <= 52u * 1024u) {
            k.duo_qo = qo;
            k.prof_stride = prof;
            k.lds_bytes = lds;
            const uint32_t items = (uint32_t)((k.n_small + 1) / 2);
            k.grid = std::max(1u, std::min((items + 3u) / 4u, (uint32_t)ctx->cus * 3u));
        }
    }
    return ALN_OK;
}

// upper bounds over the chunks of a pipelined call: a slot is sized for the largest chunk the first time it is touched, so no
// buffer grows (hipFree + hipMalloc stall every stream) in the middle of the pipeline
struct Need {
    uint64_t seq_span = 0, n = 0, dir_bytes = 0, tb_bytes = 0, tag_bytes = 0, scratch = 0, coop = 0;
};

// device buffers of a slot for this chunk (grow-only; nothing happens once the pool is warm)
static int slot_ensure(Slot &s, const Call &c, const Chunk &k, const Need *need = nullptr)
{
    int st;
    const bool sl = s.pooled;
#define ENS(buf, bytes) if ((st = dev_ensure(s.buf, (bytes), sl)) != ALN_OK) return st
    if (need) {
        ENS(seqs, need->seq_span + 64);
        ENS(descs, need->n * sizeof(PairDesc));
        ENS(order, need->n * sizeof(uint32_t));
        ENS(results, need->n * sizeof(aln_pair_result));
        ENS(dirs, need->dir_bytes);
        ENS(tb, need->tb_bytes);
        ENS(tags, need->tag_bytes);
        ENS(scratch, need->scratch);
        if (need->coop) ENS(coop, need->coop);
    }
    ENS(seqs, k.seq_span + 64);
    ENS(descs, k.n * sizeof(PairDesc));
    ENS(order, k.n * sizeof(uint32_t));
    ENS(counter, k.counter_bytes);
    if (k.coop) ENS(coop, k.coop_bytes);
    if (k.overlap) {
        const size_t before = s.walked.cap;
        ENS(walked, 4ull * k.n);
        if (s.walked.cap != before) s.walked_clean = false;
        if (!s.tb_stream) HIPCHK(hipStreamCreateWithFlags(&s.tb_stream, hipStreamNonBlocking));
        if (!s.ev_fork) HIPCHK(hipEventCreateWithFlags(&s.ev_fork, hipEventDisableTiming));
        if (!s.ev_join) HIPCHK(hipEventCreateWithFlags(&s.ev_join, hipEventDisableTiming));
    }
    ENS(dirs, k.dir_bytes);
    ENS(results, k.n * sizeof(aln_pair_result));
    ENS(tb, k.tb_bytes);
    ENS(tags, k.tag_bytes);
    {
        const size_t before = s.scratch.cap;
        const void *before_p = s.scratch.p;
        ENS(scratch, std::max<uint64_t>((uint64_t)k.grid * 4 * k.scratch_stride, k.wg_pairs.empty() ? 0 : ((uint64_t)k.max_len + 66) * 8));
        if (s.scratch.cap != before || s.scratch.p != before_p) s.scratch_clean = false;
    }
    if (!k.wg_pairs.empty()) ENS(tbmap, k.tbmap_entries * 16);
    ENS(matrix, (uint64_t)c.rows * c.cols * (c.is_int ? 4 : 8));
    if (c.pwm && c.fast) ENS(pwm_words, (uint64_t)c.cols * 4);
    if (c.want_h) ENS(hmat, k.hmat_elems * (c.is_int ? 4 : 8));
    if (!k.single_pairs.empty()) {
        ENS(granules, k.granule_bytes);
        ENS(advice1, 2ull * single_advice_bytes(k.single_max_n));
        ENS(cand, 16ull * (k.single_max_n + 64));
        ENS(ctrl, 256);
        ENS(tbmap, k.tbmap_entries * 16);
        // localized repair of the single-pair route: up to 8 strips' scratch granule rows, checkpoints, candidates
        ENS(repair, 8ull * (k.granule_stride_max * 4 + 18 * 64 * 4 + 32) + 256);
    }
#undef ENS
    // pinned staging of the small tables: descs | order | matrix | pwm words
    const size_t meta = std::max<size_t>(k.n, need ? need->n : 0) * (sizeof(PairDesc) + 4) + (size_t)c.rows * c.cols * 8 + (size_t)c.cols * 4 + 256;
    if ((st = pin_ensure(s.h_meta, meta)) != ALN_OK) return st;
    if (!k.seq_direct && (st = pin_ensure(s.h_in, k.seq_span + 64)) != ALN_OK) return st;
    return slot_init(s);
}

// H2D of one chunk on the slot's stream.  The residues come straight out of the caller's buffer (one span); the small tables
// go through pinned memory.  Returns when the copies are queued (the span copy of pageable memory is staged by the runtime).
static int slot_upload(Slot &s, const Call &c, const Chunk &k, const uint8_t *seqs, const uint64_t *q_off, const uint64_t *q_len,
                       const uint64_t *t_off, const uint64_t *t_len, hipStream_t st, bool staged = false, bool seqs_there = false)
{
    uint8_t *m = s.h_meta.as<uint8_t>();
    size_t o = 0;
    if (k.n) {
        memcpy(m + o, k.descs.data(), k.n * sizeof(PairDesc));
        HIPCHK(hipMemcpyAsync(s.descs.p, m + o, k.n * sizeof(PairDesc), hipMemcpyHostToDevice, st));
        o += k.n * sizeof(PairDesc);
        if (!k.order.empty()) {
            memcpy(m + o, k.order.data(), k.order.size() * 4);
            HIPCHK(hipMemcpyAsync(s.order.p, m + o, k.order.size() * 4, hipMemcpyHostToDevice, st));
        }
        o += k.n * 4;
    }
    o = (o + 15) & ~(size_t)15;
    const size_t nm = c.md.size();
    if (c.is_int) {
        int32_t *mi = reinterpret_cast<int32_t *>(m + o);
        for (size_t i = 0; i < nm; ++i) mi[i] = (int32_t)c.md[i];
        HIPCHK(hipMemcpyAsync(s.matrix.p, mi, nm * 4, hipMemcpyHostToDevice, st));
    } else {
        memcpy(m + o, c.md.data(), nm * 8);
        HIPCHK(hipMemcpyAsync(s.matrix.p, m + o, nm * 8, hipMemcpyHostToDevice, st));
    }
    o += nm * 8;
    if (c.pwm && c.fast) {
        uint32_t *words = reinterpret_cast<uint32_t *>(m + o);
        for (uint32_t x = 0; x < c.cols; ++x) {
            uint32_t wv = 0;
            for (uint32_t r = 0; r < 4; ++r) wv |= ((uint32_t)(int32_t)(4 * (int32_t)c.md[(size_t)r * c.cols + x] - 2) & 0xffu) << (8 * r);
            words[x] = wv;
        }
        HIPCHK(hipMemcpyAsync(s.pwm_words.p, words, (size_t)c.cols * 4, hipMemcpyHostToDevice, st));
    }
    if (k.seq_span && !seqs_there) {
        if (k.seq_direct) {
            HIPCHK(hipMemcpyAsync(s.seqs.p, seqs + k.seq_lo, k.seq_span, hipMemcpyHostToDevice, st));
        } else {
            uint8_t *h = s.h_in.as<uint8_t>();
            for (size_t i = 0; i < k.n && !staged; ++i) {     // staged: the caller has filled h_in already (aln_align_pair)
                const size_t g = k.first + i;
                if (!c.pwm && q_len[g]) memcpy(h + k.descs[i].q_off, seqs + q_off[g], q_len[g]);
                if (t_len[g]) memcpy(h + k.descs[i].t_off, seqs + t_off[g], t_len[g]);
            }
            HIPCHK(hipMemcpyAsync(s.seqs.p, h, k.seq_span, hipMemcpyHostToDevice, st));
        }
    }
    return ALN_OK;
}

// All kernels of one chunk, asynchronous on `st`: validation of the residue codes, fill (+ exact re-fills), traceback.
// ev (optional): three timing events (fill start, fill end, traceback end).
// fill_after (optional): an event the fill has to wait for (pipelined calls: the fill of the chunk two before, see below).
static int slot_launch(DevCtx *ctx, Slot &s, const Call &c, const Chunk &k, hipStream_t st, hipEvent_t *ev, uint32_t *fill_launches,
                       hipEvent_t fill_after = nullptr)
{
    if (fill_launches) *fill_launches = 0;
    if (k.n == 0) return ALN_OK;
    if (k.n_small) HIPCHK(hipMemsetAsync(s.counter.p, 0, k.counter_bytes, st));     // the batch kernel's work queue
    if (k.n_small && k.coop) HIPCHK(hipMemsetAsync(s.coop.p, 0, k.coop_bytes, st)); // counters, claim words (zero = nothing to claim), records
    if (k.n_small && c.fast) {
        s.salt = (s.salt + 1u) & 0x3ffu;
        if (!s.scratch_clean || s.salt == 0u) { HIPCHK(hipMemsetAsync(s.scratch.p, 0, s.scratch.cap, st)); s.scratch_clean = true; }
    }
    // residue codes outside the matrix: the batch fill kernels check the pairs they take; the single-pair route reads the status
    // from the descriptor, so its pairs are checked by a kernel of their own in front
    if (!k.single_pairs.empty())
        aln_launch_validate(s.seqs.as<uint8_t>(), s.descs.as<PairDesc>(), (uint32_t)k.n, c.rows, c.cols, c.pwm ? 1 : 0, st);
    FillArgs fa{};
    fa.seqs = s.seqs.as<uint8_t>(); fa.descs = s.descs.as<PairDesc>(); fa.order = s.order.as<uint32_t>(); fa.n_pairs = (uint32_t)k.n_small;
    fa.counter = s.counter.as<uint32_t>(); fa.dirs = s.dirs.as<uint8_t>(); fa.results = s.results.as<aln_pair_result>();
    fa.scratch = s.scratch.as<uint8_t>(); fa.scratch_stride = k.scratch_stride; fa.max_len = k.max_len; fa.zrow_bytes = k.zrow_bytes;
    fa.cascade_rows = k.cascade_rows;
    fa.matrix = s.matrix.p; fa.rows = c.rows; fa.cols = c.cols; fa.prof_stride = k.prof_stride;
    fa.del = c.p.del; fa.ext = c.p.ext; fa.semantics = c.semantics;
    fa.max_passes = c.p.max_passes; fa.force_serial = c.p.force_serial;
    fa.no_repair = getenv("ALN_NO_REPAIR") ? 1u : 0u;
    fa.f64_old = getenv("ALN_F64_OLD") ? 1u : 0u;
    fa.max_cells = k.max_cells;
    fa.store_dirs = c.store_dirs ? 1u : 0u;
    fa.pwm = c.pwm ? 1u : 0u;
    fa.pwm_words = s.pwm_words.as<uint32_t>();
    fa.hmat = c.want_h ? s.hmat.p : nullptr; fa.blank = c.p.blank_code;
    fa.n_descs = (uint32_t)k.n;
    // ALN_BACK_WAVES=1 (experiment, off): the youngest wave of every SIMD takes its pairs from the back of the queue (next_pair2).
    // Measured on the 8-way shard of C5: fill 6.47-6.61 ms with it, 6.25 without -- the short pairs it moves to the slow waves are
    // what filled the gaps at the end; r02 had seen the same with a queue in two segments.
    { const char *e = getenv("ALN_BACK_WAVES"); fa.back_waves = (e && atoi(e) && k.grid == (uint32_t)ctx->cus * 3u && k.n_small >= 2ull * k.grid * 4u) ? 1u : 0u; }
    fa.coop = k.coop ? s.coop.as<uint32_t>() : nullptr; fa.coop_waves = k.grid * 4u; fa.coop_tail = k.coop_tail; fa.salt = s.salt;
    { const char *e = getenv("ALN_COOP_LINGER"); fa.coop_linger = e ? (uint32_t)atoi(e) : (k.coop_linger ? 1u : 0u); }
    { const char *e = getenv("ALN_COOP_DEBUG"); fa.coop_debug = e ? (uint32_t)atoi(e) : 0u; }
    { const char *e = getenv("ALN_FAIR"); fa.fair = e ? (uint32_t)atoi(e) : 0u; }
    { const char *e = getenv("ALN_CK_LAST"); fa.ck_last = (e && atoi(e) == 512) ? 512u : ALN_CK_LAST; }
    fa.duo_qo = k.duo_qo;
    // runs of queue positions per atomic: only where pairs are many, short and alike (one strip, <= 2^18 cells), nothing is shared
    // and the queue is not two-ended; about 1.6 runs per wave or more, so that the last round stays as even as with single pairs
    fa.claim = 1;
    if (c.fast && !k.coop && !fa.back_waves && k.max_cells <= (1ull << 18) && k.n_small != 0) {
        // (queue units per resident wave; two pairs per wave: a unit is two pairs)
        const double per_wave = (double)(k.duo_qo ? (k.n_small + 1) / 2 : k.n_small) / ((double)std::min(k.grid, (uint32_t)ctx->cus * 3u) * 4.0);
        // (measured: C3, 3.3 pairs per wave: runs of 2 fill 0.232 -> 0.204 ms, runs of 3 / 4 0.218 / 0.222 -- the last round gets uneven;
        // 100 000 PWM windows, 32 per wave: runs of 4 cost 4 % -- with many pairs per wave the waves drift apart by themselves)
        fa.claim = (per_wave >= 3.2 && per_wave < 12.0) ? 2u : 1u;
    }
    if (const char *e = getenv("ALN_CLAIM")) fa.claim = (uint32_t)std::max(1, std::min(8, atoi(e)));
    if (fill_after) HIPCHK(hipStreamWaitEvent(st, fill_after, 0));
    if (ev) HIPCHK(hipEventRecord(ev[0], st));
    uint32_t launches = 0;
    TraceArgs ta{};
    ta.seqs = s.seqs.as<uint8_t>(); ta.descs = s.descs.as<PairDesc>(); ta.n_pairs = (uint32_t)k.n; ta.dirs = s.dirs.as<uint8_t>();
    ta.results = s.results.as<aln_pair_result>(); ta.tb = s.tb.as<uint8_t>(); ta.tags = s.tags.as<uint8_t>();
    ta.semantics = c.semantics; ta.blank = c.p.blank_code; ta.pwm = c.pwm ? 1 : 0;
    const bool overlap = k.overlap;
    if (overlap) {
        if (!s.walked_clean) { HIPCHK(hipMemsetAsync(s.walked.p, 0, s.walked.cap, st)); s.walked_clean = true; s.epoch = 0; }
        fa.doneq = s.counter.as<uint32_t>() + 64;
        ta.walked = s.walked.as<uint32_t>(); ta.epoch = ++s.epoch; ta.n_order = (uint32_t)k.n_small;
        ta.doneq = s.counter.as<uint32_t>() + 64; ta.head = s.counter.as<uint32_t>() + 2; ta.wait_ticks = 50000000ull;   // 0.5 s
        if (const char *w = getenv("ALN_TB_WAIT_US")) ta.wait_ticks = 100ull * strtoull(w, nullptr, 10);   // testing: 0 = give up at once
        HIPCHK(hipEventRecord(s.ev_fork, st));                                // after the memsets, before the fill
    }
    if (k.n_small) {
        aln_launch_fill(&fa, c.is_int ? 1 : 0, c.fast ? 1 : 0, k.grid, k.lds_bytes, st);
        HIPCHK(hipGetLastError());
        launches = 1;
    }
    if (overlap) {                                                            // submitted after the fill, runs beside it
        HIPCHK(hipStreamWaitEvent(s.tb_stream, s.ev_fork, 0));
        aln_launch_traceback_overlap(&ta, k.tb_waves, s.tb_stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(s.ev_join, s.tb_stream));
    }
    for (size_t j = 0; j < k.single_pairs.size(); ++j) {
        const PairDesc &d = k.descs[k.single_pairs[j]];
        SingleArgs sa{};
        sa.seqs = s.seqs.as<uint8_t>(); sa.descs = s.descs.as<PairDesc>(); sa.pair = k.single_pairs[j]; sa.dirs = s.dirs.as<uint8_t>();
        sa.results = s.results.as<aln_pair_result>(); sa.granules = s.granules.as<uint32_t>();
        sa.gstride = ((uint64_t)d.N + 64 + 63) & ~63ull;
        sa.advice = s.advice1.as<uint8_t>(); sa.zrow = s.advice1.as<uint8_t>() + single_advice_bytes(k.single_max_n);
        sa.cand = s.cand.as<int32_t>(); sa.ctrl = s.ctrl.as<uint32_t>(); sa.matrix = s.matrix.p;
        sa.rows = c.rows; sa.cols = c.cols; sa.del = c.p.del; sa.ext = c.p.ext;
        sa.semantics = c.semantics; sa.R = k.single_r[j];
        sa.ns = (d.M + 64 * sa.R - 1) / (64 * sa.R);
        sa.hazard = (sa.semantics == ALN_CORE_LOCAL && sa.del != sa.ext && d.N >= 2) ? 1u : 0u;
        sa.store_dirs = c.store_dirs ? 1u : 0u;
        { const char *td = getenv("ALN_TEST_DROP_STRIP"); sa.test_drop = td ? (uint32_t)atoi(td) : 0u; }
        if (getenv("ALN_TEST_FAIL_REPAIR")) sa.test_drop = 0xffffffffu;      // the repair run is declared failed (tests)
        sa.max_passes = sa.hazard ? std::min<uint32_t>(c.p.max_passes ? c.p.max_passes : 4u, 12u) : 1u;
        // Localized repair (hazard pairs): when pass 0's advice is wrong in leading columns only, the first rep_S strips re-run
        // their leading columns (strip s up to step rep_K + 64 (rep_S - 1 - s)) instead of the whole pipeline running again.
        sa.rep_S = 0; sa.rep_K = 0; sa.mode = 0;
        {
            const uint32_t S = sa.R >= 2 ? 4u : 8u, K = 256u;
            if (sa.hazard && !getenv("ALN_NO_SINGLE_REPAIR") && sa.R <= 2 && sa.ns > S && d.N >= K + 64u * S + 128u) { sa.rep_S = S; sa.rep_K = K; }
            uint8_t *rb = s.repair.as<uint8_t>();
            sa.rgranules = reinterpret_cast<uint32_t *>(rb);
            sa.ckpt = reinterpret_cast<int *>(rb + 8ull * k.granule_stride_max * 4);
            sa.rcand = reinterpret_cast<int32_t *>(rb + 8ull * k.granule_stride_max * 4 + 8ull * 18 * 64 * 4);
        }
        aln_launch_single_init(&sa, (uint32_t)single_advice_bytes(d.N), st);
        // the granule rows must read "not yet produced" before a pass: one memset here, later passes are zeroed by the
        // finalize kernel that arms them
        HIPCHK(hipMemsetAsync(s.granules.p, 0, (size_t)sa.ns * sa.gstride * 4, st));
        for (uint32_t pass = 0; pass < sa.max_passes; ++pass) {
            sa.pass = pass;
            aln_launch_single(&sa, d.N, pass + 1 == sa.max_passes ? 1 : 0, st);
            launches++;
            if (pass == 0 && sa.rep_S) aln_launch_single_repair(&sa, d.N, st);      // exits at once unless pass 0 armed it
        }
        HIPCHK(hipGetLastError());
    }
    for (size_t j = 0; j < k.wg_pairs.size(); ++j) {
        const PairDesc &d = k.descs[k.wg_pairs[j]];
        WgArgs wa{};
        wa.seqs = s.seqs.as<uint8_t>(); wa.descs = s.descs.as<PairDesc>(); wa.pair = k.wg_pairs[j]; wa.dirs = s.dirs.as<uint8_t>();
        wa.results = s.results.as<aln_pair_result>(); wa.matrix = s.matrix.p; wa.rows = c.rows; wa.cols = c.cols;
        wa.del = c.p.del; wa.ext = c.p.ext; wa.semantics = c.semantics; wa.R = k.wg_r[j]; wa.ns = (d.M + 64 * wa.R - 1) / (64 * wa.R);
        wa.max_passes = c.p.max_passes; wa.store_dirs = c.store_dirs ? 1u : 0u; wa.scratch = s.scratch.as<uint8_t>();
        wa.hmat = c.want_h ? s.hmat.p : nullptr;
        wa.pwm = c.pwm ? 1u : 0u;
        aln_launch_wgpipe(&wa, c.is_int ? 1 : 0, aln_wg_lds_bytes(c.rows, c.cols, c.is_int ? 4u : 8u, wa.ns, d.N), st);
        launches++;
        HIPCHK(hipGetLastError());
    }
    if (ev) HIPCHK(hipEventRecord(ev[1], st));
    if (s.ev_fill) HIPCHK(hipEventRecord(s.ev_fill, st));
    if (c.want_tb && c.store_dirs) {
        if (overlap) HIPCHK(hipStreamWaitEvent(st, s.ev_join, 0));
        // every pair except those in the uniform-R layout (handled below); pairs the single-pair route hands to the strict-order
        // kernel (row-major layout) are walked by these two as well
        const bool batch_tb = k.n_small != 0 || c.semantics == ALN_CORE_LOCAL || !k.wg_pairs.empty();
        if (batch_tb) {
            // few pairs: one WAVE per pair, its lanes fetching the direction quads ahead of the path (tb_walk_pair_wave: the walk of
            // 256 pairs of 4200 x 4200 3.7 -> ~1 ms); many pairs: one lane per pair, 64 walks in flight per wave.  ALN_TB_WAVE=0 / 1
            // forces one or the other.
            // (up to 4095 pairs when they are long: 3000 C5 pairs host to host 5.27 -> 4.56 ms, 3072 pairs of 4200 x 4200 27.8 -> 26.3;
            // from 4096 pairs on the walks of such batches run beside the fill)
            bool wave_walk = k.n <= 2048 || (k.n < 4096 && k.max_len >= 1024);
            if (const char *e = getenv("ALN_TB_WAVE")) wave_walk = atoi(e) != 0;
            if (wave_walk) aln_launch_traceback_wave(&ta, st);
            else aln_launch_traceback(&ta, st);
        }
        for (size_t j = 0; j < k.single_pairs.size(); ++j) {
            const PairDesc &d = k.descs[k.single_pairs[j]];
            TraceSingleArgs tsa{};
            tsa.seqs = s.seqs.as<uint8_t>(); tsa.descs = s.descs.as<PairDesc>(); tsa.pair = k.single_pairs[j]; tsa.dirs = s.dirs.as<uint8_t>();
            tsa.results = s.results.as<aln_pair_result>(); tsa.tb = s.tb.as<uint8_t>(); tsa.tags = s.tags.as<uint8_t>();
            tsa.semantics = c.semantics;
            tsa.R = k.single_r[j]; tsa.ns = (d.M + 64 * tsa.R - 1) / (64 * tsa.R);
            tsa.map = s.tbmap.as<uint4>(); tsa.seg = s.tbmap.as<uint4>() + (uint64_t)tsa.ns * (d.N + 1);
            aln_launch_traceback_single(&tsa, d.N, st);
            aln_launch_traceback_expand_single(&ta, tsa.pair, st);
        }
        for (size_t j = 0; j < k.wg_pairs.size(); ++j) {      // the same parallel traceback for the pairs one workgroup filled
            const PairDesc &d = k.descs[k.wg_pairs[j]];
            TraceSingleArgs tsa{};
            tsa.seqs = s.seqs.as<uint8_t>(); tsa.descs = s.descs.as<PairDesc>(); tsa.pair = k.wg_pairs[j]; tsa.dirs = s.dirs.as<uint8_t>();
            tsa.results = s.results.as<aln_pair_result>(); tsa.tb = s.tb.as<uint8_t>(); tsa.tags = s.tags.as<uint8_t>();
            tsa.semantics = c.semantics;
            tsa.R = k.wg_r[j]; tsa.ns = (d.M + 64 * tsa.R - 1) / (64 * tsa.R);
            tsa.map = s.tbmap.as<uint4>(); tsa.seg = s.tbmap.as<uint4>() + (uint64_t)tsa.ns * (d.N + 1);
            tsa.pwm = c.pwm ? 1u : 0u;
            aln_launch_traceback_single(&tsa, d.N, st);
            aln_launch_traceback_expand_single(&ta, tsa.pair, st);
        }
        if (batch_tb) aln_launch_traceback_expand(&ta, st);
        HIPCHK(hipGetLastError());
    }
    if (c.unscale != 1.0) { aln_launch_scale_results(s.results.as<aln_pair_result>(), (uint32_t)k.n, c.unscale, st); HIPCHK(hipGetLastError()); }
    if (ev) HIPCHK(hipEventRecord(ev[2], st));
    if (fill_launches) *fill_launches = launches;
    return ALN_OK;
}

// diagnostics (ALN_COOP_STATS): the control words of the cooperative passes after a run
static void coop_stats_slot(Slot &s, const Chunk &k)
{
    if (k.coop && getenv("ALN_COOP_STATS")) {
        uint32_t w[256] = {0};
        if (hipMemcpy(w, s.coop.p, sizeof w, hipMemcpyDeviceToHost) == hipSuccess) {
            if (w[14]) {
                fprintf(stderr, "coop: %u strips left rows without their tag; the first: N %u, strip %u of %u, R %u, own %u, open %u; columns (tag found):", w[14], w[16], (w[15] >> 4), w[17] >> 8,
                        w[15] & 15u, w[17] & 1u, (w[17] >> 1) & 1u);
                for (int i = 0; i < 40 && w[128 + i]; ++i) fprintf(stderr, " %u(%#x)", w[128 + i], w[192 + i]);
                fprintf(stderr, "\n");
            }
            fprintf(stderr, "coop: urgent passes %u, unclaimed strips now %d urgent %d lazy, finished %u, strips helped %u, scans %u, aborted passes %u, spins waiting for strips to finish %u; "
                    "strips that gave up waiting for the row above %u, owners that gave up waiting for a strip %u, bottom-row records not found %u, pair granules not found %u\n",
                    w[ALN_COOP_ULOGW], (int)w[ALN_COOP_UOPEN], (int)w[ALN_COOP_LOPEN], w[ALN_COOP_FINISHED], w[ALN_COOP_HELPED], w[ALN_COOP_SCANS], w[ALN_COOP_ABORTS], w[9],
                    w[10], w[11], w[12], w[13]);
        }
    }
}
// D2H of one chunk on `st`, then a stream sync: summaries into results[first ..], strings into tb_buf.  When the caller's
// tb_off is the documented cumulative layout the chunk's strings are ONE span of tb_buf and are copied there directly;
// any other layout goes through pinned staging and one memcpy per string.
static int slot_download(Slot &s, const Call &c, const Chunk &k, hipStream_t st, aln_pair_result *results, uint8_t *tb_buf,
                         const uint64_t *tb_off)
{
    if (k.n == 0) return ALN_OK;
    HIPCHK(hipMemcpyAsync(results + k.first, s.results.p, k.n * sizeof(aln_pair_result), hipMemcpyDeviceToHost, st));
    const bool want = tb_buf && tb_off && c.store_dirs && c.want_tb && k.tb_bytes;
    bool direct = want;
    if (want) {
        const uint64_t base = tb_off[k.first];
        for (size_t i = 0; i < k.n && direct; ++i) direct = tb_off[k.first + i] - base == k.descs[i].tb_off && tb_off[k.first + i] >= base;
        if (direct) HIPCHK(hipMemcpyAsync(tb_buf + base, s.tb.p, k.tb_bytes, hipMemcpyDeviceToHost, st));
        else {
            int e = pin_ensure(s.h_out, k.tb_bytes);
            if (e != ALN_OK) return e;
            HIPCHK(hipMemcpyAsync(s.h_out.p, s.tb.p, k.tb_bytes, hipMemcpyDeviceToHost, st));
        }
    }
    HIPCHK(hipStreamSynchronize(st));
    coop_stats_slot(s, k);
    if (want && !direct) {
        const uint8_t *h = s.h_out.as<uint8_t>();
        for (size_t i = 0; i < k.n; ++i) {
            const PairDesc &d = k.descs[i];
            const aln_pair_result &r = results[k.first + i];
            if (r.status != ALN_OK) continue;
            const uint64_t cap = (uint64_t)d.N + d.M + 2;
            uint8_t *dst = tb_buf + tb_off[k.first + i];
            if (c.pwm) {   // u32 column numbers, then the residue string
                memcpy(dst, h + d.tb_off, 4ull * r.aln_len);
                memcpy(dst + 4 * cap, h + d.tb_off + 4 * cap, r.aln_len);
            } else {
                memcpy(dst, h + d.tb_off, r.aln_len);
                memcpy(dst + cap, h + d.tb_off + cap, r.aln_len);
            }
        }
    }
    return ALN_OK;
}

// ---------------------------------------------------------------- chunking of a batch call
// Chunks are ranges of the caller's pair order, between 5e9 and 1.6e10 cells each (1.4-4.6 GB of packed directions, 2-6 ms
// of fill): small enough that the first upload and the last traceback + download -- the only stages nothing overlaps -- are a
// few per cent of a large call, large enough that a chunk fills the chip and that its largest pairs (a 2000 x 2000 pair takes
// ~3 ms on its wave whatever else runs) do not outlast it by much.  ALN_CHUNK_CELLS overrides.
// Measured (profiles/r02_e2e_chunking.txt): C5 100 000 pairs (47.3 ms resident): 30 chunks 60 ms, 16 chunks 54 ms, 8 chunks 52 ms;
// 25 000 pairs: 6 chunks 18.2 ms, 2-4 chunks 16.6-17.1, one chunk 21.1; 12 500 pairs: 7 chunks 15.6 ms, 3 chunks 9.7, one 10.7.
static void make_chunks(const Call &c, const uint64_t *q_len, const uint64_t *t_len, size_t n, size_t ndev,
                        std::vector<std::pair<size_t, size_t>> &out, double waves = 3072.0)
{
    out.clear();
    double total = 0;
    for (size_t i = 0; i < n; ++i) total += (double)(c.pwm ? c.cols : q_len[i]) * (double)t_len[i];
    // (several devices: the same bounds per device -- each takes about three chunks or more from the common queue)
    double target = std::min(1.6e10, std::max(5.0e9, total / (4.0 * (double)ndev)));
    // The generic (f64 / off-fast-path) kernels take ~3 x as long per cell and have no cooperative passes: a chunk ends with its
    // largest pairs alone on their waves for 10-30 ms, so what counts is pairs per wave, not overlap of the copies (r03, 20 000
    // real-valued C5 pairs: four chunks 44.6 ms, one 40.7, resident 29.9).
    if (!c.fast) target *= 3.0;
    // Large pairs: what a chunk needs is pairs, not cells -- two or more per resident wave, or the whole batch, so that either every
    // wave has pairs of its own or the chunk is alone on the chip and its waves share the strips of a pair (cooperative passes are
    // for a kernel that has the chip to itself).  r03, 1024 pairs of 4200 x 4200, score only: four chunks of 284 pairs 31 ms (one
    // wave per pair, 9 strips one after the other), one chunk 11.2 ms; 2000 pairs 31 -> 17.3 ms; 512 pairs 15.7 -> 8.9 ms.  Large
    // pairs carry few bytes per cell, so the copies that chunking would overlap are small; the cap keeps a chunk's packed directions
    // at 16 GB (four slots).
    // (pairs of C5's size -- 1.2e6 cells on average -- keep the bounds above: they were measured on them)
    if (n && total / (double)n >= 3.0e6) target = std::max(target, std::min(6.4e10, 2.0 * waves * (total / (double)n)));
    if (const char *e = getenv("ALN_CHUNK_CELLS")) target = std::max(1.0, atof(e));
    // (one chunk up to 2e10 cells: a one-chunk call uploads its residues while it plans and its kernel has the chip to itself -- 12 500
    // C5 pairs, 1.5e10 cells: 8.7 ms against 9.1-9.4 in three chunks; 25 000 pairs, 3e10: 17 against 15.5 in four)
    // -- unless the strings it brings back are many: nothing overlaps that copy in a one-chunk call (100 000 PWM windows with both
    // strings, 316 MB: 15.3 ms in one chunk, 13.2 in two)
    double out_bytes = 0;
    if (c.want_tb) for (size_t i = 0; i < n; ++i) out_bytes += (c.pwm ? 5.0 : 2.0) * ((double)(c.pwm ? c.cols : q_len[i]) + (double)t_len[i] + 2.0);
    if (total <= std::max(1.5 * target, 2.0e10) && ndev == 1 && out_bytes <= 64.0e6) { out.emplace_back(0, n); return; }
    if (total <= 1.5 * target) { out.emplace_back(0, n); return; }
    size_t first = 0;
    double acc = 0;
    for (size_t i = 0; i < n; ++i) {
        acc += (double)(c.pwm ? c.cols : q_len[i]) * (double)t_len[i];
        if (acc >= target || i + 1 - first >= (1u << 22)) { out.emplace_back(first, i + 1 - first); first = i + 1; acc = 0; }
    }
    if (first < n) {
        // a short tail joins the chunk before it
        if (!out.empty() && acc < 0.25 * target) out.back().second += n - first;
        else out.emplace_back(first, n - first);
    }
}

// The chunking aln_align_batch would use for these lengths on a context of n_devices GPUs: pure host arithmetic (no device is
// touched), exported so that the sharding can be inspected and tested anywhere.  Returns the number of chunks; writes up to
// `cap` (first pair, pair count) entries.
extern "C" size_t aln_plan_chunks(const aln_params *params, const uint64_t *q_len, const uint64_t *t_len, size_t n_pairs,
                                  int n_devices, uint64_t *first, uint64_t *count, size_t cap)
{
    if (!params || (n_pairs && (!q_len || !t_len)) || n_devices < 1) return 0;
    Call c;
    if (call_init(c, params, q_len, t_len, n_pairs, false) != ALN_OK) {      // lengths only (no matrix): planned as for the fast kernels
        c = Call();
        c.pwm = params->semantics == ALN_PWM_LOCAL;
        c.cols = params->cols;
        c.fast = true;
    }
    std::vector<std::pair<size_t, size_t>> ranges;
    if (n_pairs) make_chunks(c, q_len, t_len, n_pairs, (size_t)n_devices, ranges);
    for (size_t i = 0; i < ranges.size() && i < cap; ++i) {
        if (first) first[i] = ranges[i].first;
        if (count) count[i] = ranges[i].second;
    }
    return ranges.size();
}

// everything a pipelined call shares between its threads
struct BatchJob {
    const Call *c;
    const uint8_t *seqs;
    const uint64_t *q_off, *q_len, *t_off, *t_len;
    aln_pair_result *results;
    uint8_t *tb_buf;
    const uint64_t *tb_off;
    const std::vector<std::pair<size_t, size_t>> *ranges;
    Need need;
    std::atomic<size_t> next{0};       // the chunk queue: devices take the next range when they have a free slot
    std::atomic<int> failed{0};
    std::mutex mu;
    int status = ALN_OK;
    std::string err;
    void fail_with(int st, const std::string &e)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (status == ALN_OK) { status = st; err = e; }
        failed.store(1);
    }
};

// One device's share of a pipelined call: this thread takes chunks from the job's queue, plans, uploads and launches them; a
// second thread waits for each chunk's kernels and copies its results back; a slot is reused once its chunk has been fetched.
static void device_pipeline(DevCtx *dev, BatchJob &job)
{
    if (hipSetDevice(dev->device) != hipSuccess) { job.fail_with(ALN_ERR_DEVICE, "hipSetDevice failed"); return; }
    const Call &c = *job.c;
    const size_t nc = job.ranges->size();
    Slot *slots[ALN_POOL_SLOTS];
    const int ns = pool_lease(dev, ALN_POOL_SLOTS, slots);
    struct Release { DevCtx *c; Slot **s; int n; ~Release() { pool_release(c, s, n); } } rel{dev, slots, ns};
    struct Shared {
        std::mutex mu;
        std::condition_variable cv;
        std::deque<std::pair<size_t, int>> ready;      // (local chunk number, slot) launched, to be fetched, in order
        std::vector<char> slot_free;
        bool done_issuing = false;
    } sh;
    sh.slot_free.assign(ns, 1);
    static thread_local std::vector<Chunk> plans_tl;   // (kept from call to call: see Chunk::reset)
    if (plans_tl.size() < (size_t)ns) plans_tl.resize(ns);
    std::vector<Chunk> &plans = plans_tl;              // the plan of the chunk each slot holds
    size_t depth = 3;
    if (const char *e = getenv("ALN_FILL_DEPTH")) depth = std::max(1, atoi(e));
    std::thread fetcher([&] {
        (void)hipSetDevice(dev->device);
        for (;;) {
            int si;
            {
                std::unique_lock<std::mutex> lk(sh.mu);
                sh.cv.wait(lk, [&] { return !sh.ready.empty() || sh.done_issuing; });
                if (sh.ready.empty()) return;
                si = sh.ready.front().second;
                sh.ready.pop_front();
            }
            Slot &s = *slots[si];
            const int e = slot_download(s, c, plans[si], s.stream, job.results, job.tb_buf, job.tb_off);
            if (e != ALN_OK) job.fail_with(e, g_err);
            {
                std::lock_guard<std::mutex> lk(sh.mu);
                sh.slot_free[si] = 1;
            }
            sh.cv.notify_all();
        }
    });
    int st = ALN_OK;
    for (size_t li = 0; st == ALN_OK && !job.failed.load(); ++li) {
        const int si = (int)(li % (size_t)ns);
        {   // the slot's previous chunk must have been fetched before its plan and buffers are reused
            std::unique_lock<std::mutex> lk(sh.mu);
            sh.cv.wait(lk, [&] { return sh.slot_free[si] != 0; });
        }
        const size_t ci = job.next.fetch_add(1);
        if (ci >= nc) break;
        Chunk &k = plans[si];
        k.reset();
        Slot &s = *slots[si];
        // the first chunk of this pipeline: nothing runs yet, so its residues go up on a helper thread while it is planned (as in a
        // one-chunk call); every later chunk is planned and uploaded while the chunks before it fill
        std::thread copier;
        hipError_t copy_err = hipSuccess;
        uint64_t lo = ~0ull, hi = 0, sum = 0;
        bool early = false;
        if (li == 0 && !getenv("ALN_NO_EARLY_UPLOAD")) {
            const size_t f0 = (*job.ranges)[ci].first, f1 = f0 + (*job.ranges)[ci].second;
            for (size_t i = f0; i < f1; ++i) {
                if (!c.pwm && job.q_len[i]) { lo = std::min(lo, job.q_off[i]); hi = std::max(hi, job.q_off[i] + job.q_len[i]); sum += job.q_len[i]; }
                if (job.t_len[i]) { lo = std::min(lo, job.t_off[i]); hi = std::max(hi, job.t_off[i] + job.t_len[i]); sum += job.t_len[i]; }
            }
            early = lo != ~0ull && (hi - lo) <= 2 * sum + 65536 && hi - lo >= (4u << 20) && hi - lo <= job.need.seq_span;
            if (early) {
                if ((st = slot_init(s)) != ALN_OK || (st = dev_ensure(s.seqs, job.need.seq_span + 64, s.pooled)) != ALN_OK) break;
                copier = std::thread([&] {
                    copy_err = hipSetDevice(dev->device);
                    if (copy_err == hipSuccess) copy_err = hipMemcpyAsync(s.seqs.p, job.seqs + lo, hi - lo, hipMemcpyHostToDevice, s.stream);
                });
            }
        }
        struct Join { std::thread &t; ~Join() { if (t.joinable()) t.join(); } } join{copier};
        // (walk waves beside the LAST chunk's own fill, as a staged batch has them, were measured: 53.7 ms against 52.0 without)
        st = chunk_plan(dev, c, job.q_off, job.q_len, job.t_off, job.t_len, (*job.ranges)[ci].first, (*job.ranges)[ci].second, false, k, nc > 4);
        if (st != ALN_OK) break;
        if ((st = slot_ensure(s, c, k, &job.need)) != ALN_OK) break;
        if (copier.joinable()) copier.join();
        if (copy_err != hipSuccess) { st = fail(copy_err, "hipMemcpyAsync(residues)"); break; }
        const bool seqs_there = early && k.seq_direct && k.seq_lo == lo && k.seq_span == hi - lo;
        if ((st = slot_upload(s, c, k, job.seqs, job.q_off, job.q_len, job.t_off, job.t_len, s.stream, false, seqs_there)) != ALN_OK) break;
        // At most `depth` fills share the chip: chunk i's fill waits for the fill of chunk i - depth.  With every slot's fill
        // started at once the chunks run in lockstep -- they share the chip equally, reach their tails together (a chunk's
        // largest pairs take as long as the whole chunk), then all walk and copy while nothing fills.  Fewer at a time stay
        // staggered: the oldest is in its tail while the younger ones fill the waves it leaves free (C5, 16 chunks: depth
        // 1: 83 ms, 2: 53.7, 3: 52.6, 4 = every slot: 56.5).
        hipEvent_t after = (li >= depth && ns > (int)depth) ? slots[(li - depth) % (size_t)ns]->ev_fill : nullptr;
        if ((st = slot_launch(dev, s, c, k, s.stream, nullptr, nullptr, after)) != ALN_OK) break;
        {
            std::lock_guard<std::mutex> lk(sh.mu);
            sh.slot_free[si] = 0;
            sh.ready.emplace_back(li, si);
        }
        sh.cv.notify_all();
    }
    if (st != ALN_OK) job.fail_with(st, g_err);
    {
        std::lock_guard<std::mutex> lk(sh.mu);
        sh.done_issuing = true;
    }
    sh.cv.notify_all();
    fetcher.join();
    for (int i = 0; i < ns; ++i) (void)hipStreamSynchronize(slots[i]->stream);
}

extern "C" int aln_align_batch(aln_ctx *ctx, const aln_params *params, const uint8_t *seqs, const uint64_t *q_off,
                               const uint64_t *q_len, const uint64_t *t_off, const uint64_t *t_len, size_t n_pairs,
                               aln_pair_result *results, uint8_t *tb_buf, const uint64_t *tb_off)
{
    if (!ctx || !params || (n_pairs && (!seqs || !q_off || !q_len || !t_off || !t_len || !results))) { g_err = "null argument"; return ALN_ERR_INVALID_ARGUMENT; }
    Call c;
    int st = call_init(c, params, q_len, t_len, n_pairs, false);
    if (st != ALN_OK) return st;
    if (n_pairs == 0) return ALN_OK;
    std::vector<std::pair<size_t, size_t>> ranges;
    make_chunks(c, q_len, t_len, n_pairs, ctx->devs.size(), ranges, 12.0 * (double)ctx->devs[0]->cus);
    const size_t nc = ranges.size();

    if (nc == 1) {                                   // one chunk: everything on the caller's thread, on the next device in turn
        DevCtx *dev = ctx->next_device();
        HIPCHK(hipSetDevice(dev->device));
        Slot *sl[1];
        pool_lease(dev, 1, sl);
        struct Release { DevCtx *c; Slot **s; ~Release() { pool_release(c, s, 1); } } rel{dev, sl};
        static thread_local Chunk k_tl;              // (kept from call to call: see Chunk::reset)
        Chunk &k = k_tl;
        k.reset();
        Slot &s = *sl[0];
        // ALN_TRACE_CALL=1: where the wall clock of a one-chunk call goes (stderr)
        const bool trace = getenv("ALN_TRACE_CALL") != nullptr;
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        const auto t0 = now();
        // The residues do not wait for the plan: where they lie is a min / max over the offsets (the same arithmetic as chunk_plan's),
        // and copying pageable memory keeps the calling thread busy for its whole length (0.55 ms for the 27 MB of 12 500 C5 pairs,
        // beside 0.7 ms of planning) -- a helper thread copies while this one plans.
        uint64_t lo = ~0ull, hi = 0, sum = 0;
        for (size_t i = 0; i < n_pairs; ++i) {
            if (!c.pwm && q_len[i]) { lo = std::min(lo, q_off[i]); hi = std::max(hi, q_off[i] + q_len[i]); sum += q_len[i]; }
            if (t_len[i]) { lo = std::min(lo, t_off[i]); hi = std::max(hi, t_off[i] + t_len[i]); sum += t_len[i]; }
        }
        const bool early = lo != ~0ull && (hi - lo) <= 2 * sum + 65536 && hi - lo >= (4u << 20) && !getenv("ALN_NO_EARLY_UPLOAD");
        std::thread copier;
        hipError_t copy_err = hipSuccess;
        if (early) {
            if ((st = slot_init(s)) != ALN_OK || (st = dev_ensure(s.seqs, hi - lo + 64, s.pooled)) != ALN_OK) return st;
            copier = std::thread([&] {
                copy_err = hipSetDevice(dev->device);
                if (copy_err == hipSuccess) copy_err = hipMemcpyAsync(s.seqs.p, seqs + lo, hi - lo, hipMemcpyHostToDevice, s.stream);
            });
        }
        struct Join { std::thread &t; ~Join() { if (t.joinable()) t.join(); } } join{copier};
        if ((st = chunk_plan(dev, c, q_off, q_len, t_off, t_len, 0, n_pairs, true, k)) != ALN_OK) return st;
        const auto t1 = now();
        if ((st = slot_ensure(s, c, k)) != ALN_OK) return st;        // (the residue buffer has its size already: it is not moved)
        const auto t2 = now();
        if (copier.joinable()) copier.join();
        if (copy_err != hipSuccess) return fail(copy_err, "hipMemcpyAsync(residues)");
        const bool seqs_there = early && k.seq_direct && k.seq_lo == lo && k.seq_span == hi - lo;
        if ((st = slot_upload(s, c, k, seqs, q_off, q_len, t_off, t_len, s.stream, false, seqs_there)) != ALN_OK) return st;
        const auto t3 = now();
        if ((st = slot_launch(dev, s, c, k, s.stream, nullptr, nullptr)) != ALN_OK) { (void)hipStreamSynchronize(s.stream); return st; }
        const auto t4 = now();
        if (trace) (void)hipStreamSynchronize(s.stream);
        const auto t5 = now();
        st = slot_download(s, c, k, s.stream, results, tb_buf, tb_off);
        if (trace) fprintf(stderr, "aln_align_batch (one chunk, %zu pairs): plan %.3f ensure %.3f upload %.3f launch %.3f kernels %.3f download %.3f ms\n",
                           n_pairs, ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4), ms(t4, t5), ms(t5, now()));
        return st;
    }

    BatchJob job;
    job.c = &c; job.seqs = seqs; job.q_off = q_off; job.q_len = q_len; job.t_off = t_off; job.t_len = t_len;
    job.results = results; job.tb_buf = tb_buf; job.tb_off = tb_off; job.ranges = &ranges;
    {   // upper bounds over the chunks: every slot is sized once, before the pipeline runs
        const uint64_t sc = (c.is_int && !c.fast) ? 4 : 8;       // fast kernels: 8-byte granules (chunk_plan)
        int max_cus = 0;
        for (const DevCtx *d : ctx->devs) max_cus = std::max(max_cus, d->cus);
        Need &need = job.need;
        for (const auto &r : ranges) {
            uint64_t lo = ~0ull, hi = 0, sum = 0, dirs = 0, tb = 0, tags = 0, mlen = 1;
            for (size_t i = r.first; i < r.first + r.second; ++i) {
                const uint64_t N = c.pwm ? c.cols : q_len[i], M = t_len[i], cap = N + M + 2;
                if (!c.pwm && N) { lo = std::min(lo, q_off[i]); hi = std::max(hi, q_off[i] + N); sum += N; }
                if (M) { lo = std::min(lo, t_off[i]); hi = std::max(hi, t_off[i] + M); sum += M; }
                if (c.store_dirs) { tb += c.pwm ? ((5 * cap + 3) & ~3ull) : 2 * cap; tags += (cap + 3) & ~3ull; }
                if (N && M && c.store_dirs) dirs += aln_dir_bytes((uint32_t)N, (uint32_t)M);
                if (N && M) mlen = std::max(mlen, std::max(N, M));
            }
            if (c.fast) need.coop = std::max<uint64_t>(need.coop, 4ull * (ALN_COOP_CTL_WORDS + (uint64_t)max_cus * 16 + 64) + (uint64_t)max_cus * 16 * sizeof(CoopRec));
            const uint64_t span = hi > lo ? ((hi - lo) <= 2 * sum + 65536 ? hi - lo : sum) : 0;
            need.seq_span = std::max(need.seq_span, span);
            need.n = std::max<uint64_t>(need.n, r.second);
            need.dir_bytes = std::max(need.dir_bytes, dirs);
            need.tb_bytes = std::max(need.tb_bytes, tb);
            need.tag_bytes = std::max(need.tag_bytes, tags);
            // per wave: boundary row, advice, bottom-row record, checkpoints (chunk_plan)
            const uint64_t nrows = (c.fast && c.semantics == ALN_CORE_LOCAL && c.p.del != c.p.ext) ? ALN_CASCADE_ROWS : (c.fast ? 2u : 1u);
            const uint64_t stride = nrows * (((mlen + 66) * sc + 63) & ~63ull) + ((mlen + 66 + 63) & ~63ull) +
                                    std::max<uint64_t>((mlen + 66 + 63) & ~63ull, ((c.fast ? 8 : 4) * ((mlen + 63) / 2 + 8) + 63) & ~63ull) +
                                    (c.fast ? (uint64_t)ALN_CK_SLOTS * 18 * 64 * 4 : (((mlen + 66) * sc + 63) & ~63ull) + ((mlen + 66 + 63) & ~63ull));
            need.scratch = std::max(need.scratch, (uint64_t)max_cus * 4 * 4 * stride);
        }
    }
    // ---- the devices of the context take the chunks from one queue; device 0's pipeline runs on the caller's thread
    const size_t nd = std::min(ctx->devs.size(), nc);
    std::vector<std::thread> others;
    for (size_t d = 1; d < nd; ++d) others.emplace_back([&, d] { device_pipeline(ctx->devs[d], job); });
    device_pipeline(ctx->devs[0], job);
    for (auto &t : others) t.join();
    if (job.status != ALN_OK) { g_err = job.err; return job.status; }
    return ALN_OK;
}

// ---------------------------------------------------------------- staged batch: one chunk resident in a private slot
struct aln_batch {
    DevCtx *ctx = nullptr;           // a staged batch lives on ONE device: the context's first
    Call call;
    Chunk k;
    Slot *slot = nullptr;
    hipStream_t last_stream = nullptr;
    // timing ring: one event triple per run (fill start, fill end, traceback end), recorded on the launch stream
    bool timing = false;
    std::vector<hipEvent_t> ev;
    uint32_t ev_runs = 0;
    uint32_t fill_launches = 0;
};

static void batch_free(aln_batch *b)
{
    if (!b) return;
    (void)hipSetDevice(b->ctx->device);
    for (auto &e : b->ev) if (e) (void)hipEventDestroy(e);
    slot_destroy(b->slot);
    delete b;
}

extern "C" void aln_batch_destroy(aln_batch *b) { batch_free(b); }

extern "C" aln_batch *aln_batch_create(aln_ctx *ctx, const aln_params *params, const uint8_t *seqs,
                                       const uint64_t *q_off, const uint64_t *q_len, const uint64_t *t_off,
                                       const uint64_t *t_len, size_t n_pairs, int *status)
{
    int st = ALN_OK;
    aln_batch *b = nullptr;
    if (!ctx || !params || (n_pairs && (!seqs || !q_off || !q_len || !t_off || !t_len))) { g_err = "null argument"; st = ALN_ERR_INVALID_ARGUMENT; }
    else {
        b = new aln_batch();
        b->ctx = ctx->devs[0];
        hipError_t e = hipSetDevice(b->ctx->device);
        if (e != hipSuccess) st = fail(e, "hipSetDevice");
        if (st == ALN_OK) st = call_init(b->call, params, q_len, t_len, n_pairs, false);
        if (st == ALN_OK) st = chunk_plan(b->ctx, b->call, q_off, q_len, t_off, t_len, 0, n_pairs, true, b->k);
        if (st == ALN_OK) {
            b->slot = new Slot();
            b->slot->pooled = false;
            st = slot_ensure(*b->slot, b->call, b->k);
        }
        if (st == ALN_OK) st = slot_upload(*b->slot, b->call, b->k, seqs, q_off, q_len, t_off, t_len, b->slot->stream);
        if (st == ALN_OK) {
            e = hipMemsetAsync(b->slot->results.p, 0, std::max<size_t>(n_pairs * sizeof(aln_pair_result), 1), b->slot->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(b->slot->stream);
            if (e != hipSuccess) st = fail(e, "stage");
        }
        if (st != ALN_OK) { batch_free(b); b = nullptr; }
    }
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
    hipStream_t s = stream ? (hipStream_t)stream : b->slot->stream;
    b->last_stream = s;
    hipEvent_t *ev = b->timing ? &b->ev[3 * (b->ev_runs % ALN_TIMING_SLOTS)] : nullptr;
    int st = slot_launch(b->ctx, *b->slot, b->call, b->k, s, ev, &b->fill_launches);
    if (st == ALN_OK && ev) b->ev_runs++;
    return st;
}

static void coop_stats(aln_batch *b) { coop_stats_slot(*b->slot, b->k); }

extern "C" int aln_batch_sync(aln_batch *b)
{
    if (!b) return ALN_ERR_INVALID_ARGUMENT;
    HIPCHK(hipSetDevice(b->ctx->device));
    HIPCHK(hipStreamSynchronize(b->last_stream ? b->last_stream : b->slot->stream));
    coop_stats(b);
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
    coop_stats(b);
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
    return slot_download(*b->slot, b->call, b->k, b->slot->stream, results, tb_buf, tb_off);
}

extern "C" uint64_t aln_batch_cells(const aln_batch *b) { return b ? b->k.cells : 0; }
extern "C" size_t aln_batch_size(const aln_batch *b) { return b ? b->k.n : 0; }
extern "C" void *aln_batch_results_device(aln_batch *b) { return b ? b->slot->results.p : nullptr; }
extern "C" uint64_t aln_batch_direction_bytes(const aln_batch *b)
{
    if (!b) return 0;
    // bytes the fill kernel actually stores: per strip, ceil(steps / SPB) blocks of 256 B
    uint64_t total = 0;
    for (const PairDesc &d : b->k.descs) {
        if (d.status != ALN_OK) continue;
        const uint32_t ns = aln_num_strips(d.M);
        for (uint32_t s = 0; s < ns; ++s) {
            const bool last = s + 1 == ns;
            const uint32_t rem = d.M - s * ALN_STRIP_ROWS;
            const int R = last ? aln_pick_r(rem) : ALN_FULL_R;
            const uint32_t rows = std::min<uint32_t>(rem, 64u * R), L = (rows + R - 1) / R, spb = aln_spb((uint32_t)R);
            total += (uint64_t)aln_strip_blocks(d.N + L - 1, spb) * 256u;
        }
    }
    return total;
}

// ---------------------------------------------------------------- one pair, blocking
extern "C" int aln_align_pair(aln_ctx *ctx, const aln_params *params, const uint8_t *query, size_t N,
                              const uint8_t *target, size_t M, aln_pair_result *out, uint8_t *q_aln, uint8_t *t_aln,
                              uint8_t *directions, double *h_matrix)
{
    if (!ctx || !params || !out) { g_err = "null argument"; return ALN_ERR_INVALID_ARGUMENT; }
    const bool pwm = params->semantics == ALN_PWM_LOCAL;
    if (pwm) { N = params->cols; query = nullptr; }          // the columns are the PWM positions (pwm/mod.rs:44-46)
    const size_t nq = pwm ? 0 : N;
    aln_params p = *params;
    p.outputs = (params->outputs ? params->outputs : (ALN_OUT_SCORE | ALN_OUT_TRACEBACK));
    if (q_aln && t_aln) p.outputs |= ALN_OUT_TRACEBACK;
    if (directions) p.outputs |= ALN_OUT_DIRECTIONS;
    const uint64_t qo = 0, ql = N, to = nq, tl = M;
    Call c;
    int st = call_init(c, &p, &ql, &tl, 1, h_matrix != nullptr);
    if (st != ALN_OK) { memset(out, 0, sizeof *out); out->status = st; return st; }
    DevCtx *dev = ctx->next_device();                        // concurrent callers spread over the context's devices
    HIPCHK(hipSetDevice(dev->device));
    Slot *sl[1];
    pool_lease(dev, 1, sl);
    struct Release { DevCtx *c; Slot **s; ~Release() { pool_release(c, s, 1); } } rel{dev, sl};
    Slot &s = *sl[0];
    Chunk k;
    // (allow_overlap = true: a pair that misses the single-pair route -- more than ~77 000 columns -- opens its first pass for other waves
    // like a one-pair batch call does: 200 000 x 5000 ran on ONE wave at 1.8 GCUPS before, the reference's only limit being memory,
    // simple/mod.rs:53-57)
    if ((st = chunk_plan(dev, c, &qo, &ql, &to, &tl, 0, 1, true, k)) != ALN_OK) return st;
    k.seq_direct = false;                                    // query and target are two caller buffers: gathered into staging
    k.seq_span = nq + M;
    k.descs[0].q_off = 0; k.descs[0].t_off = nq;
    if ((st = slot_ensure(s, c, k)) != ALN_OK) return st;
    {   // the gather of slot_upload, from two pointers
        uint8_t *h = s.h_in.as<uint8_t>();
        if (nq) memcpy(h, query, nq);
        if (M) memcpy(h + nq, target, M);
    }
    if ((st = slot_upload(s, c, k, nullptr, &qo, &ql, &to, &tl, s.stream, true)) != ALN_OK) return st;
    if ((st = slot_launch(dev, s, c, k, s.stream, nullptr, nullptr)) != ALN_OK) { (void)hipStreamSynchronize(s.stream); return st; }
    const size_t cap = N + M + 2;
    const uint64_t cells = (uint64_t)(N + 1) * (M + 1);
    const bool ok_shape = k.descs[0].status == ALN_OK;
    if (directions && ok_shape) {
        if ((st = dev_ensure(s.unpack, cells, true)) != ALN_OK) return st;
        aln_launch_unpack(s.dirs.as<uint8_t>(), s.descs.as<PairDesc>(), 0, c.semantics, s.unpack.as<uint8_t>(), cells, s.stream);
    }
    // summary + strings: the strings of one pair are fetched through pinned staging and handed out as two buffers
    HIPCHK(hipMemcpyAsync(out, s.results.p, sizeof *out, hipMemcpyDeviceToHost, s.stream));
    const bool want_tb = q_aln && t_aln && c.want_tb && k.tb_bytes;
    if (want_tb) {
        if ((st = pin_ensure(s.h_out, k.tb_bytes)) != ALN_OK) return st;
        HIPCHK(hipMemcpyAsync(s.h_out.p, s.tb.p, k.tb_bytes, hipMemcpyDeviceToHost, s.stream));
    }
    HIPCHK(hipStreamSynchronize(s.stream));
    if (out->status == ALN_OK && ok_shape) {
        if (want_tb) {
            const uint8_t *h = s.h_out.as<uint8_t>();
            if (pwm) {
                memcpy(q_aln, h, 4ull * out->aln_len);          // uint32 PWM column numbers
                memcpy(t_aln, h + 4 * cap, out->aln_len);
            } else {
                memcpy(q_aln, h, out->aln_len);
                memcpy(t_aln, h + cap, out->aln_len);
            }
        }
        if (directions) HIPCHK(hipMemcpy(directions, s.unpack.p, cells, hipMemcpyDeviceToHost));
        if (h_matrix) {
            if (c.is_int) {
                std::vector<int32_t> hi(cells);
                HIPCHK(hipMemcpy(hi.data(), s.hmat.p, cells * 4, hipMemcpyDeviceToHost));
                for (uint64_t i = 0; i < cells; ++i) h_matrix[i] = (double)hi[i];
            } else {
                HIPCHK(hipMemcpy(h_matrix, s.hmat.p, cells * 8, hipMemcpyDeviceToHost));
            }
        }
    }
    return out->status;
}
