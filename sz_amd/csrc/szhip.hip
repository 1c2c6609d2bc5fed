// szhip.hip -- C-ABI HIP layer (include/szhip.h): owns device buffers, launches the kernels of
// szhip_kernels.h in stream order and calls the short serial host pieces of szhost.c between them.
// gfx950 only; no CPU fallback: every entry point returns an error if the HIP runtime/device is missing.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <ctime>
#include <condition_variable>
#include <deque>
#include <functional>
#include <future>
#include <mutex>
#include <thread>
#include <vector>
#include "../../include/szhip.h"
#include "szhost.h"
#include "szhip_kernels.h"

namespace {

struct DevBuf { void *p = nullptr; size_t cap = 0; };

double now_ms()
{
    timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

} // namespace

struct szhip_sweep_gate { std::mutex m; hipEvent_t last = nullptr; int lanes = 2; };
// compress calls of this process that are inside the library right now, whatever their context: the chain / kernel overlap of an array with
// regression blocks is only taken by a call that starts alone (see compress_impl)
static std::atomic<int> g_compress_calls{0};
struct compress_call_guard { compress_call_guard() { g_compress_calls.fetch_add(1); } ~compress_call_guard() { g_compress_calls.fetch_sub(1); } };
// The host's coefficient chains (szhost_coeff_chain_one_p: one serial chain per coefficient) on PERSISTENT threads (round 5).  Threads created per
// call started 0.1 ms late on cores that had been idle (measured on the GPU box: four chains of 1.0 - 1.2 ms each took 2.1 - 2.7 ms end to
// end, and 4.9 on some runs).  A context that has met regression blocks once keeps four workers; they are woken when a compression STARTS and
// spin -- on cores that are awake by then -- until the fit pass has delivered the coefficients (~0.4 ms), or are sent back to sleep if the array
// has no regression block.  A job has two stages: the chain (the caller waits for it: the sweep needs the decoded coefficients), then the
// coefficient's section of the stream header (collected where the header is assembled).
struct szhip_chain_pool {
    std::thread th[4];
    std::mutex m; std::condition_variable cv;
    std::atomic<int> phase{0};                    // 0 asleep, 1 awake (spinning for a job), 2 quit
    std::atomic<uint64_t> seq{0};
    std::atomic<int> stage1{0}, stage2{0};
    int n = 0;
    std::function<void(int)> chain, section;
    static void pause() {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    void run(int e)
    {
        uint64_t seen = 0;
        for (;;) {
            { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return phase.load() != 0; }); }
            if (phase.load() == 2) return;
            while (phase.load(std::memory_order_acquire) == 1 && seq.load(std::memory_order_acquire) == seen) pause();
            const uint64_t s = seq.load(std::memory_order_acquire);
            if (s != seen) {
                seen = s;
                if (e < n) { chain(e); stage1.fetch_add(1, std::memory_order_release); section(e); }
                stage2.fetch_add(1, std::memory_order_release);
            }
        }
    }
    void start() { for (int e = 0; e < 4; ++e) th[e] = std::thread([this, e] { run(e); }); }
    void arm() { { std::lock_guard<std::mutex> lk(m); if (phase.load() == 0) phase.store(1); } cv.notify_all(); }
    void disarm() { std::lock_guard<std::mutex> lk(m); if (phase.load() == 1) phase.store(0); }
    void submit(int ncoef, std::function<void(int)> c, std::function<void(int)> sec)
    {
        n = ncoef; chain = std::move(c); section = std::move(sec);
        stage1.store(0); stage2.store(0);
        arm();
        seq.fetch_add(1, std::memory_order_release);
    }
    void wait_chains() { while (stage1.load(std::memory_order_acquire) < n) pause(); }
    void wait_all() { while (stage2.load(std::memory_order_acquire) < 4) std::this_thread::yield(); }
    ~szhip_chain_pool()
    {
        { std::lock_guard<std::mutex> lk(m); phase.store(2); }
        cv.notify_all();
        for (auto &t : th) if (t.joinable()) t.join();
    }
};

struct szhip_ctx {
    int device = 0;
    int fast_stat_per_cu[2] = {0, 0}, fast_pack_per_cu[2] = {0, 0};   // resident workgroups per CU of the fast mode's persistent kernels (float, double)
    int cus = 256;                               // compute units of `device` (hipDeviceAttributeMultiprocessorCount): persistent kernels launch one workgroup per CU at most
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;   // the fit + selection pass runs here, concurrently with the interval optimiser's sampling and host decisions
    hipEvent_t ev_in = nullptr, ev_fit = nullptr;
    int settle_probes = 0, settle_rejected = 0;   // settle_streams: queue probes made, streams replaced
    int side_prio = 0;               // 1: stream2 at the lowest, stream3 at the highest stream priority (create_ctx)
    hipStream_t stream3 = nullptr;   // the block-ordering pass of finished tile rows, while the sweep is still running on `stream` (created on first use)
    hipEvent_t ev_perm = nullptr;
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    char err[512] = {0};
    unsigned epoch = 0;
    // Huffman decode: two rounds without asking the device in between (checked with the caller's next synchronisation); a call whose
    // start guesses were still moving after them is repeated once with a synchronisation per round (with_ticket_fallback)
    bool hdec_sync_rounds = false, hdec_unconverged = false;
    // arrays in flight (szhip_pool), SZ_HIP_SWEEP_GATE=1: the sweeps of the lanes take turns instead of sharing the CUs.  Measured at 512^3
    // with two lanes over 60 steps: 294 - 303 GB/s gated against 305 - 319 free-running, so it is OFF by default.  The gate belongs to the
    // pool (nullptr for a lone context).
    struct szhip_sweep_gate *gate = nullptr;
    hipEvent_t ev_gate = nullptr;
    unsigned long long *hdec_res = nullptr;      // pinned: {symbols the payload holds, starts still moving after round 1}, copied asynchronously
    // workspaces (grow-only)
    DevBuf lor_bits, reg_flags, reg_rank, coef_compact, in, out, codes_nat, codes_blk, coef, blk_lor, faceI, faceJ, rb_down, rb_right, rb_vals, pt_flags, fast_slots, fast_units, progress, trace, order, small, hist, col_zeros, col_zeros64,
        col_off, partial, samples, unpred, stream_buf, chunk_bits, chunk_off, code_tab, len_tab, dec_tab,
        starts, ends, counts, offs, dirty, zcnt, zpos, pwr_log, pwr_signs, pwr_small, coef_dec, msst_ptab, msst_cells, msst_rec, msst_pe;
    void *pinned = nullptr; size_t pinned_cap = 0;
    void *pinned2 = nullptr; size_t pinned2_cap = 0;   // target of the second stream's copies (indicator bits, regression-block count)
    // bulk copies between the caller's pageable arrays and the device: SZH_STAGE_T host threads, two pinned buffers + events each
    void *stage_buf[8][2] = {};
    hipEvent_t stage_ev[8][2] = {};
    void *pinned3 = nullptr; size_t pinned3_cap = 0;
    int streams_independent = -1;                      // -1 not probed yet; 1: work on stream2 proceeds while a kernel on stream is running
    void *coh = nullptr; size_t coh_cap = 0;           // host-coherent (uncached on the GPU) pinned memory the wavefront kernel reads while the host writes   // the regression coefficients on their way to the host chain and back
    int order_nI = -1, order_nJ = -1;
    // tile tickets: by default a workgroup's index is its ticket (SZ_HIP_TICKET_MODE=2), which assumes that the workgroups of one XCD start in index
    // order AND that every XCD gets to run; if a wavefront-kernel wait ever times out the call is repeated once with the atomic ticket, which
    // assumes neither (a GPU shared with other work), and the context keeps that mode
    bool wave_timeout = false, ticket_atomic = false;
    // the coefficient chain beside the running sweep (M-field): a sweep that gave up waiting for the coefficients (seen 5 - 6 times in 480 rounds with
    // several arrays in flight) is answered by ONE repetition with the chain finished before the sweep starts; the context keeps that order
    bool coef_late = false, no_chain_overlap = false;
    std::vector<int> chain_codes; std::vector<unsigned char> chain_unpred;   // the chains' outputs, kept across calls (fresh memory page-faults under the chain: ~1 ms for the M-field's 10 MB)
    szhip_chain_pool *chain_pool = nullptr;      // the coefficient chains' persistent threads (created with the first array that has regression blocks)
};

namespace {

#define HIPCHK(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            snprintf(ctx->err, sizeof(ctx->err), "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            fprintf(stderr, "szhip: %s\n", ctx->err);                                                  \
            return SZHIP_ERR_NODEVICE;                                                                 \
        }                                                                                              \
    } while (0)

#define FAIL(code, ...)                                                                                \
    do {                                                                                               \
        snprintf(ctx->err, sizeof(ctx->err), __VA_ARGS__);                                             \
        fprintf(stderr, "szhip: %s\n", ctx->err);                                                      \
        return (code);                                                                                 \
    } while (0)
// a failure found AFTER the stream has been handed to the caller's pointer: a host copy that this call malloc'd is released again
#define FAIL_PUBLISHED(code, ...)                                                                      \
    do {                                                                                               \
        if (!out_on_device && out && *out) { free(*out); *out = nullptr; }                              \
        if (out_size) *out_size = 0;                                                                   \
        FAIL(code, __VA_ARGS__);                                                                       \
    } while (0)


int ensure(szhip_ctx *ctx, DevBuf &b, size_t bytes, bool zero_new = false)
{
    if (bytes == 0) bytes = 16;
    if (b.cap >= bytes) return SZHIP_OK;
    if (b.p) { HIPCHK(hipStreamSynchronize(ctx->stream)); HIPCHK(hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    size_t cap = bytes + bytes / 8 + 256;
    HIPCHK(hipMalloc(&b.p, cap));
    b.cap = cap;
    if (zero_new) HIPCHK(hipMemsetAsync(b.p, 0, cap, ctx->stream));
    return SZHIP_OK;
}

int ensure_pinned(szhip_ctx *ctx, size_t bytes)
{
    if (ctx->pinned_cap >= bytes) return SZHIP_OK;
    if (ctx->pinned) { HIPCHK(hipStreamSynchronize(ctx->stream)); HIPCHK(hipHostFree(ctx->pinned)); ctx->pinned = nullptr; ctx->pinned_cap = 0; }
    size_t cap = bytes + bytes / 4 + 4096;
    HIPCHK(hipHostMalloc(&ctx->pinned, cap, hipHostMallocDefault));
    ctx->pinned_cap = cap;
    return SZHIP_OK;
}

int ensure_pinned2(szhip_ctx *ctx, size_t bytes)
{
    if (ctx->pinned2_cap >= bytes) return SZHIP_OK;
    if (ctx->pinned2) { HIPCHK(hipStreamSynchronize(ctx->stream2)); HIPCHK(hipHostFree(ctx->pinned2)); ctx->pinned2 = nullptr; ctx->pinned2_cap = 0; }
    size_t cap = bytes + bytes / 4 + 4096;
    HIPCHK(hipHostMalloc(&ctx->pinned2, cap, hipHostMallocDefault));
    ctx->pinned2_cap = cap;
    return SZHIP_OK;
}

int ensure_pinned3(szhip_ctx *ctx, size_t bytes)
{
    if (ctx->pinned3_cap >= bytes) return SZHIP_OK;
    if (ctx->pinned3) { HIPCHK(hipStreamSynchronize(ctx->stream)); HIPCHK(hipHostFree(ctx->pinned3)); ctx->pinned3 = nullptr; ctx->pinned3_cap = 0; }
    size_t cap = bytes + bytes / 4 + 4096;
    HIPCHK(hipHostMalloc(&ctx->pinned3, cap, hipHostMallocDefault));
    ctx->pinned3_cap = cap;
    return SZHIP_OK;
}

int ensure_coherent(szhip_ctx *ctx, size_t bytes)
{
    if (ctx->coh_cap >= bytes) return SZHIP_OK;
    if (ctx->coh) { HIPCHK(hipStreamSynchronize(ctx->stream)); HIPCHK(hipHostFree(ctx->coh)); ctx->coh = nullptr; ctx->coh_cap = 0; }
    size_t cap = bytes + bytes / 4 + 4096;
    HIPCHK(hipHostMalloc(&ctx->coh, cap, hipHostMallocCoherent | hipHostMallocMapped));
    memset(ctx->coh, 0, cap);                          // (epoch-tagged words live here: none may look current by accident)
    ctx->coh_cap = cap;
    return SZHIP_OK;
}

#define TRY(x) do { int rc_ = (x); if (rc_ != SZHIP_OK) return rc_; } while (0)

enum { SM_MINMAX = 0, SM_WITHIN = 2, SM_MEANCNT = 3, SM_TOTAL_UNPRED = 4, SM_TOTAL_BITS = 5, SM_TICKET = 6, SM_ERR = 7,
       SM_CHANGED = 8, SM_MEANSUM = 9, SM_TOTAL_SYM = 10, SM_NREG = 11, SM_SCRATCH = 12, SM_COUNT = 16 };

// Does a copy on the second stream complete while a kernel on the first one is still running?  HIP maps streams onto a few hardware
// queues; two streams of one context can land on the same queue (seen with several contexts + torch in one process), and then anything
// queued behind a kernel that WAITS for it is a deadlock.  Probed once per context: a kernel on `stream` spins (bounded, ~4 ms) on a
// host-coherent word; a small copy goes onto `stream2`; if it completes while the kernel is still spinning the queues are independent.
int probe_streams(szhip_ctx *ctx)
{
    if (ctx->streams_independent >= 0) return SZHIP_OK;
    TRY(ensure_coherent(ctx, 256));
    TRY(ensure(ctx, ctx->small, SM_COUNT * 8));
    volatile unsigned long long *flag = (volatile unsigned long long *)((char *)ctx->coh + 128);
    unsigned long long *src = (unsigned long long *)((char *)ctx->coh + 192);
    *flag = 0; *src = 1;
    HIPCHK(hipStreamSynchronize(ctx->stream)); HIPCHK(hipStreamSynchronize(ctx->stream2));
    u64 *sm = (u64 *)ctx->small.p;
    hipLaunchKernelGGL(k_probe_wait, dim3(1), dim3(1), 0, ctx->stream, (const unsigned long long *)flag, (unsigned long long *)(sm + SM_SCRATCH));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(sm + SM_SCRATCH + 1, src, 8, hipMemcpyHostToDevice, ctx->stream2));
    const double t0 = now_ms();
    HIPCHK(hipStreamSynchronize(ctx->stream2));
    const double waited = now_ms() - t0;
    *flag = 1;                                                      // release the kernel
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->streams_independent = waited < 2.0 ? 1 : 0;
    return SZHIP_OK;
}

int tune_int(const char *name, int def);
// Do kernels on `b` run while a kernel on `a` is still running?  (As above, with a kernel instead of the copy.)  *shared = 1: no, the two
// streams sit on one hardware queue.
static int probe_pair(szhip_ctx *ctx, hipStream_t a, hipStream_t b, int *shared)
{
    TRY(ensure_coherent(ctx, 256));
    TRY(ensure(ctx, ctx->small, SM_COUNT * 8));
    volatile unsigned long long *flag = (volatile unsigned long long *)((char *)ctx->coh + 128);
    *flag = 0;
    HIPCHK(hipStreamSynchronize(a)); HIPCHK(hipStreamSynchronize(b));
    u64 *sm = (u64 *)ctx->small.p;
    hipLaunchKernelGGL(k_probe_wait, dim3(1), dim3(1), 0, a, (const unsigned long long *)flag, (unsigned long long *)(sm + SM_SCRATCH));
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_probe_touch, dim3(1), dim3(1), 0, b, (unsigned long long *)(sm + SM_SCRATCH + 1));
    HIPCHK(hipGetLastError());
    const double t0 = now_ms();
    HIPCHK(hipStreamSynchronize(b));
    const double waited = now_ms() - t0;
    *flag = 1;                                                      // release the kernel
    HIPCHK(hipStreamSynchronize(a));
    *shared = waited < 2.0 ? 0 : 1;
    return SZHIP_OK;
}

// HIP maps the streams of a process onto a few hardware queues, and two streams on one queue run their kernels one after the other.  Which
// streams share a queue is the luck of what else the process has created.  Measured (round 4, 512^3 float32, one call after the other):
// 236 - 241 GB/s when a context's second stream (the fit pass beside the sampling chain, the histogram beside the block ordering) or third stream
// (the slices' passes beside the sweep) sat on the main stream's queue, 274 - 279 otherwise; two lanes of a pool whose main streams shared a queue:
// 248 GB/s against 340.  So a new context PROBES: a kernel that waits (bounded, ~4 ms) on the one stream, a trivial kernel on the other; a side
// stream that does not get through is replaced by a freshly created one (the rejected streams stay alive until the search is over, so that the
// runtime's next choice is another queue), a few times over.  `mains`: the main streams of the pool's earlier lanes (a lane's streams must
// not share a queue with those either).  SZ_HIP_SETTLE=0: take the streams as they come.
static void settle_one(szhip_ctx *ctx, hipStream_t *victim, const std::vector<hipStream_t> &against, std::vector<hipStream_t> &rejected, int *probes)
{
    for (int attempt = 0; attempt < 6; ++attempt) {
        bool clash = false;
        for (hipStream_t a : against) {
            if (a == *victim) continue;
            int shared = 0;
            ++*probes;
            if (probe_pair(ctx, a, *victim, &shared) != SZHIP_OK) return;
            if (shared) { clash = true; break; }
        }
        if (!clash) return;
        hipStream_t fresh = nullptr;
        if (hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking) != hipSuccess) return;
        rejected.push_back(*victim);
        *victim = fresh;
    }
}
static void settle_streams(szhip_ctx *ctx, const hipStream_t *mains, int n_mains, bool third = true)
{
#ifdef SZH_SYNC_LAUNCH
    (void)ctx; (void)mains; (void)n_mains; (void)third;             // (the CPU shim runs every kernel at its launch)
#else
    if (!tune_int("SZ_HIP_SETTLE", 1)) return;
    std::vector<hipStream_t> rejected, against(mains, mains + n_mains);
    int probes = 0;
    if (n_mains) settle_one(ctx, &ctx->stream, against, rejected, &probes);      // this lane's main stream against the earlier lanes'
    against.push_back(ctx->stream);
    settle_one(ctx, &ctx->stream2, against, rejected, &probes);
    if (third && !ctx->stream3 && hipStreamCreateWithFlags(&ctx->stream3, hipStreamNonBlocking) != hipSuccess) ctx->stream3 = nullptr;
    if (third && ctx->stream3) { against.push_back(ctx->stream2); settle_one(ctx, &ctx->stream3, against, rejected, &probes); }
    ctx->settle_probes = probes; ctx->settle_rejected = (int)rejected.size();
    for (hipStream_t r : rejected) hipStreamDestroy(r);
    if (tune_int("SZ_HIP_SETTLE_LOG", 0)) fprintf(stderr, "szhip: streams settled after %d probes, %d streams replaced\n", probes, (int)rejected.size());
#endif
}

// ---- bulk copies between pageable host memory and the device.
// hipMemcpyAsync on pageable memory makes the runtime pin and unpin the caller's pages around the transfer: measured on this pool,
// 512 MiB host-to-device ran at 11-25 GB/s depending on the box, device-to-host into a fresh malloc at 3.4 GB/s, and the deferred
// unpinning stalled LATER calls (see the coefficient transfers in compress_impl).  Here the array is cut into 8 MiB chunks; SZH_STAGE_T
// host threads each own two pinned buffers and take every SZH_STAGE_T-th chunk: memcpy into (out of) a pinned buffer, asynchronous DMA on
// the context's stream, the other buffer meanwhile.  The call returns when the last byte has arrived.
int tune_int(const char *name, int def);
constexpr int SZH_STAGE_T = 4, SZH_STAGE_TMAX = 8;     // default / most threads (SZ_HIP_STAGE_THREADS)
constexpr size_t SZH_STAGE_CHUNK = 8u << 20;
int staged_copy(szhip_ctx *ctx, void *dst, const void *src, size_t bytes, bool to_device)
{
    hipStream_t st = ctx->stream;
    // (SZ_HIP_STAGE_CHUNK_KB: smaller chunks, so that tests reach this path with small arrays)
    const size_t chunk = std::min(SZH_STAGE_CHUNK, (size_t)std::max(1, tune_int("SZ_HIP_STAGE_CHUNK_KB", (int)(SZH_STAGE_CHUNK >> 10))) << 10);
    if (bytes < 4 * chunk || tune_int("SZ_HIP_STAGED_COPY", 1) == 0) {
        HIPCHK(hipMemcpyAsync(dst, src, bytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        return SZHIP_OK;
    }
    const int nthr = std::min(SZH_STAGE_TMAX, std::max(1, tune_int("SZ_HIP_STAGE_THREADS", SZH_STAGE_T)));
    for (int w = 0; w < nthr; ++w)
        for (int k = 0; k < 2; ++k)
            if (!ctx->stage_buf[w][k]) {
                HIPCHK(hipHostMalloc(&ctx->stage_buf[w][k], SZH_STAGE_CHUNK, hipHostMallocDefault));
                HIPCHK(hipEventCreateWithFlags(&ctx->stage_ev[w][k], hipEventDisableTiming));
            }
    HIPCHK(hipStreamSynchronize(st));                              // the source (destination) is ready (free) from here on
    const size_t nchunks = (bytes + chunk - 1) / chunk;
    std::atomic<int> failed(0);
    const int device = ctx->device;
    auto worker = [&](int w) {
        if (hipSetDevice(device) != hipSuccess) { failed = 1; return; }
        size_t pend_off[2] = {0, 0}, pend_len[2] = {0, 0};        // device-to-host: the chunk in flight into buffer k
        bool pend[2] = {false, false};
        int k = 0;
        for (size_t c = (size_t)w; c < nchunks && !failed; c += (size_t)nthr, k ^= 1) {
            const size_t off = c * chunk, len = std::min(chunk, bytes - off);
            if (to_device) {
                if (pend[k] && hipEventSynchronize(ctx->stage_ev[w][k]) != hipSuccess) { failed = 1; return; }   // buffer k is free again
                memcpy(ctx->stage_buf[w][k], (const char *)src + off, len);
                if (hipMemcpyAsync((char *)dst + off, ctx->stage_buf[w][k], len, hipMemcpyHostToDevice, st) != hipSuccess ||
                    hipEventRecord(ctx->stage_ev[w][k], st) != hipSuccess) { failed = 1; return; }
                pend[k] = true;
            } else {
                if (pend[k]) {                                     // the chunk that went into buffer k two rounds ago: hand it to the caller
                    if (hipEventSynchronize(ctx->stage_ev[w][k]) != hipSuccess) { failed = 1; return; }
                    memcpy((char *)dst + pend_off[k], ctx->stage_buf[w][k], pend_len[k]);
                }
                if (hipMemcpyAsync(ctx->stage_buf[w][k], (const char *)src + off, len, hipMemcpyDeviceToHost, st) != hipSuccess ||
                    hipEventRecord(ctx->stage_ev[w][k], st) != hipSuccess) { failed = 1; return; }
                pend[k] = true; pend_off[k] = off; pend_len[k] = len;
            }
        }
        for (int q = 0; q < 2; ++q, k ^= 1) {                      // drain, oldest first
            if (!pend[k]) continue;
            if (hipEventSynchronize(ctx->stage_ev[w][k]) != hipSuccess) { failed = 1; return; }
            if (!to_device) memcpy((char *)dst + pend_off[k], ctx->stage_buf[w][k], pend_len[k]);
        }
    };
    std::vector<std::thread> th;
    for (int w = 1; w < nthr; ++w) th.emplace_back(worker, w);
    worker(0);
    for (auto &t : th) t.join();
    if (failed) FAIL(SZHIP_ERR_NODEVICE, "staged %s copy failed", to_device ? "host-to-device" : "device-to-host");
    return SZHIP_OK;
}

// launch tuning knobs (development): environment overrides of the wavefront kernel's wait parameters
int tune_int(const char *name, int def)
{
    const char *e = getenv(name);
    return e ? atoi(e) : def;
}

// device-wide exclusive scan of u64 in[0..n) -> out; total to *total_dev (device u64)
int scan_u64(szhip_ctx *ctx, const u64 *in, int64_t n, u64 *out, u64 *total_dev, hipStream_t on = nullptr)
{
    const hipStream_t sst = on ? on : ctx->stream;
    const int64_t nblk = (n + SZH_SCAN_TILE - 1) / SZH_SCAN_TILE;
    TRY(ensure(ctx, ctx->partial, (size_t)(nblk + 1) * 8));
    u64 *partial = (u64 *)ctx->partial.p;
    hipLaunchKernelGGL(k_scan_partials, dim3((unsigned)nblk), dim3(256), 0, sst, in, n, partial);
    hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(256), 0, sst, partial, nblk, total_dev);
    hipLaunchKernelGGL(k_scan_final, dim3((unsigned)nblk), dim3(256), 0, sst, in, n, (const u64 *)partial, out);
    HIPCHK(hipGetLastError());
    return SZHIP_OK;
}

// layout of the "small" device scratch (u64 slots)

int choose_segb(const szh_geom3 &G, size_t elem, size_t budget)
{
    const size_t rows = (size_t)G.g0.early * G.g1.early;
    size_t per_block = rows * (size_t)G.g2.early * elem;
    int segb = (int)(budget / (per_block ? per_block : 1));
    if (segb < 1) segb = 1;
    const int cap = std::max(1, tune_int("SZ_HIP_PERM_SEGB_MAX", 32));
    if (segb > cap) segb = cap;
    if (segb > G.g2.num) segb = G.g2.num;
    return segb;
}
// tile of k_permute: rows widened to 16-byte vector boundaries, pitch a multiple of 8 elements
size_t tile_bytes(const szh_geom3 &G, int segb, size_t elem)
{
    const size_t rows = (size_t)G.g0.early * G.g1.early;
    const size_t kp = ((size_t)segb * G.g2.early + 31) & ~(size_t)7;
    return rows * kp * elem + 16;
}

// buffers of the wavefront kernel: granule faces, progress words, start order of the tiles (tpi x tpj pencils each)
int prepare_pencil(szhip_ctx *ctx, const szh_geom3 &G, int nw, int tpi, int tpj, int *nI_out, int *nJ_out, int *ntiles_out)
{
    const int nI = (G.g0.count + 7) / 8, nJ = (G.g1.count + 7) / 8;
    if (nI > 65535 || nJ > 65535) FAIL(SZHIP_ERR_UNSUP, "dimension too large for the pencil grid");
    const size_t rowg = (size_t)G.g2.count * nw * sizeof(u64);
    TRY(ensure(ctx, ctx->faceI, (size_t)nI * nJ * 9 * rowg + 64, true));   // + 64: a 16-byte granule pair may reach one granule past a row's end
    TRY(ensure(ctx, ctx->faceJ, (size_t)nI * nJ * 8 * rowg + 64, true));
    TRY(ensure(ctx, ctx->progress, (size_t)nI * nJ * 2 * sizeof(u64), true));
    if (tune_int("SZ_HIP_TRACE", 0)) TRY(ensure(ctx, ctx->trace, ((size_t)nI * nJ * 8 + 256 + 4 * 2 * SZH_TRACE_LOG) * sizeof(u64), true));
    const int nTI = (nI + tpi - 1) / tpi, nTJ = (nJ + tpj - 1) / tpj;
    if (ctx->order_nI != nTI || ctx->order_nJ != nTJ) {
        std::vector<unsigned> ord((size_t)nTI * nTJ);
        szh_fill_pencil_order(nTI, nTJ, ord.data());
        TRY(ensure(ctx, ctx->order, ord.size() * 4));
        HIPCHK(hipMemcpyAsync(ctx->order.p, ord.data(), ord.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        ctx->order_nI = nTI; ctx->order_nJ = nTJ;
    }
    *nI_out = nI; *nJ_out = nJ; *ntiles_out = nTI * nTJ;
    return SZHIP_OK;
}

// ---- the ribbon mapping of the wavefront kernel (szh_ribbon.h): 3-D arrays whose block map is Lorenzo-only
// does it cover this call?  (SZ_HIP_RIBBON=0 sends everything to k_pencil)
template <class T> bool ribbon_applies(const szh_geom3 &G, size_t reg_count)
{
    using RS = szh_rb_shape<T>;
    if (!tune_int("SZ_HIP_RIBBON", 1)) return false;
    if (G.ndim != 3 || reg_count != 0) return false;
    if ((double)RS::W * RS::R * (double)G.d0 * sizeof(T) >= 2.0e9) return false;      // a tile's rows are addressed by 32-bit buffer offsets
    const int nTI = (G.g0.count + RS::W * RS::R - 1) / (RS::W * RS::R), nTJ = (G.g1.count + 63) / 64;
    return nTI <= 65535 && nTJ <= 65535;
}
// granule rows of the tile hand-offs + the launch; `a` carries everything that does not depend on the mapping
// k_pencil's grid: one workgroup per tile (a lone context: the inverse sweep at 512^3 takes 1.47 ms either way), or that many persistent
// workgroups drawing tiles from the ticket counter (a pool lane: fewer workgroups that only poll for their predecessors -- two M-field
// arrays in flight 111 -> 136 GB/s with 512, 127 with 256).  SZ_HIP_PENCIL_WGS overrides (0 = one per tile).
template <class QA> static unsigned pencil_grid(szhip_ctx *ctx, QA &a, int ntiles)
{
    const int cap = tune_int("SZ_HIP_PENCIL_WGS", ctx->gate ? 512 : 0);
    a.persist = cap > 0 && cap < ntiles;
    return (unsigned)(a.persist ? cap : ntiles);
}
template <class T, bool DEC>
int launch_ribbon(szhip_ctx *ctx, const szh_geom3 &G, szh_qargs<T> a, hipStream_t st)
{
    using RS = szh_rb_shape<T>;
    constexpr int WR = RS::W * RS::R, NW = szh_gran<T>::NW;
    const int nTI = (G.g0.count + WR - 1) / WR, nTJ = (G.g1.count + 63) / 64;
    const size_t NT = (size_t)szh_rb_steps_of<T>(G.g2.count), tiles = (size_t)nTI * nTJ;
    TRY(ensure(ctx, ctx->rb_down, tiles * NT * NW * 64 * sizeof(u64) + 64, true));
    TRY(ensure(ctx, ctx->rb_right, tiles * NT * NW * WR * sizeof(u64) + 64, true));
    a.faceI = (szh_u64 *)ctx->rb_down.p; a.faceJ = (szh_u64 *)ctx->rb_right.p;
    a.nI = nTI; a.nJ = nTJ;
    if (a.ticket_mode == 2) a.ticket_mode = 1;          // (the tile is always computed from the ticket here; 0 = atomic ticket)
    // Persistent workgroups with a FIXED share of the tickets (mode 1) need the whole grid resident at once.  A lone context has the
    // chip to itself; the lanes of a pool compete for CUs, and workgroups of different launches that wait for tiles of their own,
    // not yet resident, workgroups could block each other.  A pool lane therefore draws its tickets from the launch's counter: a
    // workgroup holds a ticket only while it runs the tile, so the smallest unfinished ticket always belongs to a running workgroup.
    if (ctx->gate && tune_int("SZ_HIP_RB_POOL_ATOMIC", 1)) a.ticket_mode = 0;
    // persistent workgroups (k_ribbon): no more than one per CU, or the ticket order could wait for a workgroup that is not resident.
    // A lone context takes every CU (512^3: sweep 1.02 ms with 256 workgroups, 1.18 with 96); a lane of a pool leaves half of them to
    // the other lanes' kernels (two arrays in flight, 40-step runs: 324 GB/s with 256, 351 - 354 with 128 or 112, 346 with 96, 326 with 64;
    // four in flight, round 4: 377 - 388 with 128, 391 - 395 with 96, 329 - 417 with 64: three eighths of the CUs from three lanes on)
    // (the device's own CU count, not 256: on a smaller or partitioned GPU workgroups beyond it would not be resident and every call would
    //  run into the wait bound before the atomic-ticket repetition took over)
#ifdef SZH_SYNC_LAUNCH
    const unsigned wgs = (unsigned)tiles;      // (the CPU shim runs workgroups one after the other: a workgroup per tile, or the first would wait for tiles of the second)
#else
    const unsigned wgs = (unsigned)std::min<size_t>(tiles, (size_t)std::max(1, std::min(tune_int("SZ_HIP_RB_WGS", ctx->gate ? (ctx->gate->lanes <= 2 ? ctx->cus / 2 : ctx->cus * 3 / 8) : ctx->cus), ctx->cus)));
#endif
    if (a.use_mean) hipLaunchKernelGGL((k_ribbon<T, DEC, true>), dim3(wgs), dim3((RS::W + 3 + (DEC ? 1 : 0)) * 64), 0, st, a);
    else hipLaunchKernelGGL((k_ribbon<T, DEC, false>), dim3(wgs), dim3((RS::W + 3 + (DEC ? 1 : 0)) * 64), 0, st, a);
    HIPCHK(hipGetLastError());
    return SZHIP_OK;
}

// ---- the beam mapping of the sweep (szh_beam.h, round 5): 3-D arrays whose contiguous extent is a multiple of 4 values
// does it cover this call?  (SZ_HIP_BEAM=0 sends everything to k_ribbon / k_pencil)
template <class T> bool beam_applies(const szh_geom3 &G, const void *base, size_t reg_count)
{
    if (!tune_int("SZ_HIP_BEAM", 1)) return false;
    (void)reg_count;                                                                               // (regression blocks: k_reg_points beside the sweep)
    if (G.ndim != 3) return false;
    if (G.g2.count < 4 || (G.g2.count & 3) != 0 || ((uintptr_t)base & 15) != 0) return false;      // 16-byte row pieces
    if ((double)G.n * sizeof(T) >= 4.0e9) return false;                                            // 32-bit buffer offsets
    const szh_bm::grid_t g = szh_bm::make_grid(G);
    if (g.nKB > 65535 || g.nJG > 65535) return false;
    return (double)szh_bm::kface_words<T>(G) * 8.0 < 4.0e9 && (double)szh_bm::jface_words<T>(G) * 8.0 < 4.0e9;
}
// granule buffers of the beams' faces + the launch; `a` carries everything that does not depend on the mapping
template <class T, bool DEC>
int launch_beam(szhip_ctx *ctx, const szh_geom3 &G, szh_qargs<T> a, hipStream_t st, size_t reg_count)
{
    const szh_bm::grid_t g = szh_bm::make_grid(G);
    const size_t tiles = (size_t)g.nKB * g.nJG;
    TRY(ensure(ctx, ctx->rb_down, szh_bm::kface_words<T>(G) * sizeof(u64) + 64, true));
    TRY(ensure(ctx, ctx->rb_right, szh_bm::jface_words<T>(G) * sizeof(u64) + 64, true));
    a.faceI = (szh_u64 *)ctx->rb_down.p; a.faceJ = (szh_u64 *)ctx->rb_right.p;
    a.nI = g.nKB; a.nJ = g.nJG;
    if (a.ticket_mode == 2) a.ticket_mode = 1;
    if (ctx->gate && tune_int("SZ_HIP_RB_POOL_ATOMIC", 1)) a.ticket_mode = 0;
#ifdef SZH_SYNC_LAUNCH
    const unsigned wgs = (unsigned)tiles;      // (the CPU shim runs workgroups one after the other: a workgroup per tile, in ticket order)
#else
    // persistent workgroups in ticket order: as many as are resident at once (a tile's predecessors hold smaller tickets)
    const int per_cu = std::max(1, tune_int("SZ_HIP_BEAM_WG_PER_CU", sizeof(T) == 4 ? 2 : 1));
    const unsigned wgs = (unsigned)std::min<size_t>(tiles, (size_t)std::max(1, tune_int("SZ_HIP_BEAM_WGS", ctx->cus * per_cu)));
#endif
    // arrays with regression blocks: their points are quantised / reconstructed by k_reg_points (no neighbour involved); the sweep takes
    // their reconstructions as its neighbours (compress: from a.xr; the inverse: they are in the output array already) and passes them through
    const bool hasreg = reg_count != 0;
    if (hasreg) {
        TRY(ensure(ctx, ctx->pt_flags, (size_t)G.n + 64));
        HIPCHK(hipMemsetAsync(ctx->pt_flags.p, 0, (size_t)G.n, st));
        a.ptflags = (const uint8_t *)ctx->pt_flags.p;
        const unsigned rgrid = (unsigned)(G.g0.num * G.g1.num);       // a workgroup per block column (b0, b1)
        if (!DEC) {
            TRY(ensure(ctx, ctx->rb_vals, (size_t)G.n * sizeof(T) + 64));
            a.xr = (const T *)ctx->rb_vals.p;
            hipLaunchKernelGGL((k_reg_points<T, 0>), dim3(rgrid), dim3(256), 0, st, G, a.blk_lor, a.coef, a.coef_stride, a.data, (T *)ctx->rb_vals.p, a.codes, (uint8_t *)ctx->pt_flags.p,
                               a.eb, a.recip, a.cap, a.radius);
        } else
            hipLaunchKernelGGL((k_reg_points<T, 2>), dim3(rgrid), dim3(256), 0, st, G, a.blk_lor, a.coef, a.coef_stride, (const T *)nullptr, a.out, a.codes, (uint8_t *)ctx->pt_flags.p,
                               a.eb, a.recip, a.cap, a.radius);
        HIPCHK(hipGetLastError());
    }
    const dim3 grid(wgs), block(szh_bm::WPG * 64);
    if (hasreg) {
        if (a.use_mean) hipLaunchKernelGGL((k_beam<T, DEC, true, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((k_beam<T, DEC, false, true>), grid, block, 0, st, a);
    } else {
        if (a.use_mean) hipLaunchKernelGGL((k_beam<T, DEC, true, false>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((k_beam<T, DEC, false, false>), grid, block, 0, st, a);
    }
    HIPCHK(hipGetLastError());
    return SZHIP_OK;
}

template <class T> double ord_dec(u64 e);
template <> double ord_dec<float>(u64 e)
{
    unsigned u = (unsigned)e;
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    float f; memcpy(&f, &u, 4); return (double)f;
}
template <> double ord_dec<double>(u64 e)
{
    u64 u = (e & 0x8000000000000000ull) ? (e & 0x7fffffffffffffffull) : ~e;
    double d; memcpy(&d, &u, 8); return d;
}

template <class T>
int minmax_impl(szhip_ctx *ctx, const void *data, int on_dev, size_t n, double *vmin, double *vmax)
{
    const T *d_in = (const T *)data;
    if (!on_dev) {
        TRY(ensure(ctx, ctx->in, n * sizeof(T)));
        HIPCHK(hipMemcpyAsync(ctx->in.p, data, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
        d_in = (const T *)ctx->in.p;
    }
    TRY(ensure(ctx, ctx->small, SM_COUNT * 8));
    u64 *sm = (u64 *)ctx->small.p;
    u64 init[2] = {~0ull, 0ull};
    HIPCHK(hipMemcpyAsync(sm + SM_MINMAX, init, 16, hipMemcpyHostToDevice, ctx->stream));
    int grid = (int)std::min<int64_t>(((int64_t)n + 255) / 256, 2048);
    hipLaunchKernelGGL((k_minmax<T>), dim3(grid), dim3(256), 0, ctx->stream, d_in, (int64_t)n, sm + SM_MINMAX);
    HIPCHK(hipGetLastError());
    u64 res[2];
    HIPCHK(hipMemcpyAsync(res, sm + SM_MINMAX, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    *vmin = ord_dec<T>(res[0]); *vmax = ord_dec<T>(res[1]);
    return SZHIP_OK;
}

template <class T>
int compress_impl(szhip_ctx *ctx, const void *data, int data_on_device, size_t r0, size_t r1, size_t r2, double eb_in,
                  const szhip_params *prm, const unsigned char *meta, size_t meta_len, int out_on_device,
                  unsigned char **out, size_t *out_size, szhip_stats *stats)
{
    const int is_double = sizeof(T) == 8;
    const bool two_d = r0 == 0;                                // r0 == 0: a 2-D array r1 x r2 (sz_float.c:5516)
    const szh_geom3 G = two_d ? szh_make_geom2((int)r1, (int)r2) : szh_make_geom3((int)r0, (int)r1, (int)r2);
    const int ncoef = two_d ? 3 : 4;
    const int64_t n = G.n, nb = G.nblocks;
    const T eb = (T)eb_in;
    const double t_begin = now_ms();
    double host_ms = 0;
    const int tp_on = tune_int("SZ_HIP_TIMING", 0); double tp_t[32]; const char *tp_n[32]; int tp_k = 0;
    auto TP = [&](const char *nm) { if (tp_on && tp_k < 32) { tp_t[tp_k] = now_ms() - t_begin; tp_n[tp_k++] = nm; } };
    hipStream_t st = ctx->stream;
    szhip_stats S; memset(&S, 0, sizeof(S));
    S.n_elements = (uint64_t)n; S.n_blocks = (uint64_t)nb;

    const T *d_in = (const T *)data;
    if (!data_on_device) {
        TRY(ensure(ctx, ctx->in, (size_t)n * sizeof(T)));
        TRY(staged_copy(ctx, ctx->in.p, data, (size_t)n * sizeof(T), true));
        d_in = (const T *)ctx->in.p;
    }
    TRY(ensure(ctx, ctx->small, SM_COUNT * 8));
    u64 *sm = (u64 *)ctx->small.p;
    HIPCHK(hipMemsetAsync(sm, 0, SM_COUNT * 8, st));
    static const u64 minmax_init[2] = {~0ull, 0ull};          // ordered encodings: the fit pass reduces the array's range into these
    HIPCHK(hipMemcpyAsync(sm + SM_MINMAX, minmax_init, 16, hipMemcpyHostToDevice, st));
    const bool range_from_data = (prm->flags & SZHIP_RANGE_FROM_DATA) != 0;
    TRY(ensure(ctx, ctx->coef, (size_t)nb * 4 * sizeof(T)));
    TRY(ensure(ctx, ctx->blk_lor, (size_t)nb));
    T *d_coef = (T *)ctx->coef.p;
    uint8_t *d_lor = (uint8_t *)ctx->blk_lor.p;
    HIPCHK(hipEventRecord(ctx->ev[0], st));
    TP("ev0");

    const int ncols = G.g0.num * G.g1.num;

    // ---- regression fit + predictor selection on the second stream, overlapped with the interval optimiser below.  The pass
    //      needs only the bound -- except for the mean shortcut of the selection, which the optimiser may switch on; it is run
    //      WITHOUT it here and repeated in the (rare) use_mean case.
    const T noise = (T)((double)eb * (two_d ? 0.81 : 1.22));   // sz_float.c:7056 / :5672
    HIPCHK(hipEventRecord(ctx->ev_in, st));                    // input staged, scratch cleared
    HIPCHK(hipStreamWaitEvent(ctx->stream2, ctx->ev_in, 0));
    hipLaunchKernelGGL((k_fit_select<T>), dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, ctx->stream2,
                       G, d_in, d_coef, d_lor, noise, 0, (T)0, sm + SM_MINMAX);
    HIPCHK(hipGetLastError());
    // the stream's indicator bit array and the regression-block count come from the device (no per-block host loop); they follow the
    // speculative pass on the second stream, so that they are on the host by the time the interval decision is made
    const size_t ind_bytes = ((size_t)nb + 7) / 8;
    TRY(ensure(ctx, ctx->lor_bits, ind_bytes + 8));
    TRY(ensure_pinned2(ctx, ind_bytes + 32));                   // pinned: the copies below must not block this thread
    unsigned char *const ind_bits = (unsigned char *)ctx->pinned2 + 32;
    u64 *const nreg_h = (u64 *)ctx->pinned2;
    u64 *const minmax_h = (u64 *)ctx->pinned2 + 2;
    hipLaunchKernelGGL(k_pack_lor, dim3((unsigned)((ind_bytes + 255) / 256)), dim3(256), 0, ctx->stream2, (const uint8_t *)d_lor, nb,
                       (uint8_t *)ctx->lor_bits.p, sm + SM_NREG);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(ind_bits, ctx->lor_bits.p, ind_bytes, hipMemcpyDeviceToHost, ctx->stream2));
    HIPCHK(hipMemcpyAsync(nreg_h, sm + SM_NREG, 8, hipMemcpyDeviceToHost, ctx->stream2));
    HIPCHK(hipMemcpyAsync(minmax_h, sm + SM_MINMAX, 16, hipMemcpyDeviceToHost, ctx->stream2));
    HIPCHK(hipEventRecord(ctx->ev_fit, ctx->stream2));

    // ---- interval optimiser
    unsigned intervals = prm->quantization_intervals;
    int use_mean = 0; T mean = 0;
    if (intervals == 0) {
        const unsigned max_radius = prm->max_quant_intervals / 2;
        const int64_t md = (int64_t)(int)std::sqrt((double)n);
        // 2-D: a plain stride (sz_float.c:5412-5417): no step-backs
        const szh_meanwalk w = two_d ? szh_make_meanwalk(n, INT64_MAX / 2, INT64_MAX / 2, md) : szh_make_meanwalk(n, G.d0, G.g2.count, md);
        int64_t M = 0; // number of strided samples: first m with pos >= n (positions are increasing)
        {
            int64_t lo = 0, hi = n / std::max<int64_t>(md - 2, 1) + 2;
            while (szh_meanwalk_pos(w, hi) < n) hi *= 2;
            while (lo < hi) { int64_t mid = (lo + hi) / 2; if (szh_meanwalk_pos(w, mid) >= n) hi = mid; else lo = mid + 1; }
            M = lo;
        }
        TRY(ensure(ctx, ctx->samples, (size_t)M * sizeof(T)));
        TRY(ensure_pinned(ctx, std::max<size_t>((size_t)M * sizeof(T), (size_t)(max_radius + 8192) * 4 + 64)));
        hipLaunchKernelGGL((k_gather_mean<T>), dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, d_in, w, M, (T *)ctx->samples.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(ctx->pinned, ctx->samples.p, (size_t)M * sizeof(T), hipMemcpyDeviceToHost, st));
        TP("mean enqueued");
        HIPCHK(hipStreamSynchronize(st));
        TP("mean on host");
        double h0 = now_ms();
        const double smean = szhost_seq_mean(is_double, ctx->pinned, (size_t)M);
        host_ms += now_ms() - h0;

        TRY(ensure(ctx, ctx->hist, (size_t)(max_radius + 8192) * 4 + 64));
        unsigned *d_rh = (unsigned *)ctx->hist.p, *d_fh = d_rh + max_radius;
        HIPCHK(hipMemsetAsync(d_rh, 0, (size_t)(max_radius + 8192) * 4, st));
        const int64_t nrows = szh_sample_row_limit(G, prm->sample_distance);
        if (!two_d && (G.g0.count <= 1 || G.g1.count <= 1)) {      // a degenerate 3-D array: the reference's walk, literally (k_sample_walk)
            hipLaunchKernelGGL((k_sample_walk<T, true>), dim3(1), dim3(64), 0, st, G, d_in, prm->sample_distance, (double)eb, (T)smean, max_radius, d_rh, d_fh, sm + SM_WITHIN);
            HIPCHK(hipGetLastError());
        } else if (nrows > 0) {
            int grid = (int)std::min<int64_t>((nrows + 255) / 256, 1024);
            hipLaunchKernelGGL((k_sample<T, true>), dim3(grid), dim3(256), 0, st, G, d_in, nrows, prm->sample_distance, (double)eb,
                               (T)smean, max_radius, d_rh, d_fh, sm + SM_WITHIN);
            HIPCHK(hipGetLastError());
        }
        unsigned *h_hist = (unsigned *)ctx->pinned;
        u64 within = 0;
        HIPCHK(hipMemcpyAsync(h_hist, d_rh, (size_t)(max_radius + 8192) * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(&within, sm + SM_WITHIN, 8, hipMemcpyDeviceToHost, st));
        TP("sample enqueued");
        HIPCHK(hipStreamSynchronize(st));
        TP("sample on host");
        h0 = now_ms();
        u64 sample_count = 0;
        for (unsigned i = 0; i < max_radius; ++i) sample_count += h_hist[i];
        szhost_decision dec;
        szhost_decide(is_double, h_hist, max_radius, h_hist + max_radius, sample_count, within, prm->pred_threshold,
                      (double)eb, smean, &dec);
        host_ms += now_ms() - h0;
        intervals = dec.intervals; use_mean = two_d ? 0 : dec.use_mean;   // 2-D: `use_mean = 0`, sz_float.c:5615
        if (use_mean) {
            T *d_sum = (T *)(sm + SM_MEANSUM);
            hipLaunchKernelGGL((k_mean_seq<T>), dim3(1), dim3(64), 0, st, d_in, n, (T)dec.dense_pos, eb, d_sum, sm + SM_MEANCNT);
            HIPCHK(hipGetLastError());
            T hsum = 0; u64 hcnt = 0;
            HIPCHK(hipMemcpyAsync(&hsum, d_sum, sizeof(T), hipMemcpyDeviceToHost, st));
            HIPCHK(hipMemcpyAsync(&hcnt, sm + SM_MEANCNT, 8, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            if (hcnt > 0) mean = hsum / (T)hcnt; // `mean = sum / mean_count`, sz_float.c:6668
        }
    }
    if (intervals > 65536 || intervals < 4) FAIL(SZHIP_ERR_UNSUP, "quantization interval count %u outside [4,65536]", intervals);
    S.intervals = intervals; S.use_mean = use_mean;

    HIPCHK(hipStreamWaitEvent(st, ctx->ev_fit, 0));           // join the fit + selection pass
    if (use_mean) {                                            // the selection depends on the mean after all: repeat the pass
        hipLaunchKernelGGL((k_fit_select<T>), dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st,
                           G, d_in, d_coef, d_lor, noise, use_mean, mean, sm + SM_MINMAX);
        HIPCHK(hipGetLastError());
    }
    if (use_mean) {                                            // (the repeated pass: its indicator bits replace the speculative ones)
        HIPCHK(hipMemsetAsync(sm + SM_NREG, 0, 8, st));
        hipLaunchKernelGGL(k_pack_lor, dim3((unsigned)((ind_bytes + 255) / 256)), dim3(256), 0, st, (const uint8_t *)d_lor, nb,
                           (uint8_t *)ctx->lor_bits.p, sm + SM_NREG);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(ind_bits, ctx->lor_bits.p, ind_bytes, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(nreg_h, sm + SM_NREG, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    } else HIPCHK(hipEventSynchronize(ctx->ev_fit));           // usually long done
    TP("fit joined");
    const size_t reg_count = (size_t)*nreg_h;
    S.n_reg_blocks = reg_count;
    if (reg_count == 0 && ctx->chain_pool) ctx->chain_pool->disarm();
    // the parameter bytes as they go into the stream: with SZHIP_RANGE_FROM_DATA the range field comes from the fit pass
    // (computeRangeSize_float + `max = min + valueRangeSize`, sz_float.c:2845-2849, in the data's type)
    std::vector<unsigned char> meta_own(meta, meta + meta_len);
    if (range_from_data) {
        if (meta_len < 4 + 20 + 2 * sizeof(T)) FAIL(SZHIP_ERR_ARG, "parameter bytes too short for a range field");
        const T lo = (T)ord_dec<T>(minmax_h[0]), hi = (T)ord_dec<T>(minmax_h[1]);
        const T range = hi - lo, top = lo + range;
        unsigned char *q = meta_own.data() + 4 + 20;
        if (is_double) { szhost_put_f64be(q, (double)lo); szhost_put_f64be(q + 8, (double)top); }
        else { szhost_put_f32be(q, (float)lo); szhost_put_f32be(q + 4, (float)top); }
        S.vmin = (double)lo; S.vmax = (double)hi;
        if ((double)range <= eb_in) {          // constant within the bound: the caller's business (a full compression would be thrown away)
            HIPCHK(hipStreamSynchronize(ctx->stream2)); HIPCHK(hipStreamSynchronize(st));
            *out = nullptr; *out_size = 0;
            S.ms_total = now_ms() - t_begin;
            if (stats) *stats = S;
            return SZHIP_CONSTANT;
        }
    }
    meta = meta_own.data();
    // ---- regression coefficient chain (a serial recurrence with reconstruction feedback: host) and its Huffman streams.
    //      The chains of the four (three) coefficients are independent of each other and each is bound by the latency of its own
    //      ~45-cycle dependence per block, so they run on one host thread each; a thread goes on to build its coefficient's section
    //      of the stream header (histogram, tree, payload) while the wavefront kernel is already running on the decoded values.
    szhost_coeffs cf; memset(&cf, 0, sizeof(cf));
    std::vector<unsigned char> coef_sections;
    std::vector<unsigned char> section[4];
    std::vector<std::thread> section_threads;
    std::atomic<int> section_failed(0);
    std::vector<unsigned char> all_reg_keep;   // "every block is a regression block" for the chain over the compacted coefficients
    std::vector<uint32_t> blk_of_rank;
    size_t chain_done[4] = {0, 0, 0, 0};       // regression blocks finished per coefficient (written by the chain threads)
    double chain_t0[4] = {0, 0, 0, 0}, chain_t1[4] = {0, 0, 0, 0};   // (SZ_HIP_TIMING: when each chain thread started / finished its chain)
    bool overlap = false;
    T *hcoef = nullptr;   // pinned: an asynchronous copy to or from pageable memory makes the runtime pin and unpin the pages around it,
                          // which was seen to stall later calls for ~20 ms
    auto make_section = [&](int e) {   // (lives as long as the threads that call it: declared in the function's scope)
        std::vector<uint32_t> h32(65536, 0);
        for (size_t i = 0; i < reg_count; ++i) h32[(size_t)cf.codes[e][i]]++;
        szhost_huff *ch = szhost_huff_build(131072, h32.data(), nullptr, 65536);
        if (!ch) { section_failed = 1; return; }
        const size_t tb = szhost_huff_tree_size(ch);
        const size_t enc_cap = (size_t)((ch->total_bits + 7) / 8) + 16;
        std::vector<unsigned char> &sec = section[e];
        sec.assign(sizeof(T) + 12 + tb + 8 + enc_cap + 4 + cf.unpred_count[e] * sizeof(T), 0);
        unsigned char *q = sec.data();
        if (is_double) szhost_put_f64be(q, cf.prec[e]); else szhost_put_f32be(q, (float)cf.prec[e]);
        q += sizeof(T);
        szhost_put_u32be(q, 32768); q += 4;
        szhost_put_u32be(q, (uint32_t)tb); q += 4;
        szhost_put_u32be(q, (uint32_t)ch->n_nodes); q += 4;
        szhost_huff_tree_write(ch, q); q += tb;
        const size_t enc = szhost_huff_encode_i32(ch, cf.codes[e], reg_count, q + 8);
        szhost_put_u64be(q, enc); q += 8 + enc;
        szhost_put_u32be(q, (uint32_t)cf.unpred_count[e]); q += 4;
        memcpy(q, cf.unpred[e], cf.unpred_count[e] * sizeof(T)); q += cf.unpred_count[e] * sizeof(T);
        sec.resize((size_t)(q - sec.data()));
        szhost_huff_free(ch);
    };
    bool pool_busy = false;                     // a job of this call is on the context's chain workers
    struct JoinSections {   // no early return may leave the section threads (or the pool's workers) running on this frame's data
        std::vector<std::thread> &t; szhost_coeffs &c; szhip_ctx *ctx; bool &busy;
        bool own = false;           // the output arrays belong to the context
        ~JoinSections()
        {
            for (auto &x : t) if (x.joinable()) x.join();
            if (busy && ctx->chain_pool) ctx->chain_pool->wait_all();
            if (ctx->chain_pool) ctx->chain_pool->disarm();
            if (own) for (int e = 0; e < 4; ++e) { c.codes[e] = nullptr; c.unpred[e] = nullptr; }
            szhost_coeffs_free(&c);
        }
    } join_sections{section_threads, cf, ctx, pool_busy};
    if (ctx->chain_pool) ctx->chain_pool->arm();       // (the workers wake up now and spin until the coefficients are there -- or are sent back to sleep below)
    if (reg_count > 0) {
        // only the regression blocks' coefficients travel: rank them in scan order, gather [4][reg_count], chain on the host
        // (the compact arrays are "all regression blocks" to the chain), scatter the decoded values back
        TRY(ensure(ctx, ctx->reg_flags, (size_t)nb * 8));
        TRY(ensure(ctx, ctx->reg_rank, (size_t)nb * 8));
        TRY(ensure(ctx, ctx->coef_compact, reg_count * 4 * sizeof(T)));
        hipLaunchKernelGGL(k_reg_flags, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, (const uint8_t *)d_lor, nb, (u64 *)ctx->reg_flags.p);
        TRY(scan_u64(ctx, (const u64 *)ctx->reg_flags.p, nb, (u64 *)ctx->reg_rank.p, sm + SM_SCRATCH));
        hipLaunchKernelGGL((k_move_coef<T, 0>), dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, (const uint8_t *)d_lor,
                           (const u64 *)ctx->reg_rank.p, nb, (int64_t)reg_count, d_coef, (T *)ctx->coef_compact.p);
        HIPCHK(hipGetLastError());
#ifndef SZH_SYNC_LAUNCH
        // (not in a pool lane: with several contexts at work the hand-off of coefficients to the running kernel failed 5 - 6 times in 480
        //  rounds of tools/gpu_pool_dbg.py -- errors, not wrong streams; none in 480 rounds with the serial order.  The lanes overlap one
        //  array's chain with the other's kernels anyway.
        //  A lone context takes the overlap only when no other compress call of the process is under way as this one starts.)
        overlap = !beam_applies<T>(G, d_in, reg_count) && !two_d && !ctx->gate && !ctx->no_chain_overlap && g_compress_calls.load() <= 1 && tune_int("SZ_HIP_CHAIN_THREADS", 1) && tune_int("SZ_HIP_CHAIN_OVERLAP", 1);
        if (overlap) { TRY(probe_streams(ctx)); overlap = ctx->streams_independent == 1; }
#endif
        TRY(ensure_pinned3(ctx, reg_count * 4 * sizeof(T) + (overlap ? ((size_t)nb + 64) * 4 * sizeof(T) + 256 : 0)));
        hcoef = (T *)ctx->pinned3;
        HIPCHK(hipMemcpyAsync(hcoef, ctx->coef_compact.p, reg_count * 4 * sizeof(T), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        TP("coef on host");
        double h0 = now_ms();
        all_reg_keep.assign(reg_count, 0);
        TP("chain begin");
        const std::vector<unsigned char> &all_reg = all_reg_keep;
        // 2-D planes are carried as {0, a, b, c}: the chain sees components 1..3
        T *const chain_in = hcoef + (two_d ? reg_count : 0);
        szhost_coeff_chain_begin(is_double, all_reg.data(), reg_count, (double)eb, G.g0.late, G.g1.late, G.g2.late, ncoef, &cf);
        {   // the chains write into the context's arrays (touched in earlier calls) instead of freshly allocated ones
            if (ctx->chain_codes.size() < reg_count * 4) ctx->chain_codes.resize(reg_count * 4 + reg_count / 4);
            if (ctx->chain_unpred.size() < reg_count * 4 * sizeof(T)) ctx->chain_unpred.resize((reg_count * 4 + reg_count / 4) * sizeof(T));
            for (int e = 0; e < 4; ++e) {
                free(cf.codes[e]); free(cf.unpred[e]);
                cf.codes[e] = ctx->chain_codes.data() + (size_t)e * reg_count;
                cf.unpred[e] = ctx->chain_unpred.data() + (size_t)e * reg_count * sizeof(T);
            }
            join_sections.own = true;
        }
        // (threads also for a handful of regression blocks: a section's fixed cost -- a 131 072-state code book -- is ~0.5 ms, and
        //  four of them in line delayed the wavefront kernel of BASELINE configs[3] by 2 ms)
        if (overlap) {
            // The chain runs NEXT TO the wavefront kernel: the threads publish how far they are, the kernel is launched right away and
            // makes a pencil that touches a regression block wait until the blocks it reads are final, and this thread ships the decoded
            // coefficients as they appear (after the launch, below).  blk_of_rank: scan-order block index of the r-th regression block.
            blk_of_rank.resize(reg_count);
            { size_t r = 0; for (int64_t bb = 0; bb < nb && r < reg_count; ++bb) if (!((ind_bits[bb >> 3] >> (7 - (bb & 7))) & 1)) blk_of_rank[r++] = (uint32_t)bb; if (r != reg_count) FAIL(SZHIP_ERR_INTERNAL, "indicator bits and regression-block count disagree"); }
            for (int e = 0; e < 4; ++e) chain_done[e] = 0;
            for (int e = 0; e < ncoef; ++e)
                section_threads.emplace_back([&, e, chain_in, ind = all_reg_keep.data()] {
                    szhost_coeff_chain_one_p(is_double, chain_in, ind, reg_count, use_mean, e, &cf, &chain_done[e]);
                    make_section(e);
                });
        } else if (tune_int("SZ_HIP_CHAIN_THREADS", 1) && tune_int("SZ_HIP_CHAIN_POOL", 1)) {
            if (!ctx->chain_pool) { ctx->chain_pool = new szhip_chain_pool(); ctx->chain_pool->start(); }
            pool_busy = true;
            ctx->chain_pool->submit(ncoef,
                [&, chain_in, ind = all_reg.data()](int e) {
                    chain_t0[e] = now_ms() - t_begin;
                    szhost_coeff_chain_one(is_double, chain_in, ind, reg_count, use_mean, e, &cf);
                    chain_t1[e] = now_ms() - t_begin;
                },
                [&](int e) { make_section(e); });
            ctx->chain_pool->wait_chains();           // the decoded coefficients are final: the main thread may ship them (the sections follow on the workers)
        } else if (tune_int("SZ_HIP_CHAIN_THREADS", 1)) {
            std::vector<std::promise<void>> chained(ncoef);
            std::vector<std::future<void>> chained_f;
            for (int e = 0; e < ncoef; ++e) chained_f.push_back(chained[e].get_future());
            for (int e = 0; e < ncoef; ++e)
                section_threads.emplace_back([&, e, chain_in, ind = all_reg.data()](std::promise<void> done) {
                    chain_t0[e] = now_ms() - t_begin;
                    szhost_coeff_chain_one(is_double, chain_in, ind, reg_count, use_mean, e, &cf);
                    chain_t1[e] = now_ms() - t_begin;
                    done.set_value();                     // the decoded coefficients of e are final: the main thread may ship them
                    make_section(e);
                }, std::move(chained[e]));
            for (auto &f : chained_f) f.wait();
        } else {
            for (int e = 0; e < ncoef; ++e) { szhost_coeff_chain_one(is_double, chain_in, all_reg.data(), reg_count, use_mean, e, &cf); make_section(e); }
        }
        host_ms += now_ms() - h0;
        TP("chain done");
        if (tp_on) fprintf(stderr, "chain threads: %.2f-%.2f %.2f-%.2f %.2f-%.2f %.2f-%.2f | ", chain_t0[0], chain_t1[0], chain_t0[1], chain_t1[1], chain_t0[2], chain_t1[2], chain_t0[3], chain_t1[3]);
        if (!overlap) {
            HIPCHK(hipMemcpyAsync(ctx->coef_compact.p, hcoef, reg_count * 4 * sizeof(T), hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL((k_move_coef<T, 1>), dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, (const uint8_t *)d_lor,
                               (const u64 *)ctx->reg_rank.p, nb, (int64_t)reg_count, d_coef, (T *)ctx->coef_compact.p);
            HIPCHK(hipGetLastError());
        }
    }
    HIPCHK(hipEventRecord(ctx->ev[1], st));
    TP("ev1");

    // ---- predict + quantise: the wavefront kernel
    // (the ribbon mapping writes its codes in its own order, szh_ribbon.h: tiles x steps x 1024 entries, a few per cent more than n)
    // Which sweep: arrays with regression blocks and doubles take the beam (k_pencil's 4.4 / 7.4 ms become 1.2 / 2.3 at 512^3 M-field / the f64 slab);
    // a float array whose blocks all chose Lorenzo stays on k_ribbon when that applies -- alone the two sweeps take the same time (1.05 against
    // 1.03 - 1.13 ms at 512^3), and k_ribbon's finished tile rows feed the entropy stage while it runs (one call 1.95 against 2.08 ms).
    // SZ_HIP_BEAM=2 sends everything the beam covers to it.
    const bool beam_first = reg_count > 0 || sizeof(T) == 8 || tune_int("SZ_HIP_BEAM", 1) >= 2 || !ribbon_applies<T>(G, reg_count);
    const bool use_beam = !overlap && beam_first && beam_applies<T>(G, d_in, reg_count);
    const bool use_ribbon = !use_beam && !overlap && ribbon_applies<T>(G, reg_count);
    szh_rb_layout rbl = {0, 0, 0, 0, 0, 0};
    size_t nat_elems = (size_t)n;
    if (use_ribbon) {
        using RS = szh_rb_shape<T>;
        rbl.on = 1; rbl.nTJ = (G.g1.count + 63) / 64; rbl.NT = szh_rb_steps_of<T>(G.g2.count); rbl.W = RS::W; rbl.R = RS::R; rbl.U = RS::U;
        const size_t tiles = (size_t)((G.g0.count + RS::W * RS::R - 1) / (RS::W * RS::R)) * rbl.nTJ;
        nat_elems = tiles * (size_t)szh_rb_tile_elems(rbl);
    }
    TRY(ensure(ctx, ctx->codes_nat, nat_elems * 2 + 64));
    TRY(ensure(ctx, ctx->codes_blk, (size_t)n * 2 + 64));
    uint16_t *d_nat = (uint16_t *)ctx->codes_nat.p, *d_blk = (uint16_t *)ctx->codes_blk.p;
    int nI, nJ, ntiles;
    using TS = szh_tile_shape<T>;
    TRY(prepare_pencil(ctx, G, szh_gran<T>::NW, TS::TPI, TS::TPJ, &nI, &nJ, &ntiles));
    // The entropy stage's two passes over the code array (histogram, block ordering) start on FINISHED TILE ROWS while the sweep is still
    // running (round 4): the sweep's last tile row ends ~35 % after its first one (the tile rows follow each other down dim 0), and the
    // sweep keeps the CUs busy with one workgroup each.  k_ribbon publishes every finished tile in host-coherent memory (a.tile_done);
    // this thread watches the words and launches the passes of slice after slice on two other streams.  SZ_HIP_SLICES=1: everything
    // after the sweep, as before.
    // A lane of a pool (several arrays in flight) takes two slices (SZ_HIP_SLICES_POOL; with the passes of the start of round 4 the lanes lost by
    // slicing -- two lanes at 512^3: 338 GB/s without, 324 with eight slices --, with the lighter passes of its end they gain: 349 -> 356).
    // Measured (round 4, 512^3 float, one call after the other, same box): 255 GB/s with 1 slice, 276 with 4, 251 - 275 with 8 (the slices'
    // kernels take 2 - 4 x their lone time beside the sweep and slow it by ~0.1 ms; more slices, more of that).
    const int slices_req = tune_int("SZ_HIP_SLICES", ctx->gate ? tune_int("SZ_HIP_SLICES_POOL", 2) : 4);
    const int slice_from = tune_int("SZ_HIP_SLICE_FROM", 0);          // even parts: per cent of the tile rows that the first slice covers at least
    const int slice_geom = tune_int("SZ_HIP_SLICE_GEOM", 0);
    // (round 5) the beam sweep: every wavefront publishes how many of its lines have their codes in memory (szh_beam.h, `tile_done`: a word per
    // wavefront, written every SZ_HIP_BEAM_PUB lines after write-through code stores and a vmcnt(0) -- a system-scope RELEASE per word cost ~35 us
    // each and took the sweep from 1.05 to 1.58 ms); a slice = the block rows whose lines every wavefront has passed.  Measured at 512^3, one call:
    // S-field 2.12 ms unsliced, 1.94 - 1.96 with 3 - 6 slices; M-field 3.94 -> 3.82 (profiles/r05_beam_slices.txt).  SZ_HIP_BEAM_SLICES=0: off.
    const bool sliced = (use_ribbon || (use_beam && tune_int("SZ_HIP_BEAM_SLICES", 1))) && slices_req > 1 && !tune_int("SZ_HIP_FUSE_HIST", 0);
    unsigned *tile_done = nullptr;
    const szh_bm::grid_t bgrid = szh_bm::make_grid(G);
    const size_t beam_words = (size_t)bgrid.nKB * bgrid.nJG * szh_bm::WPG;
    if (sliced) {
        const size_t tiles = use_beam ? beam_words : (size_t)((G.g0.count + szh_rb_shape<T>::W * szh_rb_shape<T>::R - 1) / (szh_rb_shape<T>::W * szh_rb_shape<T>::R)) * rbl.nTJ;
        TRY(ensure_coherent(ctx, 512 + tiles * 4));
        tile_done = (unsigned *)((char *)ctx->coh + 512);
        if (!ctx->stream3) {
            const int sprio = ctx->side_prio;
            int plo = 0, phi = 0;
            if (sprio && hipDeviceGetStreamPriorityRange(&plo, &phi) != hipSuccess) { plo = 0; phi = 0; }
            if (sprio) HIPCHK(hipStreamCreateWithPriority(&ctx->stream3, hipStreamNonBlocking, sprio == 2 ? plo : phi));
            else HIPCHK(hipStreamCreateWithFlags(&ctx->stream3, hipStreamNonBlocking));
        }
        if (!ctx->ev_perm) HIPCHK(hipEventCreateWithFlags(&ctx->ev_perm, hipEventDisableTiming));
    }
    {
        szh_qargs<T> a; memset(&a, 0, sizeof(a));
        a.tile_done = tile_done; a.pub_lines = tune_int("SZ_HIP_BEAM_PUB", 32);
        a.G = G; a.data = d_in; a.out = nullptr; a.codes = d_nat; a.blk_lor = d_lor; a.coef = d_coef; a.coef_stride = nb;
        a.eb = eb; a.recip = 1 / eb; a.mean = mean; a.cap = (int)intervals; a.radius = (int)intervals / 2; a.use_mean = use_mean;
        a.faceI = (szh_u64 *)ctx->faceI.p; a.faceJ = (szh_u64 *)ctx->faceJ.p; a.epoch = ++ctx->epoch;
        a.nI = nI; a.nJ = nJ; a.order = (const unsigned *)ctx->order.p; a.no_reg = reg_count == 0 && tune_int("SZ_HIP_NO_REG_HINT", 1);
        a.ticket = (unsigned *)(sm + SM_TICKET); a.err = (unsigned *)(sm + SM_ERR); a.ticket_mode = ctx->ticket_atomic ? 0 : tune_int("SZ_HIP_TICKET_MODE", 2);
        a.progress = (szh_u64 *)ctx->progress.p; a.backoff = tune_int("SZ_HIP_BACKOFF", 4); a.wide = tune_int("SZ_HIP_WIDE", 1) && (double)TS::TPI * nJ * 9.0 * (double)G.g2.count * szh_gran<T>::NW * 8.0 < 4.0e9 && (double)TS::TPJ * 9.0 * (double)G.g2.count * szh_gran<T>::NW * 8.0 < 4.0e9;
        a.trace = tune_int("SZ_HIP_TRACE", 0) ? (szh_u64 *)ctx->trace.p : nullptr;
        a.dbg = tune_int("SZ_HIP_DBG", 0); a.trace_tile = tune_int("SZ_HIP_TRACE_TILE", 1);
        a.coef_progress = nullptr;
        szh_u64 *coh_prog = nullptr;
        T *dec = nullptr;                                          // the decoded coefficients as the kernel reads them
        constexpr int64_t LINE = 128 / (int64_t)sizeof(T);         // values per 128-byte cache line
        const int64_t nbp = (nb + LINE - 1) / LINE * LINE;         // a coefficient's array starts on a line boundary
        if (overlap) {
            // Hand-off between the host's chain and the running kernel (per-XCD L2s are not coherent, and DMA writes do not touch them):
            //  * the progress word lives in HOST-COHERENT pinned memory (uncached on the GPU; the kernel polls it over PCIe).  In device
            //    memory a poll leaves the line in that XCD's L2 and every later poll finds the old value -- seen as a pencil that never saw
            //    the word the memory held.  It carries the launch epoch, so it is never cleared;
            //  * the coefficients go by DMA into a device buffer that no kernel ever stores to (reading them over PCIe was 45 ms: every lane
            //    of a block's rows reads them).  A line of it must never be loaded while only part of it is final -- the rest would stay
            //    stale in that L2 -- so progress is published in whole 128-byte lines of blocks, every coefficient array is line-aligned,
            //    and a pencil only loads below the published mark.  Lines of an earlier launch are gone at kernel start.
            TRY(ensure_coherent(ctx, 256));
            TRY(ensure(ctx, ctx->coef_dec, (size_t)nbp * 4 * sizeof(T)));
            coh_prog = (szh_u64 *)ctx->coh;
            dec = (T *)ctx->coef_dec.p;
            a.coef_progress = coh_prog;
            a.coef = dec; a.coef_stride = nbp;
        }
        {
        std::unique_lock<std::mutex> gate_lock;
        if (ctx->gate && tune_int("SZ_HIP_SWEEP_GATE", 0)) {
            gate_lock = std::unique_lock<std::mutex>(ctx->gate->m);
            if (ctx->gate->last && ctx->gate->last != ctx->ev_gate) HIPCHK(hipStreamWaitEvent(st, ctx->gate->last, 0));   // the other lane's sweep first
        }
        HIPCHK(hipEventRecord(ctx->ev[2], st));
        if (use_beam) { TRY((launch_beam<T, false>(ctx, G, a, st, reg_count))); S.quant_kernel = 2; }
        else if (use_ribbon) { TRY((launch_ribbon<T, false>(ctx, G, a, st))); S.quant_kernel = 1; }
        else {
        const unsigned pgrid = pencil_grid(ctx, a, ntiles);
        hipLaunchKernelGGL((k_pencil<T, false>), dim3(pgrid), dim3((TS::TPI * TS::TPJ + 2) * 64), 0, st, a);
        HIPCHK(hipGetLastError());
        }
        HIPCHK(hipEventRecord(ctx->ev[3], st));
        if (gate_lock.owns_lock()) { HIPCHK(hipEventRecord(ctx->ev_gate, st)); ctx->gate->last = ctx->ev_gate; }
        }
        TP("pencil launched");
        S.quant_kernel_launches = 1;
        if (overlap) {
            // ship the decoded coefficients while the kernel runs: DMA only (no kernel that could queue behind the waiting tiles), on the
            // second stream, in scan order: the values of the newly finished regression blocks go into a block-indexed pinned staging
            // array, the contiguous block range [first, last] of each coefficient into d_coef, then the progress word (same stream: it
            // lands after the data).  Blocks between regression blocks are Lorenzo blocks: whatever they receive is never read.
            const double h1 = now_ms();
            const szh_u64 tag = (szh_u64)(a.epoch & 0xffffffu) << 40;
            T *const full = (T *)(((uintptr_t)(hcoef + reg_count * 4) + 127) & ~(uintptr_t)127);     // pinned staging, [4][nbp] block-indexed
            const size_t chunk = std::max<size_t>(2048, reg_count / 32);
            size_t shipped = 0; int64_t sent = 0;                 // ranks scattered into the staging array; blocks [0, sent) are on the device
            int idle = 0;
            while (sent < nb) {
                size_t p = reg_count;
                for (int e = 0; e < ncoef; ++e) { const size_t d = __atomic_load_n(&chain_done[e], __ATOMIC_ACQUIRE); if (d < p) p = d; }
                if (p > shipped && (p - shipped >= chunk || p == reg_count)) {
                    for (int e = 0; e < ncoef; ++e)
                        for (size_t r = shipped; r < p; ++r) full[(size_t)e * nbp + blk_of_rank[r]] = hcoef[(size_t)e * reg_count + r];
                    shipped = p;
                    // every block below `fin` is final; whole lines of them travel
                    const int64_t fin = p == reg_count ? nb : (int64_t)blk_of_rank[p];
                    const int64_t upto = fin == nb ? nb : fin / LINE * LINE;
                    if (upto > sent) {
                        for (int e = 0; e < ncoef; ++e)
                            HIPCHK(hipMemcpyAsync(dec + (size_t)e * nbp + sent, full + (size_t)e * nbp + sent, (size_t)(upto - sent) * sizeof(T), hipMemcpyHostToDevice, ctx->stream2));
                        HIPCHK(hipStreamSynchronize(ctx->stream2));            // the values are in device memory ...
                        __atomic_store_n(coh_prog, tag | (szh_u64)upto, __ATOMIC_RELEASE);   // ... before the kernel may look for them
                        sent = upto;
                    }
                    idle = 0;
                } else if (++idle > 200000) FAIL(SZHIP_ERR_INTERNAL, "coefficient chain made no progress");
                else std::this_thread::sleep_for(std::chrono::microseconds(10));
            }
            host_ms += now_ms() - h1;
            S.chain_overlapped = 1;
            TP("coefficients shipped");
        }
    }

    // ---- histogram, block ordering, unpredictable counts
    // The histogram (and its copy to the host) runs on the second stream next to the block-ordering pass, so the host builds
    // the code book while k_permute is still running.
    TRY(ensure(ctx, ctx->hist, (size_t)(65536 + 8192) * 4 + 64));
    unsigned *d_hist = (unsigned *)ctx->hist.p;
    TRY(ensure_pinned(ctx, (size_t)intervals * 4 + 64));
    unsigned *h_hist = (unsigned *)ctx->pinned;
    auto launch_hist = [&](const uint16_t *codes) -> int {
        HIPCHK(hipEventRecord(ctx->ev_in, st));                    // codes complete
        HIPCHK(hipStreamWaitEvent(ctx->stream2, ctx->ev_in, 0));
        HIPCHK(hipMemsetAsync(d_hist, 0, (size_t)intervals * 4, ctx->stream2));
        int rshift = 0; int use_lds = intervals <= 16384;
        if (use_lds) { while ((intervals << (rshift + 1)) <= 16384u && rshift < 6) ++rshift; }
        const size_t lds = use_lds ? ((size_t)intervals << rshift) * 4 : 16;
        const int64_t nh = use_ribbon ? (int64_t)nat_elems : n;          // ribbon order: the whole padded array, positions outside skipped by geometry
        int grid = (int)std::min<int64_t>((nh / 8 + 255) / 256 + 1, 2048);
        hipLaunchKernelGGL(k_hist_u16, dim3(grid), dim3(256), lds, ctx->stream2, codes, nh, intervals, rshift, use_lds, d_hist, rbl, G.g0.count, G.g1.count, G.g2.count, (int64_t)0);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(h_hist, d_hist, (size_t)intervals * 4, hipMemcpyDeviceToHost, ctx->stream2));
        HIPCHK(hipEventRecord(ctx->ev_fit, ctx->stream2));
        return SZHIP_OK;
    };
    // (SZ_HIP_FUSE_HIST=1 takes the histogram inside k_permute instead: measured equal in kernel time -- 393 against 322 + 69 us -- and
    //  without the overlap, so it is off)
    bool fuse_hist = intervals <= 4096 && tune_int("SZ_HIP_FUSE_HIST", 0);
    if (!fuse_hist && !sliced) TRY(launch_hist((const uint16_t *)d_nat));    // next to the block-ordering pass (ribbon order: padding skipped by geometry)
    TRY(ensure(ctx, ctx->col_zeros, (size_t)ncols * 4));
    TRY(ensure(ctx, ctx->col_zeros64, (size_t)ncols * 8));
    TRY(ensure(ctx, ctx->col_off, (size_t)ncols * 8));
    int perm_segb = 1, perm_nseg = 1;
    if (sliced && use_beam) {
        // slices of block rows (dim 0): the codes are in natural order, so a slice's codes are ONE contiguous range for the histogram, and the
        // block-ordering pass takes the slice's block rows.  A slice starts when every wavefront of the sweep has published its lines.
        const int NS = std::min(slices_req, G.g0.num);
        const int segb = choose_segb(G, 2, tune_int("SZ_HIP_PERM_TILE_KB", 32) * 1024);
        const int nseg = (G.g2.num + segb - 1) / segb;
        TRY(ensure(ctx, ctx->zcnt, (size_t)ncols * nseg * 4));
        TRY(ensure(ctx, ctx->zpos, (size_t)ncols * nseg * SZH_ZCAP * 4));
        const size_t tb = tile_bytes(G, segb, 2), tile_el = (tb + 1) / 2;
        int rshift = 0; const int use_lds = intervals <= 16384;
        if (use_lds) { while ((intervals << (rshift + 1)) <= 16384u && rshift < 6) ++rshift; }
        const size_t hist_lds = use_lds ? ((size_t)intervals << rshift) * 4 : 16;
        HIPCHK(hipMemsetAsync(d_hist, 0, (size_t)intervals * 4, ctx->stream2));
        HIPCHK(hipMemsetAsync(ctx->col_zeros.p, 0, (size_t)ncols * 4, ctx->stream3));
        const unsigned tag = (ctx->epoch & 0xfffu) << 20;
        bool sweep_over = false;
        int b0_done = 0;
        size_t first_late = 0;                                     // wavefronts below this index have been seen past the current slice's rows
        int64_t hist_first = 0;
        for (int sl = 0; sl < NS; ++sl) {
            const int b0_hi = sl == NS - 1 ? G.g0.num : std::max(b0_done, (int)((int64_t)G.g0.num * (sl + 1) / NS));
            const int rows_need = b0_hi >= G.g0.num ? G.g0.count : szh_blk_start(G.g0, b0_hi);
            first_late = 0;
            unsigned spins = 0;
            while (!sweep_over && first_late < beam_words) {
                const unsigned wv = __atomic_load_n(&tile_done[first_late], __ATOMIC_ACQUIRE);
                if ((wv & 0xfff00000u) == tag && (int)(wv & 0xfffffu) >= rows_need) { ++first_late; continue; }
                szhip_chain_pool::pause();
                if ((++spins & 127u) == 0) {
                    const hipError_t q = hipEventQuery(ctx->ev[3]);
                    if (q == hipSuccess) { sweep_over = true; break; }
                    if (q != hipErrorNotReady) HIPCHK(q);
                }
            }
            if (b0_hi > b0_done) {
                // (k_hist_u16 counts groups of eight codes: a slice of it ends on a multiple of eight below the slice's last code, the last slice takes the rest)
                const int64_t h_lo = hist_first, h_hi = b0_hi >= G.g0.num ? (int64_t)n : ((int64_t)rows_need * G.d0) / 8 * 8;
                hist_first = h_hi;
                const int grid = (int)std::min<int64_t>(((h_hi - h_lo) / 8 + 255) / 256 + 1, 2048);
                if (h_hi > h_lo) hipLaunchKernelGGL(k_hist_u16, dim3(grid), dim3(256), hist_lds, ctx->stream2, (const uint16_t *)d_nat, h_hi, intervals, rshift, use_lds, d_hist, rbl,
                                   G.g0.count, G.g1.count, G.g2.count, h_lo);
                hipLaunchKernelGGL((k_permute<0>), dim3((unsigned)((b0_hi - b0_done) * G.g1.num), std::min(nseg, std::max(1, tune_int("SZ_HIP_PERM_Y", 1)))), dim3(256), ((tile_el + 1) & ~(size_t)1) * 2 + 16, ctx->stream3, G,
                                   (const uint16_t *)d_nat, d_blk, (unsigned *)ctx->col_zeros.p, segb, (unsigned *)ctx->zcnt.p, (unsigned *)ctx->zpos.p, rbl, d_hist, 0u,
                                   (int)tile_el, b0_done * G.g1.num, 0);
                HIPCHK(hipGetLastError());
                b0_done = b0_hi;
            }
        }
        HIPCHK(hipMemcpyAsync(h_hist, d_hist, (size_t)intervals * 4, hipMemcpyDeviceToHost, ctx->stream2));
        HIPCHK(hipEventRecord(ctx->ev_fit, ctx->stream2));
        hipLaunchKernelGGL(k_u32_to_u64, dim3((ncols + 255) / 256), dim3(256), 0, ctx->stream3, (const unsigned *)ctx->col_zeros.p, (int64_t)ncols, (u64 *)ctx->col_zeros64.p);
        TRY(scan_u64(ctx, (const u64 *)ctx->col_zeros64.p, ncols, (u64 *)ctx->col_off.p, sm + SM_TOTAL_UNPRED, ctx->stream3));
        HIPCHK(hipEventRecord(ctx->ev_perm, ctx->stream3));
        perm_segb = segb; perm_nseg = nseg;
    } else if (sliced) {
        using RS = szh_rb_shape<T>;
        constexpr int WR = RS::W * RS::R;
        const int nTI = (G.g0.count + WR - 1) / WR, nTJ = rbl.nTJ, NS = std::min(slices_req, nTI);
        const int segb = choose_segb(G, 2, tune_int("SZ_HIP_PERM_TILE_KB", 32) * 1024);
        const int nseg = (G.g2.num + segb - 1) / segb;
        TRY(ensure(ctx, ctx->zcnt, (size_t)ncols * nseg * 4));
        TRY(ensure(ctx, ctx->zpos, (size_t)ncols * nseg * SZH_ZCAP * 4));
        const size_t tb = tile_bytes(G, segb, 2), tile_el = (tb + 1) / 2;
        int rshift = 0; const int use_lds = intervals <= 16384;
        if (use_lds) { while ((intervals << (rshift + 1)) <= 16384u && rshift < 6) ++rshift; }
        const size_t hist_lds = use_lds ? ((size_t)intervals << rshift) * 4 : 16;
        HIPCHK(hipMemsetAsync(d_hist, 0, (size_t)intervals * 4, ctx->stream2));
        HIPCHK(hipMemsetAsync(ctx->col_zeros.p, 0, (size_t)ncols * 4, ctx->stream3));
        const unsigned ep = ctx->epoch;
        bool sweep_over = false;                                   // the sweep's end event has been seen: every tile is there
        int b0_done = 0; int64_t hist_first = 0;
        for (int sl = 0; sl < NS; ++sl) {
            // slice bounds: halving (the first slice takes half of the tile rows, the next half of the rest, ...: the early slices have the
            // rest of the sweep to hide in, the LAST one runs after the sweep's end and should be small), or even parts (SZ_HIP_SLICE_GEOM=0)
            auto bound = [&](int k) -> int {                                  // first tile row of slice k (k = NS: nTI)
                if (k <= 0) return 0;
                if (k >= NS) return nTI;
                if (slice_geom) return std::max(k, nTI - std::max(1, nTI >> k));
                const int ti0 = std::min(nTI - 1, (int)((int64_t)nTI * slice_from / 100));       // rows [0, ti0) ride with the first slice
                return ti0 + (int)((int64_t)(nTI - ti0) * k / NS);
            };
            const int ti_lo = bound(sl), ti_hi = std::max(bound(sl + 1), ti_lo);
            for (int t = ti_lo * nTJ; t < ti_hi * nTJ && !sweep_over; ++t) {
                unsigned spins = 0;
                while (__atomic_load_n(&tile_done[t], __ATOMIC_ACQUIRE) != ep) {
                    szhip_chain_pool::pause();                     // (the core's sibling thread gets the issue slots while this one watches a word)
                    if ((++spins & 127u) == 0) {
                        const hipError_t q = hipEventQuery(ctx->ev[3]);
                        if (q == hipSuccess) { sweep_over = true; break; }
                        if (q != hipErrorNotReady) HIPCHK(q);
                    }
                }
            }
            // histogram of the slice's part of the ribbon order (tile major: one contiguous range)
            const int64_t hist_end = sl == NS - 1 ? (int64_t)nat_elems : (int64_t)ti_hi * nTJ * szh_rb_tile_elems(rbl);
            if (hist_end > hist_first) {
                const int grid = (int)std::min<int64_t>(((hist_end - hist_first) / 8 + 255) / 256 + 1, 2048);
                hipLaunchKernelGGL(k_hist_u16, dim3(grid), dim3(256), hist_lds, ctx->stream2, (const uint16_t *)d_nat, hist_end, intervals, rshift, use_lds, d_hist, rbl,
                                   G.g0.count, G.g1.count, G.g2.count, hist_first);
                hist_first = hist_end;
            }
            // block ordering of the block rows that lie inside finished tile rows
            const int rows_ready = std::min(ti_hi * WR, G.g0.count);
            int b0_hi = b0_done;
            if (sl == NS - 1) b0_hi = G.g0.num;
            else while (b0_hi < G.g0.num && (b0_hi + 1 < G.g0.num ? szh_blk_start(G.g0, b0_hi + 1) : G.g0.count) <= rows_ready) ++b0_hi;
            if (b0_hi > b0_done) {
                hipLaunchKernelGGL((k_permute<0>), dim3((unsigned)((b0_hi - b0_done) * G.g1.num), std::min(nseg, std::max(1, tune_int("SZ_HIP_PERM_Y", 1)))), dim3(256), ((tile_el + 1) & ~(size_t)1) * 2 + 16, ctx->stream3, G,
                                   (const uint16_t *)d_nat, d_blk, (unsigned *)ctx->col_zeros.p, segb, (unsigned *)ctx->zcnt.p, (unsigned *)ctx->zpos.p, rbl, d_hist, 0u,
                                   (int)tile_el, b0_done * G.g1.num, 0);
                b0_done = b0_hi;
            }
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipMemcpyAsync(h_hist, d_hist, (size_t)intervals * 4, hipMemcpyDeviceToHost, ctx->stream2));
        HIPCHK(hipEventRecord(ctx->ev_fit, ctx->stream2));
        hipLaunchKernelGGL(k_u32_to_u64, dim3((ncols + 255) / 256), dim3(256), 0, ctx->stream3, (const unsigned *)ctx->col_zeros.p, (int64_t)ncols, (u64 *)ctx->col_zeros64.p);
        TRY(scan_u64(ctx, (const u64 *)ctx->col_zeros64.p, ncols, (u64 *)ctx->col_off.p, sm + SM_TOTAL_UNPRED, ctx->stream3));
        HIPCHK(hipEventRecord(ctx->ev_perm, ctx->stream3));           // `st` waits for it where it first needs the block order (below)
        perm_segb = segb; perm_nseg = nseg;
    } else {
    HIPCHK(hipMemsetAsync(ctx->col_zeros.p, 0, (size_t)ncols * 4, st));
    {
        const int segb = choose_segb(G, 2, tune_int("SZ_HIP_PERM_TILE_KB", 32) * 1024);
        const int nseg = (G.g2.num + segb - 1) / segb;
        TRY(ensure(ctx, ctx->zcnt, (size_t)ncols * nseg * 4));
        TRY(ensure(ctx, ctx->zpos, (size_t)ncols * nseg * SZH_ZCAP * 4));
        // the histogram is taken inside the block-ordering pass when its bins fit behind the tile in LDS (SZ_HIP_FUSE_HIST=0: the
        // separate pass, which is also what large alphabets get)
        fuse_hist = intervals <= 4096 && tune_int("SZ_HIP_FUSE_HIST", 0);
        const size_t tb = tile_bytes(G, segb, 2), tile_el = (tb + 1) / 2;
        if (fuse_hist) HIPCHK(hipMemsetAsync(d_hist, 0, (size_t)intervals * 4, st));
        // development (timing only): an EXTRA launch in front of the real one that stops early -- 8: after the prologue; 4: after the gather; 5: gather
        // without its loads; 6: gather without its LDS stores.  It leaves nothing behind that the real launch does not overwrite.
        if (const int pdbg = tune_int("SZ_HIP_PERM_DBG", 0)) {
            hipLaunchKernelGGL((k_permute<0>), dim3(ncols, std::min(nseg, std::max(1, tune_int("SZ_HIP_PERM_Y", 1)))), dim3(256), ((tile_el + 1) & ~(size_t)1) * 2 + 16, st, G, (const uint16_t *)d_nat,
                               d_blk, (unsigned *)ctx->col_zeros.p, segb, (unsigned *)ctx->zcnt.p, (unsigned *)ctx->zpos.p, rbl, d_hist, 0u, (int)tile_el, 0, (pdbg & 8) ? 8 : (pdbg | 4));
        }
        hipLaunchKernelGGL((k_permute<0>), dim3(ncols, std::min(nseg, std::max(1, tune_int("SZ_HIP_PERM_Y", 1)))), dim3(256), ((tile_el + 1) & ~(size_t)1) * 2 + (fuse_hist ? (size_t)intervals * 4 : 0) + 16, st, G, (const uint16_t *)d_nat,
                           d_blk, (unsigned *)ctx->col_zeros.p, segb, (unsigned *)ctx->zcnt.p, (unsigned *)ctx->zpos.p, rbl,
                           d_hist, fuse_hist ? intervals : 0u, (int)tile_el, 0, 0);
        if (fuse_hist) {
            HIPCHK(hipMemcpyAsync(h_hist, d_hist, (size_t)intervals * 4, hipMemcpyDeviceToHost, st));
            HIPCHK(hipEventRecord(ctx->ev_fit, st));
        }
        perm_segb = segb; perm_nseg = nseg;
        HIPCHK(hipGetLastError());
    }
    hipLaunchKernelGGL(k_u32_to_u64, dim3((ncols + 255) / 256), dim3(256), 0, st, (const unsigned *)ctx->col_zeros.p, (int64_t)ncols,
                       (u64 *)ctx->col_zeros64.p);
    TRY(scan_u64(ctx, (const u64 *)ctx->col_zeros64.p, ncols, (u64 *)ctx->col_off.p, sm + SM_TOTAL_UNPRED));
    }

    // ---- Huffman code book (host: heap order decides the codes), built as soon as the histogram has arrived.  This is the only host
    //      round trip of the entropy stage: the number of unpredictable values is the histogram's bin 0, so the header can be written
    //      and the remaining kernels enqueued while the block-ordering pass is still running; the kernel error flag and the device's
    //      own count of zero codes are checked after the final synchronisation.
    TP("permute launched");
    HIPCHK(hipEventSynchronize(ctx->ev_fit));
    TP("hist on host");
    double h0 = now_ms();
    szhost_huff *hf = szhost_huff_build(2 * (int)intervals, h_hist, nullptr, intervals);
    host_ms += now_ms() - h0;
    const u64 total_unpred = h_hist[0];
    S.n_unpred = total_unpred;
    if (!hf) FAIL(SZHIP_ERR_INTERNAL, "Huffman build failed");

    // ---- stream header
    TP("tree built");
    for (auto &x : section_threads) if (x.joinable()) x.join();
    if (pool_busy) { ctx->chain_pool->wait_all(); pool_busy = false; }
    TP("sections joined");
    if (section_failed) { szhost_huff_free(hf); FAIL(SZHIP_ERR_INTERNAL, "coefficient Huffman build failed"); }
    for (int e = 0; e < ncoef; ++e) coef_sections.insert(coef_sections.end(), section[e].begin(), section[e].end());
    h0 = now_ms();
    const size_t tree_bytes = szhost_huff_tree_size(hf);
    const size_t hdr_len = meta_len + 8 + 4 + sizeof(T) + 4 + 4 + 4 + tree_bytes + 1 + sizeof(T) + ind_bytes + coef_sections.size() + 8;
    const size_t unpred_bytes = (size_t)total_unpred * sizeof(T);
    const size_t pay_bytes = (size_t)((hf->total_bits + 7) / 8);
    const size_t total_len = hdr_len + unpred_bytes + pay_bytes;
    // assembled in pinned memory (the histogram that lived there has been consumed): with coefficient sections the header is megabytes,
    // and an asynchronous copy from pageable memory of that size makes the runtime pin and unpin the pages
    TRY(ensure_pinned(ctx, hdr_len + 64));
    unsigned char *const hdr = (unsigned char *)ctx->pinned;
    memset(hdr, 0, hdr_len);
    {
        unsigned char *q = hdr;
        memcpy(q, meta, meta_len); q += meta_len;
        szhost_put_u64be(q, (uint64_t)n); q += 8;
        szhost_put_u32be(q, (uint32_t)G.block_size); q += 4;
        if (is_double) szhost_put_f64be(q, (double)eb); else szhost_put_f32be(q, (float)eb);
        q += sizeof(T);
        szhost_put_u32be(q, intervals); q += 4;
        szhost_put_u32be(q, (uint32_t)tree_bytes); q += 4;
        szhost_put_u32be(q, (uint32_t)hf->n_nodes); q += 4;
        szhost_huff_tree_write(hf, q); q += tree_bytes;
        *q++ = (unsigned char)use_mean;
        memcpy(q, &mean, sizeof(T)); q += sizeof(T);
        memcpy(q, ind_bits, ind_bytes); q += ind_bytes;
        if (!coef_sections.empty()) { memcpy(q, coef_sections.data(), coef_sections.size()); q += coef_sections.size(); }
        const uint64_t tu = total_unpred; memcpy(q, &tu, 8); q += 8;
    }
    // device code tables: right-aligned code bits + lengths, one entry per symbol < intervals
    std::vector<u64> tab_code(intervals); std::vector<uint8_t> tab_len(intervals);
    for (unsigned s = 0; s < intervals; ++s) { tab_code[s] = hf->code[s]; tab_len[s] = hf->len[s]; }
    const u64 total_bits = hf->total_bits;
    szhost_huff_free(hf);
    // code words of up to 32 bits and a table that fits beside the window in LDS: k_encode32 (32 consecutive codes per thread) packs the
    // payload, from the table `code << 8 | length`; anything else stays with k_encode
    unsigned enc_maxlen = 0;
    for (unsigned s = 0; s < intervals; ++s) enc_maxlen = std::max<unsigned>(enc_maxlen, tab_len[s]);
    const size_t lds_e32 = (size_t)intervals * 8 + ((size_t)SZH_E32_ROUND * enc_maxlen / 32 + 4) * 4 + 16;
    const bool enc32 = enc_maxlen >= 1 && enc_maxlen <= 32 && lds_e32 <= 60 * 1024 && tune_int("SZ_HIP_ENC32", 1);
    if (enc32) for (unsigned s = 0; s < intervals; ++s) tab_code[s] = (tab_code[s] << 8) | tab_len[s];
    host_ms += now_ms() - h0;

    TRY(ensure(ctx, ctx->code_tab, (size_t)intervals * 8));
    TRY(ensure(ctx, ctx->len_tab, (size_t)intervals));
    HIPCHK(hipMemcpyAsync(ctx->code_tab.p, tab_code.data(), (size_t)intervals * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ctx->len_tab.p, tab_len.data(), (size_t)intervals, hipMemcpyHostToDevice, st));
    TRY(ensure(ctx, ctx->stream_buf, total_len + 64));
    unsigned char *d_stream = (unsigned char *)ctx->stream_buf.p;
    HIPCHK(hipMemsetAsync(d_stream, 0, total_len + 64, st));
    HIPCHK(hipMemcpyAsync(d_stream, hdr, hdr_len, hipMemcpyHostToDevice, st));
    if (sliced) HIPCHK(hipStreamWaitEvent(st, ctx->ev_perm, 0));      // block order and per-column offsets (third stream) from here on
    if (total_unpred > 0) {
        // the unpredictable values are gathered on the second stream while the payload is being encoded on the first (both only read
        // the block-ordered codes); their copy into the stream follows the join below
        TRY(ensure(ctx, ctx->unpred, unpred_bytes));
        HIPCHK(hipEventRecord(ctx->ev_in, st));                // block order, per-column offsets ready
        HIPCHK(hipStreamWaitEvent(ctx->stream2, ctx->ev_in, 0));
        hipLaunchKernelGGL((k_unpred<T, 0>), dim3(ncols), dim3(256), 0, ctx->stream2, G, (const uint16_t *)d_blk, (const unsigned *)ctx->col_zeros.p,
                           (const u64 *)ctx->col_off.p, d_in, (T *)ctx->unpred.p, (T *)nullptr, (const unsigned *)ctx->zcnt.p,
                           (const unsigned *)ctx->zpos.p, perm_segb, perm_nseg);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(ctx->ev_fit, ctx->stream2));
    }
    if (total_bits > 0) {
        const int64_t nchunks = (n + SZH_ENC_CHUNK - 1) / SZH_ENC_CHUNK;
        TRY(ensure(ctx, ctx->chunk_bits, (size_t)nchunks * 8));
        TRY(ensure(ctx, ctx->chunk_off, (size_t)nchunks * 8));
        hipLaunchKernelGGL(k_chunk_bits, dim3((unsigned)((nchunks + SZH_CB_PER - 1) / SZH_CB_PER)), dim3(256), 0, st, (const uint16_t *)d_blk, n, (const uint8_t *)ctx->len_tab.p,
                           intervals, (u64 *)ctx->chunk_bits.p);
        TRY(scan_u64(ctx, (const u64 *)ctx->chunk_bits.p, nchunks, (u64 *)ctx->chunk_off.p, sm + SM_TOTAL_BITS));
        if (enc32) {
            const int64_t nrounds = (n + SZH_E32_ROUND - 1) / SZH_E32_ROUND;
            hipLaunchKernelGGL(k_encode32, dim3((unsigned)((nrounds + SZH_E32_PER - 1) / SZH_E32_PER)), dim3(256), lds_e32, st, (const uint16_t *)d_blk, n, (const u64 *)ctx->code_tab.p,
                               intervals, (const u64 *)ctx->chunk_off.p, (u64)(hdr_len + unpred_bytes) * 8, (unsigned *)d_stream);
        } else
        hipLaunchKernelGGL(k_encode, dim3((unsigned)((nchunks + SZH_ENC_PER - 1) / SZH_ENC_PER)), dim3(256), 0, st, (const uint16_t *)d_blk, n, (const u64 *)ctx->code_tab.p,
                           (const uint8_t *)ctx->len_tab.p, intervals, (const u64 *)ctx->chunk_off.p, (u64)(hdr_len + unpred_bytes) * 8,
                           (unsigned *)d_stream);
        HIPCHK(hipGetLastError());
    }
    if (total_unpred > 0) {
        HIPCHK(hipStreamWaitEvent(st, ctx->ev_fit, 0));
        HIPCHK(hipMemcpyAsync(d_stream + hdr_len, ctx->unpred.p, unpred_bytes, hipMemcpyDeviceToDevice, st));
    }
    HIPCHK(hipEventRecord(ctx->ev[4], st));
    TP("encode launched");
    u64 h_small[SM_COUNT];                                     // checked after the synchronisation below
    HIPCHK(hipMemcpyAsync(h_small, sm, SM_COUNT * 8, hipMemcpyDeviceToHost, st));
    if (out_on_device == 2) { // caller-provided device buffer of capacity *out_size
        if (!*out || *out_size < total_len) FAIL(SZHIP_ERR_ARG, "caller's device buffer too small (%zu < %zu)", *out_size, total_len);
        HIPCHK(hipMemcpyAsync(*out, d_stream, total_len, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
    } else if (out_on_device) {
        HIPCHK(hipStreamSynchronize(st));
        *out = d_stream;
    } else {
        unsigned char *h = (unsigned char *)malloc(total_len ? total_len : 1);
        if (!h) FAIL(SZHIP_ERR_INTERNAL, "out of host memory");
        TRY(staged_copy(ctx, h, d_stream, total_len, false));
        *out = h;
    }
    *out_size = total_len;
    TP("final sync");
    if (tp_on) { for (int i = 0; i < tp_k; ++i) fprintf(stderr, "%s %.2f | ", tp_n[i], tp_t[i]); fprintf(stderr, "\n"); }
    // after the final synchronisation: the wavefront kernel's error flag; the shuffled bit count and the device's count of zero codes
    // must match what the histogram predicted
    if ((unsigned)h_small[SM_ERR] == 2) { ctx->coef_late = true; FAIL_PUBLISHED(SZHIP_ERR_INTERNAL, "wavefront kernel: the regression coefficients did not arrive"); }
    if ((unsigned)h_small[SM_ERR] != 0) { ctx->wave_timeout = true; FAIL_PUBLISHED(SZHIP_ERR_INTERNAL, "wavefront kernel: halo wait timed out"); }
    if ((total_bits > 0 && h_small[SM_TOTAL_BITS] != total_bits) || h_small[SM_TOTAL_UNPRED] != total_unpred)
        FAIL_PUBLISHED(SZHIP_ERR_INTERNAL, "entropy stage mismatch (bits %llu vs %llu, unpredictable %llu vs %llu)", (unsigned long long)h_small[SM_TOTAL_BITS],
             (unsigned long long)total_bits, (unsigned long long)h_small[SM_TOTAL_UNPRED], (unsigned long long)total_unpred);
    float ms = 0;
    hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]); S.ms_prequant = ms;
    hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3]); S.ms_quant = ms;
    hipEventElapsedTime(&ms, ctx->ev[3], ctx->ev[4]); S.ms_entropy = ms;
    S.ms_host = host_ms; S.ms_total = now_ms() - t_begin; S.out_bytes = total_len;
    if (stats) *stats = S;
    return SZHIP_OK;
}

// Huffman decode of `n` symbols on the device (self-synchronising sub-sequence decode, k_hdec_*): `d_bits` points at the payload,
// `dtab` is the host-built decode table of the tree, `single_symbol` >= 0 for the one-leaf tree (zero payload bits).
// after the synchronisation that follows huff_decode_device: did its two unsynchronised rounds settle every start?
#define HDEC_CHECK(ctx) do { if ((ctx)->hdec_res[1] != 0 || (tune_int("SZ_HIP_TEST_HDEC_FALLBACK", 0) && !(ctx)->hdec_sync_rounds)) { (ctx)->hdec_unconverged = true; \
        /* (no message on stderr: the caller's wrapper repeats the call with a synchronisation per round) */ \
        snprintf((ctx)->err, sizeof((ctx)->err), "Huffman decode: %llu start guesses still moving after two rounds", (ctx)->hdec_res[1]); return SZHIP_ERR_INTERNAL; } } while (0)
int huff_decode_device(szhip_ctx *ctx, u64 *sm, const unsigned char *d_bits, unsigned bytes_before, u64 total_bits, const std::vector<uint32_t> &dtab, int n_nodes,
                       int single_symbol, int64_t n, uint16_t *d_out_codes, u64 *total_sym_host)
{
    // *total_sym_host receives the number of symbols the payload holds ASYNCHRONOUSLY: the caller compares it with n after its next
    // synchronisation of the stream (the write pass below never stores beyond n, so a short payload is harmless until then)
    if (!ctx->hdec_res) HIPCHK(hipHostMalloc((void **)&ctx->hdec_res, 64, hipHostMallocDefault));
    ctx->hdec_res[0] = (u64)n; ctx->hdec_res[1] = 0;
    *total_sym_host = (u64)n;
    hipStream_t st = ctx->stream;
    if (single_symbol >= 0) {
        hipLaunchKernelGGL(k_fill_u16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_out_codes, n, (uint16_t)single_symbol);
        HIPCHK(hipGetLastError());
    } else {
        const int64_t nsub = (int64_t)((total_bits + SZH_SUBSEQ_BITS - 1) / SZH_SUBSEQ_BITS);
        if (nsub == 0) FAIL(SZHIP_ERR_STREAM, "empty Huffman payload");
        const size_t lut_off = (dtab.size() * 4 + 63) / 64 * 64;
        TRY(ensure(ctx, ctx->dec_tab, lut_off + SZH_LUT_BYTES));
        HIPCHK(hipMemcpyAsync(ctx->dec_tab.p, dtab.data(), dtab.size() * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_hdec_build_lut, dim3(SZH_LUT_SIZE / 256), dim3(256), 0, st, (const unsigned *)ctx->dec_tab.p, (uint4 *)((char *)ctx->dec_tab.p + lut_off));
        TRY(ensure(ctx, ctx->starts, (size_t)nsub * 8)); TRY(ensure(ctx, ctx->ends, (size_t)nsub * 8));
        TRY(ensure(ctx, ctx->counts, (size_t)nsub * 8)); TRY(ensure(ctx, ctx->offs, (size_t)nsub * 8));
        TRY(ensure(ctx, ctx->dirty, (size_t)nsub));
        szh_hdec_args a;
        a.bits = d_bits; a.total_bits = total_bits; a.bytes_before = bytes_before; a.table = (const unsigned *)ctx->dec_tab.p; a.n_nodes = n_nodes;
        a.table_in_lds = (size_t)n_nodes * 8 <= 13 * 1024; a.nsub = nsub;      // (the write pass: 33 KB of bits + 16 KB of table + this <= 64 KB)
        a.lut = (const uint4 *)((char *)ctx->dec_tab.p + lut_off);
        a.starts = (u64 *)ctx->starts.p; a.ends = (u64 *)ctx->ends.p; a.counts = (u64 *)ctx->counts.p;
        a.dirty = (unsigned char *)ctx->dirty.p; a.changed = (unsigned *)(sm + SM_CHANGED);
        const size_t lds_tab = a.table_in_lds ? (size_t)n_nodes * 8 : 16;
        const size_t lds_pass = SZH_HDEC_LDS + SZH_LUT_SIZE * 4 + lds_tab, lds_write = SZH_HDEC_LDS + SZH_LUT_SIZE * 16 + lds_tab;
        const unsigned gsub = (unsigned)((nsub + 255) / 256);
        hipLaunchKernelGGL(k_hdec_init, dim3(gsub), dim3(256), 0, st, a);
        const bool optimistic = !ctx->hdec_sync_rounds && tune_int("SZ_HIP_HDEC_OPTIMISTIC", 1) != 0;
        // rounds without a host round trip: round 0 (warm-up guesses), then repair rounds that only touch the sub-sequences whose start
        // moved (a workgroup without one returns at once: ~10 us per idle round).  Smooth fields settle in round 1; wide code books
        // (codes longer than the look-up window resynchronise slowly) sometimes need a third
        const int64_t opt_rounds = std::max(2, tune_int("SZ_HIP_HDEC_ROUNDS", 3));
        int64_t iter = 0;
        for (;;) {
            a.warmup = iter == 0;                                  // the first round finds its own starts (k_hdec_pass)
            hipLaunchKernelGGL(k_hdec_pass, dim3(gsub), dim3(256), lds_pass, st, a);
            HIPCHK(hipMemsetAsync(sm + SM_CHANGED, 0, 8, st));
            hipLaunchKernelGGL(k_hdec_update, dim3(gsub), dim3(256), 0, st, a);
            HIPCHK(hipGetLastError());
            if (optimistic) {
                // round 0 repairs nearly every guess (84 of 287 393 wrong at 512^3), round 1 the rest; whether anything still moved after
                // the last round is read with the caller's next synchronisation (two host round trips of ~35 us less per call)
                if (iter + 1 < opt_rounds && nsub > 1) { ++iter; continue; }
                HIPCHK(hipMemcpyAsync(&ctx->hdec_res[1], sm + SM_CHANGED, 8, hipMemcpyDeviceToHost, st));
                break;
            }
            unsigned changed = 0;
            HIPCHK(hipMemcpyAsync(&changed, sm + SM_CHANGED, 4, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            if (tune_int("SZ_HIP_HDEC_TRACE", 0)) fprintf(stderr, "[szhip] Huffman decode round %lld: %u of %lld sub-sequence starts moved\n", (long long)iter, changed, (long long)nsub);
            if (!changed) break;
            if (++iter > nsub + 2) FAIL(SZHIP_ERR_INTERNAL, "Huffman decode did not converge");
        }
        TRY(scan_u64(ctx, (const u64 *)ctx->counts.p, nsub, (u64 *)ctx->offs.p, sm + SM_TOTAL_SYM));
        HIPCHK(hipMemcpyAsync(&ctx->hdec_res[0], sm + SM_TOTAL_SYM, 8, hipMemcpyDeviceToHost, st));   // (pinned: a copy into pageable memory blocks the host)
        hipLaunchKernelGGL(k_hdec_write, dim3(gsub), dim3(256), lds_write, st, a, (const u64 *)ctx->offs.p, d_out_codes, n);
        HIPCHK(hipGetLastError());
    }

    return SZHIP_OK;
}

// everything the host reads from an SZ 2.1 regression-type stream before the unpredictable values
template <class T> struct dec_header {
    T eb = 0, mean = 0;
    unsigned intervals = 0; int use_mean = 0, n_nodes = 0, single_symbol = -1;
    size_t reg_count = 0, ind_off = 0, unpred_off = 0, pay_off = 0; uint64_t total_unpred = 0;
    std::vector<T> coef; std::vector<uint32_t> dtab;   // coef: decoded regression coefficients, compact [4][reg_count]
};

// returns 0 = parsed, 1 = needs at least *need bytes of the stream on the host, < 0 = malformed (message in err)
template <class T>
int parse_header(const unsigned char *hs, size_t avail, size_t stream_len, size_t body_off, size_t nb, int want_block_size,
                 dec_header<T> &H, size_t *need, char *err, size_t errlen)
{
    const int ncoef = want_block_size == SZH_BLOCK_SIZE_2D ? 3 : 4;
    const int is_double = sizeof(T) == 8;
#define PFAIL(code, ...) do { snprintf(err, errlen, __VA_ARGS__); if (hf) szhost_huff_free(hf); return (code); } while (0)
#define NEED(k) do { const size_t end_ = (size_t)(q - hs) + (size_t)(k); if (end_ > stream_len) PFAIL(SZHIP_ERR_STREAM, "truncated stream"); \
                     if (end_ > avail) { *need = std::min(stream_len, end_ + need_more); if (hf) szhost_huff_free(hf); return 1; } } while (0)
    size_t need_more = 0;                      // what a longer prefix should hold beyond the bytes asked for (the coefficient sections still to come)
    szhost_huff *hf = nullptr;
    const unsigned char *q = hs + body_off;
    NEED(4 + sizeof(T) + 12);
    const unsigned block_size = szhost_get_u32be(q); q += 4;
    if ((int)block_size != want_block_size) PFAIL(SZHIP_ERR_UNSUP, "block size %u", block_size);
    H.eb = is_double ? (T)szhost_get_f64be(q) : (T)szhost_get_f32be(q); q += sizeof(T);
    H.intervals = szhost_get_u32be(q); q += 4;
    const unsigned tree_size = szhost_get_u32be(q); q += 4;
    const int node_count = (int)szhost_get_u32be(q); q += 4;
    if (H.intervals < 4 || H.intervals > 65536) PFAIL(SZHIP_ERR_STREAM, "bad interval count %u", H.intervals);
    NEED(tree_size);
    if (node_count <= 0 || szhost_huff_serial_size(node_count) > tree_size) PFAIL(SZHIP_ERR_STREAM, "bad Huffman tree size");
    hf = szhost_huff_from_bytes(2 * (int)H.intervals, q, node_count);
    if (!hf) PFAIL(SZHIP_ERR_STREAM, "bad Huffman tree");
    q += tree_size;
    NEED(1 + sizeof(T));
    H.use_mean = *q++;
    memcpy(&H.mean, q, sizeof(T)); q += sizeof(T);
    const size_t ind_bytes = (nb - 1) / 8 + 1;
    NEED(ind_bytes);
    {   // regression blocks = zero bits among the first nb
        size_t ones = 0, full = nb / 8;
        for (size_t i = 0; i < full; ++i) ones += (size_t)__builtin_popcount(q[i]);
        for (size_t b = full * 8; b < nb; ++b) ones += (q[b >> 3] >> (7 - (b & 7))) & 1;
        H.reg_count = nb - ones;
    }
    H.ind_off = (size_t)(q - hs);
    q += ind_bytes;
    H.coef.clear();
    if (H.reg_count > 0) {
        // The four sections are LOCATED first (their lengths stand in front of them) and decoded afterwards, each on its own thread for a large
        // array: located first, a prefix of the stream that turns out too short costs no decoding (round 4 decoded the sections again for every
        // longer prefix it fetched: 3 x at the M-field's 3.4 MB header), and the bit-serial Huffman decode of ~300 000 codes is the long part.
        std::vector<int> ccodes[4]; int *cptr[4]; int crad[4]; double cprec[4]; const unsigned char *cun[4];
        const unsigned char *tree_at[4], *pay_at[4]; int cnc_of[4]; size_t enc_of[4]; unsigned cu_of[4];
        for (int e = 0; e < ncoef; ++e) {
            NEED(sizeof(T) + 12);
            cprec[e] = is_double ? szhost_get_f64be(q) : (double)szhost_get_f32be(q); q += sizeof(T);
            crad[e] = (int)szhost_get_u32be(q); q += 4;
            const unsigned ts = szhost_get_u32be(q); q += 4;
            const int cnc = (int)szhost_get_u32be(q); q += 4;
            NEED(ts);
            if (cnc <= 0 || crad[e] <= 0 || crad[e] > 32768 || szhost_huff_serial_size(cnc) > ts) PFAIL(SZHIP_ERR_STREAM, "bad coefficient tree size");
            tree_at[e] = q; cnc_of[e] = cnc;
            q += ts;
            NEED(8);
            const uint64_t enc64 = szhost_get_u64be(q); q += 8;
            if (enc64 > (uint64_t)(stream_len - (size_t)(q - hs))) PFAIL(SZHIP_ERR_STREAM, "truncated stream");   // before any size arithmetic on it
            enc_of[e] = (size_t)enc64;
            need_more = (size_t)(ncoef - 1 - e) * (enc_of[e] + enc_of[e] / 2 + ts + 65536) + 65536;   // (the sections are of similar size: one more fetch, not three)
            NEED(enc_of[e]);
            need_more = 0;
            pay_at[e] = q;
            q += enc_of[e];
            NEED(4);
            cu_of[e] = szhost_get_u32be(q); q += 4;
            NEED((size_t)cu_of[e] * sizeof(T));
            cun[e] = q; q += (size_t)cu_of[e] * sizeof(T);
        }
        NEED(8);                                   // (the word after the sections: nothing below asks for more of the stream)
        int sec_rc[4] = {0, 0, 0, 0};              // 1 bad tree, 2 payload too short, 3 too few verbatim values
        size_t sec_zeros[4] = {0, 0, 0, 0};
        auto decode_section = [&](int e) {
            szhost_huff *ch = szhost_huff_from_bytes(4 * crad[e], tree_at[e], cnc_of[e]);
            if (!ch) { sec_rc[e] = 1; return; }
            ccodes[e].resize(H.reg_count);
            const int ok = szhost_huff_decode_i32(ch, pay_at[e], enc_of[e], H.reg_count, ccodes[e].data());   // bounded by the section's own length
            szhost_huff_free(ch);
            if (!ok) { sec_rc[e] = 2; return; }
            // every zero code takes one verbatim coefficient (szd_float.c:5809-5820): the list must hold them all
            size_t zeros = 0;
            for (size_t i = 0; i < H.reg_count; ++i) zeros += ccodes[e][i] == 0;
            sec_zeros[e] = zeros;
            if (zeros > cu_of[e]) sec_rc[e] = 3;
        };
        if (H.reg_count >= 16384) {
            std::vector<std::thread> th;
            for (int e = 1; e < ncoef; ++e) th.emplace_back(decode_section, e);
            decode_section(0);
            for (auto &t : th) t.join();
        } else {
            for (int e = 0; e < ncoef; ++e) decode_section(e);
        }
        for (int e = 0; e < ncoef; ++e) {
            if (sec_rc[e] == 1) PFAIL(SZHIP_ERR_STREAM, "bad coefficient tree");
            if (sec_rc[e] == 2) PFAIL(SZHIP_ERR_STREAM, "coefficient payload too short");
            if (sec_rc[e] == 3) PFAIL(SZHIP_ERR_STREAM, "coefficient section lists %u verbatim values, codes need %zu", cu_of[e], sec_zeros[e]);
            cptr[e] = ccodes[e].data();
        }
        // compact [4][reg_count] in scan order; the device scatters them to the blocks (k_move_coef)
        H.coef.assign(H.reg_count * 4, (T)0);
        const std::vector<unsigned char> all_reg(H.reg_count, 0);
        // 2-D planes {a, b, c} are carried as {0, a, b, c}
        szhost_coeff_unchain(is_double, H.coef.data() + (ncoef == 3 ? H.reg_count : 0), all_reg.data(), H.reg_count, cptr, crad, cprec, cun, ncoef);
    }
    NEED(8);
    memcpy(&H.total_unpred, q, 8); q += 8;
    H.unpred_off = (size_t)(q - hs);
    if (H.total_unpred > (stream_len - H.unpred_off) / sizeof(T)) PFAIL(SZHIP_ERR_STREAM, "truncated stream");
    H.pay_off = H.unpred_off + (size_t)H.total_unpred * sizeof(T);
    H.dtab.resize((size_t)hf->n_nodes * 2);
    szhost_huff_decode_table(hf, H.dtab.data());
    H.single_symbol = hf->t[0] ? (int)hf->C[0] : -1;
    H.n_nodes = hf->n_nodes;
    szhost_huff_free(hf);
    return 0;
#undef NEED
#undef PFAIL
}

template <class T>
int decompress_impl(szhip_ctx *ctx, const unsigned char *stream_in, int stream_on_device, size_t stream_len, size_t body_off,
                    size_t r0, size_t r1, size_t r2, void *out, int out_on_device, szhip_stats *stats)
{
    const szh_geom3 G = r0 == 0 ? szh_make_geom2((int)r1, (int)r2) : szh_make_geom3((int)r0, (int)r1, (int)r2);
    const int64_t n = G.n, nb = G.nblocks;
    const double t_begin = now_ms();
    double host_ms = 0;
    hipStream_t st = ctx->stream;
    szhip_stats S; memset(&S, 0, sizeof(S));
    S.n_elements = (uint64_t)n; S.n_blocks = (uint64_t)nb;

    // ---- stream on both sides: the device gets the whole stream; the host only needs the header (everything up to the
    //      unpredictable values), fetched from a device-resident stream in growing prefixes
    std::vector<unsigned char> hcopy;
    const unsigned char *hs = stream_in;
    size_t avail = stream_len;
    TRY(ensure(ctx, ctx->stream_buf, stream_len + 64));
    unsigned char *d_stream = (unsigned char *)ctx->stream_buf.p;
    if (stream_on_device) {
        if (stream_in != d_stream) HIPCHK(hipMemcpyAsync(d_stream, stream_in, stream_len, hipMemcpyDeviceToDevice, st));
        avail = 0;
    } else {
        TRY(staged_copy(ctx, d_stream, stream_in, stream_len, true));
    }
    HIPCHK(hipMemsetAsync(d_stream + stream_len, 0, 64, st)); // the bit reader may look a few bytes past the end
    HIPCHK(hipEventRecord(ctx->ev[0], st));

    // ---- header (szd_float.c:3491-3587)
    double h0 = now_ms();
    dec_header<T> H;
    for (size_t want = std::min<size_t>(stream_len, (size_t)1 << 20);;) {
        if (stream_on_device && want > avail) {
            TRY(ensure_pinned(ctx, want));
            HIPCHK(hipMemcpyAsync(ctx->pinned, d_stream, want, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            hs = (const unsigned char *)ctx->pinned; avail = want;
        }
        size_t need = 0;
        const int rc = parse_header<T>(hs, avail, stream_len, body_off, (size_t)nb, G.block_size, H, &need, ctx->err, sizeof(ctx->err));
        if (rc == 0) break;
        if (rc < 0) { fprintf(stderr, "szhip: %s\n", ctx->err); return rc; }
        want = std::min<size_t>(stream_len, std::max<size_t>(need, avail * 2)); // rc == 1: more bytes needed
    }
    const T eb = H.eb, mean = H.mean;
    const unsigned intervals = H.intervals;
    const int use_mean = H.use_mean, n_nodes = H.n_nodes, single_symbol = H.single_symbol;
    const size_t reg_count = H.reg_count, unpred_off = H.unpred_off, pay_off = H.pay_off;
    const uint64_t total_unpred = H.total_unpred;
    std::vector<T> &hcoef = H.coef;
    std::vector<uint32_t> &dtab = H.dtab;
    const u64 total_bits = (u64)(stream_len - pay_off) * 8;
    S.intervals = intervals; S.use_mean = use_mean; S.n_reg_blocks = reg_count; S.n_unpred = total_unpred;
    host_ms += now_ms() - h0;

    TRY(ensure(ctx, ctx->small, SM_COUNT * 8));
    u64 *sm = (u64 *)ctx->small.p;
    HIPCHK(hipMemsetAsync(sm, 0, SM_COUNT * 8, st));
    // The inverse sweep.  Where k_ribbon applies (3-D, no regression block) it runs in RIBBON ORDER throughout (SZ_HIP_RIBBON_DEC=2, the
    // default): k_permute<1> writes the codes the way the forward sweep does, k_unpred<1> drops the unpredictable values into a ribbon-order
    // value array (the rest of it stays uninitialised: the sweep only looks at a value where the code says so), the sweep reads and writes
    // 64 lanes x 16 contiguous bytes per instruction, and k_unribbon turns the result into the array (LDS transposition, 256-byte runs).
    // 512^3 float: 0.86 + 0.37 ms against k_pencil's 1.47 - 1.53.  What led there (tools/gpu_dec_dbg.sh, tools/gpu_rb_trace.py RB_DEC=1):
    // the sweep alone takes 0.59 ms, with natural-order loads 1.03 ms; natural-order result stores from the compute wavefronts 2.5 ms
    // (their one memory counter makes every wait for a trip's inputs wait for the previous trip's scattered stores), transposed into
    // 64-byte pieces 2.2 ms, through a STORE helper wavefront 1.6 - 1.7 ms (= SZ_HIP_RIBBON_DEC=1, kept and tested: that wavefront is busy
    // 85 % of the time -- a store instruction that touches 16 rows takes it ~280 cycles -- and a second one does not fit the register file).
    // SZ_HIP_RIBBON_DEC=0: k_pencil on natural-order codes (rounds 1 - 2; still the path of arrays with regression blocks, 2-D, SZ 1.4).
    szh_rb_layout rbl = {0, 0, 0, 0, 0, 0};
    size_t nat_elems = (size_t)n;
    const bool dec_beam = beam_applies<T>(G, out_on_device ? (const void *)out : (const void *)nullptr, reg_count);   // (the context's own buffer is aligned)
    int dec_ribbon_mode = !dec_beam && ribbon_applies<T>(G, reg_count) ? tune_int("SZ_HIP_RIBBON_DEC", 2) : 0;           // 1: natural-order values through the STORE wavefront
    if (dec_ribbon_mode == 2 && (double)szh_rb_steps_of<T>(G.g2.count) * szh_rb_shape<T>::W * szh_rb_shape<T>::R * 64.0 * sizeof(T) >= 2.0e9)
        dec_ribbon_mode = 0;                   // (a tile's stretch of the value array is addressed with 32-bit offsets)
    const bool dec_ribbon = dec_ribbon_mode != 0;                                                            // 2: ribbon-order values + k_unribbon
    if (dec_ribbon) {
        using RS = szh_rb_shape<T>;
        rbl.on = 1; rbl.nTJ = (G.g1.count + 63) / 64; rbl.NT = szh_rb_steps_of<T>(G.g2.count); rbl.W = RS::W; rbl.R = RS::R; rbl.U = RS::U;
        const size_t tiles = (size_t)((G.g0.count + RS::W * RS::R - 1) / (RS::W * RS::R)) * rbl.nTJ;
        nat_elems = tiles * (size_t)szh_rb_tile_elems(rbl);
    }
    TRY(ensure(ctx, ctx->codes_nat, nat_elems * 2 + 64));
    TRY(ensure(ctx, ctx->codes_blk, (size_t)n * 2 + 64));
    uint16_t *d_nat = (uint16_t *)ctx->codes_nat.p, *d_blk = (uint16_t *)ctx->codes_blk.p;

    // ---- Huffman decode of the type array
    u64 total_sym = 0;
    TRY(huff_decode_device(ctx, sm, d_stream + pay_off, (unsigned)std::min<size_t>(pay_off, 4096), total_bits, dtab, n_nodes, single_symbol, n, d_blk, &total_sym));

    // ---- natural order, unpredictable values into the output array
    const int ncols = G.g0.num * G.g1.num;
    TRY(ensure(ctx, ctx->col_zeros, (size_t)ncols * 4));
    TRY(ensure(ctx, ctx->col_zeros64, (size_t)ncols * 8));
    TRY(ensure(ctx, ctx->col_off, (size_t)ncols * 8));
    HIPCHK(hipMemsetAsync(ctx->col_zeros.p, 0, (size_t)ncols * 4, st));
    int perm_segb = 1, perm_nseg = 1;
    {
        const int segb = choose_segb(G, 2, tune_int("SZ_HIP_PERM_TILE_KB", 32) * 1024);
        const int nseg = (G.g2.num + segb - 1) / segb;
        TRY(ensure(ctx, ctx->zcnt, (size_t)ncols * nseg * 4));
        TRY(ensure(ctx, ctx->zpos, (size_t)ncols * nseg * SZH_ZCAP * 4));
        hipLaunchKernelGGL((k_permute<1>), dim3(ncols, std::min(nseg, std::max(1, tune_int("SZ_HIP_PERM_Y", nseg)))), dim3(256), tile_bytes(G, segb, 2), st, G, (const uint16_t *)d_blk, d_nat,
                           (unsigned *)ctx->col_zeros.p, segb, (unsigned *)ctx->zcnt.p, (unsigned *)ctx->zpos.p, rbl, (unsigned *)nullptr, 0u, 0, 0);
        perm_segb = segb; perm_nseg = nseg;
        HIPCHK(hipGetLastError());
    }
    hipLaunchKernelGGL(k_u32_to_u64, dim3((ncols + 255) / 256), dim3(256), 0, st, (const unsigned *)ctx->col_zeros.p, (int64_t)ncols,
                       (u64 *)ctx->col_zeros64.p);
    TRY(scan_u64(ctx, (const u64 *)ctx->col_zeros64.p, ncols, (u64 *)ctx->col_off.p, sm + SM_TOTAL_UNPRED));
    u64 zeros_found = 0;
    HIPCHK(hipMemcpyAsync(&zeros_found, sm + SM_TOTAL_UNPRED, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HDEC_CHECK(ctx);
    total_sym = ctx->hdec_res[0];
    if ((int64_t)total_sym < n) FAIL(SZHIP_ERR_STREAM, "Huffman payload holds %llu symbols, need %lld", (unsigned long long)total_sym, (long long)n);
    if (zeros_found != total_unpred) FAIL(SZHIP_ERR_STREAM, "stream lists %llu unpredictable values, codes need %llu",
                                          (unsigned long long)total_unpred, (unsigned long long)zeros_found);
    T *d_out = (T *)out;
    if (!out_on_device) { TRY(ensure(ctx, ctx->out, (size_t)n * sizeof(T))); d_out = (T *)ctx->out.p; }
    T *d_sweep = d_out;                        // what the inverse sweep works on: the output array, or (mode 2) a ribbon-order value array
    if (dec_ribbon_mode == 2) { TRY(ensure(ctx, ctx->rb_vals, nat_elems * sizeof(T) + 64)); d_sweep = (T *)ctx->rb_vals.p; }
    if (total_unpred > 0) {
        TRY(ensure(ctx, ctx->unpred, (size_t)total_unpred * sizeof(T)));
        HIPCHK(hipMemcpyAsync(ctx->unpred.p, d_stream + unpred_off, (size_t)total_unpred * sizeof(T), hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL((k_unpred<T, 1>), dim3(ncols), dim3(256), 0, st, G, (const uint16_t *)d_blk, (const unsigned *)ctx->col_zeros.p,
                           (const u64 *)ctx->col_off.p, (const T *)nullptr, (T *)ctx->unpred.p, d_sweep, (const unsigned *)ctx->zcnt.p,
                           (const unsigned *)ctx->zpos.p, perm_segb, perm_nseg, dec_ribbon_mode == 2 ? rbl : szh_rb_layout{0, 0, 0, 0, 0, 0});
        HIPCHK(hipGetLastError());
    }
    TRY(ensure(ctx, ctx->coef, (size_t)nb * 4 * sizeof(T)));
    TRY(ensure(ctx, ctx->blk_lor, (size_t)nb));

    hipLaunchKernelGGL(k_unpack_lor, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, (const uint8_t *)(d_stream + H.ind_off), nb,
                       (uint8_t *)ctx->blk_lor.p);
    HIPCHK(hipGetLastError());
    if (reg_count > 0) {
        TRY(ensure(ctx, ctx->reg_flags, (size_t)nb * 8));
        TRY(ensure(ctx, ctx->reg_rank, (size_t)nb * 8));
        TRY(ensure(ctx, ctx->coef_compact, reg_count * 4 * sizeof(T)));
        TRY(ensure_pinned3(ctx, hcoef.size() * sizeof(T)));    // (asynchronous copies from pageable memory stall later calls, see compress_impl)
        memcpy(ctx->pinned3, hcoef.data(), hcoef.size() * sizeof(T));
        HIPCHK(hipMemcpyAsync(ctx->coef_compact.p, ctx->pinned3, hcoef.size() * sizeof(T), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_reg_flags, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, (const uint8_t *)ctx->blk_lor.p, nb, (u64 *)ctx->reg_flags.p);
        TRY(scan_u64(ctx, (const u64 *)ctx->reg_flags.p, nb, (u64 *)ctx->reg_rank.p, sm + SM_SCRATCH));
        hipLaunchKernelGGL((k_move_coef<T, 1>), dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, (const uint8_t *)ctx->blk_lor.p,
                           (const u64 *)ctx->reg_rank.p, nb, (int64_t)reg_count, (T *)ctx->coef.p, (T *)ctx->coef_compact.p);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(ctx->ev[1], st));

    // ---- reconstruct: the wavefront kernel
    int nI, nJ, ntiles;
    using TS = szh_tile_shape<T>;
    TRY(prepare_pencil(ctx, G, szh_gran<T>::NW, TS::TPI, TS::TPJ, &nI, &nJ, &ntiles));
    {
        szh_qargs<T> a; memset(&a, 0, sizeof(a));
        a.G = G; a.data = nullptr; a.out = d_out; a.codes = d_nat; a.blk_lor = (const uint8_t *)ctx->blk_lor.p; a.coef = (const T *)ctx->coef.p; a.coef_stride = nb;
        a.eb = eb; a.recip = 1 / eb; a.mean = mean; a.cap = (int)intervals; a.radius = (int)intervals / 2; a.use_mean = use_mean;
        a.faceI = (szh_u64 *)ctx->faceI.p; a.faceJ = (szh_u64 *)ctx->faceJ.p; a.epoch = ++ctx->epoch;
        a.nI = nI; a.nJ = nJ; a.order = (const unsigned *)ctx->order.p; a.no_reg = reg_count == 0 && tune_int("SZ_HIP_NO_REG_HINT", 1);
        a.ticket = (unsigned *)(sm + SM_TICKET); a.err = (unsigned *)(sm + SM_ERR); a.ticket_mode = ctx->ticket_atomic ? 0 : tune_int("SZ_HIP_TICKET_MODE", 2);
        a.progress = (szh_u64 *)ctx->progress.p; a.backoff = tune_int("SZ_HIP_BACKOFF", 4); a.wide = tune_int("SZ_HIP_WIDE", 1) && (double)TS::TPI * nJ * 9.0 * (double)G.g2.count * szh_gran<T>::NW * 8.0 < 4.0e9 && (double)TS::TPJ * 9.0 * (double)G.g2.count * szh_gran<T>::NW * 8.0 < 4.0e9;
        a.trace = tune_int("SZ_HIP_TRACE", 0) ? (szh_u64 *)ctx->trace.p : nullptr;
        a.dbg = tune_int("SZ_HIP_DBG", 0); a.trace_tile = tune_int("SZ_HIP_TRACE_TILE", 1);
        HIPCHK(hipEventRecord(ctx->ev[2], st));
        if (dec_beam) { TRY((launch_beam<T, true>(ctx, G, a, st, reg_count))); S.quant_kernel = 2; }
        else if (dec_ribbon) {
            a.codes_ribbon = dec_ribbon_mode == 2 ? 2 : 1;
            if (dec_ribbon_mode == 2) a.out = d_sweep;
            TRY((launch_ribbon<T, true>(ctx, G, a, st)));
            S.quant_kernel = 1;
            if (dec_ribbon_mode == 2) {            // the results into the array
                const int nchunk = (rbl.NT + SZH_UR_STEPS - 1) / SZH_UR_STEPS;
                const size_t tiles = (size_t)((G.g0.count + rbl.W * rbl.R - 1) / (rbl.W * rbl.R)) * rbl.nTJ;
                hipLaunchKernelGGL((k_unribbon<T>), dim3((unsigned)(tiles * (rbl.W * rbl.R / 4) * nchunk)), dim3(256), 0, st, G, rbl, (const T *)d_sweep, d_out);
                HIPCHK(hipGetLastError());
            }
        }
        else {
        const unsigned pgrid = pencil_grid(ctx, a, ntiles);
        hipLaunchKernelGGL((k_pencil<T, true>), dim3(pgrid), dim3((TS::TPI * TS::TPJ + 2) * 64), 0, st, a);
        HIPCHK(hipGetLastError());
        }
        HIPCHK(hipEventRecord(ctx->ev[3], st));
        S.quant_kernel_launches = 1;
    }
    unsigned kerr = 0;
    HIPCHK(hipMemcpyAsync(&kerr, sm + SM_ERR, 4, hipMemcpyDeviceToHost, st));
    if (!out_on_device) TRY(staged_copy(ctx, out, d_out, (size_t)n * sizeof(T), false));
    HIPCHK(hipEventRecord(ctx->ev[4], st));
    HIPCHK(hipStreamSynchronize(st));
    if (kerr) { ctx->wave_timeout = true; FAIL(SZHIP_ERR_INTERNAL, "wavefront kernel: halo wait timed out"); }
    float ms = 0;
    hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]); S.ms_entropy = ms;
    hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3]); S.ms_quant = ms;
    S.ms_host = host_ms; S.ms_total = now_ms() - t_begin; S.out_bytes = (uint64_t)n * sizeof(T);
    if (stats) *stats = S;
    return SZHIP_OK;
}

// =====================================================================================================================
// SZ 1.4 ("no regression") 3-D path: SZ_compress_float_3D_MDQ (sz/src/sz_float.c:946-1415) + convertTDPStoFlatBytes_float
// (TightDataPointStorageF.c:379-479,590-663), and the inverse decompressDataSeries_float_3D (szd_float.c:600-1138).
// Whole-array Lorenzo on the same wavefront kernel (fmt = 1); the code array is already in stream order; the "exact" values are
// compacted in scan order and packed by k_exact_*.
// =====================================================================================================================

// required length (bits) of an exact value and the median it is taken against (sz_float.c:45-56 / sz_double.c:44-55)
template <class T> int req_length(double eb, T range, T *median);
template <> int req_length<float>(double eb, float range, float *median)
{
    const float half = range / 2; uint32_t u; memcpy(&u, &half, 4);
    uint64_t e; memcpy(&e, &eb, 8);
    const int reqExpo = (int)((e >> 52) & 0x7ff) - 1023, radExpo = (int)((u >> 23) & 0xff) - 127;
    int req = 9 + radExpo - reqExpo + 1;
    if (req < 9) req = 9;
    if (req > 32) { req = 32; *median = 0; }
    return req;
}
template <> int req_length<double>(double eb, double range, double *median)
{
    const double half = range / 2; uint64_t u; memcpy(&u, &half, 8);
    uint64_t e; memcpy(&e, &eb, 8);
    const int reqExpo = (int)((e >> 52) & 0x7ff) - 1023, radExpo = (int)((u >> 52) & 0x7ff) - 1023;
    int req = 12 + radExpo - reqExpo;
    if (req < 12) req = 12;
    if (req > 64) { req = 64; *median = 0; }
    return req;
}

template <class T>
int launch_pencil14(szhip_ctx *ctx, const szh_geom3 &G, u64 *sm, bool dec, const T *d_in, T *d_out, uint16_t *d_codes, T eb, unsigned intervals,
                    T median, int ign_bits, const msst_tab *mt = nullptr, int ndim = 3)
{   // mt != nullptr: the table-driven point-wise-relative quantiser (fmt 2) instead of the SZ 1.4 one
    hipStream_t st = ctx->stream;
    int nI, nJ, ntiles;
    using TS = szh_tile_shape<T>;
    TRY(prepare_pencil(ctx, G, szh_gran<T>::NW, TS::TPI, TS::TPJ, &nI, &nJ, &ntiles));
    szh_qargs<T> a; memset(&a, 0, sizeof(a));
    a.G = G; a.data = d_in; a.out = d_out; a.codes = d_codes; a.blk_lor = nullptr; a.coef = nullptr; a.coef_stride = 0;
    a.eb = eb; a.recip = 1 / eb; a.mean = 0; a.cap = (int)intervals; a.radius = (int)intervals / 2; a.use_mean = 0;
    a.fmt = 1; a.median = median; a.ign_bits = ign_bits;
    if (mt) {
        a.fmt = 2; a.ptab = mt->ptab; a.cells = mt->cells; a.tbase = mt->base; a.trange = mt->range; a.tbits = mt->bits;
        a.f32arith = (sizeof(T) == 4 && ndim == 2) ? 1 : 0; a.ndim3 = ndim == 3 ? 1 : 0;
    }
    a.faceI = (szh_u64 *)ctx->faceI.p; a.faceJ = (szh_u64 *)ctx->faceJ.p; a.epoch = ++ctx->epoch;
    a.nI = nI; a.nJ = nJ; a.order = (const unsigned *)ctx->order.p;
    a.ticket = (unsigned *)(sm + SM_TICKET); a.err = (unsigned *)(sm + SM_ERR); a.ticket_mode = ctx->ticket_atomic ? 0 : tune_int("SZ_HIP_TICKET_MODE", 2);
    a.progress = (szh_u64 *)ctx->progress.p; a.backoff = tune_int("SZ_HIP_BACKOFF", 4); a.wide = tune_int("SZ_HIP_WIDE", 1) && (double)TS::TPI * nJ * 9.0 * (double)G.g2.count * szh_gran<T>::NW * 8.0 < 4.0e9 && (double)TS::TPJ * 9.0 * (double)G.g2.count * szh_gran<T>::NW * 8.0 < 4.0e9;
    a.trace = nullptr; a.dbg = 0;
    HIPCHK(hipEventRecord(ctx->ev[2], st));
    const unsigned pgrid = pencil_grid(ctx, a, ntiles);
    if (mt && dec) hipLaunchKernelGGL((k_pencil<T, true, true>), dim3(pgrid), dim3((TS::TPI * TS::TPJ + 2) * 64), 0, st, a);
    else if (mt) hipLaunchKernelGGL((k_pencil<T, false, true>), dim3(pgrid), dim3((TS::TPI * TS::TPJ + 2) * 64), 0, st, a);
    else if (dec) hipLaunchKernelGGL((k_pencil<T, true>), dim3(pgrid), dim3((TS::TPI * TS::TPJ + 2) * 64), 0, st, a);
    else hipLaunchKernelGGL((k_pencil<T, false>), dim3(pgrid), dim3((TS::TPI * TS::TPJ + 2) * 64), 0, st, a);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ctx->ev[3], st));
    return SZHIP_OK;
}


// =====================================================================================================================
// Point-wise relative bounds, table-driven ("MSST19") form: host tables and the sweep (szh_msst.h)
// =====================================================================================================================
struct MsstHostTab { std::vector<double> ptab; std::vector<uint16_t> cells; u64 base = 0, range = 0; int bits = 0; };
static inline u64 f64_bits(double v) { u64 u; memcpy(&u, &v, 8); return u; }
// precisionTable (sz_float.c:2288-2293) and MultiLevelCacheTableWideIntervalBuild (MultiLevelCacheTableWideInterval.c:53-107); pow is the host's
static int msst_build_tab(MsstHostTab &t, double precision, unsigned count, int plus_bits, bool want_cells)
{
    const int radius = (int)count / 2;
    t.ptab.resize(count);
    const double inv = 2.0 - pow(2, -plus_bits);
    for (unsigned i = 0; i < count; ++i) t.ptab[i] = pow(1 + precision, inv * ((int)i - radius));
    if (!want_cells) return 0;
    const uint16_t bits = (uint16_t)((uint16_t)(-((f64_bits(precision) >> 52) - 1023)) + plus_bits);
    if (bits < 1 || bits > 24) return -1;
    t.bits = bits;
    const double bottom = t.ptab[1] / (1 + precision), top = t.ptab[count - 1] / (1 - precision);
    const uint16_t base = (uint16_t)(f64_bits(bottom) >> 52), topi = (uint16_t)(f64_bits(top) >> 52);
    if (topi < base || (u64)(topi - base + 1) << bits > (1ull << 28)) return -1;
    t.base = base; t.range = (u64)(topi - base);
    t.cells.assign((size_t)(t.range + 1) << bits, 0);
    auto rebuild = [&](uint16_t expo, u64 manti) { u64 u = (u64)expo << 52; u += manti << (52 - bits); double r; memcpy(&r, &u, 8); return r; };
    uint32_t index = 0; bool flag = false;
    for (uint32_t i = 0; i <= (uint32_t)(topi - base); ++i) {
        const uint16_t expo = (uint16_t)(i + base);
        for (uint32_t j = 0; j < (1u << bits); ++j) {
            const double sb = rebuild(expo, j), stp = rebuild(expo, (u64)j + 1);
            const double bb = t.ptab[index] / (1 + precision), tb = t.ptab[index] / (1 - precision);
            uint16_t &cell = t.cells[((size_t)i << bits) + j];
            if (stp < tb && sb > bb) { cell = (uint16_t)index; flag = true; }
            else if (flag && index < count - 1) { ++index; cell = (uint16_t)index; }
            else cell = 0;
        }
    }
    return 0;
}
// the histogram bin of one sample: `radiusIndex = (uint64_t)fabs(log2(pred_err)/divider+0.5)` with gcc's x86-64 double -> uint64 sequence
static inline u64 msst_radius_index(double pe, double divider)
{
    const double v = fabs(log2(pe) / divider + 0.5);
    if (v != v) return 0x8000000000000000ull;
    if (v >= 9223372036854775808.0) { const double w = v - 9223372036854775808.0; return (w < 9223372036854775808.0 ? (u64)(int64_t)w : 0x8000000000000000ull) ^ 0x8000000000000000ull; }
    return (u64)(int64_t)v;
}
static unsigned msst_pick_intervals(const std::vector<u64> &hist, u64 total, float pred_threshold, unsigned floor_)
{
    const unsigned max_radius = (unsigned)hist.size();
    const size_t target = (size_t)((float)total * pred_threshold);
    size_t sum = 0; unsigned i = 0;
    for (; i < max_radius; ++i) { sum += hist[i]; if (sum > target) break; }
    if (i >= max_radius) i = max_radius - 1;
    unsigned p2 = 2 * (i + 1); p2 -= 1; p2 |= p2 >> 1; p2 |= p2 >> 2; p2 |= p2 >> 4; p2 |= p2 >> 8; p2 |= p2 >> 16; p2 += 1;
    return p2 < floor_ ? floor_ : p2;
}
// optimize_intervals_float_{1,2,3}D_opt_MSST19 (sz_float.c:4468, :4518, :4578; doubles sz_double.c:4163-).  The quotients come from the device
// (k_msst_sample: IEEE adds and divisions, the same bits as the host's), the logarithm and the bin from the host (glibc's log2, as in the reference).
// The reference's walk skips samples that are zero WITHOUT advancing its column counter; zeros are left only when the array's first element is zero
// (nearZero = 0: nothing replaces them) -- then the walk is done here, on a host copy of the array.
template <class T>
int msst_intervals(szhip_ctx *ctx, const szh_geom3 &G, int ndim, const T *d_in, const szhip_params *prm, double precision, bool zeros_left, unsigned *out)
{
    hipStream_t st = ctx->stream;
    const unsigned max_radius = prm->max_quant_intervals / 2;
    const int sd = prm->sample_distance;
    const double divider = (double)(T)(log2(1 + precision) * 2);
    std::vector<u64> hist(max_radius, 0);
    u64 total = 0;
    auto add = [&](double pe) { u64 ri = msst_radius_index(pe, divider); if (ri >= max_radius) ri = max_radius - 1; ++hist[ri]; ++total; };
    const int64_t n = G.n, r2 = G.g2.count;
    if (zeros_left) {
        std::vector<T> h((size_t)n);
        HIPCHK(hipMemcpyAsync(h.data(), d_in, (size_t)n * sizeof(T), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        const T *d = h.data();
        const int64_t r3 = r2, r23 = ndim == 3 ? G.d0 : 0, rows = ndim == 3 ? G.g1.count : 0;
        if (ndim == 1) {
            for (int64_t pos = 2; pos < n; pos += sd) { if (d[pos] == 0) continue; add(fabs((double)d[pos] / (double)d[pos - 1])); }
        } else if (ndim == 2) {
            int64_t oc = sd - 1, n1 = 1, pos = r3 + oc;
            while (pos < n) {
                if (d[pos] == 0) { pos += sd; continue; }
                const T pv = d[pos - 1] + d[pos - r3] - d[pos - r3 - 1];
                add(fabs((double)(T)(pv / d[pos])));
                oc += sd;
                if (oc >= r3) { ++n1; const int64_t oc2 = n1 % sd; pos += (r3 + sd - oc) + (sd - oc2); oc = sd - oc2; if (oc == 0) ++oc; }
                else pos += sd;
            }
        } else {
            int64_t oc = sd - 2, n1 = 1, n2 = 1, pos = r23 + r3 + oc;
            while (pos < n) {
                if (d[pos] == 0) { pos += sd; continue; }
                const T pv = d[pos - 1] + d[pos - r3] + d[pos - r23] - d[pos - 1 - r23] - d[pos - r3 - 1] - d[pos - r3 - r23] + d[pos - r3 - r23 - 1];
                add(fabs((double)(T)(d[pos] / pv)));
                oc += sd;
                if (oc >= r3) {
                    ++n2;
                    if (n2 == rows) { ++n1; n2 = 1; pos += r3; }
                    const int64_t oc2 = (n1 + n2) % sd;
                    pos += (r3 + sd - oc) + (sd - oc2); oc = sd - oc2; if (oc == 0) ++oc;
                } else pos += sd;
            }
        }
    } else {
        const int64_t nrows = ndim == 1 ? 0 : szh_sample_row_limit(G, sd);
        const int per_row = (int)(r2 / sd + 2);
        const int64_t slots = ndim == 1 ? (n > 2 ? (n - 2 + sd - 1) / sd : 0) : nrows * per_row;
        if (slots > 0) {
            TRY(ensure(ctx, ctx->msst_pe, (size_t)slots * 8));
            const int64_t work = ndim == 1 ? slots : nrows;
            const int grid = (int)std::min<int64_t>((work + 255) / 256, 4096);
            hipLaunchKernelGGL((k_msst_sample<T>), dim3(grid), dim3(256), 0, st, G, ndim, d_in, nrows, sd, per_row, (double *)ctx->msst_pe.p);
            HIPCHK(hipGetLastError());
            std::vector<double> pe((size_t)slots);
            HIPCHK(hipMemcpyAsync(pe.data(), ctx->msst_pe.p, (size_t)slots * 8, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            for (int64_t k = 0; k < slots; ++k) { if (f64_bits(pe[(size_t)k]) == 0x7ff8000000000001ull) continue; add(pe[(size_t)k]); }
        }
    }
    *out = msst_pick_intervals(hist, total, prm->pred_threshold, sizeof(T) == 8 ? 64u : 32u);
    return SZHIP_OK;
}
static int msst_upload(szhip_ctx *ctx, const MsstHostTab &ht, unsigned intervals, bool dec, msst_tab *tb)
{
    hipStream_t st = ctx->stream;
    memset(tb, 0, sizeof(*tb));
    TRY(ensure(ctx, ctx->msst_ptab, (size_t)intervals * 8));
    HIPCHK(hipMemcpyAsync(ctx->msst_ptab.p, ht.ptab.data(), (size_t)intervals * 8, hipMemcpyHostToDevice, st));
    tb->ptab = (const double *)ctx->msst_ptab.p; tb->intervals = (int)intervals;
    if (!dec) {
        TRY(ensure(ctx, ctx->msst_cells, ht.cells.size() * 2 + 16));
        HIPCHK(hipMemcpyAsync(ctx->msst_cells.p, ht.cells.data(), ht.cells.size() * 2, hipMemcpyHostToDevice, st));
        tb->cells = (const uint16_t *)ctx->msst_cells.p; tb->base = ht.base; tb->range = ht.range; tb->bits = ht.bits;
    }
    HIPCHK(hipStreamSynchronize(st));                        // the host vectors are pageable: the copies must be over before they go away
    return SZHIP_OK;
}
// the sweep: one launch per hyperplane (1-D: the one-lane chain).  codes: u16 per element; rec: the reconstruction (DEC: in place)
template <class T>
int msst_sweep(szhip_ctx *ctx, const szh_geom3 &G, int ndim, bool dec, const T *d_in, T *d_rec, uint16_t *d_codes, const MsstHostTab &ht, unsigned intervals,
               int ign_bits)
{
    hipStream_t st = ctx->stream;
    msst_tab tb;
    TRY(msst_upload(ctx, ht, intervals, dec, &tb));
    HIPCHK(hipEventRecord(ctx->ev[2], st));
    if (ndim == 1) {
        const size_t lds = (size_t)intervals * 8 + (dec ? 0 : ht.cells.size() * 2);
        const int in_lds = lds <= 60 * 1024;
        if (dec) hipLaunchKernelGGL((k_msst_chain_1d<T, true>), dim3(1), dim3(64), in_lds ? lds : 0, st, (const T *)nullptr, d_rec, d_codes, G.n, tb, (int)intervals, (int64_t)0, ign_bits, in_lds);
        else hipLaunchKernelGGL((k_msst_chain_1d<T, false>), dim3(1), dim3(64), in_lds ? lds : 0, st, d_in, (T *)nullptr, d_codes, G.n, tb, (int)intervals, (int64_t)ht.cells.size(), ign_bits, in_lds);
        HIPCHK(hipGetLastError());
    } else {
        const int r0 = ndim == 3 ? G.g0.count : 1, r1 = G.g1.count, r2 = G.g2.count;
        for (int d = 0; d <= (r0 - 1) + (r1 - 1) + (r2 - 1); ++d) {
            const int a_lo = std::max(0, d - (r1 - 1) - (r2 - 1)), a_hi = std::min(d, r0 - 1);
            const int64_t threads = (int64_t)(a_hi - a_lo + 1) * r1;
            const unsigned grid = (unsigned)((threads + 255) / 256);
            if (dec) hipLaunchKernelGGL((k_msst_plane<T, true>), dim3(grid), dim3(256), 0, st, r0, r1, r2, ndim, d, (const T *)nullptr, d_rec, d_codes, tb, ign_bits);
            else hipLaunchKernelGGL((k_msst_plane<T, false>), dim3(grid), dim3(256), 0, st, r0, r1, r2, ndim, d, d_in, d_rec, d_codes, tb, ign_bits);
        }
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(ctx->ev[3], st));
    return SZHIP_OK;
}

template <class T>
int compress14_impl(szhip_ctx *ctx, const void *data, int data_on_device, size_t r0, size_t r1, size_t r2, double eb_in, double range_in,
                    double median_in, const szhip_params *prm, const unsigned char *meta, size_t meta_len, const szhip_pwr *pw, int out_on_device,
                    unsigned char **out, size_t *out_size, szhip_stats *stats)
{
    const int is_double = sizeof(T) == 8;
    // pw != NULL: the data are log2|x| of a point-wise-relative call and the container carries the PW_REL fields
    // r0 == 0: the 2-D compressor SZ_compress_float_2D_MDQ (sz_float.c:610): its predictors are those of the 3-D one's first layer,
    // so it is carried as 1 x r1 x r2 (the block size of the carried geometry plays no role here); its optimiser has the 2-D lattice
    // r0 == 0 and r1 == 0: the 1-D compressor SZ_compress_float_1D_MDQ (sz_float.c:353): a chain through the previous reconstructed
    // value, walked by k_chain_1d; the container and everything after the code array are the same
    const bool one_d = r0 == 0 && r1 == 0;
    // pw->msst19: the table-driven form of PW_REL (szh_msst.h): `data` has its zeros replaced, eb_in is the RATIO; another optimiser, another
    // quantiser, exact values taken against 0, two more header bytes -- the entropy stage and the container are the same
    const bool msst = pw && pw->msst19;
    const int ndim = one_d ? 1 : r0 == 0 ? 2 : 3;
    const szh_geom3 G = one_d ? szh_make_geom2(1, (int)r2) : r0 == 0 ? szh_make_geom2((int)r1, (int)r2) : szh_make_geom3((int)r0, (int)r1, (int)r2);
    const int64_t n = G.n;
    const T eb = (T)eb_in;                                     // `float realPrecision` parameter of sz_float.c:946 (:353 for 1-D)
    const double t_begin = now_ms();
    double host_ms = 0;
    hipStream_t st = ctx->stream;
    szhip_stats S; memset(&S, 0, sizeof(S));
    S.n_elements = (uint64_t)n;

    const T *d_in = (const T *)data;
    if (!data_on_device) {
        TRY(ensure(ctx, ctx->in, (size_t)n * sizeof(T)));
        TRY(staged_copy(ctx, ctx->in.p, data, (size_t)n * sizeof(T), true));
        d_in = (const T *)ctx->in.p;
    }
    TRY(ensure(ctx, ctx->small, SM_COUNT * 8));
    u64 *sm = (u64 *)ctx->small.p;
    HIPCHK(hipMemsetAsync(sm, 0, SM_COUNT * 8, st));
    HIPCHK(hipEventRecord(ctx->ev[0], st));

    // ---- interval optimiser (optimize_intervals_float_3D_opt, sz_float.c:4644): the SZ 2.1 sample lattice, radius histogram only
    unsigned intervals = prm->quantization_intervals;
    if (intervals == 0 && msst) {
        bool zeros_left = false;                                // only when the array's first element is zero (szhip_msst_prepare)
        { T first; HIPCHK(hipMemcpyAsync(&first, d_in, sizeof(T), hipMemcpyDeviceToHost, st)); HIPCHK(hipStreamSynchronize(st)); zeros_left = first == 0; }
        TRY(msst_intervals<T>(ctx, G, ndim, d_in, prm, eb_in, zeros_left, &intervals));
    } else if (intervals == 0) {
        const unsigned max_radius = prm->max_quant_intervals / 2;
        TRY(ensure(ctx, ctx->hist, (size_t)(max_radius + 8192) * 4 + 64));
        TRY(ensure_pinned(ctx, (size_t)(max_radius + 8192) * 4 + 64));
        unsigned *d_rh = (unsigned *)ctx->hist.p, *d_fh = d_rh + max_radius;
        HIPCHK(hipMemsetAsync(d_rh, 0, (size_t)(max_radius + 8192) * 4, st));
        const int64_t nrows = one_d ? 0 : szh_sample_row_limit(G, prm->sample_distance);
        if (one_d) {
            const int64_t count = (n - 2 + prm->sample_distance - 1) / prm->sample_distance;
            int grid = (int)std::min<int64_t>((count + 255) / 256 + 1, 1024);
            hipLaunchKernelGGL((k_sample_1d<T>), dim3(grid), dim3(256), 0, st, d_in, n, prm->sample_distance, (double)eb, max_radius, d_rh);
            HIPCHK(hipGetLastError());
        } else if (G.ndim == 3 && (G.g0.count <= 1 || G.g1.count <= 1)) {      // a degenerate 3-D array: the reference's walk, literally
            hipLaunchKernelGGL((k_sample_walk<T, false>), dim3(1), dim3(64), 0, st, G, d_in, prm->sample_distance, (double)eb, (T)0, max_radius, d_rh, d_fh, sm + SM_WITHIN);
            HIPCHK(hipGetLastError());
        } else if (nrows > 0) {
            int grid = (int)std::min<int64_t>((nrows + 255) / 256, 1024);
            hipLaunchKernelGGL((k_sample<T, false>), dim3(grid), dim3(256), 0, st, G, d_in, nrows, prm->sample_distance, (double)eb, (T)0,
                               max_radius, d_rh, d_fh, sm + SM_WITHIN);
            HIPCHK(hipGetLastError());
        }
        unsigned *h_hist = (unsigned *)ctx->pinned;
        HIPCHK(hipMemcpyAsync(h_hist, d_rh, (size_t)max_radius * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        double h0 = now_ms();
        u64 total = 0;
        for (unsigned i = 0; i < max_radius; ++i) total += h_hist[i];
        const size_t target = (size_t)((float)total * prm->pred_threshold);       // `size_t targetCount = totalSampleSize*predThreshold`
        size_t sum = 0; unsigned i = 0;
        for (; i < max_radius; ++i) { sum += h_hist[i]; if (sum > target) break; }
        if (i >= max_radius) i = max_radius - 1;
        unsigned p2 = 2 * (i + 1); p2 -= 1; p2 |= p2 >> 1; p2 |= p2 >> 2; p2 |= p2 >> 4; p2 |= p2 >> 8; p2 |= p2 >> 16; p2 += 1;
        intervals = p2 < 32 ? 32 : p2;
        host_ms += now_ms() - h0;
    }
    if (intervals > 65536 || intervals < 4) FAIL(SZHIP_ERR_UNSUP, "quantization interval count %u outside [4,65536]", intervals);
    S.intervals = intervals;
    T median = (T)median_in;
    int req_len_ = 0;
    if (msst) {
        // computeReqLength_float_MSST19 = 9 - exponent of (float)ratio (sz_float.c:58-62), the double rule 12 - exponent (sz_double.c:57-61) --
        // which the FLOAT 2-D quantiser also uses (sz_float.c:2041); exact values are the values themselves (no median)
        const int e64 = (int)((f64_bits(eb_in) & 0x7FF0000000000000ull) >> 52) - 1023;
        const float pf = (float)eb_in; unsigned u32; memcpy(&u32, &pf, 4);
        const int e32 = (int)((u32 & 0x7F800000u) >> 23) - 127;
        req_len_ = (is_double || ndim == 2) ? 12 - e64 : 9 - e32;
        median = 0;
        if (req_len_ < 9 || req_len_ > (int)sizeof(T) * 8) FAIL(SZHIP_ERR_UNSUP, "point-wise ratio %g needs %d leading bits per exact value", eb_in, req_len_);
    } else req_len_ = req_length<T>((double)eb, (T)range_in, &median);
    const int req_len = req_len_;
    const int req_bytes = req_len / 8, resi_bits = req_len % 8, ign_bits = (int)sizeof(T) * 8 - req_len;
    HIPCHK(hipEventRecord(ctx->ev[1], st));

    // ---- predict + quantise
    TRY(ensure(ctx, ctx->codes_nat, (size_t)n * 2 + 64));
    uint16_t *d_codes = (uint16_t *)ctx->codes_nat.p;
    S.quant_kernel_launches = 1;
    if (msst) {
        MsstHostTab ht;
        double hb = now_ms();
        if (msst_build_tab(ht, eb_in, intervals, pw->plus_bits, true)) FAIL(SZHIP_ERR_UNSUP, "point-wise ratio %g with %u intervals: look-up table too large", eb_in, intervals);
        host_ms += now_ms() - hb;
        if (ndim >= 2 && !tune_int("SZ_HIP_MSST_SWEEP", 0)) {     // the wavefront kernel with its third quantiser (szh_pencil.h, fmt 2)
            msst_tab tb;
            TRY(msst_upload(ctx, ht, intervals, false, &tb));
            TRY(launch_pencil14<T>(ctx, G, sm, false, d_in, nullptr, d_codes, eb, intervals, (T)0, ign_bits, &tb, ndim));
        } else {                                                  // the plane-by-plane sweep (1-D: the one-lane chain); SZ_HIP_MSST_SWEEP=1 forces it
            TRY(ensure(ctx, ctx->msst_rec, (size_t)n * sizeof(T)));
            TRY(msst_sweep<T>(ctx, G, ndim, false, d_in, (T *)ctx->msst_rec.p, d_codes, ht, intervals, ign_bits));
            S.quant_kernel_launches = ndim == 1 ? 1 : (unsigned)((ndim == 3 ? G.g0.count : 1) + G.g1.count + G.g2.count - 2);
        }
    } else if (one_d) {
        // the chain cut at its certain restarts, one thread per segment; a segment whose successor turns out not to restart raises
        // the flag, and the array is then walked by the one-wavefront kernel (same result, by construction; SZ_HIP_1D_SERIAL=1 forces it)
        HIPCHK(hipEventRecord(ctx->ev[2], st));
        unsigned violation = 1;
        if (!tune_int("SZ_HIP_1D_SERIAL", 0)) {
            const int grid = (int)std::min<int64_t>((n + 255) / 256, 1 << 20);
            hipLaunchKernelGGL((k_chain_seg_1d<T, false>), dim3(grid), dim3(256), 0, st, d_in, (T *)nullptr, d_codes, n, eb, (T)(1 / eb), (int)intervals,
                               median, ign_bits, tune_int("SZ_HIP_1D_REACH_PCT", 100) / 100.0, (unsigned *)(sm + SM_CHANGED));
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(&violation, sm + SM_CHANGED, 4, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
        }
        if (violation) {
            hipLaunchKernelGGL((k_chain_1d<T, false>), dim3(1), dim3(64), 0, st, d_in, (T *)nullptr, d_codes, n, eb, (T)(1 / eb), (int)intervals, median, ign_bits);
            HIPCHK(hipGetLastError());
            S.quant_kernel_launches = 2;
        }
        HIPCHK(hipEventRecord(ctx->ev[3], st));
    } else
        TRY(launch_pencil14<T>(ctx, G, sm, false, d_in, nullptr, d_codes, eb, intervals, median, ign_bits));

    // ---- histogram -> code book (host), exact-value counts
    TRY(ensure(ctx, ctx->hist, (size_t)(65536 + 8192) * 4 + 64));
    unsigned *d_hist = (unsigned *)ctx->hist.p;
    TRY(ensure_pinned(ctx, (size_t)intervals * 4 + 64));
    unsigned *h_hist = (unsigned *)ctx->pinned;
    HIPCHK(hipMemsetAsync(d_hist, 0, (size_t)intervals * 4, st));
    {
        int rshift = 0; int use_lds = intervals <= 16384;
        if (use_lds) { while ((intervals << (rshift + 1)) <= 16384u && rshift < 6) ++rshift; }
        const size_t lds = use_lds ? ((size_t)intervals << rshift) * 4 : 16;
        int grid = (int)std::min<int64_t>((n / 8 + 255) / 256 + 1, 2048);
        hipLaunchKernelGGL(k_hist_u16, dim3(grid), dim3(256), lds, st, (const uint16_t *)d_codes, n, intervals, rshift, use_lds, d_hist, szh_rb_layout{0, 0, 0, 0, 0, 0}, 0, 0, 0, (int64_t)0);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipMemcpyAsync(h_hist, d_hist, (size_t)intervals * 4, hipMemcpyDeviceToHost, st));
    const int64_t nlin = (n + SZH_LIN_CHUNK - 1) / SZH_LIN_CHUNK;
    TRY(ensure(ctx, ctx->col_zeros64, (size_t)nlin * 8));
    TRY(ensure(ctx, ctx->col_off, (size_t)nlin * 8));
    hipLaunchKernelGGL(k_lin_zero_count, dim3((unsigned)nlin), dim3(256), 0, st, (const uint16_t *)d_codes, n, (u64 *)ctx->col_zeros64.p);
    TRY(scan_u64(ctx, (const u64 *)ctx->col_zeros64.p, nlin, (u64 *)ctx->col_off.p, sm + SM_TOTAL_UNPRED));
    u64 h_small[SM_COUNT];
    HIPCHK(hipMemcpyAsync(h_small, sm, SM_COUNT * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if ((unsigned)h_small[SM_ERR] != 0) { ctx->wave_timeout = true; FAIL(SZHIP_ERR_INTERNAL, "wavefront kernel: halo wait timed out"); }
    const u64 E = h_small[SM_TOTAL_UNPRED];
    S.n_unpred = E;
    double h0 = now_ms();
    szhost_huff *hf = szhost_huff_build(2 * (int)intervals, h_hist, nullptr, intervals);
    if (!hf) FAIL(SZHIP_ERR_INTERNAL, "Huffman build failed");
    const size_t tree_bytes = szhost_huff_tree_size(hf);
    const u64 total_bits = hf->total_bits;
    const size_t pay_bytes = (size_t)((total_bits + 7) / 8);
    std::vector<u64> tab_code(intervals); std::vector<uint8_t> tab_len(intervals);
    for (unsigned s2 = 0; s2 < intervals; ++s2) { tab_code[s2] = hf->code[s2]; tab_len[s2] = hf->len[s2]; }
    host_ms += now_ms() - h0;

    // ---- exact values: compact in scan order, lead numbers, mid-byte offsets
    u64 nmid = 0;
    if (E > 0) {
        TRY(ensure(ctx, ctx->unpred, (size_t)E * sizeof(T)));
        TRY(ensure(ctx, ctx->lor_bits, (size_t)E + 8));              // lead numbers, one byte each
        TRY(ensure(ctx, ctx->reg_flags, (size_t)E * 8));             // mid-byte counts
        TRY(ensure(ctx, ctx->reg_rank, (size_t)E * 8));              // mid-byte offsets
        hipLaunchKernelGGL((k_lin_zero_move<T, 0>), dim3((unsigned)nlin), dim3(256), 0, st, (const uint16_t *)d_codes, n, (const u64 *)ctx->col_off.p,
                           d_in, (T *)ctx->unpred.p, (T *)nullptr);
        hipLaunchKernelGGL((k_exact_lead<T>), dim3((unsigned)((E + 255) / 256)), dim3(256), 0, st, (const T *)ctx->unpred.p, (int64_t)E, median,
                           req_bytes, (uint8_t *)ctx->lor_bits.p, (u64 *)ctx->reg_flags.p);
        HIPCHK(hipGetLastError());
        TRY(scan_u64(ctx, (const u64 *)ctx->reg_flags.p, (int64_t)E, (u64 *)ctx->reg_rank.p, sm + SM_SCRATCH));
        HIPCHK(hipMemcpyAsync(&nmid, sm + SM_SCRATCH, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    const size_t lead_size = (size_t)((E * 2 + 7) / 8), resi_size = resi_bits ? (size_t)((E * (u64)resi_bits + 7) / 8) : 0;

    // ---- container
    h0 = now_ms();
    const size_t type_size = 8 + tree_bytes + pay_bytes;
    const size_t blob = pw ? (size_t)pw->signs_blob_size : 0;
    const size_t hdr_len = meta_len + 8 + 4 + (pw ? 1 + 8 + 4 : 0) + 4 + sizeof(T) + 1 + (msst ? 2 : 0) + 8 + 8 + 8 + 8 + (pw ? sizeof(T) : 0) + 8 + tree_bytes; // ... up to the Huffman payload
    const size_t total_len = hdr_len + pay_bytes + blob + lead_size + (size_t)nmid + resi_size;
    std::vector<unsigned char> hdr(hdr_len, 0);
    {
        unsigned char *q = hdr.data();
        memcpy(q, meta, meta_len); q += meta_len;
        szhost_put_u64be(q, (uint64_t)n); q += 8;
        szhost_put_u32be(q, prm->max_quant_intervals); q += 4;
        if (pw) {                                                // TightDataPointStorageF.c:408-419
            *q++ = pw->rad_expo;
            szhost_put_u64be(q, pw->segment_size); q += 8;
            szhost_put_u32be(q, pw->signs_blob_size); q += 4;
        }
        szhost_put_u32be(q, intervals); q += 4;
        const T median_field = msst ? (T)pw->median_stored : median;
        if (is_double) szhost_put_f64be(q, (double)median_field); else szhost_put_f32be(q, (float)median_field);
        q += sizeof(T);
        *q++ = (unsigned char)req_len;
        if (msst) {                                              // plus_bits, max_bits (TightDataPointStorageF.c:431-435; Huffman.c:828-833)
            int max_bits = 0;
            for (unsigned s2 = 0; s2 < intervals; ++s2) if (tab_len[s2] > max_bits) max_bits = tab_len[s2];
            *q++ = pw->plus_bits; *q++ = (unsigned char)max_bits;
        }
        szhost_put_f64be(q, msst ? eb_in : (double)eb); q += 8;
        szhost_put_u64be(q, (uint64_t)type_size); q += 8;
        szhost_put_u64be(q, (uint64_t)E); q += 8;
        szhost_put_u64be(q, (uint64_t)nmid); q += 8;
        if (pw) {                                                // minLogValue in the data's type (:454-459; TightDataPointStorageD.c:456-461)
            if (is_double) szhost_put_f64be(q, pw->min_log_value); else szhost_put_f32be(q, (float)pw->min_log_value);
            q += sizeof(T);
        }
        szhost_put_u32be(q, (uint32_t)hf->n_nodes); q += 4;      // encode_withTree blob (Huffman.c:790-816)
        szhost_put_u32be(q, intervals); q += 4;
        szhost_huff_tree_write(hf, q); q += tree_bytes;
    }
    szhost_huff_free(hf);
    host_ms += now_ms() - h0;

    TRY(ensure(ctx, ctx->code_tab, (size_t)intervals * 8));
    TRY(ensure(ctx, ctx->len_tab, (size_t)intervals));
    HIPCHK(hipMemcpyAsync(ctx->code_tab.p, tab_code.data(), (size_t)intervals * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ctx->len_tab.p, tab_len.data(), (size_t)intervals, hipMemcpyHostToDevice, st));
    TRY(ensure(ctx, ctx->stream_buf, total_len + 64));
    unsigned char *d_stream = (unsigned char *)ctx->stream_buf.p;
    HIPCHK(hipMemsetAsync(d_stream, 0, total_len + 64, st));
    HIPCHK(hipMemcpyAsync(d_stream, hdr.data(), hdr_len, hipMemcpyHostToDevice, st));
    if (blob) HIPCHK(hipMemcpyAsync(d_stream + hdr_len + pay_bytes, pw->signs_blob, blob, hipMemcpyHostToDevice, st));   // after the type array (:463-467)
    if (total_bits > 0) {
        const int64_t nchunks = (n + SZH_ENC_CHUNK - 1) / SZH_ENC_CHUNK;
        TRY(ensure(ctx, ctx->chunk_bits, (size_t)nchunks * 8));
        TRY(ensure(ctx, ctx->chunk_off, (size_t)nchunks * 8));
        hipLaunchKernelGGL(k_chunk_bits, dim3((unsigned)((nchunks + SZH_CB_PER - 1) / SZH_CB_PER)), dim3(256), 0, st, (const uint16_t *)d_codes, n, (const uint8_t *)ctx->len_tab.p,
                           intervals, (u64 *)ctx->chunk_bits.p);
        TRY(scan_u64(ctx, (const u64 *)ctx->chunk_bits.p, nchunks, (u64 *)ctx->chunk_off.p, sm + SM_TOTAL_BITS));
        hipLaunchKernelGGL(k_encode, dim3((unsigned)((nchunks + SZH_ENC_PER - 1) / SZH_ENC_PER)), dim3(256), 0, st, (const uint16_t *)d_codes, n, (const u64 *)ctx->code_tab.p,
                           (const uint8_t *)ctx->len_tab.p, intervals, (const u64 *)ctx->chunk_off.p, (u64)hdr_len * 8, (unsigned *)d_stream);
        HIPCHK(hipGetLastError());
    }
    if (E > 0) {
        unsigned char *lead_out = d_stream + hdr_len + pay_bytes + blob, *mid_out = lead_out + lead_size, *resi_out = mid_out + nmid;
        hipLaunchKernelGGL((k_exact_write<T>), dim3((unsigned)(((E + 7) / 8 + 255) / 256)), dim3(256), 0, st, (const T *)ctx->unpred.p, (int64_t)E,
                           median, req_bytes, resi_bits, (const uint8_t *)ctx->lor_bits.p, (const u64 *)ctx->reg_rank.p, lead_out, mid_out, resi_out,
                           (int64_t)resi_size);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(ctx->ev[4], st));
    if (out_on_device == 2) {
        if (!*out || *out_size < total_len) FAIL(SZHIP_ERR_ARG, "caller's device buffer too small (%zu < %zu)", *out_size, total_len);
        HIPCHK(hipMemcpyAsync(*out, d_stream, total_len, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
    } else if (out_on_device) {
        HIPCHK(hipStreamSynchronize(st));
        *out = d_stream;
    } else {
        unsigned char *h = (unsigned char *)malloc(total_len ? total_len : 1);
        if (!h) FAIL(SZHIP_ERR_INTERNAL, "out of host memory");
        TRY(staged_copy(ctx, h, d_stream, total_len, false));
        *out = h;
    }
    *out_size = total_len;
    {
        u64 tb = 0;
        if (total_bits > 0) { HIPCHK(hipMemcpy(&tb, sm + SM_TOTAL_BITS, 8, hipMemcpyDeviceToHost)); }
        if (tb != total_bits) FAIL_PUBLISHED(SZHIP_ERR_INTERNAL, "encoded bit count mismatch (%llu vs %llu)", (unsigned long long)tb, (unsigned long long)total_bits);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]); S.ms_prequant = ms;
    hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3]); S.ms_quant = ms;
    hipEventElapsedTime(&ms, ctx->ev[3], ctx->ev[4]); S.ms_entropy = ms;
    S.ms_host = host_ms; S.ms_total = now_ms() - t_begin; S.out_bytes = total_len;
    if (stats) *stats = S;
    return SZHIP_OK;
}

// `body_off`: offset of the max_quant_intervals field (4 + 28|36 + 8)
template <class T>
int decompress14_impl(szhip_ctx *ctx, const unsigned char *stream_in, int stream_on_device, size_t stream_len, size_t body_off, int pwr,
                      size_t r0, size_t r1, size_t r2, void *out, int out_on_device, szhip_stats *stats)
{
    const int is_double = sizeof(T) == 8;
    const bool one_d = r0 == 0 && r1 == 0;                     // decompressDataSeries_float_1D (szd_float.c:185)
    const bool msst = pwr == 2;                                // pwr: 0 plain, 1 PW_REL log-domain form, 2 PW_REL table-driven form (szh_msst.h)
    const int ndim = one_d ? 1 : r0 == 0 ? 2 : 3;
    const szh_geom3 G = one_d ? szh_make_geom2(1, (int)r2) : r0 == 0 ? szh_make_geom2((int)r1, (int)r2) : szh_make_geom3((int)r0, (int)r1, (int)r2);
    const int64_t n = G.n;
    const double t_begin = now_ms();
    double host_ms = 0;
    hipStream_t st = ctx->stream;
    szhip_stats S; memset(&S, 0, sizeof(S));
    S.n_elements = (uint64_t)n;

    TRY(ensure(ctx, ctx->stream_buf, stream_len + 64));
    unsigned char *d_stream = (unsigned char *)ctx->stream_buf.p;
    if (stream_on_device) { if (stream_in != d_stream) HIPCHK(hipMemcpyAsync(d_stream, stream_in, stream_len, hipMemcpyDeviceToDevice, st)); }
    else TRY(staged_copy(ctx, d_stream, stream_in, stream_len, true));
    HIPCHK(hipMemsetAsync(d_stream + stream_len, 0, 64, st));
    HIPCHK(hipEventRecord(ctx->ev[0], st));

    // ---- header + tree on the host (TightDataPointStorageF.c:54-265); a device-resident stream hands over a prefix
    double h0 = now_ms();
    const size_t fixed = 4 + (pwr ? 1 + 8 + 4 : 0) + 4 + sizeof(T) + 1 + (msst ? 2 : 0) + 8 + 8 + 8 + 8 + (pwr ? sizeof(T) : 0) + 8;
    if (body_off + fixed > stream_len) FAIL(SZHIP_ERR_STREAM, "truncated stream");
    std::vector<unsigned char> hbuf;
    const unsigned char *hs = stream_in;
    auto fetch = [&](size_t want) -> int {
        if (!stream_on_device) return SZHIP_OK;
        hbuf.resize(want);
        HIPCHK(hipMemcpyAsync(hbuf.data(), d_stream, want, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        hs = hbuf.data();
        return SZHIP_OK;
    };
    TRY(fetch(body_off + fixed));
    const unsigned char *q = hs + body_off;
    q += 4;                                                        // max_quant_intervals
    size_t blob = 0;
    if (pwr) { q += 1 + 8; blob = szhost_get_u32be(q); q += 4; }   // radExpo, segment_size, size of the sign bytes (TightDataPointStorageF.c:137-148)
    const unsigned intervals = szhost_get_u32be(q); q += 4;
    T median = is_double ? (T)szhost_get_f64be(q) : (T)szhost_get_f32be(q); q += sizeof(T);
    const int req_len = *q++;
    int plus_bits = 0;
    if (msst) { plus_bits = q[0]; q += 2; median = 0; }           // plus_bits, max_bits (TightDataPointStorageF.c:164-168); exact values carry no median
    const double eb_field = szhost_get_f64be(q);
    const T eb = (T)eb_field; q += 8;                              // `float realPrecision = tdps->realPrecision`, szd_float.c:610
    const uint64_t type_size = szhost_get_u64be(q); q += 8;
    const uint64_t E = szhost_get_u64be(q); q += 8;
    const uint64_t nmid = szhost_get_u64be(q); q += 8;
    if (pwr) q += sizeof(T);                                       // minLogValue (the caller read it: szhip_sz14_pwr_locate)
    const size_t type_off = body_off + fixed - 8;                  // the blob starts with nodeCount | intervals
    if (intervals < 4 || intervals > 65536) FAIL(SZHIP_ERR_STREAM, "bad interval count %u", intervals);
    if (req_len < 9 || req_len > (int)sizeof(T) * 8) FAIL(SZHIP_ERR_STREAM, "bad exact-value length %d", req_len);
    if (!(eb > 0)) FAIL(SZHIP_ERR_STREAM, "bad error bound");
    const int req_bytes = req_len / 8, resi_bits = req_len % 8;
    const size_t lead_size = (size_t)((E * 2 + 7) / 8), resi_size = resi_bits ? (size_t)((E * (uint64_t)resi_bits + 7) / 8) : 0;
    if (E >= ((uint64_t)1 << 32)) FAIL(SZHIP_ERR_UNSUP, "more than 2^32 exact values");   // the prefix counts of k_exact_* are packed in 32-bit halves
    if (E > (uint64_t)n || type_size < 8 || type_size > stream_len || nmid > stream_len ||
        blob > stream_len || type_off + type_size + blob + lead_size + nmid + resi_size > stream_len) FAIL(SZHIP_ERR_STREAM, "truncated stream");
    const int node_count = (int)szhost_get_u32be(q);
    if (node_count <= 0 || 8 + szhost_huff_serial_size(node_count) > type_size) FAIL(SZHIP_ERR_STREAM, "bad Huffman tree size");
    const size_t tree_bytes = szhost_huff_serial_size(node_count);
    TRY(fetch(type_off + 8 + tree_bytes));
    szhost_huff *hf = szhost_huff_from_bytes(2 * (int)intervals, hs + type_off + 8, node_count);
    if (!hf) FAIL(SZHIP_ERR_STREAM, "bad Huffman tree");
    std::vector<uint32_t> dtab((size_t)hf->n_nodes * 2);
    szhost_huff_decode_table(hf, dtab.data());
    const int single_symbol = hf->t[0] ? (int)hf->C[0] : -1;
    const int n_nodes = hf->n_nodes;
    szhost_huff_free(hf);
    const size_t pay_off = type_off + 8 + tree_bytes;
    const u64 total_bits = (u64)(type_size - 8 - tree_bytes) * 8;
    S.intervals = intervals; S.n_unpred = E;
    host_ms += now_ms() - h0;

    TRY(ensure(ctx, ctx->small, SM_COUNT * 8));
    u64 *sm = (u64 *)ctx->small.p;
    HIPCHK(hipMemsetAsync(sm, 0, SM_COUNT * 8, st));
    TRY(ensure(ctx, ctx->codes_nat, (size_t)n * 2 + 64));
    uint16_t *d_codes = (uint16_t *)ctx->codes_nat.p;
    u64 total_sym = 0;
    TRY(huff_decode_device(ctx, sm, d_stream + pay_off, (unsigned)std::min<size_t>(pay_off, 4096), total_bits, dtab, n_nodes, single_symbol, n, d_codes, &total_sym));

    // ---- exact values back into the output array
    const int64_t nlin = (n + SZH_LIN_CHUNK - 1) / SZH_LIN_CHUNK;
    TRY(ensure(ctx, ctx->col_zeros64, (size_t)nlin * 8));
    TRY(ensure(ctx, ctx->col_off, (size_t)nlin * 8));
    hipLaunchKernelGGL(k_lin_zero_count, dim3((unsigned)nlin), dim3(256), 0, st, (const uint16_t *)d_codes, n, (u64 *)ctx->col_zeros64.p);
    TRY(scan_u64(ctx, (const u64 *)ctx->col_zeros64.p, nlin, (u64 *)ctx->col_off.p, sm + SM_TOTAL_UNPRED));
    u64 zeros_found = 0;
    HIPCHK(hipMemcpyAsync(&zeros_found, sm + SM_TOTAL_UNPRED, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HDEC_CHECK(ctx);
    total_sym = ctx->hdec_res[0];
    if ((int64_t)total_sym < n) FAIL(SZHIP_ERR_STREAM, "Huffman payload holds %llu symbols, need %lld", (unsigned long long)total_sym, (long long)n);
    if (zeros_found != E) FAIL(SZHIP_ERR_STREAM, "stream lists %llu exact values, codes need %llu", (unsigned long long)E, (unsigned long long)zeros_found);
    T *d_out = (T *)out;
    if (!out_on_device) { TRY(ensure(ctx, ctx->out, (size_t)n * sizeof(T))); d_out = (T *)ctx->out.p; }
    if (E > 0) {
        const unsigned char *lead_in = d_stream + type_off + type_size + blob, *mid_in = lead_in + lead_size, *resi_in = mid_in + nmid;
        const unsigned gE = (unsigned)((E + 255) / 256);
        TRY(ensure(ctx, ctx->unpred, (size_t)E * sizeof(T)));
        TRY(ensure(ctx, ctx->reg_flags, (size_t)E * 8 * 3));         // flag words: f01 | f2 | mid counts
        TRY(ensure(ctx, ctx->reg_rank, (size_t)E * 8 * 3));          // their exclusive prefix sums
        TRY(ensure(ctx, ctx->lor_bits, (size_t)E * 3 + 8));          // compacted own bytes of positions 0..2
        u64 *f01 = (u64 *)ctx->reg_flags.p, *f2 = f01 + E, *mc = f2 + E;
        u64 *s01 = (u64 *)ctx->reg_rank.p, *s2 = s01 + E, *mo = s2 + E;
        uint8_t *own0 = (uint8_t *)ctx->lor_bits.p, *own1 = own0 + E, *own2 = own1 + E;
        hipLaunchKernelGGL(k_exact_flags, dim3(gE), dim3(256), 0, st, lead_in, (int64_t)E, req_bytes, resi_bits, f01, f2, mc);
        TRY(scan_u64(ctx, f01, (int64_t)E, s01, sm + SM_SCRATCH));
        TRY(scan_u64(ctx, f2, (int64_t)E, s2, sm + SM_SCRATCH));
        TRY(scan_u64(ctx, mc, (int64_t)E, mo, sm + SM_SCRATCH));
        u64 mid_need = 0;
        HIPCHK(hipMemcpyAsync(&mid_need, sm + SM_SCRATCH, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (mid_need != nmid) FAIL(SZHIP_ERR_STREAM, "stream holds %llu mid bytes, lead numbers need %llu", (unsigned long long)nmid, (unsigned long long)mid_need);
        hipLaunchKernelGGL(k_exact_own, dim3(gE), dim3(256), 0, st, lead_in, (int64_t)E, req_bytes, resi_bits, mid_in, resi_in,
                           (const u64 *)s01, (const u64 *)s2, (const u64 *)mo, own0, own1, own2);
        hipLaunchKernelGGL((k_exact_build<T>), dim3(gE), dim3(256), 0, st, lead_in, (int64_t)E, req_bytes, resi_bits, mid_in, resi_in,
                           (const u64 *)s01, (const u64 *)s2, (const u64 *)mo, (const uint8_t *)own0, (const uint8_t *)own1, (const uint8_t *)own2,
                           median, (T *)ctx->unpred.p);
        hipLaunchKernelGGL((k_lin_zero_move<T, 1>), dim3((unsigned)nlin), dim3(256), 0, st, (const uint16_t *)d_codes, n, (const u64 *)ctx->col_off.p,
                           (const T *)nullptr, (T *)ctx->unpred.p, d_out);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(ctx->ev[1], st));

    // ---- reconstruct
    if (msst) {
        if (!(eb_field > 0 && eb_field < 1) || plus_bits > 16) FAIL(SZHIP_ERR_STREAM, "bad point-wise ratio / table parameters");
        MsstHostTab ht;
        msst_build_tab(ht, eb_field, intervals, plus_bits, false);
        if (ndim >= 2 && !tune_int("SZ_HIP_MSST_SWEEP", 0)) {
            msst_tab tb;
            TRY(msst_upload(ctx, ht, intervals, true, &tb));
            TRY(launch_pencil14<T>(ctx, G, sm, true, nullptr, d_out, d_codes, eb, intervals, (T)0, 0, &tb, ndim));
        } else
            TRY(msst_sweep<T>(ctx, G, ndim, true, nullptr, d_out, d_codes, ht, intervals, 0));
    } else if (one_d) {
        HIPCHK(hipEventRecord(ctx->ev[2], st));
        if (tune_int("SZ_HIP_1D_SERIAL", 0))
            hipLaunchKernelGGL((k_chain_1d<T, true>), dim3(1), dim3(64), 0, st, (const T *)nullptr, d_out, d_codes, n, eb, (T)(1 / eb), (int)intervals, median, 0);
        else {   // decoding sees where the chain restarts (code 0): one thread per segment, nothing to verify
            const int grid = (int)std::min<int64_t>((n + 255) / 256, 1 << 20);
            hipLaunchKernelGGL((k_chain_seg_1d<T, true>), dim3(grid), dim3(256), 0, st, (const T *)nullptr, d_out, d_codes, n, eb, (T)(1 / eb), (int)intervals,
                               median, 0, 1.0, (unsigned *)nullptr);
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(ctx->ev[3], st));
    } else
        TRY(launch_pencil14<T>(ctx, G, sm, true, nullptr, d_out, d_codes, eb, intervals, median, 0));
    S.quant_kernel_launches = 1;
    unsigned kerr = 0;
    HIPCHK(hipMemcpyAsync(&kerr, sm + SM_ERR, 4, hipMemcpyDeviceToHost, st));
    if (!out_on_device) TRY(staged_copy(ctx, out, d_out, (size_t)n * sizeof(T), false));
    HIPCHK(hipEventRecord(ctx->ev[4], st));
    HIPCHK(hipStreamSynchronize(st));
    if (kerr) { ctx->wave_timeout = true; FAIL(SZHIP_ERR_INTERNAL, "wavefront kernel: halo wait timed out"); }
    float ms = 0;
    hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]); S.ms_entropy = ms;
    hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3]); S.ms_quant = ms;
    S.ms_host = host_ms; S.ms_total = now_ms() - t_begin; S.out_bytes = (uint64_t)n * sizeof(T);
    if (stats) *stats = S;
    return SZHIP_OK;
}

// =====================================================================================================================
// Point-wise relative bounds, log-domain form (szh_pwr.h)
// =====================================================================================================================
template <class T>
int pwr_prepare_impl(szhip_ctx *ctx, const void *data, int data_on_device, size_t n, double vmin, double vmax, double ratio, void **d_log_out,
                     unsigned char *signs_host, int *positive, double *real_precision, double *value_range, double *median, double *min_log_value)
{
    hipStream_t st = ctx->stream;
    const T *d_in = (const T *)data;
    if (!data_on_device) {
        TRY(ensure(ctx, ctx->in, n * sizeof(T)));
        TRY(staged_copy(ctx, ctx->in.p, data, (size_t)n * sizeof(T), true));
        d_in = (const T *)ctx->in.p;
    }
    TRY(ensure(ctx, ctx->pwr_log, n * sizeof(T)));
    TRY(ensure(ctx, ctx->pwr_signs, n));
    TRY(ensure(ctx, ctx->pwr_small, PWR_RED * 8));
    T *d_log = (T *)ctx->pwr_log.p;
    u64 *red = (u64 *)ctx->pwr_small.p;
    const u64 init[PWR_RED] = {~0ull, 0ull, ~0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
    HIPCHK(hipMemcpyAsync(red, init, sizeof(init), hipMemcpyHostToDevice, st));
    const int grid = (int)std::min<int64_t>(((int64_t)n + 255) / 256, 4096);
    hipLaunchKernelGGL((k_pwr_log<T>), dim3(grid), dim3(256), 0, st, d_in, (int64_t)n, d_log, (unsigned char *)ctx->pwr_signs.p, red);
    HIPCHK(hipGetLastError());
    u64 res[PWR_RED];
    HIPCHK(hipMemcpyAsync(res, red, sizeof(res), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    // sz_float_pwr.c:1923-1947, in the data's type where the reference's variables have it
    const T tmin = (T)vmin, tmax = (T)vmax;
    T max_abs_log;
    if (tmin == 0) max_abs_log = (T)fabs(log2(fabs((double)tmax)));
    else if (tmax == 0) max_abs_log = (T)fabs(log2(fabs((double)tmin)));
    else max_abs_log = (T)(fabs(log2(fabs((double)tmin))) > fabs(log2(fabs((double)tmax))) ? fabs(log2(fabs((double)tmin))) : fabs(log2(fabs((double)tmax))));
    T min_log = max_abs_log;
    if (res[PWR_NONZERO]) {
        const T lo = (T)ord_dec<T>(res[PWR_MINLOG]), hi = (T)ord_dec<T>(res[PWR_MAXLOG]);
        if (hi > max_abs_log) max_abs_log = hi;
        if (lo < min_log) min_log = lo;
    }
    const T amin = (T)ord_dec<T>(res[PWR_MINALL]), amax = (T)ord_dec<T>(res[PWR_MAXALL]);     // computeRangeSize_float on the log array, zeros still 0
    const T range = amax - amin;
    *value_range = (double)range;
    *median = (double)(T)(amin + range / 2);
    if (fabs((double)min_log) > (double)max_abs_log) max_abs_log = (T)fabs((double)min_log);
    const double rp = log2(1.0 + ratio) - (double)max_abs_log * (sizeof(T) == 4 ? 1.2e-7 : 2.23e-16);
    *real_precision = rp;
    const T zval = (T)((double)min_log - 2.0001 * rp);
    *min_log_value = (double)(T)((double)min_log - 1.0001 * rp);
    hipLaunchKernelGGL((k_pwr_zero<T>), dim3(grid), dim3(256), 0, st, d_in, (int64_t)n, d_log, zval);
    HIPCHK(hipGetLastError());
    *positive = res[PWR_NEG] ? 0 : 1;
    if (res[PWR_NEG] && signs_host) HIPCHK(hipMemcpyAsync(signs_host, ctx->pwr_signs.p, n, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    *d_log_out = d_log;
    return SZHIP_OK;
}


// computeRangeSize_float_MSST19 (dataCompression.c:121-166) and the zero replacement of the _MSST19 wrappers (sz_float_pwr.c:2053-2062), on a COPY
// (the reference overwrites the caller's zeros).  nearZero: the first value of least non-zero magnitude -- or 0 when element 0 is zero (the scan
// starts from it and nothing is smaller in magnitude than 0); signs: from element 1 on, as the reference's loop.
template <class T>
int msst_prepare_impl(szhip_ctx *ctx, const void *data, int data_on_device, size_t n, double vmax, double ratio, void **d_prep_out,
                      unsigned char *signs_host, int *positive, double *near_zero_out, double *median_log, double *min_log_value)
{
    hipStream_t st = ctx->stream;
    const T *d_in = (const T *)data;
    if (!data_on_device) {
        TRY(ensure(ctx, ctx->in, n * sizeof(T)));
        TRY(staged_copy(ctx, ctx->in.p, data, (size_t)n * sizeof(T), true));
        d_in = (const T *)ctx->in.p;
    }
    TRY(ensure(ctx, ctx->pwr_log, n * sizeof(T)));
    TRY(ensure(ctx, ctx->pwr_signs, n));
    TRY(ensure(ctx, ctx->pwr_small, PWR_RED * 8));
    T *d_prep = (T *)ctx->pwr_log.p;
    u64 *red = (u64 *)ctx->pwr_small.p;
    const u64 init[MS_RED] = {~0ull, ~0ull, 0ull, 0ull};
    HIPCHK(hipMemcpyAsync(red, init, sizeof(init), hipMemcpyHostToDevice, st));
    const int grid = (int)std::min<int64_t>(((int64_t)n + 255) / 256, 4096);
    hipLaunchKernelGGL((k_msst_scan<T>), dim3(grid), dim3(256), 0, st, d_in, (int64_t)n, (unsigned char *)ctx->pwr_signs.p, red);
    HIPCHK(hipGetLastError());
    u64 res[MS_RED]; T first = 0;
    HIPCHK(hipMemcpyAsync(res, red, sizeof(res), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&first, d_in, sizeof(T), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    T near_zero = 0;
    if (first != 0 && res[MS_MINMAG] != ~0ull) {
        hipLaunchKernelGGL((k_msst_minidx<T>), dim3(grid), dim3(256), 0, st, d_in, (int64_t)n, res[MS_MINMAG], red);
        HIPCHK(hipGetLastError());
        u64 idx = 0;
        HIPCHK(hipMemcpyAsync(&idx, red + MS_MINIDX, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (idx >= n) FAIL(SZHIP_ERR_INTERNAL, "nearZero position out of range");
        HIPCHK(hipMemcpyAsync(&near_zero, d_in + idx, sizeof(T), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    const T multiplier = (T)pow(1 + ratio, -3.0001);
    const T zval = (T)(near_zero * multiplier);
    hipLaunchKernelGGL((k_msst_fill<T>), dim3(grid), dim3(256), 0, st, d_in, (int64_t)n, d_prep, zval);
    HIPCHK(hipGetLastError());
    *near_zero_out = (double)near_zero;
    *median_log = (double)(T)sqrt(fabs((double)(T)(near_zero * (T)vmax)));
    *min_log_value = (double)(T)((double)near_zero / ((1 + ratio) * (1 + ratio)));
    *positive = res[MS_NEG] ? 0 : 1;
    if (res[MS_NEG] && signs_host) HIPCHK(hipMemcpyAsync(signs_host, ctx->pwr_signs.p, n, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    *d_prep_out = d_prep;
    return SZHIP_OK;
}

template <class T>
int decompress14_pwr_impl(szhip_ctx *ctx, const unsigned char *stream, int stream_on_device, size_t stream_len, size_t body_off,
                          size_t r0, size_t r1, size_t r2, const unsigned char *signs_host, double threshold, bool msst, void *out, int out_on_device, szhip_stats *stats)
{
    const size_t n = (r0 ? r0 : 1) * (r1 ? r1 : 1) * r2;
    hipStream_t st = ctx->stream;
    TRY(ensure(ctx, ctx->pwr_log, n * sizeof(T)));
    T *d_log = (T *)ctx->pwr_log.p;
    TRY(decompress14_impl<T>(ctx, stream, stream_on_device, stream_len, body_off, msst ? 2 : 1, r0, r1, r2, d_log, 1, stats));
    const unsigned char *d_signs = nullptr;
    if (signs_host) {
        TRY(ensure(ctx, ctx->pwr_signs, n));
        HIPCHK(hipMemcpyAsync(ctx->pwr_signs.p, signs_host, n, hipMemcpyHostToDevice, st));
        d_signs = (const unsigned char *)ctx->pwr_signs.p;
    }
    T *d_out = (T *)out;
    if (!out_on_device) { TRY(ensure(ctx, ctx->out, n * sizeof(T))); d_out = (T *)ctx->out.p; }
    const int grid = (int)std::min<int64_t>(((int64_t)n + 255) / 256, 4096);
    if (msst) hipLaunchKernelGGL((k_msst_post<T>), dim3(grid), dim3(256), 0, st, (const T *)d_log, (int64_t)n, (T)threshold, d_signs, d_out);
    else hipLaunchKernelGGL((k_pwr_exp<T>), dim3(grid), dim3(256), 0, st, (const T *)d_log, (int64_t)n, (T)threshold, d_signs, d_out);
    HIPCHK(hipGetLastError());
    if (!out_on_device) TRY(staged_copy(ctx, out, d_out, n * sizeof(T), false));
    HIPCHK(hipStreamSynchronize(st));
    return SZHIP_OK;
}

// =====================================================================================================================
// FAST mode (szh_fast.h).  Container (little endian, 8-byte aligned sections):
//   "SZHF" | u8 version 1 | u8 dtype | u16 0 | u64 r0 r1 r2 | f64 eb | u32 intervals | u32 0,0,0 | u64 nA | u64 nB
//   | u32 tree_bytes | u32 n_nodes | u64 payload_bytes | tree (padded to 8) | i32 listA[nA] (padded) | i32 listBd[nB] (padded)
//   | T listB[nB] (padded) | Huffman payload
// =====================================================================================================================
#define SZF_HDR_FIXED 88
static inline size_t pad8(size_t x) { return (x + 7) & ~(size_t)7; }

template <class T>
int compress_fast_impl(szhip_ctx *ctx, const void *data, int data_on_device, size_t r0, size_t r1, size_t r2, double eb_in, unsigned intervals,
                       int out_on_device, unsigned char **out, size_t *out_size, szhip_stats *stats)
{
    const szf_geom g = szf_make_geom(r0, r1, r2);
    const int64_t n = g.n;
    const T eb = (T)eb_in;
    const double t_begin = now_ms();
    double host_ms = 0;
    hipStream_t st = ctx->stream;
    szhip_stats S; memset(&S, 0, sizeof(S));
    S.n_elements = (uint64_t)n; S.intervals = intervals;
    const T *d_in = (const T *)data;
    if (!data_on_device) {
        TRY(ensure(ctx, ctx->in, (size_t)n * sizeof(T)));
        TRY(staged_copy(ctx, ctx->in.p, data, (size_t)n * sizeof(T), true));
        d_in = (const T *)ctx->in.p;
    }
    TRY(ensure(ctx, ctx->small, SM_COUNT * 8));
    u64 *sm = (u64 *)ctx->small.p;
    HIPCHK(hipMemsetAsync(sm, 0, SM_COUNT * 8, st));
    // Two-pass form (szh_fast.h, round 3; SZ_HIP_FAST2=1): the codes are recomputed instead of stored -- pass A (statistics) here, pass B
    // (packing) and pass C (compaction) below.  Measured at 512^3 f32 it is the slower of the two end to end (1.12 ms against 0.96 ms:
    // the per-unit slots of pass B and their compaction cost more than the 2 N-byte code array they replace), so the code-array form
    // stays the default; a code longer than 32 bits takes the code-array form in any case.
    const szg_geom gg = szg_make_geom(r0, r1, r2);
    bool two_pass = tune_int("SZ_HIP_FAST2", 0) != 0 && gg.ntiles < 0x7fffffff;
    const int64_t nunits = (int64_t)r0 * (int64_t)r1 * gg.n2;
    TRY(ensure(ctx, ctx->hist, (size_t)(65536 + 8192) * 4 + 64));
    unsigned *d_hist = (unsigned *)ctx->hist.p;
    TRY(ensure_pinned(ctx, (size_t)intervals * 4 + 64));
    unsigned *h_hist = (unsigned *)ctx->pinned;
    uint16_t *d_codes = nullptr;
    u64 *d_ucnt = nullptr, *d_uoffc = nullptr, *d_ubits = nullptr, *d_uoff = nullptr;
    const int64_t nchunks = (n + 2047) / 2048;
    HIPCHK(hipEventRecord(ctx->ev[0], st));
    HIPCHK(hipEventRecord(ctx->ev[1], st));
    auto code_array_front = [&]() -> int {
        TRY(ensure(ctx, ctx->codes_nat, (size_t)n * 2 + 64));
        d_codes = (uint16_t *)ctx->codes_nat.p;
        const unsigned ntiles = (unsigned)((int64_t)g.n0 * g.n1 * g.n2);
        HIPCHK(hipEventRecord(ctx->ev[2], st));
        hipLaunchKernelGGL((k_fast_quant<T>), dim3((ntiles + 7) / 8 * 8), dim3(256), 0, st, g, d_in, d_codes, eb, (int)intervals / 2);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(ctx->ev[3], st));
        // histogram -> host code book; side-list counts meanwhile
        HIPCHK(hipMemsetAsync(d_hist, 0, (size_t)intervals * 4, st));
        {
            int rshift = 0; int use_lds = intervals <= 16384;
            if (use_lds) { while ((intervals << (rshift + 1)) <= 16384u && rshift < 6) ++rshift; }
            const size_t lds = use_lds ? ((size_t)intervals << rshift) * 4 : 16;
            int grid = (int)std::min<int64_t>((n / 8 + 255) / 256 + 1, 2048);
            hipLaunchKernelGGL(k_hist_u16, dim3(grid), dim3(256), lds, st, (const uint16_t *)d_codes, n, intervals, rshift, use_lds, d_hist, szh_rb_layout{0, 0, 0, 0, 0, 0}, 0, 0, 0, (int64_t)0);
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipMemcpyAsync(h_hist, d_hist, (size_t)intervals * 4, hipMemcpyDeviceToHost, st));
        TRY(ensure(ctx, ctx->col_zeros64, (size_t)nchunks * 8)); TRY(ensure(ctx, ctx->col_off, (size_t)nchunks * 8));
        TRY(ensure(ctx, ctx->reg_flags, (size_t)nchunks * 8)); TRY(ensure(ctx, ctx->reg_rank, (size_t)nchunks * 8));
        TRY(ensure(ctx, ctx->chunk_bits, (size_t)nchunks * 8)); TRY(ensure(ctx, ctx->chunk_off, (size_t)nchunks * 8));
        hipLaunchKernelGGL(k_fast_count, dim3((unsigned)nchunks), dim3(256), 0, st, (const uint16_t *)d_codes, n, (u64 *)ctx->col_zeros64.p, (u64 *)ctx->reg_flags.p);
        TRY(scan_u64(ctx, (const u64 *)ctx->col_zeros64.p, nchunks, (u64 *)ctx->col_off.p, sm + SM_TOTAL_UNPRED));
        TRY(scan_u64(ctx, (const u64 *)ctx->reg_flags.p, nchunks, (u64 *)ctx->reg_rank.p, sm + SM_SCRATCH));
        return SZHIP_OK;
    };
    if (two_pass) {
        TRY(ensure(ctx, ctx->fast_units, (size_t)nunits * 8 * 4 + 64));
        TRY(ensure(ctx, ctx->fast_slots, (size_t)nunits * SZG_SLOT_WORDS * 4 + 64));
        d_ucnt = (u64 *)ctx->fast_units.p; d_uoffc = d_ucnt + nunits; d_ubits = d_uoffc + nunits; d_uoff = d_ubits + nunits;
        HIPCHK(hipMemsetAsync(d_hist, 0, (size_t)intervals * 4, st));
        HIPCHK(hipMemsetAsync(d_ucnt, 0, (size_t)nunits * 8, st));          // k_fast_stat adds to the few units that have side-list entries
        HIPCHK(hipEventRecord(ctx->ev[2], st));
        {   // persistent: exactly the workgroups that are resident at once (a second, half-empty round cost 40 % at 768 over 512)
            // (occupancy and CU count per CONTEXT: function-local statics would keep the first device's values for every other one and be
            //  written without synchronisation by the threads of a pool)
            if (!ctx->fast_stat_per_cu[sizeof(T) == 8]) { int nb = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_fast_stat<T>, 256, 0) != hipSuccess || nb < 1) nb = 2; ctx->fast_stat_per_cu[sizeof(T) == 8] = nb; }
            const int wgs = tune_int("SZ_HIP_FAST_STAT_WGS", ctx->fast_stat_per_cu[sizeof(T) == 8] * ctx->cus);
            hipLaunchKernelGGL((k_fast_stat<T>), dim3((unsigned)std::min<int64_t>(gg.ntiles, wgs)), dim3(256), 0, st, gg, d_in, eb, (int)intervals / 2, intervals, d_hist, d_ucnt, (unsigned *)(sm + SM_TICKET));
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(ctx->ev[3], st));
        HIPCHK(hipMemcpyAsync(h_hist, d_hist, (size_t)intervals * 4, hipMemcpyDeviceToHost, st));
        TRY(scan_u64(ctx, (const u64 *)d_ucnt, nunits, d_uoffc, sm + SM_TOTAL_UNPRED));     // both side-list ranks in one scan (low / high half)
    } else TRY(code_array_front());
    HIPCHK(hipStreamSynchronize(st));
    double h0 = now_ms();
    szhost_huff *hf = szhost_huff_build(2 * (int)intervals, h_hist, nullptr, intervals);
    if (!hf) FAIL(SZHIP_ERR_INTERNAL, "Huffman build failed");
    if (two_pass) {
        unsigned maxlen = 0;
        for (unsigned s2 = 0; s2 < intervals; ++s2) if (hf->len[s2] > maxlen) maxlen = hf->len[s2];
        if (maxlen > 32) {                                  // a unit's slot holds 64 x 32 bits: take the code-array form (same histogram, same tree)
            szhost_huff_free(hf);
            two_pass = false;
            HIPCHK(hipMemsetAsync(sm, 0, SM_COUNT * 8, st));
            TRY(code_array_front());
            HIPCHK(hipStreamSynchronize(st));
            hf = szhost_huff_build(2 * (int)intervals, h_hist, nullptr, intervals);
            if (!hf) FAIL(SZHIP_ERR_INTERNAL, "Huffman build failed");
        }
    }
    const u64 nA = h_hist[0], nB = h_hist[1];
    S.n_unpred = nA + nB;
    const size_t tree_bytes = szhost_huff_tree_size(hf);
    const u64 total_bits = hf->total_bits;
    const size_t pay_bytes = (size_t)((total_bits + 7) / 8);
    const size_t offA = SZF_HDR_FIXED + pad8(tree_bytes), offBd = offA + pad8((size_t)nA * 4), offB = offBd + pad8((size_t)nB * 4);
    const size_t pay_off = offB + pad8((size_t)nB * sizeof(T));
    const size_t total_len = pay_off + pay_bytes;
    std::vector<unsigned char> hdr(offA, 0);
    {
        unsigned char *q = hdr.data();
        memcpy(q, "SZHF", 4); q[4] = 1; q[5] = (unsigned char)(sizeof(T) == 8); q += 8;
        const uint64_t dims[3] = {r0, r1, r2}; memcpy(q, dims, 24); q += 24;
        const double ebd = (double)eb; memcpy(q, &ebd, 8); q += 8;
        const uint32_t w[4] = {intervals, 0, 0, 0}; memcpy(q, w, 16); q += 16;
        memcpy(q, &nA, 8); q += 8; memcpy(q, &nB, 8); q += 8;
        const uint32_t tw[2] = {(uint32_t)tree_bytes, (uint32_t)hf->n_nodes}; memcpy(q, tw, 8); q += 8;
        const uint64_t pb = pay_bytes; memcpy(q, &pb, 8); q += 8;
        szhost_huff_tree_write(hf, q);
    }
    std::vector<u64> tab_code(intervals); std::vector<uint8_t> tab_len(intervals);
    for (unsigned s2 = 0; s2 < intervals; ++s2) { tab_code[s2] = hf->code[s2]; tab_len[s2] = hf->len[s2]; }
    szhost_huff_free(hf);
    host_ms += now_ms() - h0;
    TRY(ensure(ctx, ctx->code_tab, (size_t)intervals * 8));
    TRY(ensure(ctx, ctx->len_tab, (size_t)intervals));
    HIPCHK(hipMemcpyAsync(ctx->code_tab.p, tab_code.data(), (size_t)intervals * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ctx->len_tab.p, tab_len.data(), (size_t)intervals, hipMemcpyHostToDevice, st));
    TRY(ensure(ctx, ctx->stream_buf, total_len + 64));
    unsigned char *d_stream = (unsigned char *)ctx->stream_buf.p;
    HIPCHK(hipMemsetAsync(d_stream, 0, total_len + 64, st));
    HIPCHK(hipMemcpyAsync(d_stream, hdr.data(), hdr.size(), hipMemcpyHostToDevice, st));
    if (two_pass) {
        {
            if (!ctx->fast_pack_per_cu[sizeof(T) == 8]) { int nb = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_fast_pack<T>, 256, 0) != hipSuccess || nb < 1) nb = 2; ctx->fast_pack_per_cu[sizeof(T) == 8] = nb; }
            const int wgs = tune_int("SZ_HIP_FAST_PACK_WGS", ctx->fast_pack_per_cu[sizeof(T) == 8] * ctx->cus);
            hipLaunchKernelGGL((k_fast_pack<T>), dim3((unsigned)std::min<int64_t>(gg.ntiles, wgs)), dim3(256), 0, st, gg, d_in, eb, (int)intervals / 2, intervals,
                               (const u64 *)ctx->code_tab.p, (const uint8_t *)ctx->len_tab.p, (const u64 *)d_uoffc,
                               (int32_t *)(d_stream + offA), (int32_t *)(d_stream + offBd), (T *)(d_stream + offB), d_ubits, (unsigned *)ctx->fast_slots.p,
                               (unsigned *)(sm + SM_TICKET) + 1);
        }
        HIPCHK(hipGetLastError());
        TRY(scan_u64(ctx, (const u64 *)d_ubits, nunits, d_uoff, sm + SM_TOTAL_BITS));
        if (total_bits > 0) {
            hipLaunchKernelGGL(k_fast_compact, dim3((unsigned)((nunits + 255) / 256)), dim3(256), 0, st, nunits, (const u64 *)d_ubits, (const u64 *)d_uoff,
                               (const unsigned *)ctx->fast_slots.p, (u64)pay_off * 8, (unsigned *)d_stream);
            HIPCHK(hipGetLastError());
        }
    } else {
    if (nA + nB > 0) {
        hipLaunchKernelGGL((k_fast_lists<T>), dim3((unsigned)nchunks), dim3(256), 0, st, g, (const uint16_t *)d_codes, (const u64 *)ctx->col_off.p,
                           (const u64 *)ctx->reg_rank.p, d_in, eb, (int32_t *)(d_stream + offA), (int32_t *)(d_stream + offBd), (T *)(d_stream + offB));
        HIPCHK(hipGetLastError());
    }
    if (total_bits > 0) {
        hipLaunchKernelGGL(k_chunk_bits, dim3((unsigned)((nchunks + SZH_CB_PER - 1) / SZH_CB_PER)), dim3(256), 0, st, (const uint16_t *)d_codes, n, (const uint8_t *)ctx->len_tab.p,
                           intervals, (u64 *)ctx->chunk_bits.p);
        TRY(scan_u64(ctx, (const u64 *)ctx->chunk_bits.p, nchunks, (u64 *)ctx->chunk_off.p, sm + SM_TOTAL_BITS));
        hipLaunchKernelGGL(k_encode, dim3((unsigned)((nchunks + SZH_ENC_PER - 1) / SZH_ENC_PER)), dim3(256), 0, st, (const uint16_t *)d_codes, n, (const u64 *)ctx->code_tab.p,
                           (const uint8_t *)ctx->len_tab.p, intervals, (const u64 *)ctx->chunk_off.p, (u64)pay_off * 8, (unsigned *)d_stream);
        HIPCHK(hipGetLastError());
    }
    }
    HIPCHK(hipEventRecord(ctx->ev[4], st));
    u64 h_small[SM_COUNT];
    HIPCHK(hipMemcpyAsync(h_small, sm, SM_COUNT * 8, hipMemcpyDeviceToHost, st));
    if (out_on_device == 2) {
        if (!*out || *out_size < total_len) FAIL(SZHIP_ERR_ARG, "caller's device buffer too small (%zu < %zu)", *out_size, total_len);
        HIPCHK(hipMemcpyAsync(*out, d_stream, total_len, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
    } else if (out_on_device) {
        HIPCHK(hipStreamSynchronize(st));
        *out = d_stream;
    } else {
        unsigned char *h = (unsigned char *)malloc(total_len ? total_len : 1);
        if (!h) FAIL(SZHIP_ERR_INTERNAL, "out of host memory");
        TRY(staged_copy(ctx, h, d_stream, total_len, false));
        *out = h;
    }
    *out_size = total_len;
    S.quant_kernel = two_pass ? 2 : 0;                         // 2: the two-pass form of the fast mode
    const bool counts_ok = two_pass ? ((h_small[SM_TOTAL_UNPRED] & 0xffffffffull) == nA && (h_small[SM_TOTAL_UNPRED] >> 32) == nB)
                                    : (h_small[SM_TOTAL_UNPRED] == nA && h_small[SM_SCRATCH] == nB);
    if ((total_bits > 0 && h_small[SM_TOTAL_BITS] != total_bits) || !counts_ok)
        FAIL_PUBLISHED(SZHIP_ERR_INTERNAL, "fast mode: entropy stage mismatch");
    float ms = 0;
    hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3]); S.ms_quant = ms;
    hipEventElapsedTime(&ms, ctx->ev[3], ctx->ev[4]); S.ms_entropy = ms;
    S.ms_host = host_ms; S.ms_total = now_ms() - t_begin; S.out_bytes = total_len;
    if (stats) *stats = S;
    return SZHIP_OK;
}

template <class T>
int decompress_fast_impl(szhip_ctx *ctx, const unsigned char *stream_in, int stream_on_device, size_t stream_len, size_t r0, size_t r1, size_t r2,
                         void *out, int out_on_device, szhip_stats *stats)
{
    const szf_geom g = szf_make_geom(r0, r1, r2);
    const int64_t n = g.n;
    const double t_begin = now_ms();
    hipStream_t st = ctx->stream;
    szhip_stats S; memset(&S, 0, sizeof(S));
    S.n_elements = (uint64_t)n;
    TRY(ensure(ctx, ctx->stream_buf, stream_len + 64));
    unsigned char *d_stream = (unsigned char *)ctx->stream_buf.p;
    if (stream_on_device) { if (stream_in != d_stream) HIPCHK(hipMemcpyAsync(d_stream, stream_in, stream_len, hipMemcpyDeviceToDevice, st)); }
    else TRY(staged_copy(ctx, d_stream, stream_in, stream_len, true));
    HIPCHK(hipMemsetAsync(d_stream + stream_len, 0, 64, st));
    HIPCHK(hipEventRecord(ctx->ev[0], st));
    if (stream_len < SZF_HDR_FIXED) FAIL(SZHIP_ERR_STREAM, "truncated stream");
    std::vector<unsigned char> hbuf;
    const unsigned char *hs = stream_in;
    auto fetch = [&](size_t want) -> int {
        if (!stream_on_device) return SZHIP_OK;
        hbuf.resize(want);
        HIPCHK(hipMemcpyAsync(hbuf.data(), d_stream, want, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        hs = hbuf.data();
        return SZHIP_OK;
    };
    TRY(fetch(SZF_HDR_FIXED));
    if (memcmp(hs, "SZHF", 4) != 0 || hs[4] != 1 || hs[5] != (unsigned char)(sizeof(T) == 8)) FAIL(SZHIP_ERR_STREAM, "not a fast-mode stream of this type");
    uint64_t dims[3]; memcpy(dims, hs + 8, 24);
    if (dims[0] != r0 || dims[1] != r1 || dims[2] != r2) FAIL(SZHIP_ERR_STREAM, "dimensions differ from the stream's");
    double ebd; memcpy(&ebd, hs + 32, 8);
    uint32_t w[4]; memcpy(w, hs + 40, 16);
    u64 nA, nB; memcpy(&nA, hs + 56, 8); memcpy(&nB, hs + 64, 8);
    uint32_t tw[2]; memcpy(tw, hs + 72, 8);
    uint64_t pay_bytes; memcpy(&pay_bytes, hs + 80, 8);
    const unsigned intervals = w[0];
    if (!(ebd > 0) || intervals < 4 || intervals > 65536 || (intervals & 1)) FAIL(SZHIP_ERR_STREAM, "bad fast-mode header");
    const size_t tree_bytes = tw[0]; const int node_count = (int)tw[1];
    if (nA > (u64)n || nB > (u64)n || tree_bytes > stream_len || pay_bytes > stream_len) FAIL(SZHIP_ERR_STREAM, "truncated stream");
    const size_t offA = SZF_HDR_FIXED + pad8(tree_bytes), offBd = offA + pad8((size_t)nA * 4), offB = offBd + pad8((size_t)nB * 4);
    const size_t pay_off = offB + pad8((size_t)nB * sizeof(T));
    if (pay_off + pay_bytes > stream_len || node_count <= 0 || szhost_huff_serial_size(node_count) > tree_bytes) FAIL(SZHIP_ERR_STREAM, "truncated stream");
    TRY(fetch(SZF_HDR_FIXED + tree_bytes));
    szhost_huff *hf = szhost_huff_from_bytes(2 * (int)intervals, hs + SZF_HDR_FIXED, node_count);
    if (!hf) FAIL(SZHIP_ERR_STREAM, "bad Huffman tree");
    std::vector<uint32_t> dtab((size_t)hf->n_nodes * 2);
    szhost_huff_decode_table(hf, dtab.data());
    const int single_symbol = hf->t[0] ? (int)hf->C[0] : -1;
    const int n_nodes = hf->n_nodes;
    szhost_huff_free(hf);
    S.intervals = intervals; S.n_unpred = nA + nB;
    TRY(ensure(ctx, ctx->small, SM_COUNT * 8));
    u64 *sm = (u64 *)ctx->small.p;
    HIPCHK(hipMemsetAsync(sm, 0, SM_COUNT * 8, st));
    TRY(ensure(ctx, ctx->codes_nat, (size_t)n * 2 + 64));
    uint16_t *d_codes = (uint16_t *)ctx->codes_nat.p;
    u64 total_sym = 0;
    TRY(huff_decode_device(ctx, sm, d_stream + pay_off, (unsigned)std::min<size_t>(pay_off, 4096), (u64)pay_bytes * 8, dtab, n_nodes, single_symbol, n, d_codes, &total_sym));
    const int64_t nchunks = (n + 2047) / 2048;
    TRY(ensure(ctx, ctx->col_zeros64, (size_t)nchunks * 8)); TRY(ensure(ctx, ctx->col_off, (size_t)nchunks * 8));
    TRY(ensure(ctx, ctx->reg_flags, (size_t)nchunks * 8)); TRY(ensure(ctx, ctx->reg_rank, (size_t)nchunks * 8));
    hipLaunchKernelGGL(k_fast_count, dim3((unsigned)nchunks), dim3(256), 0, st, (const uint16_t *)d_codes, n, (u64 *)ctx->col_zeros64.p, (u64 *)ctx->reg_flags.p);
    TRY(scan_u64(ctx, (const u64 *)ctx->col_zeros64.p, nchunks, (u64 *)ctx->col_off.p, sm + SM_TOTAL_UNPRED));
    TRY(scan_u64(ctx, (const u64 *)ctx->reg_flags.p, nchunks, (u64 *)ctx->reg_rank.p, sm + SM_SCRATCH));
    u64 h_small[SM_COUNT];
    HIPCHK(hipMemcpyAsync(h_small, sm, SM_COUNT * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HDEC_CHECK(ctx);
    total_sym = ctx->hdec_res[0];
    if ((int64_t)total_sym < n) FAIL(SZHIP_ERR_STREAM, "Huffman payload holds %llu symbols, need %lld", (unsigned long long)total_sym, (long long)n);
    if (h_small[SM_TOTAL_UNPRED] != nA || h_small[SM_SCRATCH] != nB) FAIL(SZHIP_ERR_STREAM, "side lists do not match the codes");
    T *d_out = (T *)out;
    if (!out_on_device) { TRY(ensure(ctx, ctx->out, (size_t)n * sizeof(T))); d_out = (T *)ctx->out.p; }
    HIPCHK(hipEventRecord(ctx->ev[1], st));
    // deltas + scan along dim2, scan along dim1, scan along dim0 + scaling (three passes over a uint32 workspace), raw values last
    TRY(ensure(ctx, ctx->faceI, (size_t)n * 4 + 64));
    uint32_t *d_acc = (uint32_t *)ctx->faceI.p;
    HIPCHK(hipEventRecord(ctx->ev[2], st));
    {
        const int64_t rows = (int64_t)g.r0 * g.r1;
        if (nA + nB > 0)
            hipLaunchKernelGGL(k_fast_scatter, dim3((unsigned)nchunks), dim3(256), 0, st, (const uint16_t *)d_codes, n, (const u64 *)ctx->col_off.p,
                               (const u64 *)ctx->reg_rank.p, (const int32_t *)(d_stream + offA), (const int32_t *)(d_stream + offBd), d_acc);
        hipLaunchKernelGGL(k_fast_expand_scan2, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, g, (const uint16_t *)d_codes, (int)intervals / 2, d_acc);
        hipLaunchKernelGGL(k_fast_scan1, dim3((unsigned)(((int64_t)g.r0 * g.r2 + 255) / 256)), dim3(256), 0, st, g, d_acc);
        hipLaunchKernelGGL((k_fast_scan0_out<T>), dim3((unsigned)(((int64_t)g.r1 * g.r2 + 255) / 256)), dim3(256), 0, st, g, (const uint32_t *)d_acc, d_out, (T)ebd);
    }
    HIPCHK(hipEventRecord(ctx->ev[3], st));
    if (nB > 0) hipLaunchKernelGGL((k_fast_raw<T>), dim3((unsigned)nchunks), dim3(256), 0, st, (const uint16_t *)d_codes, n, (const u64 *)ctx->reg_rank.p,
                                   (const T *)(d_stream + offB), d_out);
    HIPCHK(hipGetLastError());
    if (!out_on_device) TRY(staged_copy(ctx, out, d_out, (size_t)n * sizeof(T), false));
    HIPCHK(hipStreamSynchronize(st));
    float ms = 0;
    hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]); S.ms_entropy = ms;
    hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3]); S.ms_quant = ms;
    S.ms_total = now_ms() - t_begin; S.out_bytes = (uint64_t)n * sizeof(T);
    if (stats) *stats = S;
    return SZHIP_OK;
}


// =====================================================================================================================
// The reference's OpenMP container for 3-D arrays (szh_omp.h; sz/src/sz_omp.c:63-358, inverse :366-566).  Stream, behind the caller's
// 4 + MetaDataByteLength parameter bytes (`meta`):
//   u32be thread_num | T eb (big endian) | u32be intervals | u32be tree_bytes | u32be nodes | tree
//   | u32 ucount[nb] | T first[nb] | the boxes' verbatim values, box after box | u64 payload_bytes[nb] | the boxes' Huffman payloads
// (the tables behind the tree in the host's byte order, as the reference memcpy's them).
// =====================================================================================================================
static int omp_box_grid(szhip_ctx *ctx, int thread_num, size_t r0, size_t r1, size_t r2, size_t elem, szh_omp_geom *g)
{
    if (thread_num < 1) FAIL(SZHIP_ERR_ARG, "thread_num %d", thread_num);
    // sz_omp.c:88-117: the exponent of two is spread over the three dimensions, dim 0 first; the rest of thread_num goes to dim 2
    int order = 0; while ((2 << order) <= thread_num) ++order;
    const int bb = order / 3;
    size_t nx, ny;
    switch (order % 3) { case 0: nx = (size_t)1 << bb; ny = (size_t)1 << bb; break; case 1: nx = (size_t)1 << (bb + 1); ny = (size_t)1 << bb; break; default: nx = (size_t)1 << (bb + 1); ny = (size_t)1 << (bb + 1); }
    const size_t nz = (size_t)thread_num / (nx * ny);
    if (r0 == 0 || r1 == 0 || r2 == 0 || r0 * r1 * r2 >= ((size_t)1 << 40)) FAIL(SZHIP_ERR_UNSUP, "OpenMP container: a 3-D array is needed");
    if (r0 % nx || r1 % ny || r2 % nz)
        FAIL(SZHIP_ERR_UNSUP, "OpenMP container: the %zu x %zu x %zu box grid of thread_num %d does not divide %zu x %zu x %zu (on an uneven grid the "
             "reference's code book depends on uninitialised memory)", nx, ny, nz, thread_num, r0, r1, r2);
    g->nx = (int)nx; g->ny = (int)ny; g->nz = (int)nz;
    g->c0 = (int)(r0 / nx); g->c1 = (int)(r1 / ny); g->c2 = (int)(r2 / nz);
    g->d0 = (int64_t)(r1 * r2); g->d1 = (int64_t)r2;
    g->nb = (int)(nx * ny * nz);
    const size_t bel = (size_t)g->c0 * g->c1 * g->c2;
    if ((size_t)g->c0 * g->c1 > SZH_OMP_MAX_ROWS || bel >= ((size_t)1 << 28))
        FAIL(SZHIP_ERR_UNSUP, "OpenMP container: a box face of %d x %d rows (at most %d; raise thread_num)", g->c0, g->c1, SZH_OMP_MAX_ROWS);
    g->bel = (int)bel;
    g->cpb = (int)((bel + SZH_ENC_CHUNK - 1) / SZH_ENC_CHUNK);
    g->vec = (g->c2 % 4 == 0 && r2 % 4 == 0) ? 1 : 0;          // (the base address is looked at by the caller)
    g->tile8 = (g->c0 % 8 == 0 && g->c1 % 8 == 0) ? 1 : 0;
    g->pitch = g->c1;
    if (g->tile8) while (g->pitch % 16 != 8) ++g->pitch;       // 8 or 24 modulo 32
    (void)elem;
    return SZHIP_OK;
}

// the column-per-lane sweep of szh_ompcol.h serves 32 x 32 box faces (any number of planes), two boxes to a wavefront, rows read 16 bytes at a time
static bool omp_col_applies(const szh_omp_geom &g, const void *base, size_t row_pitch_bytes)
{
    return g.c1 == 32 && g.c2 == 32 && g.nb % 2 == 0 && ((uintptr_t)base & 15u) == 0 && row_pitch_bytes % 16 == 0 && tune_int("SZ_HIP_OMP_COL", 1) != 0;
}

template <class T>
int compress_omp_impl(szhip_ctx *ctx, const void *data, int data_on_device, size_t r0, size_t r1, size_t r2, double eb_in, int thread_num,
                      const szhip_params *prm, const unsigned char *meta, size_t meta_len, int out_on_device, unsigned char **out, size_t *out_size,
                      szhip_stats *stats)
{
    szh_omp_geom g;
    TRY(omp_box_grid(ctx, thread_num, r0, r1, r2, sizeof(T), &g));
    const szh_geom3 G = szh_make_geom3((int)r0, (int)r1, (int)r2);
    const int64_t n = G.n;
    const T eb = (T)eb_in;                                     // `float realPrecision` of sz_omp.c:63 (double: :578)
    if (!(eb > 0)) FAIL(SZHIP_ERR_ARG, "error bound %g", eb_in);
    const double t_begin = now_ms();
    double host_ms = 0;
    hipStream_t st = ctx->stream;
    szhip_stats S; memset(&S, 0, sizeof(S));
    S.n_elements = (uint64_t)n; S.n_blocks = (uint64_t)g.nb;
    const T *d_in = (const T *)data;
    if (!data_on_device) {
        TRY(ensure(ctx, ctx->in, (size_t)n * sizeof(T)));
        TRY(staged_copy(ctx, ctx->in.p, data, (size_t)n * sizeof(T), true));
        d_in = (const T *)ctx->in.p;
    }
    if ((uintptr_t)d_in & 15u) g.vec = 0;
    TRY(ensure(ctx, ctx->small, SM_COUNT * 8));
    u64 *sm = (u64 *)ctx->small.p;
    HIPCHK(hipMemsetAsync(sm, 0, SM_COUNT * 8, st));
    HIPCHK(hipEventRecord(ctx->ev[0], st));
    // ---- interval count (sz_omp.c:73-82: optimize_intervals_float_3D_opt over the whole array when it is not fixed)
    unsigned intervals = prm->quantization_intervals;
    if (intervals == 0) {
        const unsigned max_radius = prm->max_quant_intervals / 2;
        TRY(ensure(ctx, ctx->hist, (size_t)(max_radius + 8192) * 4 + 64));
        TRY(ensure_pinned(ctx, (size_t)(max_radius + 8192) * 4 + 64));
        unsigned *d_rh = (unsigned *)ctx->hist.p, *d_fh = d_rh + max_radius;
        HIPCHK(hipMemsetAsync(d_rh, 0, (size_t)(max_radius + 8192) * 4, st));
        const int64_t nrows = szh_sample_row_limit(G, prm->sample_distance);
        if (G.g0.count <= 1 || G.g1.count <= 1) {                   // a degenerate 3-D array: the reference's walk, literally (k_sample_walk)
            hipLaunchKernelGGL((k_sample_walk<T, false>), dim3(1), dim3(64), 0, st, G, d_in, prm->sample_distance, (double)eb, (T)0, max_radius, d_rh, d_fh, sm + SM_WITHIN);
            HIPCHK(hipGetLastError());
        } else if (nrows > 0) {
            int grid = (int)std::min<int64_t>((nrows + 255) / 256, 1024);
            hipLaunchKernelGGL((k_sample<T, false>), dim3(grid), dim3(256), 0, st, G, d_in, nrows, prm->sample_distance, (double)eb, (T)0,
                               max_radius, d_rh, d_fh, sm + SM_WITHIN);
            HIPCHK(hipGetLastError());
        }
        unsigned *h_rh = (unsigned *)ctx->pinned;
        HIPCHK(hipMemcpyAsync(h_rh, d_rh, (size_t)max_radius * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        double h0 = now_ms();
        u64 total = 0;
        for (unsigned i = 0; i < max_radius; ++i) total += h_rh[i];
        const size_t target = (size_t)((float)total * prm->pred_threshold);
        size_t sum = 0; unsigned i = 0;
        for (; i < max_radius; ++i) { sum += h_rh[i]; if (sum > target) break; }
        if (i >= max_radius) i = max_radius - 1;
        unsigned p2 = 2 * (i + 1); p2 -= 1; p2 |= p2 >> 1; p2 |= p2 >> 2; p2 |= p2 >> 4; p2 |= p2 >> 8; p2 |= p2 >> 16; p2 += 1;
        intervals = p2 < 32 ? 32 : p2;
        host_ms += now_ms() - h0;
    }
    if (intervals > 65536 || intervals < 4) FAIL(SZHIP_ERR_UNSUP, "quantization interval count %u outside [4,65536]", intervals);
    S.intervals = intervals;
    HIPCHK(hipEventRecord(ctx->ev[1], st));

    // ---- the boxes: predict + quantise
    TRY(ensure(ctx, ctx->codes_nat, (size_t)n * 2 + 64));
    uint16_t *d_codes = (uint16_t *)ctx->codes_nat.p;
    TRY(ensure(ctx, ctx->zcnt, (size_t)g.nb * 4));
    TRY(ensure(ctx, ctx->samples, (size_t)g.nb * sizeof(T)));
    TRY(ensure(ctx, ctx->col_zeros64, (size_t)g.nb * 8)); TRY(ensure(ctx, ctx->col_off, (size_t)g.nb * 8 + 8));
    unsigned *d_ucount = (unsigned *)ctx->zcnt.p; T *d_first = (T *)ctx->samples.p;
    u64 *d_ucount64 = (u64 *)ctx->col_zeros64.p, *d_uoff = (u64 *)ctx->col_off.p;
    const int rows = g.c0 * g.c1, box_threads = rows;      // one lane per row
    const bool lean = tune_int("SZ_HIP_OMP_LEAN", 1) != 0;                     // (0: the entropy stage of round 3, kept for comparison)
    const bool box_hist = lean && intervals <= 1024 && (size_t)g.nb * intervals * 4 <= ((size_t)64 << 20) && g.bel % 8 == 0;
    bool sweep_counted = true;
    HIPCHK(hipEventRecord(ctx->ev[2], st));
    if (omp_col_applies(g, d_in, r2 * sizeof(T))) {        // the column-per-lane sweep (szh_ompcol.h): a wavefront per pair of boxes
        szh_oc::sweep_args<T> oa;
        oa.g = g; oa.data = d_in; oa.out = nullptr; oa.eb = eb; oa.recip = (T)(1 / eb); oa.intervals = (int)intervals; oa.codes = d_codes;
        oa.ucount = d_ucount; oa.ucount64 = d_ucount64; oa.first = d_first; oa.uoff = nullptr; oa.vflags = nullptr; oa.fw = 0; oa.dbg_no_code_stores = tune_int("SZ_HIP_OMP_DBG_NOSTORE", 0);
        // (with a histogram per box coming anyway, the boxes' counts of verbatim values are its bins 0: the sweep leaves the counting out --
        //  two vector instructions per step of a kernel that is bound by exactly those)
        if (box_hist) { sweep_counted = false; hipLaunchKernelGGL((k_omp_col<T, 32, 32, false, false>), dim3((unsigned)(g.nb / 2)), dim3(64), 0, st, oa); }
        else hipLaunchKernelGGL((k_omp_col<T, 32, 32, false, true>), dim3((unsigned)(g.nb / 2)), dim3(64), 0, st, oa);
    } else if (g.vec) hipLaunchKernelGGL((k_omp_box<T, false, true>), dim3((unsigned)g.nb), dim3((unsigned)box_threads), (size_t)4 * g.c0 * g.pitch * sizeof(T), st, g, d_in, (T *)nullptr, eb, (T)(1 / eb),
                                  (int)intervals, d_codes, d_ucount, d_ucount64, d_first, (const T *)nullptr, (const u64 *)nullptr);
    else hipLaunchKernelGGL((k_omp_box<T, false, false>), dim3((unsigned)g.nb), dim3((unsigned)box_threads), (size_t)4 * g.c0 * g.pitch * sizeof(T), st, g, d_in, (T *)nullptr, eb, (T)(1 / eb),
                            (int)intervals, d_codes, d_ucount, d_ucount64, d_first, (const T *)nullptr, (const u64 *)nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ctx->ev[3], st));
    S.quant_kernel_launches = 1;

    // ---- ONE histogram over all boxes -> code book (host); the ranks of the boxes' verbatim values meanwhile.  Small alphabets: a
    // histogram per box on the way (k_omp_hist_box), from which the boxes' payload sizes follow without another pass over the codes
    TRY(ensure(ctx, ctx->hist, (size_t)(65536 + 8192) * 4 + 64));
    unsigned *d_hist = (unsigned *)ctx->hist.p;
    TRY(ensure_pinned(ctx, (size_t)intervals * 4 + 64));
    unsigned *h_hist = (unsigned *)ctx->pinned;
    HIPCHK(hipMemsetAsync(d_hist, 0, (size_t)intervals * 4, st));
    unsigned *d_hist_box = nullptr;
    if (box_hist) {
        TRY(ensure(ctx, ctx->chunk_bits, (size_t)g.nb * intervals * 4));
        d_hist_box = (unsigned *)ctx->chunk_bits.p;
        // lane-private copies of the bins against same-address atomics -- but no more copies than the box has codes to spread over them
        // (a box of 4096 codes with 64 copies of 32 bins spent its time clearing and summing 32 KB: 0.40 ms for 32 768 such boxes)
        int rshift = 0;
        while ((intervals << (rshift + 1)) <= 8192u && rshift < 6 && ((size_t)intervals << (rshift + 1)) * 16 <= (size_t)g.bel) ++rshift;
        const int hist_per_wg = std::max(1, std::min(16, 32768 / std::max(1, g.bel)));
        hipLaunchKernelGGL(k_omp_hist_box, dim3((unsigned)((g.nb + hist_per_wg - 1) / hist_per_wg)), dim3(256), ((size_t)intervals << rshift) * 4, st, g.bel, (const uint16_t *)d_codes, intervals, rshift,
                           d_hist_box, d_hist, sweep_counted ? (unsigned *)nullptr : d_ucount, sweep_counted ? (u64 *)nullptr : d_ucount64, g.nb, hist_per_wg);
        HIPCHK(hipGetLastError());
    } else {
        int rshift = 0; int use_lds = intervals <= 16384;
        if (use_lds) { while ((intervals << (rshift + 1)) <= 16384u && rshift < 6) ++rshift; }
        const size_t lds = use_lds ? ((size_t)intervals << rshift) * 4 : 16;
        int grid = (int)std::min<int64_t>((n / 8 + 255) / 256 + 1, 2048);
        hipLaunchKernelGGL(k_hist_u16, dim3(grid), dim3(256), lds, st, (const uint16_t *)d_codes, n, intervals, rshift, use_lds, d_hist, szh_rb_layout{0, 0, 0, 0, 0, 0}, 0, 0, 0, (int64_t)0);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipMemcpyAsync(h_hist, d_hist, (size_t)intervals * 4, hipMemcpyDeviceToHost, st));
    if (!box_hist) TRY(scan_u64(ctx, (const u64 *)d_ucount64, g.nb, d_uoff, sm + SM_TOTAL_UNPRED));      // (with per-box histograms: in k_omp_layout, below)
    HIPCHK(hipStreamSynchronize(st));
    const u64 E = h_hist[0];
    S.n_unpred = E;
    double h0 = now_ms();
    szhost_huff *hf = szhost_huff_build(2 * (int)intervals, h_hist, nullptr, intervals);
    if (!hf) FAIL(SZHIP_ERR_INTERNAL, "Huffman build failed");
    const size_t tree_bytes = szhost_huff_tree_size(hf);
    const u64 total_bits = hf->total_bits;
    std::vector<u64> tab_code(intervals); std::vector<uint8_t> tab_len(intervals);
    for (unsigned s2 = 0; s2 < intervals; ++s2) { tab_code[s2] = hf->code[s2]; tab_len[s2] = hf->len[s2]; }
    // ---- container: everything up to the payloads has a known size now; the payloads take at most a byte of padding per box
    const size_t hdr_len = meta_len + 4 + sizeof(T) + 4 + 4 + 4 + tree_bytes;
    const size_t off_ucount = hdr_len, off_first = off_ucount + (size_t)g.nb * 4, off_unpred = off_first + (size_t)g.nb * sizeof(T);
    const size_t off_sizes = off_unpred + (size_t)E * sizeof(T), off_pay = off_sizes + (size_t)g.nb * 8;
    const size_t cap_len = off_pay + (size_t)((total_bits + 7) / 8) + (size_t)g.nb;
    std::vector<unsigned char> hdr(hdr_len, 0);
    {
        unsigned char *q = hdr.data();
        memcpy(q, meta, meta_len); q += meta_len;
        szhost_put_u32be(q, (uint32_t)g.nb); q += 4;              // (`thread_num` after the grid has been cut: sz_omp.c:122)
        if (sizeof(T) == 8) szhost_put_f64be(q, (double)eb); else szhost_put_f32be(q, (float)eb);
        q += sizeof(T);
        szhost_put_u32be(q, intervals); q += 4;
        szhost_put_u32be(q, (uint32_t)tree_bytes); q += 4;
        szhost_put_u32be(q, (uint32_t)hf->n_nodes); q += 4;
        szhost_huff_tree_write(hf, q);
    }
    szhost_huff_free(hf);
    host_ms += now_ms() - h0;
    unsigned maxlen = 0;
    for (unsigned s2 = 0; s2 < intervals; ++s2) maxlen = std::max<unsigned>(maxlen, tab_len[s2]);
    TRY(ensure(ctx, ctx->stream_buf, cap_len + 64));
    unsigned char *d_stream = (unsigned char *)ctx->stream_buf.p;
    HIPCHK(hipMemsetAsync(d_stream, 0, cap_len + 64, st));
    TRY(ensure(ctx, ctx->reg_flags, (size_t)g.nb * 8)); TRY(ensure(ctx, ctx->reg_rank, (size_t)g.nb * 8));
    u64 *d_box_bytes = (u64 *)ctx->reg_flags.p, *d_box_off = (u64 *)ctx->reg_rank.p;
    const size_t lds3 = (size_t)intervals * 8 + ((size_t)SZH_OMP_R3 * maxlen / 32 + 4) * 4 + 16;
    const bool fast = box_hist && maxlen <= 32 && intervals <= 2048 && lds3 <= 60 * 1024 && tune_int("SZ_HIP_OMP_ENC", 3) == 3;
    if (fast) {
        // ---- the usual case (code words of at most 32 bits, a histogram per box): ONE upload -- the header and the packed code table
        // `code << 8 | len` --, one launch for the boxes' sizes and places (k_omp_layout), one that packs the codes and writes every table
        // of the stream itself (k_omp_encode_box3).  (Round 4, first form: 8 copies / fills and 8 small launches here, ~0.1 ms of gaps.)
        const size_t hdr_pad = (hdr_len + 7) / 8 * 8, blob = hdr_pad + (size_t)intervals * 8;
        TRY(ensure_pinned3(ctx, blob));
        unsigned char *hb = (unsigned char *)ctx->pinned3;
        memcpy(hb, hdr.data(), hdr_len);
        u64 *hp = (u64 *)(hb + hdr_pad);
        for (unsigned s2 = 0; s2 < intervals; ++s2) hp[s2] = ((tab_code[s2] & (tab_len[s2] >= 64 ? ~0ull : (1ull << tab_len[s2]) - 1)) << 8) | tab_len[s2];
        TRY(ensure(ctx, ctx->code_tab, blob));
        TRY(ensure(ctx, ctx->unpred, (size_t)E * sizeof(T) + 16));
        HIPCHK(hipMemcpyAsync(ctx->code_tab.p, hb, blob, hipMemcpyHostToDevice, st));
        const u64 *d_packed = (const u64 *)((const unsigned char *)ctx->code_tab.p + hdr_pad);
        const int many = g.nb > tune_int("SZ_HIP_OMP_MANY", 8192);     // (one workgroup reading every box's histogram: 0.22 ms for 32 768 boxes)
        if (many) hipLaunchKernelGGL(k_omp_box_bits_p, dim3((unsigned)((g.nb + 3) / 4)), dim3(256), 0, st, g.nb, intervals, (const unsigned *)d_hist_box, d_packed, d_box_bytes);
        hipLaunchKernelGGL(k_omp_layout, dim3(1), dim3(1024), 0, st, g.nb, intervals, (const unsigned *)d_hist_box, d_packed, (const u64 *)d_ucount64, d_box_bytes, d_box_off, d_uoff,
                           sm + SM_SCRATCH, sm + SM_TOTAL_UNPRED, many);
        HIPCHK(hipGetLastError());
        szh_omp_tables tb;
        tb.stream = d_stream; tb.hdr = (const unsigned char *)ctx->code_tab.p; tb.hdr_len = (unsigned)hdr_len;
        tb.off_ucount = off_ucount; tb.off_first = off_first; tb.off_unpred = off_unpred; tb.off_sizes = off_sizes; tb.first = d_first; tb.dbg = tune_int("SZ_HIP_OMP_DBG", 0);
        hipLaunchKernelGGL((k_omp_encode_box3<T>), dim3((unsigned)g.nb), dim3(256), lds3, st, g, d_in, (const uint16_t *)d_codes, d_packed, intervals, maxlen, (const u64 *)d_box_off,
                           (const u64 *)d_box_bytes, (const u64 *)d_uoff, (const unsigned *)d_ucount, (u64)off_pay * 8, (unsigned *)d_stream,
                           (T *)ctx->unpred.p, (unsigned *)(sm + SM_ERR), tb);
        HIPCHK(hipGetLastError());
        // (the verbatim values go through an aligned buffer: their table lies at whatever byte offset the tree's size gives it, and byte
        //  stores from the kernel were half of its 0.1 ms for them)
        if (E > 0) HIPCHK(hipMemcpyAsync(d_stream + off_unpred, ctx->unpred.p, (size_t)E * sizeof(T), hipMemcpyDeviceToDevice, st));
    } else if (lean) {
        TRY(ensure(ctx, ctx->code_tab, (size_t)intervals * 8));
        TRY(ensure(ctx, ctx->len_tab, (size_t)intervals));
        HIPCHK(hipMemcpyAsync(ctx->code_tab.p, tab_code.data(), (size_t)intervals * 8, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(ctx->len_tab.p, tab_len.data(), (size_t)intervals, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(d_stream, hdr.data(), hdr_len, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(d_stream + off_ucount, d_ucount, (size_t)g.nb * 4, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpyAsync(d_stream + off_first, d_first, (size_t)g.nb * sizeof(T), hipMemcpyDeviceToDevice, st));
        if (box_hist) TRY(scan_u64(ctx, (const u64 *)d_ucount64, g.nb, d_uoff, sm + SM_TOTAL_UNPRED));
        // the boxes' payload sizes (from their histograms, or one more pass over the codes), their places, then ONE pass that packs every
        // box's codes behind a running bit position and drops its verbatim values into the table on the way
        TRY(ensure(ctx, ctx->unpred, (size_t)E * sizeof(T) + 16));
        if (box_hist) hipLaunchKernelGGL(k_omp_box_bits_h, dim3((unsigned)((g.nb + 3) / 4)), dim3(256), 0, st, g.nb, intervals, (const unsigned *)d_hist_box, (const uint8_t *)ctx->len_tab.p, d_box_bytes);
        else hipLaunchKernelGGL(k_omp_box_bits_c, dim3((unsigned)g.nb), dim3(256), 0, st, g.bel, (const uint16_t *)d_codes, (const uint8_t *)ctx->len_tab.p, d_box_bytes);
        HIPCHK(hipGetLastError());
        TRY(scan_u64(ctx, (const u64 *)d_box_bytes, g.nb, d_box_off, sm + SM_SCRATCH));
        HIPCHK(hipMemcpyAsync(d_stream + off_sizes, d_box_bytes, (size_t)g.nb * 8, hipMemcpyDeviceToDevice, st));
        if (intervals <= 2048)
            hipLaunchKernelGGL((k_omp_encode_box<T, true>), dim3((unsigned)g.nb), dim3(256), (size_t)intervals * 9 + 16, st, g, d_in, (const uint16_t *)d_codes, (const u64 *)ctx->code_tab.p,
                               (const uint8_t *)ctx->len_tab.p, intervals, (const u64 *)d_box_off, (const u64 *)d_box_bytes, (const u64 *)d_uoff, (const unsigned *)d_ucount, (u64)off_pay * 8,
                               (unsigned *)d_stream, (T *)ctx->unpred.p, (unsigned *)(sm + SM_ERR));
        else
            hipLaunchKernelGGL((k_omp_encode_box<T, false>), dim3((unsigned)g.nb), dim3(256), 16, st, g, d_in, (const uint16_t *)d_codes, (const u64 *)ctx->code_tab.p,
                               (const uint8_t *)ctx->len_tab.p, intervals, (const u64 *)d_box_off, (const u64 *)d_box_bytes, (const u64 *)d_uoff, (const unsigned *)d_ucount, (u64)off_pay * 8,
                               (unsigned *)d_stream, (T *)ctx->unpred.p, (unsigned *)(sm + SM_ERR));
        HIPCHK(hipGetLastError());
        if (E > 0) HIPCHK(hipMemcpyAsync(d_stream + off_unpred, ctx->unpred.p, (size_t)E * sizeof(T), hipMemcpyDeviceToDevice, st));
    } else {
    TRY(ensure(ctx, ctx->code_tab, (size_t)intervals * 8));
    TRY(ensure(ctx, ctx->len_tab, (size_t)intervals));
    HIPCHK(hipMemcpyAsync(ctx->code_tab.p, tab_code.data(), (size_t)intervals * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ctx->len_tab.p, tab_len.data(), (size_t)intervals, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_stream, hdr.data(), hdr_len, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_stream + off_ucount, d_ucount, (size_t)g.nb * 4, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(d_stream + off_first, d_first, (size_t)g.nb * sizeof(T), hipMemcpyDeviceToDevice, st));
    if (E > 0) {
        TRY(ensure(ctx, ctx->unpred, (size_t)E * sizeof(T)));
        hipLaunchKernelGGL((k_omp_gather<T>), dim3((unsigned)g.nb), dim3(256), 0, st, g, d_in, (const uint16_t *)d_codes, (const unsigned *)d_ucount, (const u64 *)d_uoff, (T *)ctx->unpred.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(d_stream + off_unpred, ctx->unpred.p, (size_t)E * sizeof(T), hipMemcpyDeviceToDevice, st));
    }
    const int64_t nchunks = (int64_t)g.nb * g.cpb;
    TRY(ensure(ctx, ctx->chunk_bits, (size_t)nchunks * 8)); TRY(ensure(ctx, ctx->chunk_off, (size_t)nchunks * 8));
    hipLaunchKernelGGL(k_omp_chunk_bits, dim3((unsigned)nchunks), dim3(256), 0, st, g, (const uint16_t *)d_codes, (const uint8_t *)ctx->len_tab.p, (u64 *)ctx->chunk_bits.p);
    HIPCHK(hipGetLastError());
    TRY(scan_u64(ctx, (const u64 *)ctx->chunk_bits.p, nchunks, (u64 *)ctx->chunk_off.p, sm + SM_TOTAL_BITS));
    hipLaunchKernelGGL(k_omp_box_bytes, dim3((unsigned)((g.nb + 255) / 256)), dim3(256), 0, st, g.nb, g.cpb, (const u64 *)ctx->chunk_off.p, (const u64 *)(sm + SM_TOTAL_BITS), d_box_bytes);
    HIPCHK(hipGetLastError());
    TRY(scan_u64(ctx, (const u64 *)d_box_bytes, g.nb, d_box_off, sm + SM_SCRATCH));
    HIPCHK(hipMemcpyAsync(d_stream + off_sizes, d_box_bytes, (size_t)g.nb * 8, hipMemcpyDeviceToDevice, st));
    if (total_bits > 0) {
        hipLaunchKernelGGL(k_omp_encode, dim3((unsigned)nchunks), dim3(256), 0, st, g, (const uint16_t *)d_codes, (const u64 *)ctx->code_tab.p, (const uint8_t *)ctx->len_tab.p,
                           (const u64 *)ctx->chunk_off.p, (const u64 *)d_box_off, (u64)off_pay * 8, (unsigned *)d_stream);
        HIPCHK(hipGetLastError());
    }
    }
    HIPCHK(hipEventRecord(ctx->ev[4], st));
    u64 h_small[SM_COUNT];
    HIPCHK(hipMemcpyAsync(h_small, sm, SM_COUNT * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if ((!lean && h_small[SM_TOTAL_BITS] != total_bits) || h_small[SM_TOTAL_UNPRED] != E || (lean && (unsigned)h_small[SM_ERR] != 0) ||
        h_small[SM_SCRATCH] < (total_bits + 7) / 8 || h_small[SM_SCRATCH] > (total_bits + 7) / 8 + (u64)g.nb)
        FAIL(SZHIP_ERR_INTERNAL, "OpenMP container: entropy stage mismatch");
    const size_t total_len = off_pay + (size_t)h_small[SM_SCRATCH];
    if (total_len > cap_len) FAIL(SZHIP_ERR_INTERNAL, "OpenMP container: payloads larger than their bound");
    if (out_on_device == 2) {
        if (!*out || *out_size < total_len) FAIL(SZHIP_ERR_ARG, "caller's device buffer too small (%zu < %zu)", *out_size, total_len);
        HIPCHK(hipMemcpyAsync(*out, d_stream, total_len, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
    } else if (out_on_device) {
        *out = d_stream;
    } else {
        unsigned char *h = (unsigned char *)malloc(total_len ? total_len : 1);
        if (!h) FAIL(SZHIP_ERR_INTERNAL, "out of host memory");
        const int rc_copy = staged_copy(ctx, h, d_stream, total_len, false);
        if (rc_copy != SZHIP_OK || hipStreamSynchronize(st) != hipSuccess) { free(h); FAIL(rc_copy != SZHIP_OK ? rc_copy : SZHIP_ERR_NODEVICE, "copying the stream to the host failed"); }
        *out = h;
    }
    *out_size = total_len;
    float ms = 0;
    hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]); S.ms_prequant = ms;
    hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3]); S.ms_quant = ms;
    hipEventElapsedTime(&ms, ctx->ev[3], ctx->ev[4]); S.ms_entropy = ms;
    S.ms_host = host_ms; S.ms_total = now_ms() - t_begin; S.out_bytes = total_len;
    if (stats) *stats = S;
    return SZHIP_OK;
}

// `body_off`: offset of the thread_num field (4 + MetaDataByteLength: what decompressDataSeries_*_3D_openmp is handed)
template <class T>
int decompress_omp_impl(szhip_ctx *ctx, const unsigned char *stream_in, int stream_on_device, size_t stream_len, size_t body_off, size_t r0, size_t r1, size_t r2,
                        void *out, int out_on_device, szhip_stats *stats)
{
    const double t_begin = now_ms();
    hipStream_t st = ctx->stream;
    szhip_stats S; memset(&S, 0, sizeof(S));
    TRY(ensure(ctx, ctx->stream_buf, stream_len + 64));
    unsigned char *d_stream = (unsigned char *)ctx->stream_buf.p;
    if (stream_on_device) { if (stream_in != d_stream) HIPCHK(hipMemcpyAsync(d_stream, stream_in, stream_len, hipMemcpyDeviceToDevice, st)); }
    else TRY(staged_copy(ctx, d_stream, stream_in, stream_len, true));
    HIPCHK(hipMemsetAsync(d_stream + stream_len, 0, 64, st));
    HIPCHK(hipEventRecord(ctx->ev[0], st));
    std::vector<unsigned char> hbuf;
    const unsigned char *hs = stream_in;
    auto fetch = [&](size_t want) -> int {                    // the first `want` bytes of the stream on the host
        if (want > stream_len) FAIL(SZHIP_ERR_STREAM, "truncated stream");
        if (!stream_on_device) return SZHIP_OK;
        hbuf.resize(want);
        HIPCHK(hipMemcpyAsync(hbuf.data(), d_stream, want, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        hs = hbuf.data();
        return SZHIP_OK;
    };
    const size_t fixed = body_off + 4 + sizeof(T) + 12;
    TRY(fetch(fixed));
    const unsigned char *q = hs + body_off;
    const int thread_num = (int)szhost_get_u32be(q); q += 4;
    const T eb = sizeof(T) == 8 ? (T)szhost_get_f64be(q) : (T)szhost_get_f32be(q); q += sizeof(T);
    const unsigned intervals = szhost_get_u32be(q); q += 4;
    const size_t tree_bytes = szhost_get_u32be(q); q += 4;
    const int node_count = (int)szhost_get_u32be(q); q += 4;
    if (intervals < 4 || intervals > 65536 || !(eb > 0)) FAIL(SZHIP_ERR_STREAM, "bad OpenMP-container header");
    if (node_count <= 0 || tree_bytes > stream_len || szhost_huff_serial_size(node_count) > tree_bytes) FAIL(SZHIP_ERR_STREAM, "truncated stream");
    szh_omp_geom g;
    TRY(omp_box_grid(ctx, thread_num, r0, r1, r2, sizeof(T), &g));
    if (g.nb != thread_num) FAIL(SZHIP_ERR_STREAM, "thread_num %d is not a box grid", thread_num);
    const int64_t n = (int64_t)r0 * r1 * r2;
    S.n_elements = (uint64_t)n; S.n_blocks = (uint64_t)g.nb; S.intervals = intervals;
    const size_t off_ucount = fixed + tree_bytes, off_first = off_ucount + (size_t)g.nb * 4, off_unpred = off_first + (size_t)g.nb * sizeof(T);
    TRY(fetch(off_unpred));
    szhost_huff *hf = szhost_huff_from_bytes(2 * (int)intervals, hs + fixed, node_count);
    if (!hf) FAIL(SZHIP_ERR_STREAM, "bad Huffman tree");
    std::vector<uint32_t> dtab((size_t)hf->n_nodes * 2);
    szhost_huff_decode_table(hf, dtab.data());
    const int single_symbol = hf->t[0] ? (int)hf->C[0] : -1;
    const int n_nodes_dec = hf->n_nodes;
    szhost_huff_free(hf);
    std::vector<u64> uoff((size_t)g.nb + 1, 0);
    for (int b = 0; b < g.nb; ++b) { uint32_t c; memcpy(&c, hs + off_ucount + (size_t)b * 4, 4); if (c > (uint32_t)g.bel) FAIL(SZHIP_ERR_STREAM, "bad verbatim-value count"); uoff[b + 1] = uoff[b] + c; }
    const u64 E = uoff[g.nb];
    S.n_unpred = E;
    const size_t off_sizes = off_unpred + (size_t)E * sizeof(T), off_pay = off_sizes + (size_t)g.nb * 8;
    if (off_pay > stream_len) FAIL(SZHIP_ERR_STREAM, "truncated stream");
    std::vector<u64> bbytes((size_t)g.nb), boff((size_t)g.nb);
    if (stream_on_device) { HIPCHK(hipMemcpyAsync(bbytes.data(), d_stream + off_sizes, (size_t)g.nb * 8, hipMemcpyDeviceToHost, st)); HIPCHK(hipStreamSynchronize(st)); }
    else memcpy(bbytes.data(), stream_in + off_sizes, (size_t)g.nb * 8);
    u64 acc = 0;
    for (int b = 0; b < g.nb; ++b) { boff[b] = acc; if (bbytes[b] > stream_len || bbytes[b] >= ((u64)1 << 28)) FAIL(SZHIP_ERR_STREAM, "bad payload size"); acc += bbytes[b]; }
    if (off_pay + acc > stream_len) FAIL(SZHIP_ERR_STREAM, "truncated stream");
    // ---- device tables: payload offsets / sizes, ranks of the verbatim values, first values and verbatim values at aligned addresses
    TRY(ensure(ctx, ctx->small, SM_COUNT * 8));
    u64 *sm = (u64 *)ctx->small.p;
    HIPCHK(hipMemsetAsync(sm, 0, SM_COUNT * 8, st));
    TRY(ensure(ctx, ctx->reg_flags, (size_t)g.nb * 8)); TRY(ensure(ctx, ctx->reg_rank, (size_t)g.nb * 8)); TRY(ensure(ctx, ctx->col_off, (size_t)g.nb * 8 + 8));
    TRY(ensure(ctx, ctx->samples, (size_t)g.nb * sizeof(T))); TRY(ensure(ctx, ctx->unpred, (size_t)E * sizeof(T) + 16));
    TRY(ensure(ctx, ctx->dec_tab, (dtab.size() * 4 + 63) / 64 * 64 + SZH_LUT_BYTES + 16));
    HIPCHK(hipMemcpyAsync(ctx->reg_flags.p, bbytes.data(), (size_t)g.nb * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ctx->reg_rank.p, boff.data(), (size_t)g.nb * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ctx->col_off.p, uoff.data(), ((size_t)g.nb + 1) * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ctx->dec_tab.p, dtab.data(), dtab.size() * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ctx->samples.p, d_stream + off_first, (size_t)g.nb * sizeof(T), hipMemcpyDeviceToDevice, st));
    if (E > 0) HIPCHK(hipMemcpyAsync(ctx->unpred.p, d_stream + off_unpred, (size_t)E * sizeof(T), hipMemcpyDeviceToDevice, st));
    TRY(ensure(ctx, ctx->codes_nat, (size_t)n * 2 + 64));
    uint16_t *d_codes = (uint16_t *)ctx->codes_nat.p;
    {
        u64 max_box = 0;
        for (int b = 0; b < g.nb; ++b) max_box = std::max(max_box, bbytes[b]);
        // the look-up-table decoder (hdec_run_lut) when a box's payload, the table and the node table fit a workgroup's LDS
        const unsigned stage_bytes = (unsigned)((max_box + 30 + 15) / 16 * 16 + 16);
        const size_t stage_lds = ((size_t)SZH_HDEC_SWZ((stage_bytes + 16) / 4) * 4 + 15) / 16 * 16;
        const int tab_lds_lut = (size_t)n_nodes_dec * 8 <= 13 * 1024;
        const size_t lds_lut = stage_lds + SZH_LUT_BYTES + (tab_lds_lut ? ((size_t)n_nodes_dec * 8 + 15) / 16 * 16 : 0);
        if (single_symbol < 0 && g.bel % 8 == 0 && lds_lut <= 64 * 1024 && tune_int("SZ_HIP_OMP_LEAN", 1) != 0) {
            const size_t lut_off = (dtab.size() * 4 + 63) / 64 * 64;
            TRY(ensure(ctx, ctx->dec_tab, lut_off + SZH_LUT_BYTES));          // (grown before the table went up: see the copy above)
            hipLaunchKernelGGL(k_hdec_build_lut, dim3(SZH_LUT_SIZE / 256), dim3(256), 0, st, (const unsigned *)ctx->dec_tab.p, (uint4 *)((char *)ctx->dec_tab.p + lut_off));
            HIPCHK(hipGetLastError());
            const int per_wg = std::max(1, tune_int("SZ_HIP_OMP_HDEC_PER_WG", 1));         // (several small boxes per workgroup, sharing its copy of the tables: measured slower, 1.59 against 1.42 ms for 32 768 boxes)
            hipLaunchKernelGGL(k_omp_hdec_lut, dim3((unsigned)((g.nb + per_wg - 1) / per_wg)), dim3(256), lds_lut, st, g.bel, (const unsigned char *)(d_stream + off_pay), (unsigned)off_pay,
                               (const u64 *)ctx->reg_rank.p, (const u64 *)ctx->reg_flags.p, (const unsigned *)ctx->dec_tab.p, n_nodes_dec, tab_lds_lut,
                               (const uint4 *)((char *)ctx->dec_tab.p + lut_off), stage_bytes, d_codes, (unsigned *)(sm + SM_ERR), g.nb, per_wg);
        } else {
        const int tab_lds = dtab.size() * 4 <= 16384;            // node table and payload in LDS when they are small (the usual case: 2 - 3 bits per code)
        const unsigned pay_cap = (unsigned)std::min<u64>(max_box, 24576);
        const size_t lds = (tab_lds ? dtab.size() * 4 : 0) + (size_t)pay_cap + 16;
        hipLaunchKernelGGL(k_omp_hdec, dim3((unsigned)g.nb), dim3(256), lds, st, g.bel, (const unsigned char *)(d_stream + off_pay), (const u64 *)ctx->reg_rank.p,
                           (const u64 *)ctx->reg_flags.p, (const unsigned *)ctx->dec_tab.p, n_nodes_dec, tab_lds, pay_cap, single_symbol, d_codes, (unsigned *)(sm + SM_ERR));
        }
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ctx->ev[1], st));
    T *d_out = (T *)out;
    if (!out_on_device) { TRY(ensure(ctx, ctx->out, (size_t)n * sizeof(T))); d_out = (T *)ctx->out.p; }
    if ((uintptr_t)d_out & 15u) g.vec = 0;
    const int rows = g.c0 * g.c1, box_threads = rows;      // one lane per row
    HIPCHK(hipEventRecord(ctx->ev[2], st));
    if (omp_col_applies(g, d_out, r2 * sizeof(T))) {
        // the verbatim values go to their places in the output first (boxes that have any); the sweep picks them up where a code is zero
        int fw = (g.c0 * g.c1 / (16 / (int)sizeof(T)) + 31) / 32;        // flag words per box: a bit per group of rows one load of the sweep covers
        unsigned *d_vflags = nullptr;
        if (fw > OC_FLAG_WORDS) fw = 0;
        if (E > 0) {
            if (fw > 0) {
                TRY(ensure(ctx, ctx->chunk_bits, (size_t)g.nb * fw * 4));
                d_vflags = (unsigned *)ctx->chunk_bits.p;
                HIPCHK(hipMemsetAsync(d_vflags, 0, (size_t)g.nb * fw * 4, st));
            }
            hipLaunchKernelGGL((k_omp_scatter<T>), dim3((unsigned)g.nb), dim3(256), 0, st, g, (const uint16_t *)d_codes, (const u64 *)ctx->col_off.p, (const T *)ctx->unpred.p, d_out,
                               (unsigned *)(sm + SM_ERR), d_vflags, fw);
            HIPCHK(hipGetLastError());
        }
        szh_oc::sweep_args<T> oa;
        oa.vflags = d_vflags; oa.fw = fw; oa.dbg_no_code_stores = 0;
        oa.g = g; oa.data = nullptr; oa.out = d_out; oa.eb = eb; oa.recip = (T)(1 / eb); oa.intervals = (int)intervals; oa.codes = d_codes;
        oa.ucount = (unsigned *)(sm + SM_ERR); oa.ucount64 = nullptr; oa.first = (T *)ctx->samples.p; oa.uoff = (const u64 *)ctx->col_off.p;
        hipLaunchKernelGGL((k_omp_col<T, 32, 32, true>), dim3((unsigned)(g.nb / 2)), dim3(64), 0, st, oa);
    } else if (g.vec) hipLaunchKernelGGL((k_omp_box<T, true, true>), dim3((unsigned)g.nb), dim3((unsigned)box_threads), (size_t)4 * g.c0 * g.pitch * sizeof(T), st, g, (const T *)nullptr, d_out, eb, (T)(1 / eb),
                                  (int)intervals, d_codes, (unsigned *)(sm + SM_ERR), (u64 *)nullptr, (T *)ctx->samples.p, (const T *)ctx->unpred.p, (const u64 *)ctx->col_off.p);
    else hipLaunchKernelGGL((k_omp_box<T, true, false>), dim3((unsigned)g.nb), dim3((unsigned)box_threads), (size_t)4 * g.c0 * g.pitch * sizeof(T), st, g, (const T *)nullptr, d_out, eb, (T)(1 / eb),
                            (int)intervals, d_codes, (unsigned *)(sm + SM_ERR), (u64 *)nullptr, (T *)ctx->samples.p, (const T *)ctx->unpred.p, (const u64 *)ctx->col_off.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ctx->ev[3], st));
    unsigned bad = 0;
    HIPCHK(hipMemcpyAsync(&bad, sm + SM_ERR, 4, hipMemcpyDeviceToHost, st));
    if (!out_on_device) TRY(staged_copy(ctx, out, d_out, (size_t)n * sizeof(T), false));
    HIPCHK(hipStreamSynchronize(st));
    if (bad) FAIL(SZHIP_ERR_STREAM, "%u boxes whose payload or verbatim-value count does not fit their codes", bad);
    float ms = 0;
    hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]); S.ms_entropy = ms;
    hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3]); S.ms_quant = ms;
    S.ms_total = now_ms() - t_begin; S.out_bytes = (uint64_t)n * sizeof(T);
    if (stats) *stats = S;
    return SZHIP_OK;
}

} // namespace

// runs one call; a wavefront-kernel wait that timed out under the index ticket is answered by ONE repetition with the atomic ticket (see szhip_ctx)
template <class F>
static int with_ticket_fallback(szhip_ctx *ctx, F &&run)
{
    ctx->wave_timeout = false; ctx->hdec_unconverged = false; ctx->coef_late = false;
    int rc = run();
    if (rc == SZHIP_OK && !ctx->no_chain_overlap && tune_int("SZ_HIP_TEST_CHAIN_FALLBACK", 0)) { ctx->coef_late = true; rc = SZHIP_ERR_INTERNAL; }   // tests: exercise the repetition
    if (rc == SZHIP_ERR_INTERNAL && ctx->coef_late && !ctx->no_chain_overlap) {           // (compression of arrays with regression blocks only)
        if (ctx->stream3) hipStreamSynchronize(ctx->stream3); hipStreamSynchronize(ctx->stream2); hipStreamSynchronize(ctx->stream);
        ctx->no_chain_overlap = true; ctx->wave_timeout = false;
        rc = run();
    }
    if (rc == SZHIP_ERR_INTERNAL && ctx->hdec_unconverged && !ctx->hdec_sync_rounds) {      // (decompression only)
        if (ctx->stream3) hipStreamSynchronize(ctx->stream3); hipStreamSynchronize(ctx->stream2); hipStreamSynchronize(ctx->stream);
        ctx->hdec_sync_rounds = true;
        ctx->wave_timeout = false;
        rc = run();
        ctx->hdec_sync_rounds = false;
    }
    if (rc == SZHIP_OK && !ctx->ticket_atomic && tune_int("SZ_HIP_TEST_TICKET_FALLBACK", 0)) { ctx->wave_timeout = true; rc = SZHIP_ERR_INTERNAL; }   // tests: exercise the repetition
    if (rc == SZHIP_ERR_INTERNAL && ctx->wave_timeout && !ctx->ticket_atomic) {
        if (ctx->stream3) hipStreamSynchronize(ctx->stream3); hipStreamSynchronize(ctx->stream2); hipStreamSynchronize(ctx->stream);
        ctx->ticket_atomic = true;
        rc = run();
    }
    return rc;
}


extern "C" {

int szhip_is_fast_stream(const unsigned char *stream, size_t stream_len) { return stream && stream_len >= SZF_HDR_FIXED && memcmp(stream, "SZHF", 4) == 0; }

int szhip_compress_fast(szhip_ctx *ctx, int dtype, const void *data, int data_on_device, size_t r0, size_t r1, size_t r2, double eb,
                        unsigned intervals, int out_on_device, unsigned char **out, size_t *out_size, szhip_stats *stats)
{
    if (!ctx || !data || !out || !out_size || !(eb > 0)) return SZHIP_ERR_ARG;
    if (intervals == 0) intervals = 1024;
    if (r0 < 1 || r1 < 1 || r2 < 1 || r0 > 0x7fffffff || r1 > 0x7fffffff || r2 > 0x7fffffff || intervals < 4 || intervals > 65536 || (intervals & 1)) return SZHIP_ERR_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return SZHIP_ERR_NODEVICE;
    const int rc = dtype == SZHIP_F32 ? compress_fast_impl<float>(ctx, data, data_on_device, r0, r1, r2, eb, intervals, out_on_device, out, out_size, stats)
                                      : compress_fast_impl<double>(ctx, data, data_on_device, r0, r1, r2, eb, intervals, out_on_device, out, out_size, stats);
    if (rc != SZHIP_OK) hipStreamSynchronize(ctx->stream);
    return rc;
}

int szhip_decompress_fast(szhip_ctx *ctx, int dtype, const unsigned char *stream, int stream_on_device, size_t stream_len,
                          size_t r0, size_t r1, size_t r2, void *out, int out_on_device, szhip_stats *stats)
{
    if (!ctx || !stream || !out || r0 < 1 || r1 < 1 || r2 < 1) return SZHIP_ERR_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return SZHIP_ERR_NODEVICE;
    const int rc = with_ticket_fallback(ctx, [&]() { return dtype == SZHIP_F32
                       ? decompress_fast_impl<float>(ctx, stream, stream_on_device, stream_len, r0, r1, r2, out, out_on_device, stats)
                       : decompress_fast_impl<double>(ctx, stream, stream_on_device, stream_len, r0, r1, r2, out, out_on_device, stats); });    // (for the Huffman decoder's repetition)
    if (rc != SZHIP_OK) hipStreamSynchronize(ctx->stream);
    return rc;
}

// side_prio: the second / third stream at the lowest / highest priority the device offers.  HIP maps the streams of one priority onto a few
// hardware queues, and whether two streams of a context share one is the luck of what else the process has created (measured, round 4, one call
// after the other at 512^3: 236 - 241 GB/s when the fit pass and the sampling chain, or the sweep and the slices' passes, sat on one queue,
// 274 - 279 otherwise); streams of different priorities never share one.  Lanes of a pool keep equal priorities (a lane whose main stream
// outranks the other's would always be dispatched first).
static int create_ctx(szhip_ctx **out, int device, int side_prio, int main_high = 0)
{
    if (!out) return SZHIP_ERR_ARG;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        fprintf(stderr, "szhip: no HIP device available (%s); this library has no CPU fallback\n", hipGetErrorString(e));
        return SZHIP_ERR_NODEVICE;
    }
    if (device < 0 || device >= count) return SZHIP_ERR_ARG;
    szhip_ctx *ctx = new szhip_ctx();
    ctx->device = device;
    { int c = 0; if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && c > 0) ctx->cus = c; }
    int mlo = 0, mhi = 0;
    if (hipSetDevice(device) != hipSuccess || (main_high && hipDeviceGetStreamPriorityRange(&mlo, &mhi) != hipSuccess) ||
        (main_high ? hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, mhi) : hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) {
        delete ctx; return SZHIP_ERR_NODEVICE;
    }
    for (int i = 0; i < 6; ++i) if (hipEventCreate(&ctx->ev[i]) != hipSuccess) { delete ctx; return SZHIP_ERR_NODEVICE; }
    // (SZ_HIP_STREAM_PRIO: 0 = equal priorities, 1 = second stream low / third high, 2 = the other way round)
    const int sprio = tune_int("SZ_HIP_STREAM_PRIO", side_prio);
    ctx->side_prio = sprio;
    int plo = 0, phi = 0;
    if (sprio && hipDeviceGetStreamPriorityRange(&plo, &phi) != hipSuccess) { plo = 0; phi = 0; }
    if ((sprio ? hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, sprio == 2 ? phi : plo) : hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking)) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_fit, hipEventDisableTiming) != hipSuccess) { delete ctx; return SZHIP_ERR_NODEVICE; }
    *out = ctx;
    return SZHIP_OK;
}

int szhip_create(szhip_ctx **out, int device)
{
    const int rc = create_ctx(out, device, 0);
    if (rc == SZHIP_OK) settle_streams(*out, nullptr, 0);
    return rc;
}

void szhip_destroy(szhip_ctx *ctx)
{
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    delete ctx->chain_pool; ctx->chain_pool = nullptr;
    DevBuf *bufs[] = {&ctx->lor_bits, &ctx->reg_flags, &ctx->reg_rank, &ctx->coef_compact, &ctx->in, &ctx->out, &ctx->codes_nat, &ctx->codes_blk, &ctx->coef, &ctx->blk_lor, &ctx->faceI, &ctx->faceJ, &ctx->rb_down, &ctx->rb_right, &ctx->rb_vals, &ctx->pt_flags, &ctx->fast_slots, &ctx->fast_units, &ctx->progress, &ctx->trace,
                      &ctx->order, &ctx->small, &ctx->hist, &ctx->col_zeros, &ctx->col_zeros64, &ctx->col_off, &ctx->partial,
                      &ctx->samples, &ctx->unpred, &ctx->stream_buf, &ctx->chunk_bits, &ctx->chunk_off, &ctx->code_tab,
                      &ctx->len_tab, &ctx->dec_tab, &ctx->starts, &ctx->ends, &ctx->counts, &ctx->offs, &ctx->dirty, &ctx->zcnt, &ctx->zpos,
                      &ctx->pwr_log, &ctx->pwr_signs, &ctx->pwr_small, &ctx->coef_dec, &ctx->msst_ptab, &ctx->msst_cells, &ctx->msst_rec, &ctx->msst_pe};
    for (DevBuf *b : bufs) if (b->p) hipFree(b->p);
    if (ctx->pinned) hipHostFree(ctx->pinned);
    if (ctx->pinned2) hipHostFree(ctx->pinned2);
    if (ctx->pinned3) hipHostFree(ctx->pinned3);
    if (ctx->coh) hipHostFree(ctx->coh);
    if (ctx->hdec_res) hipHostFree(ctx->hdec_res);
    if (ctx->ev_gate) hipEventDestroy(ctx->ev_gate);
    for (int w = 0; w < SZH_STAGE_TMAX; ++w) for (int k = 0; k < 2; ++k) { if (ctx->stage_buf[w][k]) hipHostFree(ctx->stage_buf[w][k]); if (ctx->stage_ev[w][k]) hipEventDestroy(ctx->stage_ev[w][k]); }
    for (int i = 0; i < 6; ++i) if (ctx->ev[i]) hipEventDestroy(ctx->ev[i]);
    if (ctx->ev_in) hipEventDestroy(ctx->ev_in);
    if (ctx->ev_fit) hipEventDestroy(ctx->ev_fit);
    if (ctx->stream2) hipStreamDestroy(ctx->stream2);
    if (ctx->stream3) hipStreamDestroy(ctx->stream3);
    if (ctx->ev_perm) hipEventDestroy(ctx->ev_perm);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *szhip_last_error(szhip_ctx *ctx) { return ctx ? ctx->err : "no context"; }

// ---- several arrays in flight (szhip.h): `lanes` contexts, one host thread each, one queue
struct szhip_pool_job {
    int dtype; const void *data; int on_dev; size_t r0, r1, r2; double eb; szhip_params prm; const unsigned char *meta; size_t meta_len;
    int out_on_device; unsigned char *out; size_t out_size; szhip_stats stats; int rc; bool done;
};
struct szhip_pool {
    szhip_sweep_gate gate;
    std::vector<szhip_ctx *> ctx;
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::deque<int> queue;                      // tickets waiting for a lane
    std::vector<szhip_pool_job> jobs;           // ring of tickets
    std::vector<bool> used;
    bool stop = false;
};
static void szhip_pool_worker(szhip_pool *p, int lane)
{
    for (;;) {
        int t;
        {
            std::unique_lock<std::mutex> lk(p->mu);
            p->cv_work.wait(lk, [&] { return p->stop || !p->queue.empty(); });
            if (p->queue.empty()) return;
            t = p->queue.front(); p->queue.pop_front();
        }
        szhip_pool_job &j = p->jobs[t];
        j.rc = szhip_compress(p->ctx[lane], j.dtype, j.data, j.on_dev, j.r0, j.r1, j.r2, j.eb, &j.prm, j.meta, j.meta_len, j.out_on_device,
                              &j.out, &j.out_size, &j.stats);
        { std::lock_guard<std::mutex> lk(p->mu); j.done = true; }
        p->cv_done.notify_all();
    }
}
int szhip_pool_create(szhip_pool **out, int device, int lanes)
{
    if (!out || lanes < 1 || lanes > 16) return SZHIP_ERR_ARG;
    szhip_pool *p = new szhip_pool();
    for (int i = 0; i < lanes; ++i) {
        szhip_ctx *c = nullptr;
        // (SZ_HIP_LANE_PRIO=1, development: every other lane's main stream at the highest priority, so that two lanes never share a hardware queue)
        const int rc = create_ctx(&c, device, 0, lanes > 1 && tune_int("SZ_HIP_LANE_PRIO", 0) ? (i & 1) : 0);
        if (rc != SZHIP_OK) { for (szhip_ctx *x : p->ctx) szhip_destroy(x); delete p; return rc; }
        p->gate.lanes = lanes;
        if (lanes > 1) { c->gate = &p->gate; if (hipEventCreateWithFlags(&c->ev_gate, hipEventDisableTiming) != hipSuccess) c->gate = nullptr; }
        // (against the earlier lanes' main AND second streams: a lane's fit pass behind another lane's sweep is as bad as two sweeps in a row)
        { std::vector<hipStream_t> mains; for (szhip_ctx *x : p->ctx) { mains.push_back(x->stream); mains.push_back(x->stream2); } settle_streams(c, mains.data(), (int)mains.size(), lanes == 1); }
        p->ctx.push_back(c);
    }
    p->jobs.resize(64); p->used.assign(64, false);
    for (int i = 0; i < lanes; ++i) p->workers.emplace_back(szhip_pool_worker, p, i);
    *out = p;
    return SZHIP_OK;
}
void szhip_pool_destroy(szhip_pool *p)
{
    if (!p) return;
    { std::lock_guard<std::mutex> lk(p->mu); p->stop = true; }
    p->cv_work.notify_all();
    for (auto &w : p->workers) if (w.joinable()) w.join();
    for (szhip_ctx *c : p->ctx) szhip_destroy(c);
    delete p;
}
int szhip_pool_submit(szhip_pool *p, int dtype, const void *data, int data_on_device, size_t r0, size_t r1, size_t r2, double eb,
                      const szhip_params *params, const unsigned char *meta, size_t meta_len, int out_on_device,
                      unsigned char *out_buf, size_t out_cap, int *ticket)
{
    if (!p || !data || !params || !meta || !ticket) return SZHIP_ERR_ARG;
    std::lock_guard<std::mutex> lk(p->mu);
    int t = -1;
    for (size_t i = 0; i < p->used.size(); ++i) if (!p->used[i]) { t = (int)i; break; }
    if (t < 0) return SZHIP_ERR_ARG;                         // 64 calls waiting to be collected: wait for some first
    szhip_pool_job &j = p->jobs[t];
    j = szhip_pool_job();
    j.dtype = dtype; j.data = data; j.on_dev = data_on_device; j.r0 = r0; j.r1 = r1; j.r2 = r2; j.eb = eb; j.prm = *params;
    j.meta = meta; j.meta_len = meta_len; j.out_on_device = out_on_device; j.out = out_on_device == 2 ? out_buf : nullptr;
    j.out_size = out_on_device == 2 ? out_cap : 0; j.rc = SZHIP_ERR_INTERNAL; j.done = false;
    p->used[t] = true;
    p->queue.push_back(t);
    *ticket = t;
    p->cv_work.notify_one();
    return SZHIP_OK;
}
int szhip_pool_wait(szhip_pool *p, int ticket, unsigned char **out, size_t *out_size, szhip_stats *stats)
{
    if (!p || ticket < 0 || ticket >= (int)p->jobs.size()) return SZHIP_ERR_ARG;
    std::unique_lock<std::mutex> lk(p->mu);
    if (!p->used[ticket]) return SZHIP_ERR_ARG;
    p->cv_done.wait(lk, [&] { return p->jobs[ticket].done; });
    szhip_pool_job &j = p->jobs[ticket];
    if (out) *out = j.out;
    if (out_size) *out_size = j.out_size;
    if (stats) *stats = j.stats;
    p->used[ticket] = false;
    return j.rc;
}


int szhip_stage_input(szhip_ctx *ctx, const void *host_data, size_t bytes, void **device_ptr)
{
    if (!ctx || !host_data || !device_ptr) return SZHIP_ERR_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return SZHIP_ERR_NODEVICE;
    TRY(ensure(ctx, ctx->in, bytes));
    TRY(staged_copy(ctx, ctx->in.p, host_data, bytes, true));
    *device_ptr = ctx->in.p;
    return SZHIP_OK;
}

int szhip_minmax(szhip_ctx *ctx, int dtype, const void *data, int on_dev, size_t n, double *vmin, double *vmax)
{
    if (!ctx || !data || !n || !vmin || !vmax) return SZHIP_ERR_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return SZHIP_ERR_NODEVICE;
    return dtype == SZHIP_F32 ? minmax_impl<float>(ctx, data, on_dev, n, vmin, vmax)
                              : minmax_impl<double>(ctx, data, on_dev, n, vmin, vmax);
}

int szhip_compress(szhip_ctx *ctx, int dtype, const void *data, int data_on_device, size_t r0, size_t r1, size_t r2, double eb,
                   const szhip_params *params, const unsigned char *meta, size_t meta_len, int out_on_device,
                   unsigned char **out, size_t *out_size, szhip_stats *stats)
{
    if (!ctx || !data || !params || !meta || !out || !out_size) return SZHIP_ERR_ARG;
    if ((r0 != 0 && r0 < 2) || r1 < 2 || r2 < 2 || r0 > 0x7fffffff || r1 > 0x7fffffff || r2 > 0x7fffffff) return SZHIP_ERR_ARG;
    if (!(eb > 0)) return SZHIP_ERR_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return SZHIP_ERR_NODEVICE;
    const compress_call_guard in_flight;
    unsigned char *const out0 = *out; const size_t cap0 = *out_size;     // (a failed first attempt clears them: the repetition starts from the caller's values)
    const int rc = with_ticket_fallback(ctx, [&]() { *out = out0; *out_size = cap0; return dtype == SZHIP_F32
               ? compress_impl<float>(ctx, data, data_on_device, r0, r1, r2, eb, params, meta, meta_len, out_on_device, out, out_size, stats)
               : compress_impl<double>(ctx, data, data_on_device, r0, r1, r2, eb, params, meta, meta_len, out_on_device, out, out_size, stats); });
    if (rc != SZHIP_OK) { if (ctx->stream3) hipStreamSynchronize(ctx->stream3); hipStreamSynchronize(ctx->stream2); hipStreamSynchronize(ctx->stream); } // an early exit must not leave work in flight
    return rc;
}

int szhip_compress_sz14(szhip_ctx *ctx, int dtype, const void *data, int data_on_device, size_t r0, size_t r1, size_t r2, double eb,
                        double value_range, double median, const szhip_params *params, const unsigned char *meta, size_t meta_len,
                        int out_on_device, unsigned char **out, size_t *out_size, szhip_stats *stats)
{
    if (!ctx || !data || !params || !meta || !out || !out_size) return SZHIP_ERR_ARG;
    if ((r0 != 0 && r0 < 2) || (r1 < 2 && !(r0 == 0 && r1 == 0)) || r2 < 2 || r0 > 0x7fffffff || r1 > 0x7fffffff || r2 > 0x7fffffff) return SZHIP_ERR_ARG;
    if (!(eb > 0)) return SZHIP_ERR_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return SZHIP_ERR_NODEVICE;
    const compress_call_guard in_flight;
    unsigned char *const out0 = *out; const size_t cap0 = *out_size;     // (a failed first attempt clears them: the repetition starts from the caller's values)
    const int rc = with_ticket_fallback(ctx, [&]() { *out = out0; *out_size = cap0; return dtype == SZHIP_F32
               ? compress14_impl<float>(ctx, data, data_on_device, r0, r1, r2, eb, value_range, median, params, meta, meta_len, nullptr, out_on_device, out, out_size, stats)
               : compress14_impl<double>(ctx, data, data_on_device, r0, r1, r2, eb, value_range, median, params, meta, meta_len, nullptr, out_on_device, out, out_size, stats); });
    if (rc != SZHIP_OK) { if (ctx->stream3) hipStreamSynchronize(ctx->stream3); hipStreamSynchronize(ctx->stream2); hipStreamSynchronize(ctx->stream); }
    return rc;
}

int szhip_decompress_sz14(szhip_ctx *ctx, int dtype, const unsigned char *stream, int stream_on_device, size_t stream_len, size_t body_off,
                          size_t r0, size_t r1, size_t r2, void *out, int out_on_device, szhip_stats *stats)
{
    if (!ctx || !stream || !out || body_off >= stream_len) return SZHIP_ERR_ARG;
    if ((r0 != 0 && r0 < 2) || (r1 < 2 && !(r0 == 0 && r1 == 0)) || r2 < 2 || r0 > 0x7fffffff || r1 > 0x7fffffff || r2 > 0x7fffffff) return SZHIP_ERR_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return SZHIP_ERR_NODEVICE;
    const int rc = with_ticket_fallback(ctx, [&]() { return dtype == SZHIP_F32
               ? decompress14_impl<float>(ctx, stream, stream_on_device, stream_len, body_off, 0, r0, r1, r2, out, out_on_device, stats)
               : decompress14_impl<double>(ctx, stream, stream_on_device, stream_len, body_off, 0, r0, r1, r2, out, out_on_device, stats); });
    if (rc != SZHIP_OK) { if (ctx->stream3) hipStreamSynchronize(ctx->stream3); hipStreamSynchronize(ctx->stream2); hipStreamSynchronize(ctx->stream); }
    return rc;
}

int szhip_compress_omp(szhip_ctx *ctx, int dtype, const void *data, int data_on_device, size_t r0, size_t r1, size_t r2, double eb, int thread_num,
                       const szhip_params *params, const unsigned char *meta, size_t meta_len, int out_on_device, unsigned char **out, size_t *out_size,
                       szhip_stats *stats)
{
    if (!ctx || !data || !params || !meta || !out || !out_size) return SZHIP_ERR_ARG;
    if (r0 < 1 || r1 < 1 || r2 < 1 || r0 > 0x7fffffff || r1 > 0x7fffffff || r2 > 0x7fffffff || thread_num < 1) return SZHIP_ERR_ARG;
    if (!(eb > 0)) return SZHIP_ERR_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return SZHIP_ERR_NODEVICE;
    const int rc = dtype == SZHIP_F32
               ? compress_omp_impl<float>(ctx, data, data_on_device, r0, r1, r2, eb, thread_num, params, meta, meta_len, out_on_device, out, out_size, stats)
               : compress_omp_impl<double>(ctx, data, data_on_device, r0, r1, r2, eb, thread_num, params, meta, meta_len, out_on_device, out, out_size, stats);
    if (rc != SZHIP_OK) hipStreamSynchronize(ctx->stream);
    return rc;
}

int szhip_decompress_omp(szhip_ctx *ctx, int dtype, const unsigned char *stream, int stream_on_device, size_t stream_len, size_t body_off,
                         size_t r0, size_t r1, size_t r2, void *out, int out_on_device, szhip_stats *stats)
{
    if (!ctx || !stream || !out || body_off >= stream_len) return SZHIP_ERR_ARG;
    if (r0 < 1 || r1 < 1 || r2 < 1 || r0 > 0x7fffffff || r1 > 0x7fffffff || r2 > 0x7fffffff) return SZHIP_ERR_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return SZHIP_ERR_NODEVICE;
    const int rc = dtype == SZHIP_F32
               ? decompress_omp_impl<float>(ctx, stream, stream_on_device, stream_len, body_off, r0, r1, r2, out, out_on_device, stats)
               : decompress_omp_impl<double>(ctx, stream, stream_on_device, stream_len, body_off, r0, r1, r2, out, out_on_device, stats);
    if (rc != SZHIP_OK) hipStreamSynchronize(ctx->stream);
    return rc;
}

int szhip_pwr_prepare(szhip_ctx *ctx, int dtype, const void *data, int data_on_device, size_t n, double vmin, double vmax, double pwr_ratio,
                      void **d_log, unsigned char *signs_host, int *positive, double *real_precision, double *value_range, double *median,
                      double *min_log_value)
{
    if (!ctx || !data || !n || !d_log || !positive || !real_precision || !value_range || !median || !min_log_value || !(pwr_ratio > 0)) return SZHIP_ERR_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return SZHIP_ERR_NODEVICE;
    const int rc = dtype == SZHIP_F32
        ? pwr_prepare_impl<float>(ctx, data, data_on_device, n, vmin, vmax, pwr_ratio, d_log, signs_host, positive, real_precision, value_range, median, min_log_value)
        : pwr_prepare_impl<double>(ctx, data, data_on_device, n, vmin, vmax, pwr_ratio, d_log, signs_host, positive, real_precision, value_range, median, min_log_value);
    if (rc != SZHIP_OK) hipStreamSynchronize(ctx->stream);
    return rc;
}

int szhip_msst_prepare(szhip_ctx *ctx, int dtype, const void *data, int data_on_device, size_t n, double vmax, double pwr_ratio, void **d_prepared,
                       unsigned char *signs_host, int *positive, double *near_zero, double *median_log, double *min_log_value)
{
    if (!ctx || !data || !d_prepared || !positive || !near_zero || !median_log || !min_log_value || n == 0 || !(pwr_ratio > 0)) return SZHIP_ERR_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return SZHIP_ERR_NODEVICE;
    const int rc = dtype == SZHIP_F32
        ? msst_prepare_impl<float>(ctx, data, data_on_device, n, vmax, pwr_ratio, d_prepared, signs_host, positive, near_zero, median_log, min_log_value)
        : msst_prepare_impl<double>(ctx, data, data_on_device, n, vmax, pwr_ratio, d_prepared, signs_host, positive, near_zero, median_log, min_log_value);
    if (rc != SZHIP_OK) hipStreamSynchronize(ctx->stream);
    return rc;
}

int szhip_compress_sz14_pwr(szhip_ctx *ctx, int dtype, const void *data, int data_on_device, size_t r0, size_t r1, size_t r2, double eb,
                            double value_range, double median, const szhip_params *params, const unsigned char *meta, size_t meta_len,
                            const szhip_pwr *pwr, int out_on_device, unsigned char **out, size_t *out_size, szhip_stats *stats)
{
    if (!ctx || !data || !params || !meta || !out || !out_size || !pwr || (pwr->signs_blob_size && !pwr->signs_blob)) return SZHIP_ERR_ARG;
    if ((r0 != 0 && r0 < 2) || (r1 < 2 && !(r0 == 0 && r1 == 0)) || r2 < 2 || r0 > 0x7fffffff || r1 > 0x7fffffff || r2 > 0x7fffffff) return SZHIP_ERR_ARG;
    if (!(eb > 0)) return SZHIP_ERR_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return SZHIP_ERR_NODEVICE;
    const compress_call_guard in_flight;
    unsigned char *const out0 = *out; const size_t cap0 = *out_size;     // (a failed first attempt clears them: the repetition starts from the caller's values)
    const int rc = with_ticket_fallback(ctx, [&]() { *out = out0; *out_size = cap0; return dtype == SZHIP_F32
               ? compress14_impl<float>(ctx, data, data_on_device, r0, r1, r2, eb, value_range, median, params, meta, meta_len, pwr, out_on_device, out, out_size, stats)
               : compress14_impl<double>(ctx, data, data_on_device, r0, r1, r2, eb, value_range, median, params, meta, meta_len, pwr, out_on_device, out, out_size, stats); });
    if (rc != SZHIP_OK) { if (ctx->stream3) hipStreamSynchronize(ctx->stream3); hipStreamSynchronize(ctx->stream2); hipStreamSynchronize(ctx->stream); }
    return rc;
}

int szhip_sz14_pwr_locate(int dtype, const unsigned char *stream, size_t stream_len, size_t body_off, size_t *blob_off, size_t *blob_size, double *min_log_value)
{
    if (!stream || !blob_off || !blob_size || !min_log_value) return SZHIP_ERR_ARG;
    const size_t es = dtype == SZHIP_F32 ? 4 : 8;
    if (stream_len < 4) return SZHIP_ERR_STREAM;
    const size_t x = (stream[3] & 0x08) ? 2 : 0;                                    // table-driven form: plus_bits, max_bits after reqLength (:164-168)
    const size_t fixed = 4 + 1 + 8 + 4 + 4 + es + 1 + x + 8 + 8 + 8 + 8 + es;      // ... up to the type array (TightDataPointStorageF.c:133-240)
    if (body_off + fixed > stream_len) return SZHIP_ERR_STREAM;
    const unsigned char *q = stream + body_off + 4 + 1 + 8;
    *blob_size = szhost_get_u32be(q); q += 4 + 4 + es + 1 + x + 8;
    const uint64_t type_size = szhost_get_u64be(q); q += 8 + 8 + 8;
    *min_log_value = dtype == SZHIP_F32 ? (double)szhost_get_f32be(q) : szhost_get_f64be(q);
    if (type_size > stream_len || *blob_size > stream_len || body_off + fixed + type_size + *blob_size > stream_len) return SZHIP_ERR_STREAM;
    *blob_off = body_off + fixed + (size_t)type_size;
    return SZHIP_OK;
}

int szhip_decompress_sz14_pwr(szhip_ctx *ctx, int dtype, const unsigned char *stream, int stream_on_device, size_t stream_len, size_t body_off,
                              size_t r0, size_t r1, size_t r2, const unsigned char *signs_host, void *out, int out_on_device, szhip_stats *stats)
{
    if (!ctx || !stream || !out || body_off >= stream_len || stream_on_device) return SZHIP_ERR_ARG;        // the header is read on the host
    if ((r0 != 0 && r0 < 2) || (r1 < 2 && !(r0 == 0 && r1 == 0)) || r2 < 2 || r0 > 0x7fffffff || r1 > 0x7fffffff || r2 > 0x7fffffff) return SZHIP_ERR_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return SZHIP_ERR_NODEVICE;
    size_t bo, bs; double thr;
    if (szhip_sz14_pwr_locate(dtype, stream, stream_len, body_off, &bo, &bs, &thr) != SZHIP_OK) return SZHIP_ERR_STREAM;
    const bool msst = (stream[3] & 0x08) != 0;                   // TightDataPointStorageF.c:81
    const int rc = with_ticket_fallback(ctx, [&]() { return dtype == SZHIP_F32
               ? decompress14_pwr_impl<float>(ctx, stream, 0, stream_len, body_off, r0, r1, r2, signs_host, thr, msst, out, out_on_device, stats)
               : decompress14_pwr_impl<double>(ctx, stream, 0, stream_len, body_off, r0, r1, r2, signs_host, thr, msst, out, out_on_device, stats); });
    if (rc != SZHIP_OK) { if (ctx->stream3) hipStreamSynchronize(ctx->stream3); hipStreamSynchronize(ctx->stream2); hipStreamSynchronize(ctx->stream); }
    return rc;
}

int szhip_decompress(szhip_ctx *ctx, int dtype, const unsigned char *stream, int stream_on_device, size_t stream_len, size_t body_off,
                     size_t r0, size_t r1, size_t r2, void *out, int out_on_device, szhip_stats *stats)
{
    if (!ctx || !stream || !out || body_off >= stream_len) return SZHIP_ERR_ARG;
    if ((r0 != 0 && r0 < 2) || r1 < 2 || r2 < 2 || r0 > 0x7fffffff || r1 > 0x7fffffff || r2 > 0x7fffffff) return SZHIP_ERR_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return SZHIP_ERR_NODEVICE;
    return with_ticket_fallback(ctx, [&]() { return dtype == SZHIP_F32
               ? decompress_impl<float>(ctx, stream, stream_on_device, stream_len, body_off, r0, r1, r2, out, out_on_device, stats)
               : decompress_impl<double>(ctx, stream, stream_on_device, stream_len, body_off, r0, r1, r2, out, out_on_device, stats); });
}

int szhip_debug_fetch(szhip_ctx *ctx, int which, void *dst, size_t bytes)
{
    if (!ctx || !dst) return SZHIP_ERR_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return SZHIP_ERR_NODEVICE;
    DevBuf *bufs[] = {&ctx->coef, &ctx->blk_lor, &ctx->codes_nat, &ctx->codes_blk, &ctx->hist, &ctx->col_zeros, &ctx->col_off,
                      &ctx->unpred, &ctx->stream_buf, &ctx->trace, &ctx->rb_down, &ctx->rb_right, &ctx->small};
    if (which < 0 || which >= (int)(sizeof(bufs) / sizeof(bufs[0]))) return SZHIP_ERR_ARG;
    if (!bufs[which]->p || bufs[which]->cap < bytes) return SZHIP_ERR_ARG;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpy(dst, bufs[which]->p, bytes, hipMemcpyDeviceToHost));
    return SZHIP_OK;
}

} // extern "C"
