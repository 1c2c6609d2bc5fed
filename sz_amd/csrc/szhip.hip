// szhip.hip -- C-ABI HIP layer (include/szhip.h): owns device buffers, launches the kernels of
// szhip_kernels.h in stream order and calls the short serial host pieces of szhost.c between them.
// gfx950 only; no CPU fallback: every entry point returns an error if the HIP runtime/device is missing.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <array>
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
// (round 6) A sweep that is FED while the host's chains run (arrays with regression blocks, a call that starts alone) must not be joined half-way by another lone context's
// call: its slices' kernels then wait behind the other call's sweep, the fed sweep runs into its bounded waits, and the call is repeated -- streams stayed right in 20 000
// calls of two and four contexts on threads, but single calls took up to 5.7 s (profiles/r06_multi_context_stress_before_the_fed_call_guard.txt).  So: the fed call raises
// g_fed_active for its duration, and a compress call that starts meanwhile waits at its door (a few milliseconds at most); a call that meets another one inside the library
// notes the time, and for a while after such a meeting nobody feeds (several contexts at work overlap their calls anyway).  Sequentially consistent: either the fed call sees
// the newcomer's count (and does not feed), or the newcomer sees the flag (and waits).
static std::atomic<int> g_fed_active{0};
static std::atomic<long long> g_last_meeting_us{-(1ll << 60)};
static long long mono_us() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (long long)ts.tv_sec * 1000000ll + ts.tv_nsec / 1000; }
struct compress_call_guard {
    compress_call_guard()
    {
        if (g_compress_calls.fetch_add(1) > 0) g_last_meeting_us.store(mono_us());
        const long long t0 = mono_us();
        while (g_fed_active.load() != 0 && mono_us() - t0 < 200000) std::this_thread::yield();       // (bounded: 0.2 s, far beyond a fed call)
    }
    ~compress_call_guard() { g_compress_calls.fetch_sub(1); }
};
// the fed call's side: raise the flag, then look again whether the call is still alone
struct fed_call_guard {
    bool on = false;
    bool try_raise()
    {
        if (mono_us() - g_last_meeting_us.load() < 100000) return false;       // (another context was at work within the last 0.1 s)
        int expected = 0;
        if (!g_fed_active.compare_exchange_strong(expected, 1)) return false;
        if (g_compress_calls.load() > 1) { g_fed_active.store(0); return false; }
        on = true;
        return true;
    }
    ~fed_call_guard() { if (on) g_fed_active.store(0); }
};
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
    int cus = 256;                               // compute units of `device` (hipDeviceAttributeMultiprocessorCount): persistent kernels launch one workgroup per CU at most
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;   // the fit + selection pass runs here, concurrently with the interval optimiser's sampling and host decisions
    hipEvent_t ev_in = nullptr, ev_fit = nullptr, ev_feed = nullptr, ev_sec = nullptr;
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
    DevBuf lor_bits, reg_flags, reg_rank, coef_compact, in, out, codes_nat, codes_blk, coef, blk_lor, faceI, faceJ, rb_down, rb_right, rb_vals, pt_flags, feed_word, progress, trace, order, small, hist, col_zeros, col_zeros64,
        seg_bits, seg_zeros, seg_bitoff, seg_zoff, seg_hist, seg_tab, col_off, partial, samples, unpred, stream_buf, chunk_bits, chunk_off, code_tab, len_tab, dec_tab,
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
    unsigned char *sec_pin[4] = {nullptr, nullptr, nullptr, nullptr}; size_t sec_pin_cap[4] = {0, 0, 0, 0};   // the coefficient sections of the stream header as the chain threads build them: pinned, they go to the device from where they are
    szhip_chain_pool *chain_pool = nullptr;      // the coefficient chains' persistent threads (created with the first array that has regression blocks)
};

namespace {

// The host side of every path, one file per path (all inside this anonymous namespace, one translation unit: the kernels' templates are
// instantiated once):
#include "szhip_rt.inc"      // buffers, streams, staging copies, launchers of the sweep kernels
#include "szhip_sz21.inc"    // SZ 2.1: the hot path (SZ_compress_args / SZ_decompress of float and double arrays)
#include "szhip_sz14.inc"    // SZ 1.4 container, MSST19
#include "szhip_pwr.inc"     // point-wise relative bounds
#include "szhip_omp.inc"     // the reference's OpenMP container

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
        hipEventCreateWithFlags(&ctx->ev_fit, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_feed, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_sec, hipEventDisableTiming) != hipSuccess) { delete ctx; return SZHIP_ERR_NODEVICE; }
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
    DevBuf *bufs[] = {&ctx->lor_bits, &ctx->reg_flags, &ctx->reg_rank, &ctx->coef_compact, &ctx->in, &ctx->out, &ctx->codes_nat, &ctx->codes_blk, &ctx->coef, &ctx->blk_lor, &ctx->faceI, &ctx->faceJ, &ctx->rb_down, &ctx->rb_right, &ctx->rb_vals, &ctx->pt_flags, &ctx->feed_word, &ctx->progress, &ctx->trace,
                      &ctx->order, &ctx->small, &ctx->hist, &ctx->col_zeros, &ctx->col_zeros64, &ctx->seg_bits, &ctx->seg_zeros, &ctx->seg_bitoff, &ctx->seg_zoff, &ctx->seg_hist, &ctx->seg_tab, &ctx->col_off, &ctx->partial,
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
    if (ctx->ev_feed) hipEventDestroy(ctx->ev_feed);
    if (ctx->ev_sec) hipEventDestroy(ctx->ev_sec);
    for (int e = 0; e < 4; ++e) if (ctx->sec_pin[e]) hipHostFree(ctx->sec_pin[e]);
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
