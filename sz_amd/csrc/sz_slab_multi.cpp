// sz_slab_multi.cpp -- the multi-GPU path of SURVEY 8(e) for C callers, in ONE process (include/sz_slab.h: sz_slab_compress_multi).
// One host thread and one HIP context per device; every thread runs the ordinary SZ_compress_args on its slab of the slowest dimension
// (the streams are therefore exactly those of sz_slab_compress, and the container the bytes of sz_slab_pack).  What crosses devices:
//   * range-based bound modes need the value range of the WHOLE array (sz_float.c:2845-2866): every device scans its slab (szhip_minmax)
//     and the two scalars are all-reduced -- ncclAllReduce(ncclMin / ncclMax) over xGMI;
//   * the sub-streams are concatenated on every device: ncclAllGather of the sizes, then the payloads (one ncclBroadcast per slab inside a
//     group: slabs differ in size) -- the "single all-gather of the bitstream" of north_star; the host container is assembled from the
//     threads' own host copies, which SZ_compress_args returns anyway.
// RCCL is looked up at run time (dlopen("librccl.so"): the library must load on a box without it); without it, or with a device list that
// names one device twice (tests), the same exchange goes through host memory behind a barrier -- same results, no xGMI.
// The reference has no counterpart (no multi-device path); the Python twin is sz_amd/slab.py over torch.distributed.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <thread>
#include <vector>
#include <chrono>
extern "C" {
#include "sz.h"
#include "sz_slab.h"
#include "szhip.h"
void sz_slab_thread_bind(struct szhip_ctx *ctx, sz_params *cpr_copy, sz_exedata *exe_copy);     // sz_api.c
void sz_slab_preload_lossless(void);
}

namespace {

// ---- RCCL, resolved at run time.  The few declarations needed, as rccl.h has them (ncclDataType_t / ncclRedOp_t values are ABI)
typedef struct ncclComm *ncclComm_t;
enum { NCCL_UINT8 = 1, NCCL_UINT64 = 5, NCCL_FLOAT64 = 8 };
enum { NCCL_MAX = 2, NCCL_MIN = 3 };
struct rccl_api {
    void *h = nullptr;
    int (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    bool load()
    {
#ifdef SZH_HIPSIM
        return false;
#else
        const char *names[] = {"librccl.so", "librccl.so.1", nullptr};
        for (int i = 0; names[i] && !h; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
        if (!h) return false;
        CommInitAll = (decltype(CommInitAll))dlsym(h, "ncclCommInitAll"); CommDestroy = (decltype(CommDestroy))dlsym(h, "ncclCommDestroy");
        AllReduce = (decltype(AllReduce))dlsym(h, "ncclAllReduce"); AllGather = (decltype(AllGather))dlsym(h, "ncclAllGather");
        Broadcast = (decltype(Broadcast))dlsym(h, "ncclBroadcast"); GroupStart = (decltype(GroupStart))dlsym(h, "ncclGroupStart"); GroupEnd = (decltype(GroupEnd))dlsym(h, "ncclGroupEnd");
        return CommInitAll && CommDestroy && AllReduce && AllGather && Broadcast && GroupStart && GroupEnd;
#endif
    }
};

struct barrier_t {
    std::mutex mu; std::condition_variable cv; int n = 0, waiting = 0; unsigned gen = 0;
    void wait() { std::unique_lock<std::mutex> lk(mu); const unsigned g = gen; if (++waiting == n) { waiting = 0; ++gen; cv.notify_all(); } else cv.wait(lk, [&] { return gen != g; }); }
};

struct shared_t {
    int ndev = 0; bool rccl = false; rccl_api api; std::vector<ncclComm_t> comms;
    barrier_t bar;
    std::vector<double> lo, hi;                    // host exchange of the ranges
    std::vector<unsigned char *> streams; std::vector<size_t> bytes; std::vector<int> rc;
    std::vector<double> t_compress;
    std::mutex sim_mu;                             // the CPU shim runs one launch at a time
    int gather_ok = 1; size_t gathered = 0;
    std::vector<int> xbuf_ok;                      // per rank: its device exchange buffers exist (decided before any collective is entered)
    std::vector<int> pbuf_ok;                      // per rank: its payload buffer exists (the second decision of a call: words of its own, so that a rank still reading the first one never sees it)
    int coll_failed = 0;                           // some RCCL / HIP call of the exchange returned an error: the call fails as a whole
};

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct job_t {
    int dataType, mode; double abs_eb, rel, pw; const unsigned char *data; size_t r2, r1; const size_t *bounds; const int *devices;
    const sz_params *cpr0; const sz_exedata *exe0; bool verify_gather;
};

// What a call needs per device and what is expensive to make -- the communicator (ncclCommInitAll: hundreds of milliseconds), the contexts (streams probed
// for shared hardware queues, gigabytes of workspaces on first use), the exchange buffers -- is kept from call to call for the same device list (round 5).
// sz_slab_multi_release (also called by SZ_Finalize) gives it back; SZ_SLAB_MULTI_CACHE=0: everything per call, as before.
struct multi_cache_t {
    std::mutex mu;                                 // one multi-device call at a time
    std::vector<int> devs;
    rccl_api api; bool rccl = false; std::vector<ncclComm_t> comms;
    std::vector<szhip_ctx *> ctx; std::vector<hipStream_t> st; std::vector<double *> d_rng; std::vector<unsigned long long *> d_sizes;
    void release()
    {
        for (size_t r = 0; r < devs.size(); ++r) {
            if (hipSetDevice(devs[r]) != hipSuccess) continue;
            if (r < d_rng.size() && d_rng[r]) (void)hipFree(d_rng[r]);
            if (r < d_sizes.size() && d_sizes[r]) (void)hipFree(d_sizes[r]);
            if (r < st.size() && st[r]) (void)hipStreamDestroy(st[r]);
            if (r < ctx.size() && ctx[r]) szhip_destroy(ctx[r]);
        }
        if (rccl) for (ncclComm_t c : comms) if (c) api.CommDestroy(c);
        devs.clear(); comms.clear(); ctx.clear(); st.clear(); d_rng.clear(); d_sizes.clear(); rccl = false;
    }
};
multi_cache_t g_cache;

void worker(shared_t *S, const job_t *J, int r, multi_cache_t *C)
{
    const size_t esz = J->dataType == SZ_FLOAT ? 4 : 8, plane = J->r2 * J->r1;
    const size_t z0 = J->bounds[2 * r], h = J->bounds[2 * r + 1] - z0, count = h * plane;
    const unsigned char *slab = J->data + z0 * plane * esz;
    S->rc[r] = SZ_NSCS;
    szhip_ctx *ctx = C ? C->ctx[r] : nullptr;
    hipStream_t st = C ? C->st[r] : nullptr;
    bool alive = hipSetDevice(J->devices[r]) == hipSuccess && (ctx || szhip_create(&ctx, J->devices[r]) == SZHIP_OK) && (st || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess);
    if (C) { C->ctx[r] = ctx; C->st[r] = st; }                       // (kept for the next call, whatever becomes of this one)
    sz_params cpr = *J->cpr0; sz_exedata exe = *J->exe0;
    sz_slab_thread_bind(ctx, &cpr, &exe);
    // ---- the exchange buffers of this rank, BEFORE any collective: a rank that entered a collective the others skip -- or the other way round -- hangs
    // the whole call, so all ranks decide together (host barrier) whether the device transport is used; if any rank could not allocate, all take the
    // host transport.  Every RCCL / HIP return of the exchange is checked; an error fails the call (S->coll_failed), it never leaves a garbage range.
    double *d_rng = C ? C->d_rng[r] : nullptr; unsigned long long *d_sizes = C ? C->d_sizes[r] : nullptr;
    if (S->rccl) {
        const bool got = alive && (d_rng || hipMalloc((void **)&d_rng, 4 * sizeof(double)) == hipSuccess) && (d_sizes || hipMalloc((void **)&d_sizes, ((size_t)S->ndev + 1) * 8) == hipSuccess);
        if (C) { C->d_rng[r] = d_rng; C->d_sizes[r] = d_sizes; }
        S->xbuf_ok[r] = got ? 1 : 0;
        S->bar.wait();
    }
    bool use_rccl = S->rccl;
    if (S->rccl) for (int q = 0; q < S->ndev; ++q) if (!S->xbuf_ok[q]) use_rccl = false;
    auto cfail = [&](bool bad) { if (bad) { std::lock_guard<std::mutex> lk(S->sim_mu); S->coll_failed = 1; } return bad; };
    // ---- the range of the whole array (range-based modes): local scan, then min / max over the devices
    int mode = J->mode; double abs_eb = J->abs_eb;
    const bool ranged = mode == REL || mode == ABS_AND_REL || mode == ABS_OR_REL;
    if (ranged) {
        double lo = 0, hi = 0; bool have = false;
        if (alive && count) {
            void *d_in = nullptr;
#ifdef SZH_HIPSIM
            std::lock_guard<std::mutex> lk(S->sim_mu);
#endif
            have = szhip_stage_input(ctx, slab, count * esz, &d_in) == SZHIP_OK &&
                   szhip_minmax(ctx, J->dataType == SZ_FLOAT ? SZHIP_F32 : SZHIP_F64, d_in, 1, count, &lo, &hi) == SZHIP_OK;
            if (!have) alive = false;
        }
        if (!have) { lo = 1.0 / 0.0; hi = -1.0 / 0.0; }           // (an empty slab takes no part)
        if (use_rccl) {
            // ONE all-reduce: min over {lo, -hi}
            const double v[2] = {lo, -hi};
            double wv[2] = {lo, -hi};
            bool bad = hipMemcpyAsync(d_rng, v, 16, hipMemcpyHostToDevice, st) != hipSuccess;
            bad = S->api.AllReduce(d_rng, d_rng + 2, 2, NCCL_FLOAT64, NCCL_MIN, S->comms[r], st) != 0 || bad;
            bad = hipMemcpyAsync(wv, d_rng + 2, 16, hipMemcpyDeviceToHost, st) != hipSuccess || bad;
            bad = hipStreamSynchronize(st) != hipSuccess || bad;
            cfail(bad);
            lo = wv[0]; hi = -wv[1];
        } else {
            S->lo[r] = lo; S->hi[r] = hi;
            S->bar.wait();
            for (int q = 0; q < S->ndev; ++q) { if (S->lo[q] < lo) lo = S->lo[q]; if (S->hi[q] > hi) hi = S->hi[q]; }
            S->bar.wait();
        }
        if (!(lo <= hi) || lo != lo || hi != hi || lo - lo != 0 || hi - hi != 0) { cfail(true); lo = 0; hi = 0; }     // no rank saw a finite value: nothing to derive a bound from
        // getRealPrecision_float / _double on the whole array's range (dataCompression.c:288-332; the float helpers narrow both operands)
        const double range = J->dataType == SZ_FLOAT ? (double)((float)hi - (float)lo) : hi - lo, rel = J->rel * range;
        if (mode == REL) abs_eb = rel;
        else if (J->dataType == SZ_FLOAT) { const float fa = (float)J->abs_eb, fb = (float)rel; abs_eb = mode == ABS_AND_REL ? (fa < fb ? fa : fb) : (fa > fb ? fa : fb); }
        else abs_eb = mode == ABS_AND_REL ? (J->abs_eb < rel ? J->abs_eb : rel) : (J->abs_eb > rel ? J->abs_eb : rel);
        mode = ABS;
    }
    // ---- this device's slab: the ordinary entry point, on this thread's context and its own copy of the configuration
    const double t0 = now_s();
    if (alive) {
        if (h == 0) { S->streams[r] = (unsigned char *)malloc(1); S->bytes[r] = 0; S->rc[r] = S->streams[r] ? SZ_SCES : SZ_NSCS; }
        else {
#ifdef SZH_HIPSIM
            std::lock_guard<std::mutex> lk(S->sim_mu);
#endif
            S->streams[r] = SZ_compress_args(J->dataType, (void *)slab, &S->bytes[r], mode, abs_eb, J->rel, J->pw, 0, 0, h, J->r2, J->r1);
            S->rc[r] = S->streams[r] ? SZ_SCES : SZ_NSCS;
        }
    }
    S->t_compress[r] = now_s() - t0;
    // ---- the sub-streams on every device: sizes, then payloads (north_star's all-gather over xGMI).  Every rank takes part even when its
    // own slab failed (a collective that one rank skips hangs the others): a failed slab contributes an empty stream
    if (use_rccl) {
        unsigned char *d_all = nullptr;
        const unsigned long long mine = S->rc[r] == SZ_SCES ? (unsigned long long)S->bytes[r] : 0ull;
        std::vector<unsigned long long> sizes((size_t)S->ndev, 0ull);
        {
            bool bad = hipMemcpyAsync(d_sizes + S->ndev, &mine, 8, hipMemcpyHostToDevice, st) != hipSuccess;
            bad = S->api.AllGather(d_sizes + S->ndev, d_sizes, 1, NCCL_UINT64, S->comms[r], st) != 0 || bad;
            bad = hipMemcpyAsync(sizes.data(), d_sizes, (size_t)S->ndev * 8, hipMemcpyDeviceToHost, st) != hipSuccess || bad;
            bad = hipStreamSynchronize(st) != hipSuccess || bad;
            if (cfail(bad)) std::fill(sizes.begin(), sizes.end(), 0ull);
        }
        size_t total = 0, my_off = 0;
        for (int q = 0; q < S->ndev; ++q) { if (q == r) my_off = total; total += (size_t)sizes[q]; }
        // the payload buffer: again decided together (a rank without it cannot take part in the broadcasts)
        S->pbuf_ok[r] = hipMalloc((void **)&d_all, total ? total : 1) == hipSuccess ? 1 : 0;
        S->bar.wait();
        bool all_have = true;
        for (int q = 0; q < S->ndev; ++q) if (!S->pbuf_ok[q]) all_have = false;
        if (!all_have) cfail(true);
        else {
            bool bad = false;
            if (mine && sizes[r] == mine) bad = hipMemcpyAsync(d_all + my_off, S->streams[r], (size_t)mine, hipMemcpyHostToDevice, st) != hipSuccess;
            bad = S->api.GroupStart() != 0 || bad;
            size_t off = 0;
            for (int q = 0; q < S->ndev; ++q) { if (sizes[q]) bad = S->api.Broadcast(d_all + off, d_all + off, (size_t)sizes[q], NCCL_UINT8, q, S->comms[r], st) != 0 || bad; off += (size_t)sizes[q]; }
            bad = S->api.GroupEnd() != 0 || bad;
            bad = hipStreamSynchronize(st) != hipSuccess || bad;
            cfail(bad);
            if (r == 0) S->gathered = total;
            if (J->verify_gather) {                                 // what every device now holds = the slabs' streams back to back
                std::vector<unsigned char> back(total ? total : 1);
                if (hipMemcpy(back.data(), d_all, total, hipMemcpyDeviceToHost) != hipSuccess) cfail(true);
                S->bar.wait();                                      // (all host streams are final)
                size_t o2 = 0; bool ok = true;
                for (int q = 0; q < S->ndev; ++q) { if (sizes[q] && memcmp(back.data() + o2, S->streams[q], (size_t)sizes[q]) != 0) ok = false; o2 += (size_t)sizes[q]; }
                if (!ok) { std::lock_guard<std::mutex> lk(S->sim_mu); S->gather_ok = 0; }
            }
        }
        if (d_all) (void)hipFree(d_all);
    }
    sz_slab_thread_bind(nullptr, nullptr, nullptr);
    if (!C) {
        if (d_rng) (void)hipFree(d_rng);
        if (d_sizes) (void)hipFree(d_sizes);
        if (st) hipStreamDestroy(st);
        if (ctx) szhip_destroy(ctx);
    }
}

} // namespace

extern "C" unsigned char *sz_slab_compress_multi(int dataType, void *data, size_t *outSize, int errBoundMode, double absErrBound, double relBoundRatio,
                                                 double pwrBoundRatio, size_t r3, size_t r2, size_t r1, int ndev, const int *devices, sz_slab_multi_info *info)
{
    if (info) memset(info, 0, sizeof(*info));
    if (!data || !outSize || ndev < 1 || ndev > 64 || r3 < 1 || r2 < 1 || r1 < 1 || (dataType != SZ_FLOAT && dataType != SZ_DOUBLE)) return nullptr;
    if (errBoundMode == PSNR || errBoundMode == NORM || errBoundMode >= PW_REL) {
        printf("Error: sz_slab_compress_multi takes ABS, REL, ABS_AND_REL and ABS_OR_REL (PSNR / NORM / point-wise bounds: sz_slab_compress).\n");
        return nullptr;
    }
    if (confparams_cpr == nullptr && SZ_Init(nullptr) != SZ_SCES) return nullptr;
    sz_slab_preload_lossless();
    std::vector<int> devs((size_t)ndev);
    bool distinct = true;
    for (int r = 0; r < ndev; ++r) { devs[r] = devices ? devices[r] : r; for (int q = 0; q < r; ++q) if (devs[q] == devs[r]) distinct = false; }
    std::vector<size_t> bounds(2 * (size_t)ndev);
    sz_slab_bounds(r3, ndev, 6, bounds.data());
    const char *nocache = getenv("SZ_SLAB_MULTI_CACHE");
    multi_cache_t *const C = (nocache && atoi(nocache) == 0) ? nullptr : &g_cache;
    std::unique_lock<std::mutex> cache_lock;
    if (C) {
        cache_lock = std::unique_lock<std::mutex>(C->mu);
        if (C->devs != devs) {                     // another device list: what was kept is given back
            C->release();
            C->devs = devs;
            C->ctx.assign(ndev, nullptr); C->st.assign(ndev, nullptr); C->d_rng.assign(ndev, nullptr); C->d_sizes.assign(ndev, nullptr);
        }
    }
    shared_t S;
    S.ndev = ndev; S.bar.n = ndev;
    S.lo.assign(ndev, 0); S.hi.assign(ndev, 0); S.xbuf_ok.assign(ndev, 0); S.pbuf_ok.assign(ndev, 0); S.streams.assign(ndev, nullptr); S.bytes.assign(ndev, 0); S.rc.assign(ndev, SZ_NSCS); S.t_compress.assign(ndev, 0);
    const char *no = getenv("SZ_SLAB_NO_RCCL");
    if (distinct && !(no && atoi(no)) && S.api.load()) {
        if (C && C->rccl) { S.comms = C->comms; S.rccl = true; }      // the communicator of the last call with these devices
        else {
            S.comms.assign(ndev, nullptr);
            S.rccl = S.api.CommInitAll(S.comms.data(), ndev, devs.data()) == 0;
            if (!S.rccl) S.comms.clear();
            else if (C) { C->api = S.api; C->comms = S.comms; C->rccl = true; }
        }
    }
    const char *vg = getenv("SZ_SLAB_VERIFY_GATHER");
    const sz_params cpr0 = *confparams_cpr; const sz_exedata exe0 = *exe_params;
    job_t J = {dataType, errBoundMode, absErrBound, relBoundRatio, pwrBoundRatio, (const unsigned char *)data, r2, r1, bounds.data(), devs.data(), &cpr0, &exe0, vg && atoi(vg)};
    const double t0 = now_s();
    std::vector<std::thread> th;
    for (int r = 0; r < ndev; ++r) th.emplace_back(worker, &S, &J, r, C);
    for (auto &t : th) t.join();
    const double t1 = now_s();
    if (S.rccl && !(C && C->rccl)) for (ncclComm_t c : S.comms) if (c) S.api.CommDestroy(c);
    if (C && S.coll_failed) C->release();          // (a communicator that failed a collective is not used again)
    unsigned char *out = nullptr;
    bool ok = S.gather_ok != 0 && S.coll_failed == 0;
    for (int r = 0; r < ndev; ++r) if (S.rc[r] != SZ_SCES) ok = false;
    if (ok) {
        const size_t dims[3] = {r3, r2, r1};
        out = sz_slab_pack(dataType, dims, ndev, bounds.data(), (const unsigned char *const *)S.streams.data(), S.bytes.data(), outSize);
    }
    for (int r = 0; r < ndev; ++r) free(S.streams[r]);
    if (info) {
        info->devices = ndev; info->used_rccl = S.rccl ? 1 : 0; info->gathered_bytes = S.gathered; info->seconds_total = t1 - t0;
        double m = 0; for (double t : S.t_compress) if (t > m) m = t;
        info->seconds_slowest_slab = m;
    }
    return out;
}

extern "C" void sz_slab_multi_release(void)
{
    std::lock_guard<std::mutex> lk(g_cache.mu);
    g_cache.release();
}
