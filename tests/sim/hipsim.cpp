// hipsim.cpp -- TEST INFRASTRUCTURE ONLY: thread-pool execution engine of the HIP-on-CPU shim
// (tests/sim/hip_shim/hip/hip_runtime.h), plus the product's szhip.hip compiled against it.
#include <hip/hip_runtime.h>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

namespace hipsim {
thread_local TIdx t_threadIdx;
dim3 g_blockIdx, g_blockDim, g_gridDim;
char *g_dyn_smem = nullptr;

namespace {
struct Barrier {
    std::mutex m; std::condition_variable cv; int count = 0, waiting = 0; unsigned gen = 0;
    void reset(int n) { count = n; waiting = 0; }
    void wait()
    {
        std::unique_lock<std::mutex> lk(m);
        unsigned g = gen;
        if (++waiting == count) { waiting = 0; ++gen; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g; });
    }
};
constexpr int MAXT = 1024;
Barrier g_block_bar, g_wave_bar[MAXT / 64], g_start, g_done;
alignas(16) unsigned char g_xbuf[MAXT / 64][64][16];
int g_nthreads = 0;
const std::function<void()> *g_body = nullptr;
std::vector<std::thread> g_pool;
bool g_quit = false;

void worker(int tid)
{
    for (;;) {
        g_start.wait();
        if (g_quit) return;
        if (tid < g_nthreads) { t_threadIdx = TIdx{(unsigned)tid, 0, 0}; (*g_body)(); }
        g_done.wait();
    }
}
void ensure_pool()
{
    if (!g_pool.empty()) return;
    g_start.reset(MAXT + 1); g_done.reset(MAXT + 1);
    for (int t = 0; t < MAXT; ++t) g_pool.emplace_back(worker, t);
    atexit([] { g_quit = true; g_start.wait(); for (auto &t : g_pool) t.join(); });
}
} // namespace

int lanes_in_wave()
{
    const int w = t_threadIdx.x >> 6;
    const int rem = g_nthreads - w * 64;
    return rem < 64 ? rem : 64;
}
void sync_block() { g_block_bar.wait(); }
void wave_exchange(const void *src, void *dst_all, size_t elem)
{
    const int w = t_threadIdx.x >> 6, l = t_threadIdx.x & 63;
    memcpy(g_xbuf[w][l], src, elem);
    g_wave_bar[w].wait();
    for (int i = 0; i < 64; ++i) memcpy((char *)dst_all + i * elem, g_xbuf[w][i], elem);
    g_wave_bar[w].wait();
}
void launch(const std::function<void()> &body, dim3 grid, dim3 block, size_t shmem)
{
    ensure_pool();
    if ((int)block.x > MAXT || block.y != 1 || block.z != 1) { fprintf(stderr, "hipsim: unsupported block shape\n"); abort(); }
    std::vector<char> smem(shmem + 64);
    g_dyn_smem = (char *)(((uintptr_t)smem.data() + 15) & ~(uintptr_t)15);
    g_nthreads = (int)block.x; g_blockDim = block; g_gridDim = grid; g_body = &body;
    g_block_bar.reset(g_nthreads);
    for (int w = 0; w < MAXT / 64; ++w) { int rem = g_nthreads - w * 64; g_wave_bar[w].reset(rem < 0 ? 0 : (rem < 64 ? rem : 64)); }
    for (unsigned by = 0; by < grid.y; ++by)
        for (unsigned bx = 0; bx < grid.x; ++bx) {
            g_blockIdx = dim3(bx, by, 0);
            memset(g_dyn_smem, 0xCD, shmem); // poison dynamic LDS per workgroup
            g_start.wait();
            g_done.wait();
        }
}
} // namespace hipsim

#include "../../sz_amd/csrc/szhip.hip"
