// hipsim.cpp -- TEST INFRASTRUCTURE ONLY: thread-pool execution engine of the HIP-on-CPU shim
// (tests/sim/hip_shim/hip/hip_runtime.h), plus the product's szhip.hip compiled against it.
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <ucontext.h>
#include <vector>

// Execution model: the lanes of ONE workgroup are fibres (ucontext) of the calling thread, switched round-robin at every barrier,
// wavefront exchange and s_sleep (the kernels' spin-waits all nap through s_sleep, so a waiting "wavefront" hands the processor on).
// Workgroups run one after the other in launch order.  (The first engine gave every lane an OS thread: a 704-lane workgroup of
// k_pencil spinning on 8 cores spent its time in the kernel's scheduler -- 35 s for an 8 x 8 x 128 array.)
namespace hipsim {
TIdx t_threadIdx;
dim3 t_blockIdx;
dim3 g_blockDim, g_gridDim;
char *g_dyn_smem = nullptr;

namespace {
constexpr int MAXT = 1024;
constexpr size_t STACK = 256 << 10;
struct Fiber { ucontext_t ctx; char *stack = nullptr; bool done = true; };
Fiber g_f[MAXT];
ucontext_t g_main;
int g_cur = 0, g_nthreads = 0, g_live = 0;
const std::function<void()> *g_body = nullptr;
alignas(16) unsigned char g_xbuf[MAXT / 64][64][16];

inline void enter(int f) { g_cur = f; t_threadIdx = TIdx{(unsigned)f, 0, 0}; }
int next_runnable(int me)
{
    for (int k = 1; k <= g_nthreads; ++k) { const int f = (me + k) % g_nthreads; if (!g_f[f].done) return f; }
    return -1;
}
void yield()
{
    const int me = g_cur, nx = next_runnable(me);
    if (nx < 0 || nx == me) return;
    enter(nx);
    swapcontext(&g_f[me].ctx, &g_f[nx].ctx);
}
void fiber_main()
{
    (*g_body)();
    const int me = g_cur;
    g_f[me].done = true;
    --g_live;
    const int nx = next_runnable(me);
    if (nx < 0) { swapcontext(&g_f[me].ctx, &g_main); return; }
    enter(nx);
    swapcontext(&g_f[me].ctx, &g_f[nx].ctx);
}
struct Barrier {
    int count = 0, waiting = 0; unsigned gen = 0;
    void reset(int n) { count = n; waiting = 0; }
    void wait()
    {
        const unsigned g = gen;
        if (++waiting == count) { waiting = 0; ++gen; }
        else while (gen == g) yield();
    }
};
Barrier g_block_bar, g_wave_bar[MAXT / 64];
} // namespace

void fiber_yield() { yield(); }
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
    if ((int)block.x > MAXT || block.y != 1 || block.z != 1) { fprintf(stderr, "hipsim: unsupported block shape\n"); abort(); }
    if (grid.x == 0 || grid.y == 0 || block.x == 0) return;
    if (g_nthreads != 0) { fprintf(stderr, "hipsim: nested or concurrent launch\n"); abort(); }
    std::vector<char> smem(shmem + 64);
    g_dyn_smem = (char *)(((uintptr_t)smem.data() + 15) & ~(uintptr_t)15);
    g_nthreads = (int)block.x; g_blockDim = block; g_gridDim = grid; g_body = &body;
    for (int f = 0; f < g_nthreads; ++f)
        if (!g_f[f].stack) {
            g_f[f].stack = (char *)mmap(nullptr, STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
            if (g_f[f].stack == MAP_FAILED) { fprintf(stderr, "hipsim: cannot map a fibre stack\n"); abort(); }
        }
    for (unsigned by = 0; by < grid.y; ++by)
        for (unsigned bx = 0; bx < grid.x; ++bx) {
            t_blockIdx = dim3(bx, by, 0);
            memset(g_dyn_smem, 0xCD, shmem);                     // poison dynamic LDS per workgroup
            g_block_bar.reset(g_nthreads);
            for (int w = 0; w < MAXT / 64; ++w) { int rem = g_nthreads - w * 64; g_wave_bar[w].reset(rem < 0 ? 0 : (rem < 64 ? rem : 64)); }
            for (int f = 0; f < g_nthreads; ++f) {
                getcontext(&g_f[f].ctx);
                g_f[f].ctx.uc_stack.ss_sp = g_f[f].stack; g_f[f].ctx.uc_stack.ss_size = STACK; g_f[f].ctx.uc_link = nullptr;
                makecontext(&g_f[f].ctx, fiber_main, 0);
                g_f[f].done = false;
            }
            g_live = g_nthreads;
            enter(0);
            swapcontext(&g_main, &g_f[0].ctx);
            if (g_live != 0) { fprintf(stderr, "hipsim: workgroup left %d lanes behind\n", g_live); abort(); }
        }
    g_nthreads = 0;
}
} // namespace hipsim

#include "../../sz_amd/csrc/szhip.hip"
