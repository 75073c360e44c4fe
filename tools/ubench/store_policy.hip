// store_policy.hip — does the cache policy of the stores change the launch-to-launch time of a kernel with the step kernel's
// footprint (11.3 MB read, 14.5 MB written, 4096 J6M6E2 instances)?  The step kernel's in-kernel time is 7.7 us, its event time
// 12.2 us; the end of a kernel writes the dirty lines of eight L2s back, and the question is how much of the difference that is and
// whether write-through / non-temporal stores move it under the kernel's own run time.
//   POLICY 0: plain global_store_dwordx4            1: nt (non-temporal)
//          2: sc1 (agent scope: written through)    3: sc0 sc1 (system scope)         4: sc1 nt
// Each policy: dependent back-to-back launches on one stream (dst of launch i is src of launch i+1, like the rollout's kernels),
// HIP events around every launch, average of the timed launches.  Also an empty kernel (the launch floor) and a read-only one.
//   hipcc --offload-arch=gfx950 -O3 -o store_policy store_policy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int POLICY>
__device__ __forceinline__ void st(u4 *p, u4 v)
{
    if (POLICY == 0) *p = v;
    else if (POLICY == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
    else if (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    else if (POLICY == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(p), "v"(v) : "memory");
}
template <int POLICY>
__global__ __launch_bounds__(256) void k_copy(const u4 *__restrict__ src, u4 *__restrict__ dst, size_t nr, size_t nw, unsigned *sink)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < nr; i += stride) { const u4 v = src[i]; st<POLICY>(dst + i, v); acc ^= v.x; }
    for (; i < nw; i += stride) st<POLICY>(dst + i, u4{acc, acc, acc, acc});
    if (acc == 0x9e3779b9u) *sink = acc;
}
__global__ void k_empty(unsigned *sink) { if (threadIdx.x == 9999) *sink = 1; }
// a kernel that just takes `ticks` of the 100 MHz clock in every wave (a long kernel whose own work does not touch memory)
__global__ void k_spin(unsigned *sink, unsigned ticks)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
    if (threadIdx.x == 9999) *sink = 1;
}
__global__ __launch_bounds__(256) void k_read(const u4 *__restrict__ src, size_t nr, unsigned *sink)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nr; i += stride) acc ^= src[i].x;
    if (acc == 0x9e3779b9u) *sink = acc;
}

template <typename F>
static int timed(const char *what, F launch, int reps)
{
    std::vector<hipEvent_t> ev(2 * (size_t)reps);
    for (auto &e : ev) CHK(hipEventCreate(&e));
    for (int i = 0; i < 20; i++) launch(i);
    for (int i = 0; i < reps; i++) { CHK(hipEventRecord(ev[2 * i], 0)); launch(i); CHK(hipEventRecord(ev[2 * i + 1], 0)); }
    CHK(hipDeviceSynchronize());
    double tot = 0, mn = 1e30;
    for (int i = 0; i < reps; i++) { float ms; CHK(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1])); tot += ms; mn = ms < mn ? ms : mn; }
    // and the whole train between two events (what a rollout sees: launch i+1 queued behind launch i)
    CHK(hipEventRecord(ev[0], 0));
    for (int i = 0; i < reps; i++) launch(i);
    CHK(hipEventRecord(ev[1], 0));
    CHK(hipDeviceSynchronize());
    float tr; CHK(hipEventElapsedTime(&tr, ev[0], ev[1]));
    printf("%-44s per-launch events: avg %6.2f us  min %6.2f us   | train of %d: %6.2f us per launch\n", what, tot / reps * 1e3, mn * 1e3, reps, tr / reps * 1e3);
    for (auto &e : ev) (void)hipEventDestroy(e);
    return 0;
}

// the same train as ONE hipGraph (stream capture of `n` dependent launches), launched `reps` times
template <typename F>
static int timed_graph(const char *what, F launch, int n, int reps)
{
    hipStream_t st; CHK(hipStreamCreate(&st));
    hipGraph_t g; hipGraphExec_t ge;
    CHK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < n; i++) launch(i, st);
    CHK(hipStreamEndCapture(st, &g));
    CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) CHK(hipGraphLaunch(ge, st));
    CHK(hipStreamSynchronize(st));
    CHK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; i++) CHK(hipGraphLaunch(ge, st));
    CHK(hipEventRecord(e1, st));
    CHK(hipStreamSynchronize(st));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s hipGraph of %d dependent launches x %d: %6.2f us per launch\n", what, n, reps, ms / reps / n * 1e3);
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g); (void)hipStreamDestroy(st);
    return 0;
}

int main()
{
    const size_t rb = 11337728, wb = 14528512;           // 4096 x (64 T + 100 M + 176), 4096 x (72 T + 56 M + 307) with T = 36, M = 6
    const size_t nr = rb / 16, nw = wb / 16;
    u4 *a, *b; unsigned *sink;
    CHK(hipMalloc((void **)&a, wb + 64)); CHK(hipMalloc((void **)&b, wb + 64)); CHK(hipMalloc((void **)&sink, 4));
    CHK(hipMemset(a, 1, wb + 64)); CHK(hipMemset(b, 2, wb + 64));
    const int reps = 100;
    timed("empty kernel (256 x 256)", [&](int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, 0, sink); }, reps);
    timed_graph("empty kernel (256 x 256)", [&](int, hipStream_t st) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, sink); }, 100, 20);
    for (unsigned us : {20u, 100u}) {
        char nm[96];
        snprintf(nm, sizeof nm, "spin %u us (256 x 256)", us);
        timed(nm, [&](int) { hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, 0, sink, us * 100); }, reps);
        timed_graph(nm, [&](int, hipStream_t st) { hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, st, sink, us * 100); }, 100, 10);
    }
    timed_graph("empty kernel, graphs of FOUR launches", [&](int, hipStream_t st) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, sink); }, 4, 500);
    timed_graph("copy, graphs of FOUR launches", [&](int i, hipStream_t st) { hipLaunchKernelGGL(k_copy<0>, dim3(2048), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, nr, nw, sink); }, 4, 500);
    timed_graph("copy, plain stores, grid 2048", [&](int i, hipStream_t st) { hipLaunchKernelGGL(k_copy<0>, dim3(2048), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, nr, nw, sink); }, 100, 20);
    for (int grid : {1024, 2048, 4096}) {
        char nm[96];
        snprintf(nm, sizeof nm, "read only, grid %d", grid);
        timed(nm, [&](int) { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, nr, sink); }, reps);
#define RUN(P, label) snprintf(nm, sizeof nm, "copy, %s, grid %d", label, grid); \
        timed(nm, [&](int i) { hipLaunchKernelGGL(k_copy<P>, dim3(grid), dim3(256), 0, 0, (i & 1) ? b : a, (i & 1) ? a : b, nr, nw, sink); }, reps);
        RUN(0, "plain stores")
        RUN(1, "nt stores")
        RUN(2, "sc1 stores")
        RUN(3, "sc0 sc1 stores")
        RUN(4, "sc1 nt stores")
    }
    return 0;
}
