// mall_order.hip — the streaming GIN is a chain of kernels that each read one [rows,128] f32 matrix (419 MB at 819 200 rows) and write
// the next.  MI355X has a 256 MB memory-side cache: does the ORDER in which a kernel walks what its predecessor has just written
// matter, and do non-temporal loads (read once, never again) leave more of the written matrix in it?
// Chain a -> b -> a -> ... of `k_pass` kernels; per pass: direction (same as the producer | alternating), load / store flavour.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mall_order tools/ubench/mall_order.hip && /tmp/mall_order
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int CHUNK = 64 * 1024;          // bytes per block iteration: 256 threads x 16 x 16 B
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NTL, int NTS>
__global__ __launch_bounds__(256) void k_pass(const f4 *x, f4 *y, long nchunk, int rev)
{
    for (long c = blockIdx.x; c < nchunk; c += gridDim.x) {
        const long cc = rev ? nchunk - 1 - c : c;
        const f4 *p = x + cc * (CHUNK / 16);
        f4 *q = y + cc * (CHUNK / 16);
        f4 v[16];
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = NTL ? __builtin_nontemporal_load(p + threadIdx.x + 256 * k) : p[threadIdx.x + 256 * k];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const f4 o = {v[k].y + 1.f, v[k].x, v[k].w, v[k].z};
            if (NTS) __builtin_nontemporal_store(o, q + threadIdx.x + 256 * k); else q[threadIdx.x + 256 * k] = o;
        }
    }
}
int main()
{
    const long sizes[] = {128l << 20, 256l << 20, 419430400l, 838860800l};
    for (long bytes : sizes) {
        f4 *a, *b;
        CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
        CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
        const long nchunk = bytes / CHUNK;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int inplace = 0; inplace < 2; inplace++)
        for (int flavour = 0; flavour < (inplace ? 2 : 4); flavour++)
            for (int alt = 0; alt < 2; alt++) {
                const int passes = 24;
                for (int r = 0; r < 2; r++) {                      // r = 0: warm-up
                    CK(hipEventRecord(e0, 0));
                    for (int p = 0; p < passes; p++) {
                        const f4 *src = inplace ? a : p & 1 ? b : a; f4 *dst = inplace ? a : p & 1 ? a : b;
                        const int rev = alt ? (p & 1) : 0;
                        switch (flavour) {
                        case 0: hipLaunchKernelGGL((k_pass<0, 0>), dim3(2048), dim3(256), 0, 0, src, dst, nchunk, rev); break;
                        case 1: hipLaunchKernelGGL((k_pass<1, 0>), dim3(2048), dim3(256), 0, 0, src, dst, nchunk, rev); break;
                        case 2: hipLaunchKernelGGL((k_pass<0, 1>), dim3(2048), dim3(256), 0, 0, src, dst, nchunk, rev); break;
                        default: hipLaunchKernelGGL((k_pass<1, 1>), dim3(2048), dim3(256), 0, 0, src, dst, nchunk, rev); break;
                        }
                    }
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                }
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const double us = ms * 1e3 / passes;
                printf("%7.1f MB per matrix %s loads %s stores %s  direction %s: %.1f us per pass (%.2f TB/s read+write)\n", bytes / 1048576.0, inplace ? "IN PLACE" : "a -> b  ",
                       flavour & 1 ? "non-temporal" : "plain       ", flavour & 2 ? "non-temporal" : "plain       ", alt ? "ALTERNATING" : "same       ", us, 2.0 * bytes / us * 1e-6);
            }
        CK(hipFree(a)); CK(hipFree(b));
    }
    return 0;
}
