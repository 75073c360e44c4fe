// Micro-benchmark: does VALU work overlap with matrix instructions on gfx950?  f32 16x16x4 vs bf16 16x16x32.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu mfma_valu.hip ; run: ./mfma_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int NV>   // MODE 0: f32 16x16x4, 1: bf16 16x16x32 ; NV = VALU fma per matrix instruction
__global__ __launch_bounds__(1024) void k(float *out, unsigned long long *cyc, int iters, float seed)
{
    f32x4 acc[8];
    for (int t = 0; t < 8; t++) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = seed + threadIdx.x, b = seed * 0.5f;
    bf16x8 ah, bh;
    for (int i = 0; i < 8; i++) { ah[i] = (__bf16)(a + i); bh[i] = (__bf16)(b + i); }
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = seed + i;
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int t = 0; t < 8; t++) {
            if (MODE == 0) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
            else acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[t], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; j++) v[(t + j) & 7] = __builtin_fmaf(v[(t + j) & 7], 1.0001f, 0.5f);
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0.f;
    for (int t = 0; t < 8; t++) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    for (int i = 0; i < 8; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE, int NV>
static void run(const char *name, int threads)
{
    float *out; unsigned long long *cyc, h;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NV>), dim3(256), dim3(threads), 0, 0, out, cyc, iters, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<MODE, NV>), dim3(256), dim3(threads), 0, 0, out, cyc, iters, 1.0f);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const int waves_per_simd = threads / 256;
    const double flops = 256.0 * (threads / 64) * iters * 8.0 * (MODE == 0 ? 2048.0 : 16384.0);
    printf("%-28s waves/SIMD %d  NV %d : %.1f ticks per matrix instr per wave, %.1f per SIMD-instr | %.3f ms  %.1f TFLOP/s  (%.2f GHz s_memtime)\n", name, waves_per_simd, NV,
           (double)h / (iters * 8.0), (double)h / (iters * 8.0 * waves_per_simd), ms, flops / (ms * 1e-3) / 1e12, (double)h / (ms * 1e-3) / 1e9);
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<0, 0>("f32 16x16x4", 256);  run<0, 0>("f32 16x16x4", 512); run<0, 0>("f32 16x16x4", 1024);
    run<0, 2>("f32 16x16x4 + valu", 256);  run<0, 2>("f32 16x16x4 + valu", 512);
    run<0, 6>("f32 16x16x4 + valu", 256);  run<0, 6>("f32 16x16x4 + valu", 512);
    run<1, 0>("bf16 16x16x32", 256); run<1, 0>("bf16 16x16x32", 512);
    run<1, 2>("bf16 16x16x32 + valu", 256); run<1, 2>("bf16 16x16x32 + valu", 512);
    run<1, 3>("bf16 16x16x32 + valu", 256); run<1, 3>("bf16 16x16x32 + valu", 512);
    run<1, 6>("bf16 16x16x32 + valu", 256); run<1, 6>("bf16 16x16x32 + valu", 512);
    return 0;
}
