// Micro-benchmark (diagnostic, not part of the product): how many vector / LDS instructions of the SAME wave hide behind a
// v_mfma_f32_32x32x16_f16 (and 16x16x32) on gfx950, with one and with two waves per SIMD.  Every instruction is an asm volatile
// statement, so the stream is exactly as written: MFMA, then NV fillers, repeated.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_fill mfma_fill.hip ; run: ./mfma_fill
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// FILL: 0 v_fma_f32 (independent registers), 1 v_pk_fma_f32, 2 v_cvt_pk_f16_f32 (v_cvt_pkrtz), 3 ds_read_b128, 4 ds_write_b64, 5 v_max_f32,
//       6 mix of the split sequence (fma, max, cvt, fma_mix, cvt, add, fma) on independent registers
template <int SHAPE, int NV, int FILL, int NACC>
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *cyc, int iters, float seed)
{
    __shared__ __align__(16) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = seed + i;
    f32x16 acc[NACC];
    f32x4 acc4[NACC];
    for (int t = 0; t < NACC; t++) { for (int i = 0; i < 16; i++) acc[t][i] = 0.f; acc4[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    h8 ah, bh;
    for (int i = 0; i < 8; i++) { ah[i] = (_Float16)(0.001f * (threadIdx.x + i)); bh[i] = (_Float16)(0.002f * (seed + i)); }
    float v[16];
    for (int i = 0; i < 16; i++) v[i] = seed + i + threadIdx.x;
    f32x2 w[8];
    for (int i = 0; i < 8; i++) w[i] = f32x2{seed + i, seed - i};
    f32x4 ld[4];
    for (int i = 0; i < 4; i++) ld[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned la = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 1024;
    unsigned hp[8];
    for (int i = 0; i < 8; i++) hp[i] = i;
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int t = 0; t < 8; t++) {
            if (SHAPE == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[t % NACC]) : "v"(ah), "v"(bh));
            else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc4[t % NACC]) : "v"(ah), "v"(bh));
#pragma unroll
            for (int j = 0; j < NV; j++) {
                const int r = (t * NV + j) & 15;
                if (FILL == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[r]) : "v"(seed));
                else if (FILL == 1) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(w[r & 7]));
                else if (FILL == 2) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(hp[r & 7]) : "v"(v[r]), "v"(v[(r + 1) & 15]));
                else if (FILL == 3) { asm volatile("ds_read_b128 %0, %1" : "=v"(ld[r & 3]) : "v"(la)); if ((j & 3) == 3 || j == NV - 1) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory"); }
                else if (FILL == 4) asm volatile("ds_write_b64 %0, %1" :: "v"(la), "v"(w[r & 7]) : "memory");
                else if (FILL == 5) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[r]) : "v"(seed));
                else {
                    switch ((t * NV + j) % 7) {
                    case 0: asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[r]) : "v"(seed)); break;
                    case 1: asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[r]) : "v"(seed)); break;
                    case 2: asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(hp[r & 7]) : "v"(v[r]), "v"(v[(r + 1) & 15])); break;
                    case 3: asm volatile("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(v[r]) : "v"(hp[r & 7])); break;
                    case 4: asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(hp[(r + 3) & 7]) : "v"(v[r]), "v"(v[(r + 5) & 15])); break;
                    case 5: asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[r]) : "v"(seed)); break;
                    default: asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(v[r]) : "v"(seed)); break;
                    }
                }
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0.f;
    for (int t = 0; t < NACC; t++) { for (int i = 0; i < 16; i++) s += acc[t][i]; s += acc4[t][0] + acc4[t][3]; }
    for (int i = 0; i < 16; i++) s += v[i];
    for (int i = 0; i < 8; i++) s += w[i][0] + w[i][1] + (float)hp[i];
    for (int i = 0; i < 4; i++) s += ld[i][0] + ld[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

static const char *FN[] = {"v_fma_f32", "v_pk_fma_f32", "v_cvt_pkrtz", "ds_read_b128", "ds_write_b64", "v_max_f32", "split mix"};
template <int SHAPE, int NV, int FILL, int NACC>
static void run(int threads)
{
    float *out; unsigned long long *cyc, h;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SHAPE, NV, FILL, NACC>), dim3(256), dim3(threads), 0, 0, out, cyc, iters, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<SHAPE, NV, FILL, NACC>), dim3(256), dim3(threads), 0, 0, out, cyc, iters, 1.0f);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const int wps = threads / 256;
    const double ns_per = ms * 1e6 / (iters * 8.0 * wps);         // wall ns per MFMA per SIMD
    printf("%s acc %d  %-13s x%d  waves/SIMD %d : %6.1f ticks/MFMA/wave  | %6.2f ns per MFMA per SIMD (pure MFMA = %s)\n", SHAPE ? "16x16x32" : "32x32x16", NACC, FN[FILL], NV, wps,
           (double)h / (iters * 8.0), ns_per, SHAPE ? "~6.7" : "~13.3");
    hipFree(out); hipFree(cyc);
}

template <int SHAPE, int FILL>
static void sweep()
{
    run<SHAPE, 0, FILL, 1>(256); run<SHAPE, 2, FILL, 1>(256); run<SHAPE, 4, FILL, 1>(256); run<SHAPE, 5, FILL, 1>(256); run<SHAPE, 6, FILL, 1>(256);
    run<SHAPE, 8, FILL, 1>(256); run<SHAPE, 12, FILL, 1>(256);
    run<SHAPE, 4, FILL, 1>(512); run<SHAPE, 6, FILL, 1>(512); run<SHAPE, 8, FILL, 1>(512); run<SHAPE, 12, FILL, 1>(512);
}
int main()
{
    sweep<0, 0>(); sweep<0, 6>(); sweep<0, 1>(); sweep<0, 3>(); sweep<0, 4>();
    run<0, 6, 0, 2>(256); run<0, 8, 0, 2>(256); run<0, 8, 6, 2>(256);
    sweep<1, 0>(); sweep<1, 6>();
    return 0;
}
