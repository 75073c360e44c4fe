// mfma_overlap.hip — may the destination of v_mfma_f32_16x16x32_f16 overlap its A or B source registers on gfx950?
// (Round-4 root cause candidate for the function-form k_gat3x miscomputation: hipcc allocated `v_mfma_f32_16x16x32_f16 v[186:189],
// v[214:217], v[186:189], v[202:205]` — destination = SrcB — for the one matrix instruction whose weight fragment died there; the wrong
// values sat in lanes 48..63, the rows of the LAST write-back pass.  LLVM marks the destination early-clobber only for results wider
// than 128 bits.)  A = B = ones, C = 0: every element of D must be 32.  Variants: D = fresh registers, D = SrcB, D = SrcA, D = SrcC.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_overlap mfma_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int MODE, int PREV>      // MODE 0: D fresh, 1: D = SrcB, 2: D = SrcA, 3: D = SrcC; PREV: independent matrix instructions issued right before (pipe busy)
__global__ void k(unsigned *bad_rows, int iters)
{
    const int lane = threadIdx.x & 63;
    unsigned nbad = 0;
    for (int it = 0; it < iters; it++) {
        float r0, r1, r2, r3;
        asm volatile(
            "v_mov_b32 v32, 0x3c003c00\n\tv_mov_b32 v33, 0x3c003c00\n\tv_mov_b32 v34, 0x3c003c00\n\tv_mov_b32 v35, 0x3c003c00\n\t"
            "v_mov_b32 v36, 0x3c003c00\n\tv_mov_b32 v37, 0x3c003c00\n\tv_mov_b32 v38, 0x3c003c00\n\tv_mov_b32 v39, 0x3c003c00\n\t"
            "v_mov_b32 v40, 0\n\tv_mov_b32 v41, 0\n\tv_mov_b32 v42, 0\n\tv_mov_b32 v43, 0\n\t"
            "v_mov_b32 v44, 0\n\tv_mov_b32 v45, 0\n\tv_mov_b32 v46, 0\n\tv_mov_b32 v47, 0\n\t"
            "v_mov_b32 v48, 0x3c003c00\n\tv_mov_b32 v49, 0x3c003c00\n\tv_mov_b32 v50, 0x3c003c00\n\tv_mov_b32 v51, 0x3c003c00\n\t"
            "v_mov_b32 v52, 0\n\tv_mov_b32 v53, 0\n\tv_mov_b32 v54, 0\n\tv_mov_b32 v55, 0\n\t"
            "s_nop 7\n\ts_nop 7" ::: "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
        if (PREV >= 1) asm volatile("v_mfma_f32_16x16x32_f16 v[52:55], v[48:51], v[48:51], v[52:55]" ::: "v52", "v53", "v54", "v55");
        if (PREV >= 2) asm volatile("v_mfma_f32_16x16x32_f16 v[52:55], v[48:51], v[48:51], v[52:55]" ::: "v52", "v53", "v54", "v55");
        if (MODE == 0) asm volatile("v_mfma_f32_16x16x32_f16 v[44:47], v[32:35], v[36:39], v[40:43]\n\ts_nop 7\n\ts_nop 7\n\tv_mov_b32 %0, v44\n\tv_mov_b32 %1, v45\n\tv_mov_b32 %2, v46\n\tv_mov_b32 %3, v47"
                                    : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) :: "v44", "v45", "v46", "v47");
        if (MODE == 1) asm volatile("v_mfma_f32_16x16x32_f16 v[36:39], v[32:35], v[36:39], v[40:43]\n\ts_nop 7\n\ts_nop 7\n\tv_mov_b32 %0, v36\n\tv_mov_b32 %1, v37\n\tv_mov_b32 %2, v38\n\tv_mov_b32 %3, v39"
                                    : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) :: "v36", "v37", "v38", "v39");
        if (MODE == 2) asm volatile("v_mfma_f32_16x16x32_f16 v[32:35], v[32:35], v[36:39], v[40:43]\n\ts_nop 7\n\ts_nop 7\n\tv_mov_b32 %0, v32\n\tv_mov_b32 %1, v33\n\tv_mov_b32 %2, v34\n\tv_mov_b32 %3, v35"
                                    : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) :: "v32", "v33", "v34", "v35");
        if (MODE == 3) asm volatile("v_mfma_f32_16x16x32_f16 v[40:43], v[32:35], v[36:39], v[40:43]\n\ts_nop 7\n\ts_nop 7\n\tv_mov_b32 %0, v40\n\tv_mov_b32 %1, v41\n\tv_mov_b32 %2, v42\n\tv_mov_b32 %3, v43"
                                    : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) :: "v40", "v41", "v42", "v43");
        nbad += (r0 != 32.f) || (r1 != 32.f) || (r2 != 32.f) || (r3 != 32.f);
    }
    if (nbad) atomicAdd(&bad_rows[lane >> 4], nbad);
}

template <int MODE, int PREV>
static int run(unsigned *d_bad, int waves_per_simd, int iters)
{
    CHK(hipMemset(d_bad, 0, 16));
    hipLaunchKernelGGL((k<MODE, PREV>), dim3(256), dim3(256 * waves_per_simd), 0, 0, d_bad, iters);
    CHK(hipDeviceSynchronize());
    unsigned h[4];
    CHK(hipMemcpy(h, d_bad, 16, hipMemcpyDeviceToHost));
    const char *names[4] = {"D fresh ", "D = SrcB", "D = SrcA", "D = SrcC"};
    printf("v_mfma_f32_16x16x32_f16  %s  %d matrix instruction(s) right before  %d wave(s)/SIMD : wrong results by lane row [0-15] %u  [16-31] %u  [32-47] %u  [48-63] %u   (of %lld per row)\n",
           names[MODE], PREV, waves_per_simd, h[0], h[1], h[2], h[3], (long long)iters * 256 * 4 * waves_per_simd * 16);
    return 0;
}
int main()
{
    unsigned *d_bad;
    CHK(hipMalloc(&d_bad, 16));
    const int iters = 20000;
    run<0, 0>(d_bad, 1, iters); run<3, 0>(d_bad, 1, iters); run<1, 0>(d_bad, 1, iters); run<2, 0>(d_bad, 1, iters);
    run<1, 1>(d_bad, 1, iters); run<2, 1>(d_bad, 1, iters); run<1, 2>(d_bad, 1, iters); run<2, 2>(d_bad, 1, iters);
    run<0, 2>(d_bad, 2, iters); run<3, 2>(d_bad, 2, iters); run<1, 0>(d_bad, 2, iters); run<2, 0>(d_bad, 2, iters); run<1, 2>(d_bad, 2, iters); run<2, 2>(d_bad, 2, iters);
    return 0;
}
