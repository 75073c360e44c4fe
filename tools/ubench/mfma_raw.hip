// mfma_raw.hip — how many wait states does gfx950 need between v_mfma_f32_16x16x32_f16 / v_mfma_f32_32x32x16_f16 and a VALU
// instruction that READS the result?  (Round-4 bisection of the k_gat3x function-form miscomputation: wrong values only in tile rows
// 12..15 = lanes 48..63 = the LAST write-back pass of the matrix instruction, only in the accumulator read first after the matrix
// chain, at a distance of 8 wait states chosen by hipcc's hazard recogniser.)
// A = B = all ones, C = 0: every element of D is 32 (16x16x32) / 16 (32x32x16); D's registers hold a sentinel beforehand.  K wait
// states (K-1 as s_nop, or K independent VALU instructions) separate the matrix instruction from v_mov reads of D; lanes that read
// the sentinel are counted per 16-lane row.  One wave per SIMD (back-to-back issue) and two (arbitration bubbles).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_raw mfma_raw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int SHAPE, int K, int FILL, int SRCC, int CONS = 0>     // CONS 0: v_mov_b32 reads the result, 1: v_pk_mul_f32 (by 1.0) does; SHAPE 0: 16x16x32 (4 result registers), 1: 32x32x16 (16); FILL 0: s_nop, 1: VALU; SRCC 0: C = D registers, 1: C = other registers
__global__ void k(unsigned *bad_rows, int iters)
{
    const int lane = threadIdx.x & 63;
    unsigned nbad = 0;
    unsigned filler = lane;
    for (int it = 0; it < iters; it++) {
        float r0, r1, r2, r3;
        // operands: v[32:35] = A (8 x f16 1.0), v[36:39] = B, v[40:55] = D (sentinel), v[56:71] = zero C
        asm volatile(
            "v_mov_b32 v32, 0x3c003c00\n\tv_mov_b32 v33, 0x3c003c00\n\tv_mov_b32 v34, 0x3c003c00\n\tv_mov_b32 v35, 0x3c003c00\n\t"
            "v_mov_b32 v36, 0x3c003c00\n\tv_mov_b32 v37, 0x3c003c00\n\tv_mov_b32 v38, 0x3c003c00\n\tv_mov_b32 v39, 0x3c003c00\n\t"
            "v_mov_b32 v40, 0x4640e400\n\tv_mov_b32 v41, 0x4640e400\n\tv_mov_b32 v42, 0x4640e400\n\tv_mov_b32 v43, 0x4640e400\n\t"
            "v_mov_b32 v52, 0x4640e400\n\tv_mov_b32 v53, 0x4640e400\n\tv_mov_b32 v54, 0x4640e400\n\tv_mov_b32 v55, 0x4640e400\n\t"
            "v_mov_b32 v56, 0\n\tv_mov_b32 v57, 0\n\tv_mov_b32 v58, 0\n\tv_mov_b32 v59, 0\n\t"
            "v_mov_b32 v60, 0\n\tv_mov_b32 v61, 0\n\tv_mov_b32 v62, 0\n\tv_mov_b32 v63, 0\n\t"
            "v_mov_b32 v64, 0\n\tv_mov_b32 v65, 0\n\tv_mov_b32 v66, 0\n\tv_mov_b32 v67, 0\n\t"
            "v_mov_b32 v68, 0\n\tv_mov_b32 v69, 0\n\tv_mov_b32 v70, 0\n\tv_mov_b32 v71, 0\n\t"
            "v_mov_b32 v78, 1.0\n\tv_mov_b32 v79, 1.0\n\t"
            "s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "v78", "v79", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48",
            "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71");
        if (SHAPE == 0) {
            if (SRCC) asm volatile("v_mfma_f32_16x16x32_f16 v[40:43], v[32:35], v[36:39], v[56:59]" ::: "v40", "v41", "v42", "v43");
            else { asm volatile("v_mov_b32 v40, 0\n\tv_mov_b32 v41, 0\n\tv_mov_b32 v42, 0\n\tv_mov_b32 v43, 0\n\ts_nop 7\n\tv_mfma_f32_16x16x32_f16 v[40:43], v[32:35], v[36:39], v[40:43]" ::: "v40", "v41", "v42", "v43"); }
        } else {
            if (SRCC) asm volatile("v_mfma_f32_32x32x16_f16 v[40:55], v[32:35], v[36:39], v[56:71]" ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
            else asm volatile("v_mfma_f32_32x32x16_f16 v[40:55], v[32:35], v[36:39], 0" ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
        }
        if (FILL == 2) {          // K independent single-width multiplies, back to back (what the failing schedule had between the two)
            asm volatile(".rept %c0\n\tv_mul_f32_e64 v72, s2, v73\n\t.endr" :: "n"(K) : "v72");
        } else if (FILL == 3) {   // K independent two-wide multiplies
            asm volatile(".rept %c0\n\tv_pk_mul_f32 v[72:73], s[2:3], v[74:75]\n\t.endr" :: "n"(K) : "v72", "v73");
        } else {
#pragma unroll
        for (int g = 0; g < K; g++) {
            if (FILL) asm volatile("v_add_u32 %0, %0, %0" : "+v"(filler));
            else asm volatile("s_nop 0");
        }
        }
        // the first reads: the LAST registers of the result (what the failing schedule read first), then the first ones
        if (SHAPE == 0 && CONS == 1) {
            asm volatile("v_pk_mul_f32 v[76:77], v[42:43], v[78:79]\n\tv_pk_mul_f32 v[80:81], v[40:41], v[78:79]\n\ts_nop 7\n\tv_mov_b32 %0, v76\n\tv_mov_b32 %1, v77\n\tv_mov_b32 %2, v80\n\tv_mov_b32 %3, v81"
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) :: "v40", "v41", "v42", "v43", "v76", "v77", "v80", "v81");
        } else if (SHAPE == 0) asm volatile("v_mov_b32 %0, v42\n\tv_mov_b32 %1, v43\n\tv_mov_b32 %2, v40\n\tv_mov_b32 %3, v41" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) :: "v40", "v41", "v42", "v43");
        else asm volatile("v_mov_b32 %0, v54\n\tv_mov_b32 %1, v55\n\tv_mov_b32 %2, v40\n\tv_mov_b32 %3, v41" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) :: "v40", "v41", "v54", "v55");
        const float want = SHAPE == 0 ? 32.f : 16.f;
        nbad += (r0 != want) || (r1 != want) || (r2 != want) || (r3 != want);
        if (filler == 0x12345u) nbad += 1000;
    }
    if (nbad) atomicAdd(&bad_rows[lane >> 4], nbad);
}

template <int SHAPE, int K, int FILL, int SRCC, int CONS = 0>
static int run(unsigned *d_bad, int waves_per_simd, int iters)
{
    CHK(hipMemset(d_bad, 0, 16));
    hipLaunchKernelGGL((k<SHAPE, K, FILL, SRCC, CONS>), dim3(256), dim3(256 * waves_per_simd), 0, 0, d_bad, iters);
    CHK(hipDeviceSynchronize());
    unsigned h[4];
    CHK(hipMemcpy(h, d_bad, 16, hipMemcpyDeviceToHost));
    printf("%s  read by %s  C=%s  gap %2d %s  %d wave(s)/SIMD : stale reads by lane row [0-15] %u  [16-31] %u  [32-47] %u  [48-63] %u\n", SHAPE ? "32x32x16" : "16x16x32", CONS ? "v_pk_mul" : "v_mov   ", SRCC ? "other" : "D    ", K,
           FILL == 3 ? "pkmul" : FILL == 2 ? "mul  " : FILL ? "VALU " : "s_nop", waves_per_simd, h[0], h[1], h[2], h[3]);
    return 0;
}
#define SWEEP(SHAPE, FILL, SRCC, W) \
    run<SHAPE, 1, FILL, SRCC>(d_bad, W, iters); run<SHAPE, 2, FILL, SRCC>(d_bad, W, iters); run<SHAPE, 3, FILL, SRCC>(d_bad, W, iters); run<SHAPE, 4, FILL, SRCC>(d_bad, W, iters); run<SHAPE, 5, FILL, SRCC>(d_bad, W, iters); \
    run<SHAPE, 6, FILL, SRCC>(d_bad, W, iters); run<SHAPE, 7, FILL, SRCC>(d_bad, W, iters); run<SHAPE, 8, FILL, SRCC>(d_bad, W, iters); run<SHAPE, 9, FILL, SRCC>(d_bad, W, iters); \
    run<SHAPE, 10, FILL, SRCC>(d_bad, W, iters); run<SHAPE, 12, FILL, SRCC>(d_bad, W, iters); run<SHAPE, 16, FILL, SRCC>(d_bad, W, iters); run<SHAPE, 20, FILL, SRCC>(d_bad, W, iters);
int main()
{
    unsigned *d_bad;
    CHK(hipMalloc(&d_bad, 16));
    const int iters = 20000;
    SWEEP(0, 0, 1, 1) SWEEP(0, 2, 1, 1) SWEEP(0, 2, 1, 2) SWEEP(0, 3, 1, 1)
    SWEEP(1, 0, 1, 1) SWEEP(1, 2, 1, 1)
#define SWEEPC(FILL, W) run<0, 4, FILL, 1, 1>(d_bad, W, iters); run<0, 5, FILL, 1, 1>(d_bad, W, iters); run<0, 6, FILL, 1, 1>(d_bad, W, iters); run<0, 7, FILL, 1, 1>(d_bad, W, iters); \
    run<0, 8, FILL, 1, 1>(d_bad, W, iters); run<0, 9, FILL, 1, 1>(d_bad, W, iters); run<0, 10, FILL, 1, 1>(d_bad, W, iters); run<0, 12, FILL, 1, 1>(d_bad, W, iters); run<0, 16, FILL, 1, 1>(d_bad, W, iters);
    SWEEPC(0, 1) SWEEPC(0, 2) SWEEPC(2, 1) SWEEPC(2, 2) SWEEPC(3, 2)
    return 0;
}
