// valu_after_mfma.hip — which vector instructions lose results in lanes 48..63 when they execute shortly behind the wave's own matrix
// instructions with a second wave on the SIMD?  (gat_epilogue_seq.hip reproduced the function-form k_gat3x error in isolation:
// 0.28 % of the sequences differ from their spaced-out twin, only in lanes 48..63, only with matrix instructions issued right before
// and two waves per SIMD.)  Kinds: 0 plain f32 chain, 1 packed-f32 chain (v_pk_mul/add), 2 transcendental + consumer with the
// compiler's one wait state, 3 packed with op_sel.  GAP = s_nop states between the matrix instructions and the chain.
//   hipcc --offload-arch=gfx950 -O3 -o valu_after_mfma valu_after_mfma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define W7 "s_nop 7\n\t"
#define CH0(N) "v_add_f32 v6, v6, v7\n\t" N "v_mul_f32 v7, v6, v10\n\t" N "v_add_f32 v6, v7, v11\n\t" N "v_mul_f32 v7, v6, v10\n\t" N "v_add_f32 v6, v6, v7\n\t" N "v_mul_f32 v7, v7, v11\n\t" N "v_add_f32 v6, v6, v7\n\t" N "v_add_f32 v7, v6, v10\n\t" N
#define CH1(N) "v_pk_mul_f32 v[186:187], v[6:7], v[10:11]\n\t" N "v_pk_add_f32 v[6:7], v[6:7], v[186:187]\n\t" N "v_pk_mul_f32 v[186:187], v[6:7], v[10:11]\n\t" N "v_pk_add_f32 v[6:7], v[186:187], v[6:7]\n\t" N \
               "v_pk_mul_f32 v[6:7], v[6:7], 0.5 op_sel_hi:[1,0]\n\t" N "v_pk_add_f32 v[6:7], v[6:7], v[10:11]\n\t" N
#define CH2(N) "v_exp_f32 v186, v6\n\t" N "s_nop 0\n\tv_add_f32 v7, v186, v7\n\t" N "v_rcp_f32 v187, v7\n\t" N "s_nop 0\n\tv_mul_f32 v6, v187, v186\n\t" N "v_exp_f32 v186, v6\n\t" N "s_nop 0\n\tv_add_f32 v6, v186, v6\n\t" N
#define CH3(N) "v_mov_b32 v186, v11\n\t" N "v_mov_b32 v187, v6\n\t" N "v_mov_b32 v6, v10\n\t" N "v_pk_mul_f32 v[186:187], v[186:187], v[10:11] op_sel:[0,1] op_sel_hi:[1,0]\n\t" N \
               "v_pk_mul_f32 v[188:189], v[6:7], v[10:11]\n\t" N "v_mov_b32 v6, v11\n\t" N "v_pk_add_f32 v[186:187], v[188:189], v[186:187]\n\t" N "v_pk_add_f32 v[6:7], v[6:7], v[186:187]\n\t" N
#define MV3 "v_mov_b32 v186, v11\n\t" "v_mov_b32 v187, v6\n\t" "v_mov_b32 v6, v10\n\t"
#define PM1 "v_pk_mul_f32 v[186:187], v[186:187], v[10:11] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
#define PM1N "v_pk_mul_f32 v[186:187], v[186:187], v[10:11]\n\t"
#define PM2 "v_pk_mul_f32 v[188:189], v[6:7], v[10:11]\n\t"
#define MV4 "v_mov_b32 v6, v11\n\t"
#define PA "v_pk_add_f32 v[186:187], v[188:189], v[186:187]\n\t" "v_pk_add_f32 v[6:7], v[6:7], v[186:187]\n\t"
#define G "s_nop 1\n\t"
// sub-variants of the failing chain (N = the spaced twin's filler; unused fillers are "")
#define CH4(N) "v_mov_b32 v186, v11\n\t" N "v_mov_b32 v187, v6\n\t" N "v_mov_b32 v6, v10\n\t" N G PM1 N PM2 N MV4 N "v_pk_add_f32 v[186:187], v[188:189], v[186:187]\n\t" N "v_pk_add_f32 v[6:7], v[6:7], v[186:187]\n\t" N   /* gap after the three moves */
#define CH5(N) "v_mov_b32 v186, v11\n\t" N "v_mov_b32 v187, v6\n\t" N "v_mov_b32 v6, v10\n\t" N PM1 N PM2 N G MV4 N "v_pk_add_f32 v[186:187], v[188:189], v[186:187]\n\t" N "v_pk_add_f32 v[6:7], v[6:7], v[186:187]\n\t" N   /* gap before the move that overwrites a source of the packed multiply */
#define CH6(N) "v_mov_b32 v186, v11\n\t" N "v_mov_b32 v187, v6\n\t" N "v_mov_b32 v6, v10\n\t" N PM1 N PM2 N MV4 N G "v_pk_add_f32 v[186:187], v[188:189], v[186:187]\n\t" N "v_pk_add_f32 v[6:7], v[6:7], v[186:187]\n\t" N   /* gap after that move */
#define CH7(N) "v_mov_b32 v186, v11\n\t" N "v_mov_b32 v187, v6\n\t" N "v_mov_b32 v6, v10\n\t" N PM1N N PM2 N MV4 N "v_pk_add_f32 v[186:187], v[188:189], v[186:187]\n\t" N "v_pk_add_f32 v[6:7], v[6:7], v[186:187]\n\t" N   /* no op_sel */
#define CH8(N) "v_mov_b32 v186, v11\n\t" N "v_mov_b32 v187, v6\n\t" N "v_mov_b32 v6, v10\n\t" N PM1 N PM2 N MV4 N "v_pk_add_f32 v[186:187], v[188:189], v[186:187]\n\t" N G "v_pk_add_f32 v[6:7], v[6:7], v[186:187]\n\t" N   /* gap between the two packed adds */
#define CHV(N, OPS) "v_mov_b32 v186, v11\n\t" N "v_mov_b32 v187, v6\n\t" N "v_mov_b32 v6, v10\n\t" N "v_pk_mul_f32 v[186:187], v[186:187], v[10:11] " OPS "\n\t" N PM2 N MV4 N "v_pk_add_f32 v[186:187], v[188:189], v[186:187]\n\t" N "v_pk_add_f32 v[6:7], v[6:7], v[186:187]\n\t" N
#define CH9(N) CHV(N, "op_sel_hi:[1,0]")
#define CH10(N) CHV(N, "op_sel:[0,1]")
#define CH11(N) CHV(N, "op_sel:[1,0] op_sel_hi:[0,1]")
#define CH12(N) CHV(N, "op_sel:[1,1] op_sel_hi:[0,0]")
#define CH13(N) "v_mov_b32 v186, v11\n\t" N "v_mov_b32 v187, v6\n\t" N "v_mov_b32 v6, v10\n\t" N "v_pk_add_f32 v[186:187], v[186:187], v[10:11] op_sel:[0,1] op_sel_hi:[1,0]\n\t" N PM2 N MV4 N "v_pk_add_f32 v[186:187], v[188:189], v[186:187]\n\t" N "v_pk_add_f32 v[6:7], v[6:7], v[186:187]\n\t" N
#define CH14(N) "v_mov_b32 v186, v11\n\t" N "v_mov_b32 v187, v6\n\t" N "v_mov_b32 v6, v10\n\t" N "v_pk_fma_f32 v[186:187], v[186:187], v[10:11], v[6:7] op_sel:[0,1,0] op_sel_hi:[1,0,1]\n\t" N PM2 N MV4 N "v_pk_add_f32 v[186:187], v[188:189], v[186:187]\n\t" N "v_pk_add_f32 v[6:7], v[6:7], v[186:187]\n\t" N
#define CH15(N) CHV(N, "op_sel:[1,0]")
#define CH16(N) "v_mov_b32 v186, v11\n\t" N "v_mov_b32 v187, v6\n\t" N "v_mov_b32 v6, v10\n\t" N "v_pk_fma_f32 v[186:187], v[186:187], v[10:11], v[6:7] op_sel:[1,0,0]\n\t" N PM2 N MV4 N "v_pk_add_f32 v[186:187], v[188:189], v[186:187]\n\t" N "v_pk_add_f32 v[6:7], v[6:7], v[186:187]\n\t" N
#define CH17(N) "v_mov_b32 v186, v11\n\t" N "v_mov_b32 v187, v6\n\t" N "v_mov_b32 v6, v10\n\t" N "v_pk_fma_f32 v[186:187], v[186:187], v[10:11], v[6:7] op_sel:[0,0,1]\n\t" N PM2 N MV4 N "v_pk_add_f32 v[186:187], v[188:189], v[186:187]\n\t" N "v_pk_add_f32 v[6:7], v[6:7], v[186:187]\n\t" N
#define CH18(N) "v_mov_b32 v186, v11\n\t" N "v_mov_b32 v187, v6\n\t" N "v_mov_b32 v6, v10\n\t" N "v_pk_fma_f32 v[186:187], v[10:11], v[186:187], v[6:7] op_sel:[1,0,0]\n\t" N PM2 N MV4 N "v_pk_add_f32 v[186:187], v[188:189], v[186:187]\n\t" N "v_pk_add_f32 v[6:7], v[6:7], v[186:187]\n\t" N
#define CLOB "v6", "v7", "v10", "v11", "v186", "v187", "v188", "v189"
#define LD "v_mov_b32 v6, %2\n\tv_mov_b32 v7, %3\n\tv_mov_b32 v10, %4\n\tv_mov_b32 v11, %5\n\t"
#define OUT "s_nop 7\n\tv_mov_b32 %0, v6\n\tv_mov_b32 %1, v7"

template <int KIND, int GAP, int NM>
__global__ void k(const float *in, unsigned *bad_rows, int iters)
{
    const int lane = threadIdx.x & 63, gid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned nbad = 0;
    for (int it = 0; it < iters; it++) {
        const float *p = in + ((size_t)(it & 63) * gridDim.x * blockDim.x + gid) * 4;
        const float a = p[0], b = p[1], c = p[2], d = p[3];
        float f0, f1, s0, s1;
#define BODY(CH)                                                                                                                                         \
        asm volatile(LD "s_nop 7\n\ts_nop 7" :: "v"(a), "v"(b), "v"(a), "v"(b), "v"(c), "v"(d) : CLOB);                                                    \
        _Pragma("unroll") for (int m = 0; m < NM; m++) asm volatile("v_mfma_f32_16x16x32_f16 v[200:203], v[204:207], v[208:211], v[200:203]" ::: "v200", "v201", "v202", "v203"); \
        _Pragma("unroll") for (int g = 0; g < GAP; g++) asm volatile("s_nop 0");                                                                          \
        asm volatile(CH("") OUT : "=v"(f0), "=v"(f1) :: CLOB);                                                                                            \
        asm volatile(LD "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t" CH(W7) OUT : "=v"(s0), "=v"(s1) : "v"(a), "v"(b), "v"(c), "v"(d) : CLOB);
        if (KIND == 0) { BODY(CH0) } else if (KIND == 1) { BODY(CH1) } else if (KIND == 2) { BODY(CH2) } else if (KIND == 3) { BODY(CH3) } else if (KIND == 4) { BODY(CH4) } else if (KIND == 5) { BODY(CH5) } else if (KIND == 6) { BODY(CH6) } else if (KIND == 7) { BODY(CH7) } else if (KIND == 8) { BODY(CH8) } else if (KIND == 9) { BODY(CH9) } else if (KIND == 10) { BODY(CH10) } else if (KIND == 11) { BODY(CH11) } else if (KIND == 12) { BODY(CH12) } else if (KIND == 13) { BODY(CH13) } else if (KIND == 14) { BODY(CH14) } else if (KIND == 15) { BODY(CH15) } else if (KIND == 16) { BODY(CH16) } else if (KIND == 17) { BODY(CH17) } else { BODY(CH18) }
        nbad += (__float_as_uint(f0) != __float_as_uint(s0)) || (__float_as_uint(f1) != __float_as_uint(s1));
    }
    if (nbad) atomicAdd(&bad_rows[lane >> 4], nbad);
}
template <int KIND, int GAP, int NM>
static int run(const float *d_in, unsigned *d_bad, int wps, int iters)
{
    CHK(hipMemset(d_bad, 0, 16));
    hipLaunchKernelGGL((k<KIND, GAP, NM>), dim3(256), dim3(256 * wps), 0, 0, d_in, d_bad, iters);
    CHK(hipDeviceSynchronize());
    unsigned h[4];
    CHK(hipMemcpy(h, d_bad, 16, hipMemcpyDeviceToHost));
    const char *nm[19] = {"plain f32 chain     ", "packed f32 chain    ", "transcendental chain", "packed with op_sel  ", "  + gap after moves ", "  + gap before WAR mov", "  + gap after WAR mov", "  without op_sel    ", "  + gap between adds", "mul op_sel_hi:[1,0] (src1 low twice)", "mul op_sel:[0,1] (src1 high twice)", "mul src0 halves swapped", "mul both sources swapped", "add src1 halves swapped", "fma src1 halves swapped", "mul op_sel:[1,0] (src0 high twice)", "fma op_sel:[1,0,0] (src0 high twice)", "fma op_sel:[0,0,1] (src2 high twice)", "fma op_sel:[1,0,0], src0 = old pair"};
    printf("%s  %d matrix instr before, gap %2d, %d wave(s)/SIMD: differ in lane row [0-15] %u  [16-31] %u  [32-47] %u  [48-63] %u  (of %lld per row)\n", nm[KIND], NM, GAP, wps, h[0], h[1], h[2], h[3],
           (long long)iters * 256 * 4 * wps * 16);
    return 0;
}
int main()
{
    const size_t nthreads = 256 * 512, n = nthreads * 64 * 4;
    std::vector<float> h(n);
    srand(5);
    for (size_t i = 0; i < n; i++) h[i] = 2.f * rand() / RAND_MAX - 1.f;
    float *d_in; unsigned *d_bad;
    CHK(hipMalloc(&d_in, n * 4)); CHK(hipMemcpy(d_in, h.data(), n * 4, hipMemcpyHostToDevice)); CHK(hipMalloc(&d_bad, 16));
    const int it = 8000;
#define ALLK(GAP, NM, W) run<0, GAP, NM>(d_in, d_bad, W, it); run<1, GAP, NM>(d_in, d_bad, W, it); run<2, GAP, NM>(d_in, d_bad, W, it); run<3, GAP, NM>(d_in, d_bad, W, it);
    run<3, 0, 2>(d_in, d_bad, 2, it); run<7, 0, 2>(d_in, d_bad, 2, it); run<9, 0, 2>(d_in, d_bad, 2, it); run<10, 0, 2>(d_in, d_bad, 2, it); run<11, 0, 2>(d_in, d_bad, 2, it); run<12, 0, 2>(d_in, d_bad, 2, it);
    run<13, 0, 2>(d_in, d_bad, 2, it); run<14, 0, 2>(d_in, d_bad, 2, it); run<3, 0, 2>(d_in, d_bad, 1, it); run<3, 0, 0>(d_in, d_bad, 2, it);
    run<15, 0, 2>(d_in, d_bad, 2, it); run<16, 0, 2>(d_in, d_bad, 2, it); run<17, 0, 2>(d_in, d_bad, 2, it); run<18, 0, 2>(d_in, d_bad, 2, it);
    return 0;
}
