// gat_epilogue_seq.hip — the exact vector-instruction sequence hipcc emitted for the last-pass epilogue of the function-form k_gat3x
// (attention weights from two exponentials and a full-precision reciprocal, then the first two column blocks' node means through
// v_pk_mul_f32 / v_pk_add_f32 with op_sel), executed (a) as emitted and (b) with s_nop 7 between all instructions, on the same random
// inputs; the results must be bit-identical.  Differences are counted per 16-lane row.  Round-4 bisection: the function form's wrong
// node values were one 16-column block of the rows held by lanes 48..63, exactly 0.75 x the right value.
//   hipcc --offload-arch=gfx950 -O3 -o gat_epilogue_seq gat_epilogue_seq.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#define SEQ6(N1, N2, N3, N4, N5, N6)                                                                                    \
    "v_exp_f32_e32 v122, v122\n\t" N1 "v_exp_f32_e32 v123, v49\n\t" N1 "s_nop 0\n\t"                                       \
    "v_add_f32_e32 v49, v122, v123\n\t" N2 "v_div_scale_f32 v186, s[0:1], v49, v49, 1.0\n\t" N2                            \
    "v_rcp_f32_e32 v187, v186\n\ts_nop 0\n\t" N2 "v_fma_f32 v188, -v186, v187, 1.0\n\t" N2 "v_fmac_f32_e32 v187, v188, v187\n\t" N2     \
    "v_div_scale_f32 v188, vcc, 1.0, v49, 1.0\n\t" N2 "v_mul_f32_e32 v189, v188, v187\n\t" N2                              \
    "v_fma_f32 v190, -v186, v189, v188\n\t" N2 "v_fmac_f32_e32 v189, v190, v187\n\t" N2 "v_fma_f32 v186, -v186, v189, v188\n\t" N2 \
    "v_div_fmas_f32 v186, v186, v187, v189\n\t" N3 "v_div_fixup_f32 v186, v186, v49, 1.0\n\t" N3                           \
    "v_pk_mul_f32 v[122:123], v[122:123], v[186:187] op_sel_hi:[1,0]\n\t" N4 "s_nop 0\n\t"                                \
    "v_mul_f32_e32 v2, v2, v122\n\t" N4 "v_mul_f32_e32 v49, v3, v123\n\t" N4 "v_add_f32_e32 v2, v2, v49\n\t" N4             \
    "v_mov_b32_e32 v186, v11\n\t" N5 "v_mov_b32_e32 v187, v6\n\t" N5 "v_mov_b32_e32 v6, v10\n\t" N5                         \
    "v_pk_mul_f32 v[186:187], v[186:187], v[122:123] op_sel:[0,1] op_sel_hi:[1,0]\n\t" N5                                 \
    "v_pk_mul_f32 v[188:189], v[6:7], v[122:123]\n\t" N5 "v_mov_b32_e32 v6, v11\n\t" N5                                    \
    "v_pk_add_f32 v[186:187], v[188:189], v[186:187]\n\t" N6 "v_add_f32_e32 v2, v3, v2\n\t" N6                             \
    "v_pk_add_f32 v[6:7], v[6:7], v[186:187]\n\t" N6 "v_mul_f32_e32 v2, 0.5, v2\n\t" N6                                    \
    "v_pk_mul_f32 v[6:7], v[6:7], 0.5 op_sel_hi:[1,0]\n\t" N6
#define W7 "s_nop 7\n\t"
#define CLOB "v2", "v3", "v6", "v7", "v10", "v11", "v49", "v122", "v123", "v186", "v187", "v188", "v189", "v190", "vcc", "s0", "s1"
#define LOADS "v_mov_b32 v122, %3\n\tv_mov_b32 v49, %4\n\tv_mov_b32 v2, %5\n\tv_mov_b32 v3, %6\n\tv_mov_b32 v10, %7\n\tv_mov_b32 v11, %8\n\tv_mov_b32 v6, %9\n\tv_mov_b32 v7, %10\n\ts_nop 7\n\t"
#define OUTS "s_nop 7\n\tv_mov_b32 %0, v2\n\tv_mov_b32 %1, v7\n\tv_mov_b32 %2, v6"

template <int PRE, int VAR>      // VAR: which segments of the first sequence are spaced too (bit i = segment i+1); PRE: matrix instructions issued right before the sequence (the epilogue follows the pass's last products)
__global__ void k(const float *in, unsigned *bad_rows, int iters)
{
    const int lane = threadIdx.x & 63;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned nbad = 0;
    for (int it = 0; it < iters; it++) {
        const float *p = in + ((size_t)(it & 63) * gridDim.x * blockDim.x + gid) * 8;
        const float a0 = p[0], a1 = p[1], z00 = p[2], z01 = p[3], zA0 = p[4], zA1 = p[5], zB0 = p[6], zB1 = p[7];
        float f0, f1, f2, s0, s1, s2;
        if (PRE) asm volatile("v_mfma_f32_16x16x32_f16 v[200:203], v[204:207], v[208:211], v[200:203]\n\tv_mfma_f32_16x16x32_f16 v[212:215], v[204:207], v[208:211], v[212:215]"
                              ::: "v200", "v201", "v202", "v203", "v212", "v213", "v214", "v215");
#define RUNVAR(V, A, B, C, D, E, F) if (VAR == V) asm volatile(LOADS SEQ6(A, B, C, D, E, F) OUTS : "=v"(f0), "=v"(f1), "=v"(f2) : "v"(a0), "v"(a1), "v"(z00), "v"(z01), "v"(zA0), "v"(zA1), "v"(zB0), "v"(zB1) : CLOB)
        RUNVAR(0, "", "", "", "", "", ""); RUNVAR(1, W7, "", "", "", "", ""); RUNVAR(2, "", W7, "", "", "", ""); RUNVAR(3, "", "", W7, "", "", "");
        RUNVAR(4, "", "", "", W7, "", ""); RUNVAR(5, "", "", "", "", W7, ""); RUNVAR(6, "", "", "", "", "", W7);
        asm volatile(LOADS SEQ6(W7, W7, W7, W7, W7, W7) OUTS : "=v"(s0), "=v"(s1), "=v"(s2) : "v"(a0), "v"(a1), "v"(z00), "v"(z01), "v"(zA0), "v"(zA1), "v"(zB0), "v"(zB1) : CLOB);
        nbad += (__float_as_uint(f0) != __float_as_uint(s0)) || (__float_as_uint(f1) != __float_as_uint(s1)) || (__float_as_uint(f2) != __float_as_uint(s2));
    }
    if (nbad) atomicAdd(&bad_rows[lane >> 4], nbad);
}

template <int PRE, int VAR>
static int run(const float *d_in, unsigned *d_bad, int wps, int iters)
{
    CHK(hipMemset(d_bad, 0, 16));
    hipLaunchKernelGGL((k<PRE, VAR>), dim3(256), dim3(256 * wps), 0, 0, d_in, d_bad, iters);
    CHK(hipDeviceSynchronize());
    unsigned h[4];
    CHK(hipMemcpy(h, d_bad, 16, hipMemcpyDeviceToHost));
    printf("variant %d (segment spaced; 0 = none) vs all spaced  (%d matrix instructions before, %d wave(s)/SIMD): results differ in lane row [0-15] %u  [16-31] %u  [32-47] %u  [48-63] %u  (of %lld per row)\n",
           VAR, 2 * PRE, wps, h[0], h[1], h[2], h[3], (long long)iters * 256 * 4 * wps * 16);
    return 0;
}
int main()
{
    const size_t nthreads = 256 * 512, n = nthreads * 64 * 8;
    std::vector<float> h(n);
    srand(3);
    for (size_t i = 0; i < n; i += 8) {
        const int sat = rand() % 3;
        const float d = sat == 0 ? -200.f * rand() / RAND_MAX : sat == 1 ? -5.f * rand() / RAND_MAX : 0.f;   // exponent argument of the smaller logit
        const bool first = rand() & 1;
        h[i] = first ? 0.f : d; h[i + 1] = first ? d : 0.f;
        for (int j = 2; j < 8; j++) h[i + j] = 4.f * rand() / RAND_MAX - 2.f;
    }
    float *d_in; unsigned *d_bad;
    CHK(hipMalloc(&d_in, n * 4)); CHK(hipMemcpy(d_in, h.data(), n * 4, hipMemcpyHostToDevice)); CHK(hipMalloc(&d_bad, 16));
    run<0, 0>(d_in, d_bad, 2, 8000); run<1, 0>(d_in, d_bad, 1, 8000); run<1, 0>(d_in, d_bad, 2, 8000);
    run<1, 1>(d_in, d_bad, 2, 8000); run<1, 2>(d_in, d_bad, 2, 8000); run<1, 3>(d_in, d_bad, 2, 8000); run<1, 4>(d_in, d_bad, 2, 8000); run<1, 5>(d_in, d_bad, 2, 8000); run<1, 6>(d_in, d_bad, 2, 8000);
    return 0;
}
