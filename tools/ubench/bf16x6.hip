// Accuracy + operand-layout check: f32 GEMM tile via 3-way bf16 split (6 products) on v_mfma_f32_16x16x32_bf16 and via 2-way
// f16 split (3 products) on v_mfma_f32_16x16x32_f16 (gfx950), against f64 on the host
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ inline void split3(float x, __bf16 &h, __bf16 &m, __bf16 &l)
{
    h = (__bf16)x; float r = x - (float)h; m = (__bf16)r; r = r - (float)m; l = (__bf16)r;
}
// X [16][128], W [128 out][128 in] -> out[16][16 cols c0..] = X @ W^T ; mode 0: f32 mfma, 1: bf16x6, 2: bf16x3, 3: bf16x9,
// 4: f16x3 = 2-way f16 split (round to nearest), weights pre-scaled by `wscale` (a power of two), 3 products (k_gin_res)
__global__ void k(const float *X, const float *W, float *out, int mode, float wscale)
{
    const int lane = threadIdx.x, m = lane & 15, q = lane >> 4;
    f32x4 acc = {0, 0, 0, 0};
    if (mode == 0) {
        for (int s = 0; s < 32; s++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(X[m * 128 + 4 * s + q], W[m * 128 + 4 * s + q], acc, 0, 0, 0);
    } else if (mode == 4) {
        for (int ks = 0; ks < 4; ks++) {
            f16x8 a[2], b[2];
            for (int i = 0; i < 8; i++) {
                const int kk = 32 * ks + 8 * q + i;
                const float x = X[m * 128 + kk], w = W[m * 128 + kk] * wscale;
                a[0][i] = (_Float16)x; a[1][i] = (_Float16)(x - (float)a[0][i]);
                b[0][i] = (_Float16)w; b[1][i] = (_Float16)(w - (float)b[0][i]);
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[0], acc, 0, 0, 0);
        }
        for (int i = 0; i < 4; i++) acc[i] *= 1.0f / wscale;
    } else {
        for (int ks = 0; ks < 4; ks++) {
            bf16x8 a[3], b[3];
            for (int i = 0; i < 8; i++) {
                const int kk = 32 * ks + 8 * q + i;
                __bf16 h, mm, l;
                split3(X[m * 128 + kk], h, mm, l); a[0][i] = h; a[1][i] = mm; a[2][i] = l;
                split3(W[m * 128 + kk], h, mm, l); b[0][i] = h; b[1][i] = mm; b[2][i] = l;
            }
            // small terms first
            if (mode == 3) { acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[2], acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[2], acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[1], acc, 0, 0, 0); }
            if (mode != 2) { acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc, 0, 0, 0); }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
        }
    }
    for (int i = 0; i < 4; i++) out[(4 * q + i) * 16 + m] = acc[i];   // C[row 4q+i][col m]
}
int main()
{
    float hX[16 * 128], hW[16 * 128], hO[256];
    srand(1);
    const char *names[5] = {"f32 mfma 16x16x4", "bf16x6", "bf16x3", "bf16x9", "f16x3 (w * 2^16)"};
    for (int scale = 0; scale < 2; scale++) {
        for (int i = 0; i < 16 * 128; i++) {
            float u = (float)rand() / RAND_MAX;
            hX[i] = scale ? fmaxf(0.f, (u - 0.3f) * 3.f) : (u - 0.5f) * 4.f;      // relu-like / symmetric
            hW[i] = ((float)rand() / RAND_MAX - 0.5f) * 0.18f;
        }
        float *dX, *dW, *dO; hipMalloc(&dX, sizeof hX); hipMalloc(&dW, sizeof hW); hipMalloc(&dO, sizeof hO);
        hipMemcpy(dX, hX, sizeof hX, hipMemcpyHostToDevice); hipMemcpy(dW, hW, sizeof hW, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 5; mode++) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dX, dW, dO, mode, 65536.0f);
            hipMemcpy(hO, dO, sizeof hO, hipMemcpyDeviceToHost);
            double maxe = 0, sume = 0, maxref = 0;
            for (int r = 0; r < 16; r++) for (int c = 0; c < 16; c++) {
                double ref = 0; for (int kk = 0; kk < 128; kk++) ref += (double)hX[r * 128 + kk] * (double)hW[c * 128 + kk];
                const double e = fabs(hO[r * 16 + c] - ref); if (e > maxe) maxe = e; sume += e; if (fabs(ref) > maxref) maxref = fabs(ref);
            }
            printf("inputs %d  %-18s max|err| %.3e  mean|err| %.3e  (max|ref| %.3f)\n", scale, names[mode], maxe, sume / 256, maxref);
        }
    }
    return 0;
}
