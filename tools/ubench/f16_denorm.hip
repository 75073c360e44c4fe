// Does v_mfma_f32_32x32x16_f16 honour f16 subnormal inputs?  (decides whether the 2-way f16 operand split needs pre-scaling)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(float a, float b, float *out)
{
    h8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = (_Float16)a; B[i] = (_Float16)b; }
    f16v c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}
int main()
{
    float *d; hipMalloc(&d, 4);
    const float cases[][2] = {{9.5367431640625e-07f, 1024.f}, {1024.f, 9.5367431640625e-07f}, {5.9604644775390625e-08f, 16384.f}, {1.f, 1.f}};
    for (auto &c : cases) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, c[0], c[1], d);
        float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("a=%g b=%g  ->  c=%g  (expected %g)\n", c[0], c[1], h, 16.0 * (double)c[0] * (double)c[1]);
    }
    return 0;
}
