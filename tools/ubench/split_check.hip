// split_check.hip — the low operand piece f16(x - (float)f16(x)) three ways, bit for bit: conversions + subtraction (the definition),
// v_fma_mix_f32 + conversion (round 3), v_fma_mixlo_f16 / v_fma_mixhi_f16 (round 4).  hipcc --offload-arch=gfx950 -O3 -o split_check split_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(const float *x, unsigned *out_def, unsigned *out_mix, unsigned *out_mix16, unsigned *hi_out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const f2 v = {x[2 * i], x[2 * i + 1]};
    const h2 a = __builtin_convertvector(v, h2);
    const unsigned pa = __builtin_bit_cast(unsigned, a);
    hi_out[i] = pa;
    {   // definition (kept as written: volatile stops the optimiser from re-fusing)
        volatile float e0 = (float)a[0], e1 = (float)a[1];
        const f2 r = {v[0] - e0, v[1] - e1};
        out_def[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, h2));
    }
    {
        float r0, r1;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(pa), "v"(v[0]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(pa), "v"(v[1]));
        out_mix[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(f2{r0, r1}, h2));
    }
    {
        unsigned r = 0xdeadbeefu;
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "+v"(r) : "v"(pa), "v"(v[0]));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(pa), "v"(v[1]));
        out_mix16[i] = r;
    }
}
int main()
{
    const int n = 1 << 22;
    std::vector<float> h(n);
    srand(1);
    for (int i = 0; i < n; i++) {
        const int kind = i & 7;
        const float u = (float)rand() / RAND_MAX * 2.f - 1.f;
        h[i] = kind == 0 ? u : kind == 1 ? u * 1e-3f : kind == 2 ? u * 60000.f : kind == 3 ? u * 1e-7f : kind == 4 ? u * 300.f : kind == 5 ? 0.f : kind == 6 ? u * 6.2e-5f : u * 8.f;
    }
    float *d; unsigned *o[4];
    hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    for (auto &p : o) hipMalloc(&p, n / 2 * 4);
    hipLaunchKernelGGL(k, dim3(n / 2 / 256), dim3(256), 0, 0, d, o[0], o[1], o[2], o[3], n);
    std::vector<unsigned> a(n / 2), b(n / 2), c(n / 2);
    hipMemcpy(a.data(), o[0], n / 2 * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), o[1], n / 2 * 4, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), o[2], n / 2 * 4, hipMemcpyDeviceToHost);
    long d1 = 0, d2 = 0;
    for (int i = 0; i < n / 2; i++) { d1 += a[i] != b[i]; d2 += a[i] != c[i]; }
    printf("pairs %d: v_fma_mix_f32 form differs from the definition in %ld, v_fma_mixlo/hi_f16 form in %ld\n", n / 2, d1, d2);
    return (d1 || d2) ? 1 : 0;
}
