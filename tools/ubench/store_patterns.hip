// Write-only bandwidth of two 16-byte-per-lane store patterns over a [rows][128] f32 matrix (512-byte rows), as the epilogue of
// k_gemm_x6 could issue them:  A = the accumulator layout's stores (a wave owns 32 columns: per instruction 16 rows x 64 bytes,
// the other 64 bytes of those 128-byte lines by the next instruction);  B = full rows (per instruction 2 rows x 512 bytes).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k_a(float *out, int rows_per_wg)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, m = lane & 15, q = lane >> 4;
    if (wave >= 4) return;                                        // 4 consumer waves, each 32 columns
    const size_t row0 = (size_t)blockIdx.x * rows_per_wg;
    for (int t = 0; t < rows_per_wg / 16; t++) {
        float *ob = out + (row0 + t * 16 + m) * 128 + 32 * wave + 4 * q;
        const float4 v = make_float4(1.f, 2.f, 3.f, (float)t);
        *reinterpret_cast<float4 *>(ob) = v;
        *reinterpret_cast<float4 *>(ob + 16) = v;
    }
}
__global__ __launch_bounds__(512) void k_b(float *out, int rows_per_wg)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= 4) return;
    const size_t row0 = (size_t)blockIdx.x * rows_per_wg;
    for (int t = 0; t < rows_per_wg / 16; t++)
        for (int i = 0; i < 2; i++) {                              // this wave's quarter of the 16-row tile: 4 rows = 2 instructions x 2 rows
            float *ob = out + (row0 + t * 16 + 4 * wave + 2 * i) * 128 + lane * 4;
            *reinterpret_cast<float4 *>(ob) = make_float4(1.f, 2.f, 3.f, (float)t);
        }
}
// C = whole 128-byte lines: per instruction 8 rows x 128 bytes (a wave's 32 columns of 8 rows, after a cross-lane exchange)
__global__ __launch_bounds__(512) void k_c(float *out, int rows_per_wg)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= 4) return;
    const size_t row0 = (size_t)blockIdx.x * rows_per_wg;
    for (int t = 0; t < rows_per_wg / 16; t++)
        for (int i = 0; i < 2; i++) {
            float *ob = out + (row0 + t * 16 + 8 * i + (lane >> 3)) * 128 + 32 * wave + 4 * (lane & 7);
            *reinterpret_cast<float4 *>(ob) = make_float4(1.f, 2.f, 3.f, (float)t);
        }
}
int main()
{
    const size_t rows = 819200;
    float *d; hipMalloc(&d, rows * 512);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 3; which++) {
        const int grid = 256, rpw = (int)(rows / grid);
        for (int rep = 0; rep < 3; rep++) { if (which == 2) hipLaunchKernelGGL(k_c, dim3(grid), dim3(512), 0, 0, d, rpw); else if (which) hipLaunchKernelGGL(k_b, dim3(grid), dim3(512), 0, 0, d, rpw); else hipLaunchKernelGGL(k_a, dim3(grid), dim3(512), 0, 0, d, rpw); }
        hipEventRecord(e0, 0);
        for (int rep = 0; rep < 10; rep++) { if (which == 2) hipLaunchKernelGGL(k_c, dim3(grid), dim3(512), 0, 0, d, rpw); else if (which) hipLaunchKernelGGL(k_b, dim3(grid), dim3(512), 0, 0, d, rpw); else hipLaunchKernelGGL(k_a, dim3(grid), dim3(512), 0, 0, d, rpw); }
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.1f us per launch, %.2f TB/s (%zu MB, 256 workgroups x 4 storing waves)\n", which == 2 ? "C whole lines (8 x 128 B per instruction)" : which ? "B full rows (1 KB contiguous per instruction)" : "A accumulator layout (16 x 64 B per instruction)",
               ms * 100, rows * 512 / (ms / 10 * 1e-3) / 1e12, rows * 512 >> 20);
    }
    return 0;
}
