// Prototype (diagnostic, not part of the product): one BatchNorm+ReLU -> 128x128 Linear layer of the resident GIN kernel with
// EIGHT waves per workgroup (two per SIMD, <= 256 registers each, every register a VGPR: no accumulation-file moves).
// wave w = (column block c = w & 3: output columns 32c..32c+31, parity p = w >> 2: the 16-row tiles u = 2i + p, i = 0..17).
// v_mfma_f32_16x16x32_f16, A := weight fragment (16 columns x 32 k), B := activation rows (32 k x 16 rows) from LDS planes
// (2-way f16 split, 3 piece products); lane (n = l & 15, q = l >> 4) owns row n, columns 16b + 4q + r of its block b = 0, 1.
// The planes of tile u are written by the four waves (., u & 1) and read by the same four: they synchronise through one LDS
// counter per tile with a whole step of slack (tile i+2 is produced during step i), never through s_barrier.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o gin_layer8 gin_layer8.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>
#include <type_traits>
#include <utility>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#ifndef ABL
#define ABL 0   // 1 no production, 2 no matrix instructions, 4 no statistics, 8 no tile waits/signals, 16 no operand reads
#endif
#define NT 18                 // own tiles per wave
#define RING 4                // plane slots per parity
#define SLOT 8192             // 2 planes x 16 rows x 256 B
#define OFF_CNT (2 * RING * SLOT)
#define OFF_BN (OFF_CNT + 2 * NT * 4 + 112)
#define LDS_BYTES (OFF_BN + 2 * 128 * 4)

template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }
#define FENCE() __builtin_amdgcn_sched_barrier(0)

struct Args { const float *z0; const void *wimg; const float *scale, *shift; float *zout; float *stats; int layers; };

__global__ __launch_bounds__(512) void k_layer8(Args A)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int c = wave & 3, p = wave >> 2, n = lane & 15, q = lane >> 4;
    unsigned *s_cnt = reinterpret_cast<unsigned *>(smem + OFF_CNT) + p * NT;
    float *s_bn = reinterpret_cast<float *>(smem + OFF_BN);
    unsigned char *ring = smem + p * (RING * SLOT);
    if (tid < 2 * NT) reinterpret_cast<unsigned *>(smem + OFF_CNT)[tid] = 0u;
    if (tid < 128) { s_bn[tid] = A.scale[tid]; s_bn[128 + tid] = A.shift[tid]; }
    // accumulators: z of the previous layer / of this one
    f32x4 acc[NT][2];
    const size_t row0 = (size_t)blockIdx.x * (NT * 32);
    static_for<NT>([&](auto Ic) { constexpr int i = decltype(Ic)::value;
        for (int b = 0; b < 2; b++) acc[i][b] = *reinterpret_cast<const f32x4 *>(A.z0 + (row0 + (2 * i + p) * 16 + n) * 128 + 32 * c + 16 * b + 4 * q); });
    h8 wf[2][4][2];
    {
        const float4 *wi = reinterpret_cast<const float4 *>(A.wimg) + (size_t)c * (2 * 4 * 2 * 64) + lane;
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int ks = 0; ks < 4; ks++)
#pragma unroll
                for (int pl = 0; pl < 2; pl++) wf[b][ks][pl] = __builtin_bit_cast(h8, wi[((b * 4 + ks) * 2 + pl) * 64]);
    }
    float ts[2][4], tq[2][4];
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) { ts[b][r] = 0.f; tq[b][r] = 0.f; }
    __syncthreads();
    // lane constants: read address of (row n, k chunk q) and write address of (row n, columns 32c + 4q) in a slot (chunks XOR-swizzled by n)
    const unsigned rd0 = n * 256;                                  // + (((4 ks + q) ^ n) << 4) + plane * 4096
    const unsigned wr_chunk0 = 4 * c + (q >> 1), wr_half = 8 * (q & 1);
    const float *bnp = s_bn + 32 * c + 4 * q;

    auto produce_block = [&](const f32x4 &a, int b, unsigned char *slot, const float4 &s4, const float4 &h4) __attribute__((always_inline)) {
        float v0 = fmaxf(__builtin_fmaf(a[0], s4.x, h4.x), 0.f), v1 = fmaxf(__builtin_fmaf(a[1], s4.y, h4.y), 0.f);
        float v2 = fmaxf(__builtin_fmaf(a[2], s4.z, h4.z), 0.f), v3 = fmaxf(__builtin_fmaf(a[3], s4.w, h4.w), 0.f);
        const h2 p01 = __builtin_convertvector((__attribute__((ext_vector_type(2))) float){v0, v1}, h2);
        const h2 p23 = __builtin_convertvector((__attribute__((ext_vector_type(2))) float){v2, v3}, h2);
        const unsigned u01 = __builtin_bit_cast(unsigned, p01), u23 = __builtin_bit_cast(unsigned, p23);
        float r0, r1, r2, r3;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(u01), "v"(v0));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(u01), "v"(v1));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r2) : "v"(u23), "v"(v2));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r3) : "v"(u23), "v"(v3));
        const h2 q01 = __builtin_convertvector((__attribute__((ext_vector_type(2))) float){r0, r1}, h2);
        const h2 q23 = __builtin_convertvector((__attribute__((ext_vector_type(2))) float){r2, r3}, h2);
        unsigned char *d = slot + n * 256 + (((wr_chunk0 + 2 * b) ^ n) << 4) + wr_half;
        *reinterpret_cast<uint2 *>(d) = make_uint2(u01, u23);
        *reinterpret_cast<uint2 *>(d + 4096) = make_uint2(__builtin_bit_cast(unsigned, q01), __builtin_bit_cast(unsigned, q23));
    };
    auto stats_block = [&](const f32x4 &a, int b) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 4; r++) { ts[b][r] += a[r]; tq[b][r] = __builtin_fmaf(a[r], a[r], tq[b][r]); asm volatile("" : "+v"(ts[b][r]), "+v"(tq[b][r])); }
    };
    auto signal = [&](int i) __attribute__((always_inline)) {       // this wave's part of tile i is in its slot (and its reads of that slot's previous tenant are done)
        // (LDS executes one wave's operations in issue order: the add lands after this wave's plane writes and slot reads)
        if (lane == 0) __hip_atomic_fetch_add(&s_cnt[i], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto wait_tile = [&](int i, unsigned target) __attribute__((always_inline)) {
        while (true) {
            const unsigned v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&s_cnt[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            if (v >= target) break;
            __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
    };

    for (int layer = 0; layer < A.layers; layer++) {
        const unsigned target = 4u * (layer + 1);
        // prologue: tiles 0, 1 into their slots
        {
            const float4 sa = *reinterpret_cast<const float4 *>(bnp), ha = *reinterpret_cast<const float4 *>(bnp + 128);
            const float4 sb = *reinterpret_cast<const float4 *>(bnp + 16), hb = *reinterpret_cast<const float4 *>(bnp + 128 + 16);
            produce_block(acc[0][0], 0, ring + 0 * SLOT, sa, ha); produce_block(acc[0][1], 1, ring + 0 * SLOT, sb, hb); signal(0);
            produce_block(acc[1][0], 0, ring + 1 * SLOT, sa, ha); produce_block(acc[1][1], 1, ring + 1 * SLOT, sb, hb); signal(1);
        }
        unsigned peek = 0;
        FENCE();
        static_for<NT>([&](auto Ic) __attribute__((always_inline)) {
            constexpr int i = decltype(Ic)::value;
            constexpr bool NEXT = i + 2 < NT && !(ABL & 1);
            constexpr bool STATS = i > 0 && !(ABL & 4);
            const unsigned char *slot = ring + (i % RING) * SLOT;
            unsigned char *nslot = ring + ((i + 2) % RING) * SLOT;
            if (!(ABL & 8)) if (__builtin_amdgcn_readfirstlane(peek) < target) wait_tile(i, target);
            float4 s4 = *reinterpret_cast<const float4 *>(bnp), h4 = *reinterpret_cast<const float4 *>(bnp + 128);
            if constexpr (i + 1 < NT) peek = __hip_atomic_load(&s_cnt[i + 1 < NT ? i + 1 : 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            h8 xh = *reinterpret_cast<const h8 *>(slot + rd0 + (((0 + q) ^ n) << 4));
            h8 xl = *reinterpret_cast<const h8 *>(slot + rd0 + (((0 + q) ^ n) << 4) + 4096);
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
            static_for<4>([&](auto Kc) __attribute__((always_inline)) {
                constexpr int ks = decltype(Kc)::value;
                h8 nh = xh, nl = xl;
                if constexpr (ks < 3 && !(ABL & 16)) {
                    nh = *reinterpret_cast<const h8 *>(slot + rd0 + (((4 * (ks + 1) + q) ^ n) << 4));
                    nl = *reinterpret_cast<const h8 *>(slot + rd0 + (((4 * (ks + 1) + q) ^ n) << 4) + 4096);
                }
                if (!(ABL & 2)) a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[0][ks][0], xl, a0, 0, 0, 0);
                if constexpr (NEXT && ks == 0) produce_block(acc[NEXT ? i + 2 : 0][0], 0, nslot, s4, h4);
                if constexpr (NEXT && ks == 1) { s4 = *reinterpret_cast<const float4 *>(bnp + 16); h4 = *reinterpret_cast<const float4 *>(bnp + 128 + 16); }
                if constexpr (STATS && ks == 1) stats_block(acc[STATS ? i - 1 : 0][0], 0);
                FENCE();
                if (!(ABL & 2)) a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[1][ks][0], xl, a1, 0, 0, 0);
                FENCE();
                if (!(ABL & 2)) a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[0][ks][1], xh, a0, 0, 0, 0);
                if constexpr (NEXT && ks == 2) produce_block(acc[NEXT ? i + 2 : 0][1], 1, nslot, s4, h4);
                if constexpr (STATS && ks == 3) stats_block(acc[STATS ? i - 1 : 0][1], 1);
                FENCE();
                if (!(ABL & 2)) a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[1][ks][1], xh, a1, 0, 0, 0);
                FENCE();
                if (!(ABL & 2)) a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[0][ks][0], xh, a0, 0, 0, 0);
                FENCE();
                if (!(ABL & 2)) a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[1][ks][0], xh, a1, 0, 0, 0);
                FENCE();
                xh = nh; xl = nl;
            });
            if (ABL & 2) { a0[0] += xh[0]; a1[0] += xl[0]; }
            acc[i][0] = a0; acc[i][1] = a1;
            if constexpr (i + 2 < NT && !(ABL & 8)) signal(i + 2);
            FENCE();
        });
        stats_block(acc[NT - 1][0], 0); stats_block(acc[NT - 1][1], 1);
        __syncthreads();
    }
    static_for<NT>([&](auto Ic) { constexpr int i = decltype(Ic)::value;
        for (int b = 0; b < 2; b++) *reinterpret_cast<f32x4 *>(A.zout + (row0 + (2 * i + p) * 16 + n) * 128 + 32 * c + 16 * b + 4 * q) = acc[i][b]; });
    float s = 0.f;
    for (int b = 0; b < 2; b++) for (int r = 0; r < 4; r++) s += ts[b][r] + tq[b][r];
    A.stats[(size_t)blockIdx.x * 512 + tid] = s;
}

static unsigned short f2h(float x) { _Float16 h = (_Float16)x; unsigned short u; memcpy(&u, &h, 2); return u; }
static float h2f(unsigned short u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }
int main(int argc, char **argv)
{
    const int layers = argc > 1 ? atoi(argv[1]) : 1, grid = 256, rows = grid * NT * 32;
    std::vector<float> z0((size_t)rows * 128), W(128 * 128), sc(128), sh(128);
    srand(1);
    for (auto &x : z0) x = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    for (auto &x : W) x = ((rand() / (float)RAND_MAX) * 2.f - 1.f) * 0.088f;
    for (int i = 0; i < 128; i++) { sc[i] = 0.5f + (i % 7) * 0.1f; sh[i] = 0.1f * ((i % 5) - 2); }
    // register image: [c 4][b 2][ks 4][plane 2][lane 64] x 8 f16: lane (m, q): W[col 32c + 16b + m][k = 32 ks + 8 q + j]
    std::vector<unsigned short> img((size_t)4 * 2 * 4 * 2 * 64 * 8);
    for (int c = 0; c < 4; c++) for (int b = 0; b < 2; b++) for (int ks = 0; ks < 4; ks++) for (int pl = 0; pl < 2; pl++) for (int l = 0; l < 64; l++) for (int j = 0; j < 8; j++) {
        const float w = W[(32 * c + 16 * b + (l & 15)) * 128 + 32 * ks + 8 * (l >> 4) + j];
        const unsigned short hi = f2h(w); const unsigned short lo = f2h(w - h2f(hi));
        img[(((((size_t)c * 2 + b) * 4 + ks) * 2 + pl) * 64 + l) * 8 + j] = pl == 0 ? hi : lo;
    }
    float *dz0, *dzo, *dsc, *dsh, *dst; void *dimg;
    hipMalloc(&dz0, z0.size() * 4); hipMalloc(&dzo, z0.size() * 4); hipMalloc(&dsc, 512); hipMalloc(&dsh, 512); hipMalloc(&dst, (size_t)grid * 512 * 4); hipMalloc(&dimg, img.size() * 2);
    hipMemcpy(dz0, z0.data(), z0.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dsc, sc.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dsh, sh.data(), 512, hipMemcpyHostToDevice);
    hipMemcpy(dimg, img.data(), img.size() * 2, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)k_layer8, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    Args a{dz0, dimg, dsc, dsh, dzo, dst, 1};
    hipLaunchKernelGGL(k_layer8, dim3(grid), dim3(512), LDS_BYTES, 0, a);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    std::vector<float> zo(z0.size());
    hipMemcpy(zo.data(), dzo, zo.size() * 4, hipMemcpyDeviceToHost);
    double maxerr = 0, maxv = 0;
    for (int rr = 0; rr < 2000; rr++) {
        const int r = (int)(((long long)rr * 7919) % rows);
        for (int col = 0; col < 128; col += 5) {
            double s = 0;
            for (int k = 0; k < 128; k++) { const float x = fmaxf(fmaf(z0[(size_t)r * 128 + k], sc[k], sh[k]), 0.f); s += (double)x * W[col * 128 + k]; }
            maxerr = fmax(maxerr, fabs(s - zo[(size_t)r * 128 + col])); maxv = fmax(maxv, fabs(s));
        }
    }
    printf("one layer: max |err| %.3g (max |z| %.3g)\n", maxerr, maxv);
    for (int L : {1, 3, 5, 9}) {
        a.layers = L;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int w = 0; w < 3; w++) hipLaunchKernelGGL(k_layer8, dim3(grid), dim3(512), LDS_BYTES, 0, a);
        hipEventRecord(e0, 0);
        for (int w = 0; w < 20; w++) hipLaunchKernelGGL(k_layer8, dim3(grid), dim3(512), LDS_BYTES, 0, a);
        hipEventRecord(e1, 0); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("layers %d: %.2f us per launch\n", L, ms * 1e3 / 20);
    }
    return 0;
}
