// mtfjsp_enc_shared.h — what the encoder kernels of BOTH translation units use (mtfjsp_encoder.hip, mtfjsp_gin_res.hip): sizes, the LDS-only
// barrier, the margin behind matrix instructions, the BatchNorm forms, vector types and the exact 3-way bf16 split.
#pragma once
#include <hip/hip_runtime.h>
#define HD 128
#define BN_EPS 1e-5

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would wait for every
// global prefetch in flight (the next weight block, the next tile) and serialise what is meant to overlap.
#define LDS_BARRIER()                                              \
    do {                                                           \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         \
        __builtin_amdgcn_s_barrier();                              \
        asm volatile("" ::: "memory");                             \
    } while (0)

// A vector instruction must not read the result of a matrix instruction before the matrix pipe has written it back, and gfx950 does NOT
// interlock that: tools/ubench/mfma_raw.hip reads stale registers up to 6 wait states behind v_mfma_f32_16x16x32_f16 (5 when the
// fillers are single-width vector instructions, where 15-45 % of the reads are still stale: a marginal regime that depends on how the
// SIMD's two waves interleave) and up to 10 behind v_mfma_f32_32x32x16_f16.  hipcc's hazard recogniser inserts the required s_nops with
// NO margin, counting every instruction in between as one wait state.  (This was the first suspect of the round-4 bisection of the
// function-form GAT miscomputation; the culprit turned out to be another one — mtfjsp_gat3x_body.h — but the margin is cheap.)
// MFMA_SETTLE puts four real wait states behind the last matrix instruction of a chain and ties the accumulators to them.
#define MFMA_SETTLE1(a) asm volatile("s_nop 3" : "+v"(a))
#define MFMA_SETTLE2(a, b) asm volatile("s_nop 3" : "+v"(a), "+v"(b))
#define MFMA_SETTLE3(a, b, c) asm volatile("s_nop 3" : "+v"(a), "+v"(b), "+v"(c))
#define MFMA_SETTLE8(a) asm volatile("s_nop 3" : "+v"((a)[0]), "+v"((a)[1]), "+v"((a)[2]), "+v"((a)[3]), "+v"((a)[4]), "+v"((a)[5]), "+v"((a)[6]), "+v"((a)[7]))

// sum over the 16 lanes of a DPP row, result in every lane: xor 1, xor 2 (quad permutes), half-row mirror, row mirror —
// four VALU instructions with DPP operands instead of four dependent ds_bpermute round trips
__device__ __forceinline__ float row_sum16(float x)
{
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true));
    return x;
}

__device__ __forceinline__ float bn_relu(float x, float mean, float rstd, float g, float b)
{
    float y = (x - mean) * rstd * g + b;
    return y > 0.f ? y : 0.f;
}
// the same BatchNorm + ReLU as one FMA: y = max(x * sc + sh, 0) with sc = rstd * gamma, sh = beta - mean * sc (the form
// torch's CPU kernel uses as well: alpha = invstd * weight, beta' = bias - mean * alpha)
__device__ __forceinline__ float bn_relu_ss(float x, float sc, float sh) { return fmaxf(fmaf(x, sc, sh), 0.f); }

#define STAT_REP 8            // replicated BatchNorm accumulators: <=32 adders per address keeps f64 atomics at full rate

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
// exact 3-way split of four f32 values into bf16 pieces (round-to-nearest-even), two packed dwords per plane
__device__ __forceinline__ void split3x4(const float (&v)[4], uint2 &p0, uint2 &p1, uint2 &p2)
{
    unsigned o[3][2];
#pragma unroll
    for (int hlf = 0; hlf < 2; hlf++) {
        float a = v[2 * hlf], b = v[2 * hlf + 1];
#pragma unroll
        for (int lvl = 0; lvl < 3; lvl++) {
            const unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));   // v_cvt_pk_bf16_f32
            o[lvl][hlf] = pk;
            if (lvl < 2) {                                        // remainders are exact: a - bf16(a) has <= 16 significant bits
                a -= __builtin_bit_cast(float, pk << 16);
                b -= __builtin_bit_cast(float, pk & 0xffff0000u);
            }
        }
    }
    p0 = make_uint2(o[0][0], o[0][1]); p1 = make_uint2(o[1][0], o[1][1]); p2 = make_uint2(o[2][0], o[2][1]);
}
