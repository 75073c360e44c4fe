// mtfjsp_encoder.hip — rollout forward passes of the job actor (GIN encoder + candidate scorer + local critic)
// and the machine actor (3x shared 2-node GAT + BatchNorm + scorer + local critic) for MI355X (gfx950 / CDNA4).
//
// Storage and accumulation are f32.  Every [rows,128] x [128,128] product runs on the bf16 matrix cores at f32 accuracy:
// both operands are split exactly into three bf16 pieces (round-to-nearest) and the six significant piece products are
// accumulated in f32 by v_mfma_f32_16x16x32_bf16 (k_gemm_x6 explains and cites the measurements).
//   k_gemm_x6<PRO>  the GIN products (gcn:95-153): 4 producer waves (BatchNorm+ReLU / neighbour aggregation / first-layer
//                   feature aggregation, split, bf16 planes to LDS) + 4 consumer waves (weights in registers, products,
//                   16-byte stores, BatchNorm column sums of the output); one barrier per 4-tile step; HBM-bound
//   k_gat3x         the three applications of the shared 2-node GAT layer of the machine actor (gat:82-159) in one launch
//   k_headsx        scorer + local critic of an actor for 16 instances per workgroup, incl. BatchNorm/pool/gather of the
//                   encoder output, masked softmax, action selection and m_fea1 of the selected task
// Training-mode BatchNorm (batch statistics over all rows) couples all rows at every layer: each boundary
// costs exactly one write + one read of the [rows,128] f32 activations (SURVEY.md §8d ENC_BYTES); the column sums are
// accumulated by the producing kernel and applied by the consuming one — no separate normalisation pass exists.
// The f32-instruction kernels of the earlier design (k_gemm16p, k_gat3, k_heads: v_mfma_f32_16x16x4_f32) are kept as the A/B
// reference behind mtfjsp_encoder_set_product_mode(); k_gin_inst / k_gat_inst are the per-instance-BatchNorm (evaluation) path.
//
// "gcn:" = model/gcn_mlp.py, "gat:" = model/gat.py, "ac:" = model/actor_critic.py, "agent:" = algorithm/agent_func.py
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mtfjsp.h"


#include "mtfjsp_enc_shared.h"

enum { PRO_PLAIN = 0, PRO_BNRELU = 1, PRO_AGG = 2, PRO_GIN0 = 3, PRO_GIN0BN = 4 };
enum { EPI_PLAIN = 0, EPI_STATS = 1, EPI_TANH = 2 };

// All [rows,128] activation buffers the GEMM kernels read or write are INTERNAL workspaces allocated with the row count
// rounded up to a whole 32-row tile (rows_padded()), so tiles are loaded, multiplied and stored without per-row bounds
// checks; only the BatchNorm column sums mask the rows >= N of the last tile.
struct GemmArgs {
    const float *in;        // [N,128] producer output (pre-activation for PRO_BNRELU / PRO_AGG)
    int N;
    const float *Wt;        // [128(k),128(n)] = W^T of a torch Linear weight [out,in]
    const void *Wx6;        // the same weight as 16-bit operand-piece register images (mtfjsp_encoder::wx6), k_gemm_x6
    float w_sinv;           // 1 / the power-of-two scale folded into that image (f16 images; 1 for the bf16 image of the first Linear)
    const float *bias;      // [128] or NULL
    float *out;             // [N,128]
    // prologue: BatchNorm of `in` from the producer's column sums
    const double *pro_stats;   // [STAT_REP][256] sum | sumsq
    const float *pro_gamma, *pro_beta;
    double pro_inv_rows;
    const int *ell_col;     // PRO_AGG: [N,2] (caller buffer: exactly N rows)
    const float *ell_val;   // PRO_AGG: [N,2]
    int T;                  // PRO_AGG: rows per instance
    const void *tfea;       // PRO_GIN0: raw task features [N,12] (f32 or f64), aggregated over the ELL adjacency and multiplied by the 12 -> 128 Linear
    int feat_f64;
    const void *Wx6_0;      // PRO_GIN0BN: the 12 -> 128 first Linear's bf16 x 3-plane register image (what PRO_GIN0 takes as Wx6) and its bias: the producers form
    const float *bias0;     // z0 = Linear0(aggregated raw features) of their tile themselves (f32 features), then BatchNorm + ReLU as PRO_BNRELU
    const float *W0;        // PRO_GIN0BN, moments mode (non-NULL): the first Linear's weight [128,12]; pro_stats then holds k_gin0_moments' sums (MOM_* below) and the
                            // BatchNorm statistics of z0 = W0 x + b follow from them: sum z = w.Sx + n b, sum z^2 = w'Sxx w + 2 b w.Sx + n b^2 (f64)
    double *epi_stats;      // EPI_STATS: [STAT_REP][256] accumulated with atomics (zeroed by the host per forward); NULL: no sums
    // PRO_BNRELU, pooling epilogue (round 6; out == NULL): the output is the LAST Linear's — it is normalised with the BatchNorm sums a statistics-only pass
    // of the same product left in pool_stats, ReLU, and only its per-instance column means (gcn:192) and the candidates' rows (ac:197-207) leave the chip
    float *pooled;          // [B,128], zero on entry: partial means are added (at most two contributions per element wherever a workgroup's rows span an instance)
    float *cand_feat;       // [B*J,128] or NULL
    const int *cand;        // [B,J]: candidate j of an instance is taken from job j's block of rows [j M, (j+1) M) (k_cand_fixup serves any other)
    int pool_J;
    const double *pool_stats; const float *pool_gamma, *pool_beta; double pool_inv_rows;
    float *zero_f32; int zero_count;        // a buffer this launch clears on the way (the statistics-only pass in front of the pooling pass: h_pooled)
    unsigned long long *stamps;   // diagnostic build only (-DMTFJSP_STAMP): per-wave phase cycle sums [waves][8]
    int dbg;                // diagnostic build only: timing ablations of k_gemm16p (1 no stores/sums, 2 no row requests/transform)
    unsigned *range_flag;             // host-mapped word: raised when the BatchNorm sums this launch consumes are not numbers (an f16 operand piece overflowed upstream; ReLU would hide the NaN)
    // k_gemm_x6, matrices beyond the 256 MB memory-side cache (tools/ubench/mall_order.hip): a workgroup walks its tile range
    // from the END when `rev` is set — the launches of a chain alternate, so that each starts on what its predecessor wrote
    // last — and reads `in` with non-temporal loads (read once: it must not displace the matrix being written)
    int rev, nt;
};

// k_gin0_moments' layout inside a [256]-double BatchNorm accumulator replica: the column sums of the aggregated raw features and their second moments
#define MOM_SX 0            // [12]
#define MOM_SXX 16          // [12][12] (full, symmetric)
#define MOM_N 160
#ifdef MTFJSP_STAMP
#define STAMP(slot)                                                                      \
    do {                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                               \
        unsigned long long t_;                                                           \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");       \
        __builtin_amdgcn_sched_barrier(0);                                               \
        ph[slot] += t_ - t_last; t_last = t_;                                            \
    } while (0)
#else
#define STAMP(slot) do { } while (0)
#endif

// tanh(x) = 1 - 2/(exp(2x)+1) on the hardware exp2/rcp units: |error| < 3e-7 absolute (the scorer/critic heads are
// checked against the reference at 1e-4); saturates correctly for |x| large, a NaN stays a NaN.  Five instructions: multiply, v_exp_f32, add,
// v_rcp_f32 (1 ulp), FMA.  (Until round 5 the quotient was written __fdividef(2, e + 1), which hipcc expands into the full IEEE division —
// two v_div_scale, v_rcp, four FMAs, v_div_fmas, v_div_fixup: 16 instructions per tanh, ~60 tanh per wave and heads part, a third of
// the heads' vector instructions.)
__device__ __forceinline__ float fast_tanh(float x)
{
    const float e = __builtin_amdgcn_exp2f(x * 2.885390081777926815f);        // exp(2x) = 2^(2 x log2 e)
    return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}





// ---- shared pieces of the matrix-core kernels -----------------------------------------------------------------
// Layout of every matrix kernel below (v_mfma_f32_16x16x4_f32, 16-row tiles, 8 waves per workgroup = 2 per SIMD):
//   lane = (m = lane & 15, q = lane >> 4):  A[row m][k = 4s+q],  B[k = 4s+q][col 16c+m],  C[row 4q+i][col 16c+m], i = 0..3
//   W^T [k][n] in LDS, XOR-swizzled (n ^ 16*(k&1)) so the B reads of lane quarters q = 0/1 fall on opposite halves of the
//   bank row; tile row stride LDA16 = 130 words makes the A reads conflict-free
//   load/transform mapping: lane = (j = lane & 31, h = lane >> 5) owns rows 2p+h (p = 0..7), columns 4j..4j+3
#define LDA16 130
__device__ __forceinline__ void stage_w16(float *s_w, const float *Wt, int tid)
{
    const float4 *src = reinterpret_cast<const float4 *>(Wt);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int ii = tid + ((i + blockIdx.x) & 7) * 512;       // rotated chunk order: 256 CUs streaming the same 64 KB
        const int k = ii >> 5, n4 = (ii & 31) * 4;                // would otherwise hit one L2 channel in lock-step
        *reinterpret_cast<float4 *>(s_w + k * HD + (n4 ^ (16 * (k & 1)))) = src[ii];
    }
}
// BatchNorm scale/shift of 128 columns from the producer's replicated f64 column sums -> s_bn[0..127] = sc, [128..255] = sh
__device__ __forceinline__ void stage_bn(float *s_bn, const double *stats, double inv_rows, const float *gamma, const float *beta, int tid)
{
    if (tid < HD) {
        double su = 0, sq = 0;
        for (int r = 0; r < STAT_REP; r++) { su += stats[r * 256 + tid]; sq += stats[r * 256 + HD + tid]; }
        const double mean = su * inv_rows;
        double var = sq * inv_rows - mean * mean;                 // biased variance (training-mode BN)
        if (var < 0) var = 0;
        const float rstd = 1.0f / sqrtf((float)(var + BN_EPS));
        const float sc = rstd * gamma[tid];
        s_bn[tid] = sc;
        s_bn[HD + tid] = beta[tid] - (float)mean * sc;
    }
}
// the same from workgroup-local sums (LDS, no replicas): per-instance BatchNorm
__device__ __forceinline__ void stage_bn_local(float *s_bn, const double *st, double inv_rows, const float *gamma, const float *beta, int tid)
{
    if (tid < HD) {
        const double mean = st[tid] * inv_rows;
        double var = st[HD + tid] * inv_rows - mean * mean;
        if (var < 0) var = 0;
        const float rstd = 1.0f / sqrtf((float)(var + BN_EPS));
        const float sc = rstd * gamma[tid];
        s_bn[tid] = sc;
        s_bn[HD + tid] = beta[tid] - (float)mean * sc;
    }
}
// 16-row tile x W^T: 32 k-steps x 8 column blocks.  The operands of step s+1 are requested before the 8 products of step
// s are issued (two register sets), so the LDS latency hides behind 256 cycles of matrix work.
__device__ __forceinline__ void mfma_tile16(const float *ap, const float *bp, const int (&bo)[8], f32x4 (&acc)[8])
{
    float a0 = ap[0], a1, b0[8], b1[8];
#pragma unroll
    for (int c = 0; c < 8; c++) b0[c] = bp[bo[c]];
#pragma unroll 2
    for (int s = 0; s < 32; s += 2) {
        const float *bs1 = bp + (s + 1) * 4 * HD;
        a1 = ap[4 * (s + 1)];
#pragma unroll
        for (int c = 0; c < 8; c++) b1[c] = bs1[bo[c]];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 8; c++) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0[c], acc[c], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        const int s2 = s + 2 < 32 ? s + 2 : 0;                   // (the last request is a harmless re-read of step 0)
        const float *bs2 = bp + s2 * 4 * HD;
        a0 = ap[4 * s2];
#pragma unroll
        for (int c = 0; c < 8; c++) b0[c] = bs2[bo[c]];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 8; c++) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1[c], acc[c], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    MFMA_SETTLE8(acc);
}
// per-wave f64 column sums (lane (m,q), block c) -> one f64 atomic per column per workgroup (into one of STAT_REP replicas);
// s_red may alias the LDS tiles: the first barrier makes sure every wave is done with them
__device__ __forceinline__ void flush_stats16(double *s_red, double *epi_stats, const double (&st_sum)[8], const double (&st_sq)[8],
                                              int tid, int wave, int m, int q)
{
    __syncthreads();
    for (int c = 0; c < 8; c++) {
        double a = st_sum[c], b = st_sq[c];
        a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
        b += __shfl_xor(b, 16); b += __shfl_xor(b, 32);
        if (q == 0) { s_red[wave * 256 + c * 16 + m] = a; s_red[wave * 256 + HD + c * 16 + m] = b; }
    }
    __syncthreads();
    if (tid < 256) {
        double v = 0;
        for (int w = 0; w < 8; w++) v += s_red[w * 256 + tid];
        atomicAdd(&epi_stats[(blockIdx.x % STAT_REP) * 256 + tid], v);
    }
}

// ---------------------------------------------------------------------------------------------
// k_gemm16 — plain [N,128] x [128,128] (+bias, optional tanh, optional accumulate into `out`) for the small critic-head
// products: load tile -> LDS -> 256 products -> store.  Not pipelined; the GIN encoder uses k_gemm16p below.
template <int EPI, bool ACC>
__global__ __launch_bounds__(512) void k_gemm16(GemmArgs A)
{
    extern __shared__ __align__(16) unsigned char smem[];
    float *s_w = reinterpret_cast<float *>(smem);                 // 128*128, swizzled
    float *s_a = s_w + HD * HD;                                   // 8 * 16 * LDA16
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int j = lane & 31, h = lane >> 5, c4 = j * 4;
    const int m = lane & 15, q = lane >> 4, qo = q & 1;
    stage_w16(s_w, A.Wt, tid);
    __syncthreads();
    float *my_a = s_a + wave * 16 * LDA16;
    const int ntiles = (A.N + 15) / 16;
    float bias8[8];
    for (int c = 0; c < 8; c++) bias8[c] = (A.bias && !ACC) ? A.bias[c * 16 + m] : 0.f;
    const float *ap = my_a + m * LDA16 + q;
    const float *bp = s_w + q * HD + m;
    int bo[8];
    for (int c = 0; c < 8; c++) bo[c] = (c ^ qo) * 16;
    for (int tile = blockIdx.x * 8 + wave; tile < ntiles; tile += gridDim.x * 8) {
        const float *tb = A.in + (size_t)tile * 16 * HD + h * HD + c4;
#pragma unroll
        for (int p = 0; p < 8; p++) {
            const float4 x = *reinterpret_cast<const float4 *>(tb + p * 2 * HD);
            float *d = my_a + (2 * p + h) * LDA16 + c4;
            *reinterpret_cast<float2 *>(d) = make_float2(x.x, x.y);
            *reinterpret_cast<float2 *>(d + 2) = make_float2(x.z, x.w);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        f32x4 acc[8];
        float *ob = A.out + (size_t)tile * 16 * HD + (4 * q) * HD + m;
#pragma unroll
        for (int c = 0; c < 8; c++)
#pragma unroll
            for (int i = 0; i < 4; i++) acc[c][i] = ACC ? ob[i * HD + c * 16] : bias8[c];
        mfma_tile16(ap, bp, bo, acc);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int c = 0; c < 8; c++)
#pragma unroll
            for (int i = 0; i < 4; i++) ob[i * HD + c * 16] = EPI == EPI_TANH ? fast_tanh(acc[c][i]) : acc[c][i];
    }
}
static size_t gemm16_lds_bytes() { return (size_t)(HD * HD + 8 * 16 * LDA16 + 2 * HD) * 4 + 64 + 8 * 16 * 8 * 4; }

// ---------------------------------------------------------------------------------------------
// k_gemm16p — the GIN-encoder product as ONE matrix-dense instruction stream per wave (16-row tiles, 8 waves per CU).
// Measured on k_gemm16: a wave alone on its SIMD reaches ~70 % of the f32 matrix rate, two waves multiplying at the same
// time saturate the pipe, and two free-running waves fall into lock-step — so whatever a wave does outside its matrix
// phase is time the pipe idles.  Here nothing is outside: while the 256 products of tile t are issued, the same wave
//     k-steps  0.. 7 : stores tile t-1 (previous accumulator set) and adds its BatchNorm column sums,
//     k-step   8     : requests the rows of tile t+1 (and, PRO_AGG, its neighbour rows; ELL entries one tile ahead),
//     k-step  16     : draws the next tile index from the CU's LDS counter,
//     k-steps 24..31 : BatchNorm+ReLU / neighbour aggregation of tile t+1 in registers,
// and only the 16 ds_write_b64 that put tile t+1 into the LDS tile (free once tile t's products are issued) are exposed.
// vmcnt counts loads and stores together and any wait drains both, hence stores first, loads 8 k-steps later, first use
// another 16 k-steps later.  Both waves of a SIMD run this stream, so the pipe always has two takers.
template <int PRO>
__global__ __launch_bounds__(512) void k_gemm16p(GemmArgs A)
{
    extern __shared__ __align__(16) unsigned char smem[];
    float *s_w = reinterpret_cast<float *>(smem);                 // 128*128, swizzled
    float *s_a = s_w + HD * HD;                                   // 8 * 16 * LDA16
    float *s_bn = s_a + 8 * 16 * LDA16;                           // scale | shift
    double *s_red = reinterpret_cast<double *>(s_a);              // reused after the last tile
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int j = lane & 31, h = lane >> 5, c4 = j * 4;           // load / transform mapping: rows 2p+h, 4 columns
    const int m = lane & 15, q = lane >> 4, qo = q & 1;           // matrix mapping: A[row m][k = 4s+q], B[k][col 16c+m], C[row 4q+i][col 16c+m]
#ifdef MTFJSP_STAMP
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last, rt0, rt1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_last)::"memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt0)::"memory");
    const unsigned long long t_first = t_last;
#endif
    // the first tile's rows are requested before anything else: their HBM latency hides behind the W^T / BatchNorm staging
    const int ntiles = (A.N + 15) / 16;
    const int per = (ntiles + gridDim.x - 1) / gridDim.x;
    const int first = blockIdx.x * per;
    const int last = first + per < ntiles ? first + per : ntiles;
    const int lane_off = h * HD + c4;
    float4 pre[8];
    if (first + wave < last) {
        const float *tb0 = A.in + (size_t)(first + wave) * 16 * HD;
#pragma unroll
        for (int p = 0; p < 8; p++) pre[p] = *reinterpret_cast<const float4 *>(tb0 + p * 2 * HD + lane_off);
    }
    stage_w16(s_w, A.Wt, tid);
    stage_bn(s_bn, A.pro_stats, A.pro_inv_rows, A.pro_gamma, A.pro_beta, tid);
    __syncthreads();
    STAMP(0);
    float *my_a = s_a + wave * 16 * LDA16;
    double st_sum[8], st_sq[8];
    for (int c = 0; c < 8; c++) { st_sum[c] = 0; st_sq[c] = 0; }
    const float sc0 = s_bn[c4], sc1 = s_bn[c4 + 1], sc2 = s_bn[c4 + 2], sc3 = s_bn[c4 + 3];
    const float sh0 = s_bn[HD + c4], sh1 = s_bn[HD + c4 + 1], sh2 = s_bn[HD + c4 + 2], sh3 = s_bn[HD + c4 + 3];
    float bias8[8];
    for (int c = 0; c < 8; c++) bias8[c] = A.bias ? A.bias[c * 16 + m] : 0.f;
    constexpr int NA = (PRO == PRO_AGG) ? 8 : 1;
    float4 nb0[NA], nb1[NA];
    int e_ox = 0, e_oy = 0, en_ox = 0, en_oy = 0;
    float e_vx = 0.f, e_vy = 0.f, en_vx = 0.f, en_vy = 0.f, e_dg = 1.f, en_dg = 1.f;
    auto fetch_ell = [&](int tile) __attribute__((always_inline)) {
        const int g = tile * 16 + m;
        int2 cc = make_int2(-1, -1); float2 vv = make_float2(0.f, 0.f);
        if (g < A.N) { cc = *reinterpret_cast<const int2 *>(A.ell_col + (size_t)g * 2); vv = *reinterpret_cast<const float2 *>(A.ell_val + (size_t)g * 2); }
        const int base = (g / A.T) * A.T - tile * 16;             // instance's first row relative to the tile
        en_ox = cc.x >= 0 ? base + cc.x : m; en_vx = cc.x >= 0 ? vv.x : 0.f;
        en_oy = cc.y >= 0 ? base + cc.y : m; en_vy = cc.y >= 0 ? vv.y : 0.f;
        en_dg = (float)(1 + (cc.x >= 0) + (cc.y >= 0));
    };
    auto prefetch = [&](int tile) __attribute__((always_inline)) {
        const float *tb = A.in + (size_t)tile * 16 * HD;          // wave-uniform
        if (PRO == PRO_AGG) { e_ox = en_ox; e_oy = en_oy; e_vx = en_vx; e_vy = en_vy; e_dg = en_dg; }
#pragma unroll
        for (int p = 0; p < 8; p++) {
            pre[p] = *reinterpret_cast<const float4 *>(tb + p * 2 * HD + lane_off);
            if (PRO == PRO_AGG) {
                const int pp = p < NA ? p : 0;
                const int ox = __shfl(e_ox, 2 * p + h), oy = __shfl(e_oy, 2 * p + h);
                nb0[pp] = *reinterpret_cast<const float4 *>(tb + (ptrdiff_t)ox * HD + c4);
                nb1[pp] = *reinterpret_cast<const float4 *>(tb + (ptrdiff_t)oy * HD + c4);
            }
        }
    };
    // rows 2p+h of the requested tile: raw -> BatchNorm+ReLU (-> neighbour aggregation), in place in pre[p]
    auto transform_rows = [&](int p) __attribute__((always_inline)) {
        float v0 = pre[p].x, v1 = pre[p].y, v2 = pre[p].z, v3 = pre[p].w;
        if (PRO == PRO_BNRELU) {
            v0 = bn_relu_ss(v0, sc0, sh0); v1 = bn_relu_ss(v1, sc1, sh1); v2 = bn_relu_ss(v2, sc2, sh2); v3 = bn_relu_ss(v3, sc3, sh3);
        } else {
            // gcn:125-149: (A_w @ h) / nnz_row, A_w includes the self loop (1); f64 accumulate, then cast
            const int pp = p < NA ? p : 0;
            const int r = 2 * p + h;
            const double wx = (double)__shfl(e_vx, r), wy = (double)__shfl(e_vy, r);
            const float dg = __shfl(e_dg, r);
            const double inv = dg == 1.f ? 1.0 : dg == 2.f ? 0.5 : (1.0 / 3.0);
            const double a0 = (double)bn_relu_ss(v0, sc0, sh0) + wx * (double)bn_relu_ss(nb0[pp].x, sc0, sh0) + wy * (double)bn_relu_ss(nb1[pp].x, sc0, sh0);
            const double a1 = (double)bn_relu_ss(v1, sc1, sh1) + wx * (double)bn_relu_ss(nb0[pp].y, sc1, sh1) + wy * (double)bn_relu_ss(nb1[pp].y, sc1, sh1);
            const double a2 = (double)bn_relu_ss(v2, sc2, sh2) + wx * (double)bn_relu_ss(nb0[pp].z, sc2, sh2) + wy * (double)bn_relu_ss(nb1[pp].z, sc2, sh2);
            const double a3 = (double)bn_relu_ss(v3, sc3, sh3) + wx * (double)bn_relu_ss(nb0[pp].w, sc3, sh3) + wy * (double)bn_relu_ss(nb1[pp].w, sc3, sh3);
            v0 = (float)(a0 * inv); v1 = (float)(a1 * inv); v2 = (float)(a2 * inv); v3 = (float)(a3 * inv);
        }
        pre[p] = make_float4(v0, v1, v2, v3);
    };
    auto tile_to_lds = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 8; p++) {
            float *d = my_a + (2 * p + h) * LDA16 + c4;
            *reinterpret_cast<float2 *>(d) = make_float2(pre[p].x, pre[p].y);
            *reinterpret_cast<float2 *>(d + 2) = make_float2(pre[p].z, pre[p].w);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    // previous tile: its accumulators, first output row, output pointer
    f32x4 acc[8], accp[8];
    float *obp = A.out;
    int rowp = 0;
    bool have_prev = false;
    auto store_prev = [&](int c) __attribute__((always_inline)) {   // column block c of the previous tile: 4 rows per lane
        float ts = 0.f, tq = 0.f;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float v = accp[c][i];
            obp[i * HD + c * 16] = v;
            v = (rowp + 4 * q + i < A.N) ? v : 0.f;
            ts += v; tq += v * v;
        }
        st_sum[c] += (double)ts; st_sq[c] += (double)tq;
    };
    // tile queue: t_cur (in the LDS tile), t_n1 (next; its ELL entries in en_* for PRO_AGG), t_new (drawn mid-tile)
    // static round-robin inside the CU's contiguous range (waves w and w+4 share a SIMD: a remainder of <= 4 tiles lands on
    // four different SIMDs)
    int t_cur = first + wave, t_n1 = t_cur + 8;
    if (t_cur < last) {
        if (PRO == PRO_AGG) {                                     // own rows are already in flight; ELL entries, then the neighbour rows
            fetch_ell(t_cur);
            e_ox = en_ox; e_oy = en_oy; e_vx = en_vx; e_vy = en_vy; e_dg = en_dg;
            const float *tb = A.in + (size_t)t_cur * 16 * HD;
#pragma unroll
            for (int p = 0; p < 8; p++) {
                const int ox = __shfl(e_ox, 2 * p + h), oy = __shfl(e_oy, 2 * p + h);
                nb0[p < NA ? p : 0] = *reinterpret_cast<const float4 *>(tb + (ptrdiff_t)ox * HD + c4);
                nb1[p < NA ? p : 0] = *reinterpret_cast<const float4 *>(tb + (ptrdiff_t)oy * HD + c4);
            }
        }
        if (PRO == PRO_AGG && t_n1 < last) fetch_ell(t_n1);
#pragma unroll
        for (int p = 0; p < 8; p++) transform_rows(p);
        tile_to_lds();
    }
    STAMP(1);
    const float *ap = my_a + m * LDA16 + q;
    const float *bp = s_w + q * HD + m;
    int bo[8];
    for (int c = 0; c < 8; c++) bo[c] = (c ^ qo) * 16;
    while (t_cur < last) {
        const int row0 = t_cur * 16;
        float *ob = A.out + (size_t)row0 * HD + (4 * q) * HD + m;   // C[row 4q+i][col 16c+m] = ob[i*HD + 16c]
#ifdef MTFJSP_STAMP
        const bool have_next = t_n1 < last && !(A.dbg & 2);
        if (A.dbg & 1) have_prev = false;
#else
        const bool have_next = t_n1 < last;
#endif
        const int t_new = t_n1 + 8;
#pragma unroll
        for (int c = 0; c < 8; c++)
#pragma unroll
            for (int i = 0; i < 4; i++) acc[c][i] = bias8[c];
        float av[2], bv[2][8];
        av[0] = ap[0];
#pragma unroll
        for (int c = 0; c < 8; c++) bv[0][c] = bp[bo[c]];
#pragma unroll
        for (int s = 0; s < 32; s++) {
            // operands of k-step s+1 (the last request is a harmless re-read of step 0)
            const int sn = s + 1 < 32 ? s + 1 : 0;
            av[(s + 1) & 1] = ap[4 * sn];
#pragma unroll
            for (int c = 0; c < 8; c++) bv[(s + 1) & 1][c] = bp[sn * 4 * HD + bo[c]];
            // side work of this k-step
            if (s < 8) { if (have_prev) store_prev(s); }
            if (s == 8 && have_next) {
                prefetch(t_n1);
                if (PRO == PRO_AGG && t_new < last) fetch_ell(t_new);
            }
            if (s >= 24 && have_next) transform_rows(s - 24);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < 8; c++) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s & 1], bv[s & 1][c], acc[c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        MFMA_SETTLE8(acc);
        asm volatile("" ::: "memory");
        if (have_next) tile_to_lds();
#pragma unroll
        for (int c = 0; c < 8; c++) accp[c] = acc[c];
        obp = ob; rowp = row0; have_prev = true;
#ifdef MTFJSP_STAMP
        ph[7] += 1;
#endif
        t_cur = t_n1; t_n1 = t_new;
    }
    STAMP(4);
    if (have_prev)
#pragma unroll
        for (int c = 0; c < 8; c++) store_prev(c);
    flush_stats16(s_red, A.epi_stats, st_sum, st_sq, tid, wave, m, q);
#ifdef MTFJSP_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(6);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1)::"memory");
    ph[2] = t_last - t_first; ph[3] = rt1 - rt0;                  // whole kernel: shader ticks | 100 MHz ticks
    if (A.stamps && lane == 0)
        for (int i = 0; i < 8; i++) A.stamps[((size_t)blockIdx.x * 8 + wave) * 8 + i] = ph[i];
#endif
}

// ---------------------------------------------------------------------------------------------
// k_gemm_x6 — the GIN product on the bf16 matrix cores at f32 accuracy.
//
// Measured on gfx950 (tools/ubench/mfma_valu.hip): v_mfma_f32_16x16x4_f32 runs at the packed-f32 VALU rate (one per 32
// cycles per SIMD, 155 TFLOP/s chip-wide) and does NOT overlap with VALU work — every VALU instruction of either wave
// adds its 4 cycles to the SIMD's time — so an f32-input product kernel is bounded by matrix time + all side work.
// v_mfma_f32_16x16x32_bf16 does 8x the flops in half the cycles.  Each f32 operand is therefore split exactly into three
// bf16 pieces (round-to-nearest: x = x0 + x1 + x2, |x1| <= 2^-9 |x|, |x2| <= 2^-18 |x|) and the product is formed from
// the six piece products of weight >= 2^-18 (x0w0, x0w1, x1w0, x1w1, x0w2, x2w0), accumulated in f32 by the matrix core,
// smallest terms first.  The dropped terms are <= 2^-26 relative — below the 2^-24 rounding of an f32 FMA — and
// tools/ubench/bf16x6.hip measures the result against f64: mean |error| 6.8e-8 vs 8.3e-8 for the f32 instruction.
// 6 products x 4 k-steps x 16 cycles = 384 cycles per 16x16 output tile instead of 1024.
//
// Layout: 4 consumer waves + 4 producer waves, one of each per SIMD, stepping through groups of four 16-row tiles with two
// LDS plane buffers and ONE barrier per step:
//   * producer wave w (4..7): BatchNorm+ReLU (PRO_AGG: + neighbour aggregation over the ELL adjacency, f64 accumulate) of
//     tile 4s + (w-4) from registers, exact split, three bf16 planes to buffer s&1 (row pitch 272 B: conflict-free
//     16-byte operand reads); its rows are requested two steps ahead, the aggregation's neighbour rows one step ahead.
//   * consumer wave w (0..3) owns output columns 32w..32w+31 (2 x 3 planes x 4 k-steps of weight fragments = 96
//     registers, loaded once from the weight's bf16 register image): in step s it multiplies the four tiles of buffer
//     (s-1)&1 — 4 x 2 x 24 products, operand fragments of the next k-step in flight — stores the transposed accumulators
//     (lane = row m, 4 consecutive columns: 16-byte stores) and keeps the BatchNorm column sums of the output.
//   The two roles never hold each other's registers (weights vs. rows in flight), a transform's memory latency is covered
//   by the consumer on the same SIMD, and the consumers are the older waves, which the issue arbiter favours.
#define X6_ROWB 272
#define X6_PLANE (16 * X6_ROWB)
#define X6_TILE (3 * X6_PLANE)
#define X6_TRB (16 * 144)                   // k_gemm_x6: a consumer wave's output transposition buffer (16 rows x 128 B, 144-byte pitch)
#define X6_TR_OFF (8 * X6_TILE + 2 * HD * 8 + 2 * HD * 4 + 64)
// 2-way split into f16 pieces (round to nearest both times): v = hi + lo up to 2^-22 |v|; with the three significant piece
// products as accurate as the exact bf16 split with six (tools/ubench/bf16x6.hip).  |v| < 65 504.
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
#define X2_TILE (2 * X6_PLANE)
// the low pieces f16(x - (float)high piece) of a value pair in two instructions: v_fma_mixlo_f16 / v_fma_mixhi_f16 take the f16
// high piece as an f32 operand and round the (exact) f32 remainder to f16 themselves — the same single rounding as a subtraction
// followed by a conversion, two instructions less per pair
__device__ __forceinline__ unsigned rem16x2(unsigned hi_pair, float x0, float x1)
{
    unsigned r;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hi_pair), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(hi_pair), "v"(x1));
    return r;
}
#ifdef MTFJSP_SPLIT_PLAIN    // A/B: the split written as conversions and subtractions
__device__ __forceinline__ void split2x4(const float (&v)[4], uint2 &p0, uint2 &p1)
{
    const f32x2 v01 = {v[0], v[1]}, v23 = {v[2], v[3]};
    const h16x2 a = __builtin_convertvector(v01, h16x2), b = __builtin_convertvector(v23, h16x2);
    const f32x2 r01 = v01 - __builtin_convertvector(a, f32x2), r23 = v23 - __builtin_convertvector(b, f32x2);
    const h16x2 c = __builtin_convertvector(r01, h16x2), d = __builtin_convertvector(r23, h16x2);
    p0 = make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b));
    p1 = make_uint2(__builtin_bit_cast(unsigned, c), __builtin_bit_cast(unsigned, d));
}
#else
__device__ __forceinline__ void split2x4(const float (&v)[4], uint2 &p0, uint2 &p1)
{
    const f32x2 v01 = {v[0], v[1]}, v23 = {v[2], v[3]};
    const h16x2 a = __builtin_convertvector(v01, h16x2), b = __builtin_convertvector(v23, h16x2);
    const unsigned pa = __builtin_bit_cast(unsigned, a), pb = __builtin_bit_cast(unsigned, b);
    p0 = make_uint2(pa, pb);
    p1 = make_uint2(rem16x2(pa, v[0], v[1]), rem16x2(pb, v[2], v[3]));
}
#endif
// the same split with the remainders taken by v_fma_mix_f32 (f16 piece as an f32 operand: x - (float)hi in ONE instruction
// instead of two conversions and a subtraction; exactly rounded either way, see gr_rem2 in mtfjsp_gin_resident.h)
__device__ __forceinline__ void split2x4m(const float (&v)[4], uint2 &p0, uint2 &p1)
{
    const f32x2 v01 = {v[0], v[1]}, v23 = {v[2], v[3]};
    const h16x2 a = __builtin_convertvector(v01, h16x2), b = __builtin_convertvector(v23, h16x2);
    const unsigned pa = __builtin_bit_cast(unsigned, a), pb = __builtin_bit_cast(unsigned, b);
    p0 = make_uint2(pa, pb);
    p1 = make_uint2(rem16x2(pa, v[0], v[1]), rem16x2(pb, v[2], v[3]));
}
template <int PRO>
__global__ __launch_bounds__(512) void k_gemm_x6(GemmArgs A)
{
    // operand pieces: the first Linear (raw, unbounded features) keeps the exact 3-way bf16 split (6 products); the 128 -> 128
    // products, whose operands are BatchNorm outputs, use the 2-way f16 split (3 products, weights pre-scaled: A.w_sinv)
    constexpr int NP = (PRO == PRO_GIN0) ? 3 : 2;
    constexpr int XT = NP * X6_PLANE;                             // bytes of a tile's planes
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char *s_tiles = smem;                                // [2 buffers][4 tiles][NP planes][16 rows x 272 B]
    double *s_stat = reinterpret_cast<double *>(smem + 8 * X6_TILE);   // column sums | sums of squares of this workgroup
    float *s_bn = reinterpret_cast<float *>(s_stat + 2 * HD);     // scale | shift
    float *s_b0 = reinterpret_cast<float *>(smem + 8 * X2_TILE);  // PRO_GIN0BN: the first Linear's bias (the two-plane tiles leave the rest of the tile area free)
    float *s_bn2 = reinterpret_cast<float *>(smem + 8 * X2_TILE + 2048);   // PRO_BNRELU, pooling epilogue: scale | shift of the output's BatchNorm
    int *s_cand = reinterpret_cast<int *>(smem + 8 * X2_TILE + 3072);      // ... and [2 buffers][4 tiles][16 rows]: the candidate (row inside its instance) of the job each row belongs to
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int ntiles = (A.N + 15) / 16;
    const int per = (ntiles + gridDim.x - 1) / gridDim.x;
    const int first = blockIdx.x * per;
    const int last = first + per < ntiles ? first + per : ntiles;
    const int nsteps = last > first ? (last - first + 3) >> 2 : 0;
    const int nsteps_c = (nsteps + 3) & ~3;                       // steps the barriers are counted in (see the producers)
    // logical tile t of [first, last) (the order of the steps) -> the tile of the matrix it stands for
    const int rev_sum = A.rev ? first + last - 1 : 0;
    auto PT = [&](int t) __attribute__((always_inline)) { return A.rev ? rev_sum - t : t; };
    constexpr int KS = (PRO == PRO_GIN0) ? 1 : 4;                 // k-steps of 32: the 12 -> 128 first Linear is one (k >= 12 are zero)
#ifdef MTFJSP_STAMP
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last, rt0, rt1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_last)::"memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt0)::"memory");
    const unsigned long long t_first = t_last;
#endif
    // requests in the order their data is needed (vmcnt retires in order): BatchNorm sums of the input first
    double bsu[STAT_REP], bsq[STAT_REP];
    float bga = 0.f, bbe = 0.f;
    if (PRO == PRO_BNRELU && A.zero_f32)
        for (int i = blockIdx.x * 512 + tid; i < A.zero_count; i += (int)gridDim.x * 512) A.zero_f32[i] = 0.f;
    const bool mom = PRO == PRO_GIN0BN && A.W0 != nullptr;        // (workgroup-uniform)
    double *s_mom = reinterpret_cast<double *>(smem + 8 * X2_TILE + HD * 4);      // PRO_GIN0BN, moments mode: [MOM_N] sums over the replicas
    double msum = 0;
    float w0r[12];
#pragma unroll
    for (int k = 0; k < 12; k++) w0r[k] = 0.f;
    if (PRO != PRO_GIN0 && tid < HD) {
        if (!mom) {
#pragma unroll
            for (int r = 0; r < STAT_REP; r++) { bsu[r] = A.pro_stats[r * 256 + tid]; bsq[r] = A.pro_stats[r * 256 + HD + tid]; }
        } else {
#pragma unroll
            for (int r = 0; r < STAT_REP; r++) { bsu[r] = 0; bsq[r] = 0; }
#pragma unroll
            for (int k4 = 0; k4 < 3; k4++) {
                const float4 w = *reinterpret_cast<const float4 *>(A.W0 + tid * 12 + 4 * k4);
                w0r[4 * k4] = w.x; w0r[4 * k4 + 1] = w.y; w0r[4 * k4 + 2] = w.z; w0r[4 * k4 + 3] = w.w;
            }
        }
        bga = A.pro_gamma[tid]; bbe = A.pro_beta[tid];
    }
    if (mom && tid < MOM_N) {                                      // (threads of the consumer waves: the producers' call of stage_scale_shift stores nothing)
#pragma unroll
        for (int r = 0; r < STAT_REP; r++) msum += A.pro_stats[r * 256 + tid];
    }
    static_assert(MOM_N <= 256, "moments: collected by consumer threads");
    auto stage_scale_shift = [&](auto Consc) __attribute__((always_inline)) {   // stage_bn() from the registers requested above; Consc: called by a consumer wave (threads
        constexpr bool CONS = decltype(Consc)::value;             // 0..255, which hold those registers: in the producers' code they are dead, and so are their registers)
        if (mom) {                                                // (every wave of the workgroup passes here exactly once)
            if constexpr (CONS) { if (tid < MOM_N) s_mom[tid] = msum; }
            LDS_BARRIER();
        }
        if (PRO == PRO_GIN0) {                                    // the planes' k = 12..31 stay zero for the whole kernel
            for (int i = tid; i < 8 * XT / 16; i += 512) reinterpret_cast<float4 *>(s_tiles)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        } else if (CONS && tid < HD) {
            double su = 0, sq = 0;
#pragma unroll
            for (int r = 0; r < STAT_REP; r++) { su += bsu[r]; sq += bsq[r]; }
            if (mom) {                                            // column tid of z0 = W0 x + b from the moments of x
                const double nrow = 1.0 / A.pro_inv_rows, b = A.bias0 ? (double)A.bias0[tid] : 0.0;
                double wsx = 0, q = 0;
#pragma unroll
                for (int i = 0; i < 12; i++) {
                    wsx = __builtin_fma((double)w0r[i], s_mom[MOM_SX + i], wsx);
                    double row = 0;
#pragma unroll
                    for (int k = 0; k < 12; k++) row = __builtin_fma((double)w0r[k], s_mom[MOM_SXX + i * 12 + k], row);
                    q = __builtin_fma((double)w0r[i], row, q);
                }
                su = wsx + nrow * b;
                sq = q + 2.0 * b * wsx + nrow * b * b;
            }
            if (A.range_flag && (su != su || sq != sq)) __hip_atomic_store(A.range_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const double mean = su * A.pro_inv_rows;
            double var = sq * A.pro_inv_rows - mean * mean;       // biased variance (training-mode BN)
            if (var < 0) var = 0;
            const float rstd = 1.0f / sqrtf((float)(var + BN_EPS));
            const float sc = rstd * bga;
            s_bn[tid] = sc;
            s_bn[HD + tid] = bbe - (float)mean * sc;
        }
        if (CONS && PRO == PRO_BNRELU && A.pooled && tid < HD) {  // the pooling epilogue's BatchNorm of this launch's OUTPUT (one exposed round trip per launch)
            double su = 0, sq = 0;
#pragma unroll
            for (int r = 0; r < STAT_REP; r++) { su += A.pool_stats[r * 256 + tid]; sq += A.pool_stats[r * 256 + HD + tid]; }
            if (A.range_flag && (su != su || sq != sq)) __hip_atomic_store(A.range_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const double mean = su * A.pool_inv_rows;
            double var = sq * A.pool_inv_rows - mean * mean;
            if (var < 0) var = 0;
            const float sc = (1.0f / sqrtf((float)(var + BN_EPS))) * A.pool_gamma[tid];
            s_bn2[tid] = sc;
            s_bn2[HD + tid] = A.pool_beta[tid] - (float)mean * sc;
        }
        if (CONS && tid < 2 * HD) s_stat[tid] = 0.0;
        if (CONS && PRO == PRO_GIN0BN && tid >= HD && tid < 2 * HD) s_b0[tid - HD] = A.bias0 ? A.bias0[tid - HD] : 0.f;
    };
    if (wave >= 4) {
        // ================================ producer ================================
        const int pw = wave - 4;
        if (A.dbg & 16) __builtin_amdgcn_s_setprio(2);
        const int j = lane & 31, h = lane >> 5, c4 = j * 4;       // rows 2p+h of the tile, 4 columns
        const int m = lane & 15;
        const int lane_off = h * HD + c4;
        float4 preA[8], preB[8];
        constexpr int NA = (PRO == PRO_AGG) ? 8 : 1;
        // PRO_AGG: neighbour rows TWO steps ahead like the tile's own rows (two register stages), the ELL entries they depend on
        // three (ring of 4).  (Round 2 requested the neighbour rows of step s+1 behind the transform of step s: with the consumers
        // waiting for the producers — 258 k of 466 k cycles at their barrier — that round trip was exposed in every step.)
#ifndef X6_NB_AHEAD
#define X6_NB_AHEAD 1                       // steps the aggregation's neighbour rows are requested ahead (2: a second register stage = 64 registers more, 256 + 52 B of scratch: 203 / 196 us against 198 / 193 at J20M20 / J10M10)
#endif
        struct NbRows { float4 r0[NA], r1[NA]; } nbA, nbB_;
        NbRows &nbB = X6_NB_AHEAD == 2 ? nbB_ : nbA;
        struct EllSlot { int ox, oy; float vx, vy, dg; } el[4];
#pragma unroll
        for (int i = 0; i < 4; i++) { el[i].ox = 0; el[i].oy = 0; el[i].vx = 0.f; el[i].vy = 0.f; el[i].dg = 1.f; }
        // (32-bit byte offsets from a workgroup-uniform base: scalar base + vector offset addressing, no 64-bit vector arithmetic
        // per request.  The base is the first row this workgroup can touch — its first tile's, less one instance for the
        // aggregation's neighbour rows — so the offsets are small and never negative whatever the size of the matrix.)
        const int base_row = (PRO == PRO_AGG) ? (first * 16 > A.T ? first * 16 - A.T : 0) : first * 16;
        const char *inb = reinterpret_cast<const char *>(A.in + (size_t)base_row * HD);
        auto request_rows = [&](float4 (&pre)[8], int tile) __attribute__((always_inline)) {
            const unsigned tb = ((unsigned)(PT(tile) * 16 - base_row) * HD + lane_off) * 4u;
            if (A.nt) {
#pragma unroll
                for (int p = 0; p < 8; p++) {
                    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(inb + (tb + p * 2 * HD * 4)));
                    pre[p] = make_float4(v[0], v[1], v[2], v[3]);
                }
            } else {
#pragma unroll
                for (int p = 0; p < 8; p++) pre[p] = *reinterpret_cast<const float4 *>(inb + (tb + p * 2 * HD * 4));
            }
        };
        auto fetch_ell = [&](EllSlot &E, int ltile) __attribute__((always_inline)) {
            const int tile = PT(ltile);
            const int g = tile * 16 + m;
            int2 cc = make_int2(-1, -1); float2 vv = make_float2(0.f, 0.f);
            if (g < A.N) { cc = *reinterpret_cast<const int2 *>(A.ell_col + (size_t)g * 2); vv = *reinterpret_cast<const float2 *>(A.ell_val + (size_t)g * 2); }
            const int base = (g / A.T) * A.T - tile * 16;         // instance's first row relative to the tile
            E.ox = cc.x >= 0 ? base + cc.x : m; E.vx = cc.x >= 0 ? vv.x : 0.f;
            E.oy = cc.y >= 0 ? base + cc.y : m; E.vy = cc.y >= 0 ? vv.y : 0.f;
            E.dg = (float)(1 + (cc.x >= 0) + (cc.y >= 0));
        };
        // lane m's value (m = lane & 15: the same in all four 16-lane rows) of the tile row 2p + h this lane works on: two DPP row
        // broadcasts, row_newbcast:2p into lanes 0..31 and row_newbcast:2p+1 into lanes 32..63 — no LDS round trip (the ds_bpermute
        // form: 40 per tile, each with its own wait on an LDS pipe the consumer waves keep busy), no scalar detour (v_readlane: 4.5
        // instructions per value)
        auto row_pick = [&](int x, int p) __attribute__((always_inline)) {
#define RP_(P) case P: { const int t_ = __builtin_amdgcn_update_dpp(0, x, 0x150 + 2 * P, 0x3, 0xF, false); return __builtin_amdgcn_update_dpp(t_, x, 0x150 + 2 * P + 1, 0xC, 0xF, false); }
            switch (p) { RP_(0) RP_(1) RP_(2) RP_(3) RP_(4) RP_(5) RP_(6) default: RP_(7) }
#undef RP_
        };
        auto request_nb = [&](NbRows &nb, const EllSlot &E, int tile) __attribute__((always_inline)) {   // neighbour rows of `tile`, whose ELL entries are in E
            const int tb = (PT(tile) * 16 - base_row) * HD + c4;  // (neighbour offsets are relative to the tile and may be negative; the row relative to base_row is not)
#pragma unroll
            for (int p = 0; p < 8; p++) {
                int ox = row_pick(E.ox, p), oy = row_pick(E.oy, p);
#ifdef X6_ABL_NB                            // timing ablation (wrong results): MTFJSP_GEMM_DBG & 32: the first neighbour is the row itself, & 64: the second
                if (A.dbg & 32) ox = 2 * p + h;
                if (A.dbg & 64) oy = 2 * p + h;
#endif
                nb.r0[p < NA ? p : 0] = *reinterpret_cast<const float4 *>(inb + (unsigned)(tb + ox * HD) * 4u);
                nb.r1[p < NA ? p : 0] = *reinterpret_cast<const float4 *>(inb + (unsigned)(tb + oy * HD) * 4u);
            }
        };
        const int t0 = first + pw;
        if constexpr (PRO == PRO_GIN0) {
            // lane = (row r = lane >> 2 of this wave's tile, features 3*(lane & 3)..+2): raw features of the row and of its <= 2
            // neighbours (gcn:125-153: f64 accumulate, divided by the row's entry count — 1, 2 or 3, as a multiplication).
            // The consumers' step is short here (one k-step), shorter than an HBM round trip, so the requests run FOUR steps
            // ahead (ring of 4 register stages) and the ELL entries they depend on eight.
            // (the whole producer once per feature type: a run-time branch per request is a join per request, see below)
            auto gin0_producer = [&](auto ft_tag) __attribute__((always_inline)) {
            using FT = decltype(ft_tag);
            const int r = lane >> 2, k0 = 3 * (lane & 3);
            struct Stage { float fo[3], fx[3], fy[3]; int2 cc; float2 vv; int gt; } st[4];
            struct Ell { int2 cc; float2 vv; } el[4];
            auto feat3 = [&](size_t row, float (&o)[3]) __attribute__((always_inline)) {
                const FT *p = reinterpret_cast<const FT *>(A.tfea) + row * 12 + k0; o[0] = (float)p[0]; o[1] = (float)p[1]; o[2] = (float)p[2];
            };
            // Requests are unconditional (see the other producers below: a request behind an `if` makes the compiler drain every
            // outstanding one at the join — the four-steps-ahead ring had never been in effect): a tile beyond the range or a row
            // beyond N is clamped to the last one and its values are zeroed afterwards, an absent neighbour reads the row itself
            const int lastm1 = last - 1;
            auto req_ell = [&](Ell &e, int tile) __attribute__((always_inline)) {
                const int gt = PT(tile < lastm1 ? tile : lastm1) * 16 + r, g = gt < A.N ? gt : A.N - 1;
                e.cc = *reinterpret_cast<const int2 *>(A.ell_col + (size_t)g * 2);
                e.vv = *reinterpret_cast<const float2 *>(A.ell_val + (size_t)g * 2);
            };
            auto req_feat = [&](Stage &x, const Ell &e, int tile) __attribute__((always_inline)) {
                const int gt = PT(tile < lastm1 ? tile : lastm1) * 16 + r, g = gt < A.N ? gt : A.N - 1;
                x.cc = e.cc; x.vv = e.vv; x.gt = gt;               // (rows beyond N are zeroed where the values are used, not here: a select
                const int base = (g / A.T) * A.T;                 // on the loaded values is turned back into a branch around the loads)
                feat3((size_t)g, x.fo);
                feat3((size_t)(x.cc.x >= 0 ? base + x.cc.x : g), x.fx);
                feat3((size_t)(x.cc.y >= 0 ? base + x.cc.y : g), x.fy);
            };
            if (nsteps == 0) { stage_scale_shift(std::false_type{}); LDS_BARRIER(); LDS_BARRIER(); }
            else {
#pragma unroll
            for (int i = 0; i < 4; i++) req_ell(el[i], t0 + 4 * i);
#pragma unroll
            for (int i = 0; i < 4; i++) { req_feat(st[i], el[i], t0 + 4 * i); req_ell(el[i], t0 + 4 * (i + 4)); }
            stage_scale_shift(std::false_type{});
            LDS_BARRIER();
            STAMP(0);
            for (int s0 = 0; s0 < nsteps_c; s0 += 4) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int s = s0 + i;
                    {
                        const int tile = t0 + 4 * s;
                        if (tile < last) {
                            const Stage &x = st[i];
                            const int deg = 1 + (x.cc.x >= 0) + (x.cc.y >= 0);
                            const double inv = deg == 1 ? 1.0 : deg == 2 ? 0.5 : (1.0 / 3.0);
                            float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int f = 0; f < 3; f++) {
                                // (an absent neighbour has weight 0 and read the row itself: the sum is formed unconditionally, so that
                                // the compiler cannot sink the neighbour requests under the condition of their use)
                                double acc = (double)x.fo[f];
                                acc += (double)x.vv.x * (double)x.fx[f];
                                acc += (double)x.vv.y * (double)x.fy[f];
                                v[f] = x.gt < A.N ? (float)(acc * inv) : 0.f;
                            }
                            uint2 p0, p1, p2;
                            split3x4(v, p0, p1, p2);                  // element 3 is padding
                            unsigned char *d = s_tiles + ((s & 1) * 4 + pw) * XT + r * X6_ROWB + k0 * 2;
                            unsigned short *d0 = reinterpret_cast<unsigned short *>(d), *d1 = reinterpret_cast<unsigned short *>(d + X6_PLANE),
                                           *d2 = reinterpret_cast<unsigned short *>(d + 2 * X6_PLANE);
                            d0[0] = (unsigned short)p0.x; d0[1] = (unsigned short)(p0.x >> 16); d0[2] = (unsigned short)p0.y;
                            d1[0] = (unsigned short)p1.x; d1[1] = (unsigned short)(p1.x >> 16); d1[2] = (unsigned short)p1.y;
                            d2[0] = (unsigned short)p2.x; d2[1] = (unsigned short)(p2.x >> 16); d2[2] = (unsigned short)p2.y;
                        }
                        req_feat(st[i], el[i], tile + 16);            // step s + 4 (its ELL entries arrived 4 steps ago)
                        req_ell(el[i], tile + 32);                    // step s + 8
                        STAMP(1);
                        LDS_BARRIER();
                        STAMP(4);
                    }
                }
            }
            LDS_BARRIER();                                        // the consumers' last step
            }
            };
            if (A.feat_f64) gin0_producer(double{}); else gin0_producer(float{});
        } else if constexpr (PRO == PRO_GIN0BN) {
            // Round 6: the first Linear's output never goes to memory.  A statistics-only PRO_GIN0 launch (out == NULL) has left the BatchNorm
            // sums of z0; here the producer forms z0 of its tile again — the same aggregation (gcn:125-153, f64 accumulate), the same exact
            // 3-way bf16 split and the same six matrix instructions per column block in the same order as PRO_GIN0's consumers, so the
            // same bits — adds the bias, and goes on as PRO_BNRELU's producer does (BatchNorm + ReLU, 2-way f16 split, planes).  The
            // 419 MB write of z0 and its 419 MB read are replaced by a second read of the raw features (48 B per row).
            // lane = (row m = lane & 15, k-quarter q = lane >> 4) holds the B operand fragment k = 8q..8q+7 of its row directly — features 0..7 (q = 0),
            // 8..11 + zeros (q = 1), zeros (q >= 2) — no LDS round trip for the operand.  The aggregation is spread over three quarters: q = 0 forms
            // features 0..3, q = 1 features 8..11, q = 2 features 4..7, which q = 0 takes from lane + 32 after the split (v_permlane32_swap).
            const int m = lane & 15, q = lane >> 4, fsel = q == 1 ? 8 : q == 2 ? 4 : 0;
            const int keep_lo = q < 2 ? -1 : 0, keep_hi = q == 0 ? -1 : 0;
            const float invT = 1.0f / (float)A.T;
            float4 w0f[8][3];                                     // [column block][plane]: pieces of W0[16 cb + m][8q..8q+7]
            {
                const float4 *wi = reinterpret_cast<const float4 *>(A.Wx6_0) + lane;
#pragma unroll
                for (int cb = 0; cb < 8; cb++)
#pragma unroll
                    for (int p = 0; p < 3; p++) w0f[cb][p] = wi[(cb * 3 + p) * 64];
            }
            struct Stage { float4 fo, fx, fy; int2 cc; float2 vv; int gt; } st[2];
            struct Ell { int2 cc; float2 vv; } el[4];
            const float *tf = reinterpret_cast<const float *>(A.tfea);
            const int lastm1 = last - 1;
            // (requests unconditional and clamped, as in the other producers: a request behind an `if` drains the queue at the join)
            auto req_ell = [&](Ell &e, int tile) __attribute__((always_inline)) {
                const int gt = PT(tile < lastm1 ? tile : lastm1) * 16 + m, g = gt < A.N ? gt : A.N - 1;
                e.cc = *reinterpret_cast<const int2 *>(A.ell_col + (size_t)g * 2);
                e.vv = *reinterpret_cast<const float2 *>(A.ell_val + (size_t)g * 2);
            };
            auto req_feat = [&](Stage &x, const Ell &e, int tile) __attribute__((always_inline)) {
                const int gt = PT(tile < lastm1 ? tile : lastm1) * 16 + m, g = gt < A.N ? gt : A.N - 1;
                x.cc = e.cc; x.vv = e.vv; x.gt = gt;
                int qi = (int)((float)g * invT);                  // g / T for g < 2^24: the f32 quotient is off by at most one
                int rem = g - qi * A.T;
                qi += rem >= A.T ? 1 : rem < 0 ? -1 : 0;
                const int base = qi * A.T;
                x.fo = *reinterpret_cast<const float4 *>(tf + (size_t)g * 12 + fsel);
                x.fx = *reinterpret_cast<const float4 *>(tf + (size_t)(x.cc.x >= 0 ? base + x.cc.x : g) * 12 + fsel);
                x.fy = *reinterpret_cast<const float4 *>(tf + (size_t)(x.cc.y >= 0 ? base + x.cc.y : g) * 12 + fsel);
            };
            if (nsteps == 0) { stage_scale_shift(std::false_type{}); LDS_BARRIER(); LDS_BARRIER(); }
            else {
#pragma unroll
            for (int i = 0; i < 4; i++) req_ell(el[i], t0 + 4 * i);
            req_feat(st[0], el[0], t0); req_ell(el[0], t0 + 16);
            req_feat(st[1], el[1], t0 + 4); req_ell(el[1], t0 + 20);
            stage_scale_shift(std::false_type{});
            LDS_BARRIER();
            STAMP(0);
            auto produce0 = [&](Stage &x, Ell &e, int s) __attribute__((always_inline)) {
                const int tile = t0 + 4 * s;
                if (tile < last) {
                    const int deg = 1 + (x.cc.x >= 0) + (x.cc.y >= 0);
                    const double inv = deg == 1 ? 1.0 : deg == 2 ? 0.5 : (1.0 / 3.0);
                    const float fo[4] = {x.fo.x, x.fo.y, x.fo.z, x.fo.w}, fx[4] = {x.fx.x, x.fx.y, x.fx.z, x.fx.w}, fy[4] = {x.fy.x, x.fy.y, x.fy.z, x.fy.w};
                    const int keep = x.gt < A.N ? -1 : 0;             // (a mask, not a select: hipcc turns the select into a branch around the f64 work)
                    float va[4];
#pragma unroll
                    for (int f = 0; f < 4; f++) {
                        double acc = (double)fo[f];                   // (the same three f64 operations as PRO_GIN0's producer)
                        acc += (double)x.vv.x * (double)fx[f];
                        acc += (double)x.vv.y * (double)fy[f];
                        va[f] = __builtin_bit_cast(float, __builtin_bit_cast(int, (float)(acc * inv)) & keep);
                    }
                    uint2 a[3];
                    split3x4(va, a[0], a[1], a[2]);
                    bf16x8 xb[3];
#pragma unroll
                    for (int p = 0; p < 3; p++) {                     // lanes 0..31 <- lanes 32..63: q = 0 receives q = 2's features 4..7
                        const unsigned bx = __builtin_amdgcn_permlane32_swap(a[p].x, a[p].x, false, false)[1], by = __builtin_amdgcn_permlane32_swap(a[p].y, a[p].y, false, false)[1];
                        xb[p] = __builtin_bit_cast(bf16x8, make_uint4(a[p].x & (unsigned)keep_lo, a[p].y & (unsigned)keep_lo, bx & (unsigned)keep_hi, by & (unsigned)keep_hi));
                    }
                    unsigned char *dst = s_tiles + ((s & 1) * 4 + pw) * XT + m * X6_ROWB + 8 * q;
#pragma unroll
                    for (int cp = 0; cp < 4; cp++) {
                        f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
                        auto MB = [&](int wp, int xp) __attribute__((always_inline)) {
                            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w0f[2 * cp][wp]), xb[xp], c0, 0, 0, 0);
                            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w0f[2 * cp + 1][wp]), xb[xp], c1, 0, 0, 0);
                        };
                        MB(0, 2); MB(2, 0); MB(1, 1); MB(0, 1); MB(1, 0); MB(0, 0);     // smallest terms first
                        MFMA_SETTLE2(c0, c1);
#pragma unroll
                        for (int c = 0; c < 2; c++) {
                            const int col = 16 * (2 * cp + c) + 4 * q;
                            const f32x4 z = c ? c1 : c0;
                            const float4 bb = *reinterpret_cast<const float4 *>(s_b0 + col);
                            const float4 sc = *reinterpret_cast<const float4 *>(s_bn + col), sh = *reinterpret_cast<const float4 *>(s_bn + HD + col);
                            const f32x2 z01 = f32x2{z[0], z[1]} + f32x2{bb.x, bb.y}, z23 = f32x2{z[2], z[3]} + f32x2{bb.z, bb.w};      // Linear0's bias (PRO_GIN0's epilogue)
                            const f32x2 ya = __builtin_elementwise_fma(z01, f32x2{sc.x, sc.y}, f32x2{sh.x, sh.y});                  // PRO_BNRELU's bnr4
                            const f32x2 yb = __builtin_elementwise_fma(z23, f32x2{sc.z, sc.w}, f32x2{sh.z, sh.w});
                            const float v[4] = {fmaxf(ya[0], 0.f), fmaxf(ya[1], 0.f), fmaxf(yb[0], 0.f), fmaxf(yb[1], 0.f)};
                            uint2 p0, p1;
                            split2x4m(v, p0, p1);
                            *reinterpret_cast<uint2 *>(dst + 2 * 16 * (2 * cp + c)) = p0;
                            *reinterpret_cast<uint2 *>(dst + 2 * 16 * (2 * cp + c) + X6_PLANE) = p1;
                        }
                    }
                }
                req_feat(x, e, tile + 8);                         // step s + 2 (its ELL entries were requested four steps before it)
                req_ell(e, tile + 24);                            // step s + 6
                STAMP(1);
                LDS_BARRIER();
                STAMP(4);
            };
            for (int s = 0; s < nsteps_c; s += 4) {
                produce0(st[0], el[2], s);
                produce0(st[1], el[3], s + 1);
                produce0(st[0], el[0], s + 2);
                produce0(st[1], el[1], s + 3);
            }
            LDS_BARRIER();                                        // the consumers' last step
            }
        } else {
        // Every step issues the SAME requests, unconditionally: a tile beyond the range is clamped to the last one (a few wasted
        // cache hits at the end of a range) and the steps are padded to a multiple of four (the consumers run the same number of
        // barriers).  With `if (tile < last)` around them, the compiler's wait counters had to assume the shorter path at every
        // join — "nothing younger in flight" — and drained ALL outstanding requests at the top of every step (s_waitcnt vmcnt(7..0)
        // in the ISA): the two-step prefetch was never in effect and the aggregation producer sat 8.6 k cycles per tile in here.
        if (nsteps == 0) { stage_scale_shift(std::false_type{}); LDS_BARRIER(); LDS_BARRIER(); }
        else {
        const int lastm1 = last - 1;
        auto CL = [&](int t) __attribute__((always_inline)) { return t < lastm1 ? t : lastm1; };
        if (PRO == PRO_AGG) { fetch_ell(el[0], CL(t0)); fetch_ell(el[1], CL(t0 + 4)); fetch_ell(el[2], CL(t0 + 8)); }
        // pooling epilogue: the candidate index the consumers compare each row of a tile with travels with the tile — requested by the producers with the
        // rows, handed over in LDS.  (In the consumers it would be a vector-memory load among their stores and atomics: every comparison would wait for
        // all of them.)  Always requested, from a valid address, so that no request sits behind a branch.
        const bool pcand = PRO == PRO_BNRELU && A.pooled && A.cand_feat;
        const int *candp = pcand ? A.cand : reinterpret_cast<const int *>(A.in);
        const int qT = A.T > 0 ? A.T : 1, qJ = pcand ? A.pool_J : 0, qM = qJ > 0 ? qT / qJ : qT;
        const float qinvT = 1.0f / (float)qT;
        const unsigned qinvM = (unsigned)((0x100000000ull + (unsigned)qM - 1) / (unsigned)qM);
        auto request_cand = [&](int &cv, int tile) __attribute__((always_inline)) {
            int row = PT(tile) * 16 + m; row = row < A.N ? row : A.N - 1;
            int b = (int)((float)row * qinvT);
            const int rem = row - b * qT;
            b += rem >= qT ? 1 : rem < 0 ? -1 : 0;
            const int v = row - b * qT;
            cv = candp[pcand ? (size_t)b * qJ + (int)__umulhi((unsigned)v, qinvM) : (size_t)0];
        };
        int cndA = -1, cndB = -1;
        request_rows(preA, CL(t0));
        if (PRO == PRO_BNRELU) request_cand(cndA, CL(t0));
        if (PRO == PRO_AGG) request_nb(nbA, el[0], CL(t0));
        request_rows(preB, CL(t0 + 4));
        if (PRO == PRO_BNRELU) request_cand(cndB, CL(t0 + 4));
        if (PRO == PRO_AGG && X6_NB_AHEAD == 2) request_nb(nbB, el[1], CL(t0 + 4));
        stage_scale_shift(std::false_type{});
        LDS_BARRIER();
        STAMP(0);
        const float sc0 = s_bn[c4], sc1 = s_bn[c4 + 1], sc2 = s_bn[c4 + 2], sc3 = s_bn[c4 + 3];
        const float sh0 = s_bn[HD + c4], sh1 = s_bn[HD + c4 + 1], sh2 = s_bn[HD + c4 + 2], sh3 = s_bn[HD + c4 + 3];
        auto produce = [&](float4 (&pre)[8], NbRows &nb, int &cv, auto Kc, int s) __attribute__((always_inline)) {
            constexpr int K = decltype(Kc)::value;                // s & 3: the step's ELL slot
            const EllSlot &E = el[K];
            const int tile = t0 + 4 * s;
            if (tile < last) {
                unsigned char *dst = s_tiles + ((s & 1) * 4 + pw) * XT + h * X6_ROWB + j * 8;
                if (PRO == PRO_BNRELU && pcand && lane < 16) s_cand[((s & 1) * 4 + pw) * 16 + lane] = cv;
#pragma unroll
                for (int p = 0; p < 8; p++) {
                    // BatchNorm of a row quad as two packed FMAs, ReLU per element (no packed f32 max on this target)
                    auto bnr4 = [&](const float4 &x, float (&o)[4]) __attribute__((always_inline)) {
                        const f32x2 a = __builtin_elementwise_fma(f32x2{x.x, x.y}, f32x2{sc0, sc1}, f32x2{sh0, sh1});
                        const f32x2 b = __builtin_elementwise_fma(f32x2{x.z, x.w}, f32x2{sc2, sc3}, f32x2{sh2, sh3});
                        o[0] = fmaxf(a[0], 0.f); o[1] = fmaxf(a[1], 0.f); o[2] = fmaxf(b[0], 0.f); o[3] = fmaxf(b[1], 0.f);
                    };
                    float v[4];
                    bnr4(pre[p], v);
                    if (PRO == PRO_AGG) {
                        // gcn:125-149: (A_w @ h) / nnz_row, A_w includes the self loop (1).  Small-integer edge weights, <= 3 terms: an f32
                        // FMA chain — the form k_gin_res uses — is within 2 ulp of the reference's f64-then-cast (the f64 form cost this
                        // producer twelve double-rate instructions per row quad; k_gemm16p, the f32-instruction A/B path, keeps it)
                        const int pp = p < NA ? p : 0;
                        const float wx = __builtin_bit_cast(float, row_pick(__builtin_bit_cast(int, E.vx), p)), wy = __builtin_bit_cast(float, row_pick(__builtin_bit_cast(int, E.vy), p));
                        const float dg = __builtin_bit_cast(float, row_pick(__builtin_bit_cast(int, E.dg), p));
                        const float inv = dg == 1.f ? 1.0f : dg == 2.f ? 0.5f : (1.0f / 3.0f);
                        float n0[4], n1[4];
#ifdef X6_ABL_NB                            // timing ablation (wrong results): MTFJSP_GEMM_DBG & 128: the neighbour rows are requested but not normalised or added
                        if (A.dbg & 128) { asm volatile("" :: "v"(nb.r0[pp].x), "v"(nb.r0[pp].w), "v"(nb.r1[pp].x), "v"(nb.r1[pp].w), "v"(wx), "v"(wy), "v"(inv)); }
                        else
#endif
                        {
                        bnr4(nb.r0[pp], n0); bnr4(nb.r1[pp], n1);
#pragma unroll
                        for (int i = 0; i < 4; i++) v[i] = __builtin_fmaf(wy, n1[i], __builtin_fmaf(wx, n0[i], v[i])) * inv;
                        }
                    }
                    uint2 p0, p1;
                    split2x4m(v, p0, p1);
                    *reinterpret_cast<uint2 *>(dst + p * 2 * X6_ROWB) = p0;
                    *reinterpret_cast<uint2 *>(dst + p * 2 * X6_ROWB + X6_PLANE) = p1;
                }
            }
            // requests (vmcnt retires in order): the ELL entries of step s+3 first — step s+1 turns them into addresses and must not
            // wait for the rows behind them —, then the rows and neighbour rows of step s+2 into the registers this tile just left
            if (PRO == PRO_AGG) fetch_ell(el[(K + 3) & 3], CL(tile + 12));
            if (PRO == PRO_AGG && X6_NB_AHEAD == 1) request_nb(nb, el[(K + 1) & 3], CL(tile + 4));    // (needed first at the next step: ahead of the rows of s+2)
            request_rows(pre, CL(tile + 8));
            if (PRO == PRO_BNRELU) request_cand(cv, CL(tile + 8));
            if (PRO == PRO_AGG && X6_NB_AHEAD == 2) request_nb(nb, el[(K + 2) & 3], CL(tile + 8));
            STAMP(1);
            LDS_BARRIER();
            STAMP(4);
        };
        for (int s = 0; s < nsteps_c; s += 4) {
            produce(preA, nbA, cndA, std::integral_constant<int, 0>{}, s);
            produce(preB, nbB, cndB, std::integral_constant<int, 1>{}, s + 1);
            produce(preA, nbA, cndA, std::integral_constant<int, 2>{}, s + 2);
            produce(preB, nbB, cndB, std::integral_constant<int, 3>{}, s + 3);
        }
        LDS_BARRIER();                                            // the consumers' last step
        }
        }
    } else {
        // ================================ consumer ================================
        const int m = lane & 15, q = lane >> 4;                   // operands swapped: A[col c0+m][k = 8q..], B[k = 8q..][row m], C[col c0+4q+i][row m]
        const int cg = wave;
        float4 wf[2][NP][KS];                                     // [column block][plane][k-step]: pieces of W[32cg + 16c + m][32ks + 8q .. +7] (8 x 16 bit)
        {
            const float4 *wi = reinterpret_cast<const float4 *>(A.Wx6) + (size_t)cg * (2 * NP * KS * 64) + lane;
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int p = 0; p < NP; p++)
#pragma unroll
                    for (int ks = 0; ks < KS; ks++) wf[c][p][ks] = wi[((c * NP + p) * KS + ks) * 64];
        }
        const float wsinv = PRO == PRO_GIN0 ? 1.0f : A.w_sinv;
        f32x4 biasv[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        if (A.bias) {
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const float4 b = *reinterpret_cast<const float4 *>(A.bias + 32 * cg + 16 * c + 4 * q);
                biasv[c] = f32x4{b.x, b.y, b.z, b.w};
            }
        }
        stage_scale_shift(std::true_type{});
        LDS_BARRIER();
        STAMP(0);
        f32x2 ts[2][2], tq[2][2];                                 // per-lane column sums (row m of the tiles multiplied so far), two columns per register pair
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int i = 0; i < 2; i++) { ts[c][i] = f32x2{0.f, 0.f}; tq[c][i] = f32x2{0.f, 0.f}; }
        const f32x4 wsinv4 = {wsinv, wsinv, wsinv, wsinv};
        const unsigned char *xa0 = s_tiles + m * X6_ROWB + 16 * q; // operand fragment (slot t, plane p, k-step ks): + t*XT + p*X6_PLANE + 64*ks
        unsigned char *s_tr = smem + X6_TR_OFF + cg * 2 * X6_TRB;  // this wave's two output transposition buffers
        // ---- pooling epilogue (PRO_BNRELU, A.pooled): BatchNorm + ReLU of the output tile in registers; a lane keeps the running column sums of ITS row slot
        // (m) for the instance the wave is in, the 16 slots are folded and added to pooled[] when the instance changes (and at the end of the range)
        const bool pool = PRO == PRO_BNRELU && A.pooled != nullptr;
        f32x4 sc2[2], sh2[2], psum[2];
        int cur = 0, cur_lo = 0, cur_hi = 0;                        // the instance the wave is in and its rows [cur_lo, cur_hi) (wave-uniform)
        const int pT = A.T > 0 ? A.T : 1, pJ = A.pool_J, pM = pJ > 0 ? pT / pJ : pT;
        const float invTf = 1.0f / (float)pT;
        const unsigned invMu = (unsigned)((0x100000000ull + (unsigned)pM - 1) / (unsigned)pM);      // v / M = umulhi(v, invMu) for v < 65536
#pragma unroll
        for (int c = 0; c < 2; c++) { sc2[c] = f32x4{0.f, 0.f, 0.f, 0.f}; sh2[c] = sc2[c]; psum[c] = sc2[c]; }
        if (pool) {
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const float4 a = *reinterpret_cast<const float4 *>(s_bn2 + 32 * cg + 16 * c + 4 * q), b = *reinterpret_cast<const float4 *>(s_bn2 + HD + 32 * cg + 16 * c + 4 * q);
                sc2[c] = f32x4{a.x, a.y, a.z, a.w}; sh2[c] = f32x4{b.x, b.y, b.z, b.w};
            }
        }
        // row -> (instance, row inside it) for rows < 2^24: the f32 quotient is off by at most one
        auto inst_of = [&](int row, int &v) __attribute__((always_inline)) {
            int b = (int)((float)row * invTf);
            int rem = row - b * pT;
            b += rem >= pT ? 1 : rem < 0 ? -1 : 0;
            v = row - b * pT;
            return b;
        };
        auto pool_flush = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const float t = row_sum16(psum[c][i]);
                    if (m == 0) atomicAdd(&A.pooled[(size_t)cur * HD + 32 * cg + 16 * c + 4 * q + i], t * invTf);     // sparse mm with 1/T entries (gcn:192)
                }
        };
        if (pool && nsteps > 0) {                                 // the instance of the first row this wave meets (rev: the last valid row of its last tile)
            int r0 = PT(first) * 16 + (A.rev ? 15 : 0); r0 = r0 < A.N ? r0 : A.N - 1;
            int v0;
            cur = __builtin_amdgcn_readfirstlane(inst_of(r0, v0));
            cur_lo = cur * pT; cur_hi = cur_lo + pT;
        }
        LDS_BARRIER();                                            // step 0: the producers fill buffer 0
        STAMP(4);
        for (int s = 1; s <= nsteps_c; s++) {
            if (s > nsteps) { LDS_BARRIER(); continue; }           // padding step: nothing was produced
            int cnd[4] = {-1, -1, -1, -1};                         // the candidates of this step's rows (from the producers, with the planes)
            if (pool && A.cand_feat) {
#pragma unroll
                for (int t = 0; t < 4; t++) cnd[t] = s_cand[(((s - 1) & 1) * 4 + t) * 16 + m];
            }
            const int tb = first + 4 * (s - 1);
            const unsigned char *xa = xa0 + ((s - 1) & 1) * 4 * XT;
            auto tiles4 = [&](auto FULLc) __attribute__((always_inline)) {
                constexpr bool FULL = decltype(FULLc)::value;    // FULL: all four tiles exist and none holds rows >= N
                float4 xf[3][NP];                                 // fragments of (tile, k-step) units u, u+1, u+2: two units of LDS latency cover
                constexpr int NU = 4 * KS;
#pragma unroll
                for (int p = 0; p < NP; p++) xf[0][p] = *reinterpret_cast<const float4 *>(xa + p * X6_PLANE);
#pragma unroll
                for (int p = 0; p < NP; p++) xf[1][p] = *reinterpret_cast<const float4 *>(xa + (1 / KS) * XT + p * X6_PLANE + 64 * (1 % KS));
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    if (!FULL && tb + t >= last) break;
                    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                    for (int ks = 0; ks < KS; ks++) {
                        const int u = t * KS + ks;
                        if (u + 2 < NU) {                         // a stale slot beyond the last tile is read but never used
                            const int tn = (u + 2) / KS, kn = (u + 2) % KS;
#pragma unroll
                            for (int p = 0; p < NP; p++) xf[(u + 2) % 3][p] = *reinterpret_cast<const float4 *>(xa + tn * XT + p * X6_PLANE + 64 * kn);
                        }
                        const float4 *x = xf[u % 3];
                        if constexpr (NP == 3) {
                            auto MB = [&](int wp, int xp) __attribute__((always_inline)) {
#pragma unroll
                                for (int c = 0; c < 2; c++)
                                    acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[c][wp][ks]), __builtin_bit_cast(bf16x8, x[xp]), acc[c], 0, 0, 0);
                            };
                            MB(0, 2); MB(2, 0); MB(1, 1); MB(0, 1); MB(1, 0); MB(0, 0);     // smallest terms first
                        } else {
                            auto MH = [&](int wp, int xp) __attribute__((always_inline)) {
#pragma unroll
                                for (int c = 0; c < 2; c++)
                                    acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, wf[c][wp][ks]), __builtin_bit_cast(h16x8, x[xp]), acc[c], 0, 0, 0);
                            };
                            MH(0, 1); MH(1, 0); MH(0, 0);
                        }
                    }
                    MFMA_SETTLE2(acc[0], acc[1]);
#ifdef MTFJSP_STAMP_TILE                       // diagnostic split of the consumer's tile: 5 = fragments + products, 1 = epilogue up to the LDS write + sums, 6 = LDS read-back + stores issued
                    STAMP(5);
#endif
                    // (one fused operation per pair: wsinv is a power of two, the product is exact either way — the same bits as a multiplication and an addition)
#pragma unroll
                    for (int c = 0; c < 2; c++) acc[c] = __builtin_elementwise_fma(acc[c], wsinv4, biasv[c]);
                    const int ptile = FULL || tb + t < last ? PT(tb + t) : 0;
                    const int row = ptile * 16 + m;
                    const bool ok = FULL || row < A.N;
                    if (pool) {                                   // (workgroup-uniform)
                        f32x4 hv[2];
#pragma unroll
                        for (int c = 0; c < 2; c++) {
                            const f32x4 y = __builtin_elementwise_fma(acc[c], sc2[c], sh2[c]);
                            hv[c] = f32x4{fmaxf(y[0], 0.f), fmaxf(y[1], 0.f), fmaxf(y[2], 0.f), fmaxf(y[3], 0.f)};
                        }
                        const int R0 = ptile * 16;
                        int b = cur, v = row - cur_lo;
                        if (FULL && R0 >= cur_lo && R0 + 16 <= cur_hi) {             // the whole tile inside the current instance (scalar test): plain sums
#pragma unroll
                            for (int c = 0; c < 2; c++) psum[c] = psum[c] + hv[c];
                        } else {
                            // T >= 16: a tile holds rows of the current instance and of at most one other — the next one in processing order
                            const bool in_cur = row >= cur_lo && row < cur_hi;
                            const float k0 = (ok && in_cur) ? 1.0f : 0.0f;
#pragma unroll
                            for (int c = 0; c < 2; c++) psum[c] = __builtin_elementwise_fma(hv[c], f32x4{k0, k0, k0, k0}, psum[c]);
                            const bool other = A.rev ? R0 < cur_lo : (R0 + 16 > cur_hi && cur_hi < A.N);     // (wave-uniform; valid rows of it exist)
                            if (other) {
                                pool_flush();
                                cur += A.rev ? -1 : 1; cur_lo = cur * pT; cur_hi = cur_lo + pT;
                                const float k1 = (ok && !in_cur) ? 1.0f : 0.0f;
#pragma unroll
                                for (int c = 0; c < 2; c++) psum[c] = hv[c] * f32x4{k1, k1, k1, k1};
                                if (!in_cur) { b = cur; v = row - cur_lo; }
                            }
                        }
                        if (A.cand_feat && ok && cnd[t] == v) {                                      // candidate gather (ac:197-207)
                            float *d = A.cand_feat + ((size_t)b * pJ + (int)__umulhi((unsigned)v, invMu)) * HD + 32 * cg + 4 * q;
#pragma unroll
                            for (int c = 0; c < 2; c++) *reinterpret_cast<float4 *>(d + 16 * c) = make_float4(hv[c][0], hv[c][1], hv[c][2], hv[c][3]);
                        }
                        continue;
                    }
                    // stores as WHOLE 128-byte lines: the accumulator layout gives a lane two 16-byte chunks (q and 4 + q) of row m's 128
                    // bytes of this wave, i.e. 16 rows x 64 bytes per store instruction — measured 2.35 TB/s against 3.24 TB/s for 8
                    // rows x 128 bytes (tools/ubench/store_patterns.hip).  The wave's tile goes through a private LDS buffer
                    // (144-byte row pitch: conflict-free both ways; LDS executes a wave's instructions in order).
                    unsigned char *trb = s_tr + (t & 1) * X6_TRB;
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        const f32x4 v = acc[c];
                        if (A.out) *reinterpret_cast<float4 *>(trb + m * 144 + 16 * (4 * c + q)) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
                        for (int i = 0; i < 2; i++) {                 // two-wide: the same additions
                            const f32x2 x = {ok ? v[2 * i] : 0.f, ok ? v[2 * i + 1] : 0.f};
                            ts[c][i] = ts[c][i] + x; tq[c][i] = __builtin_elementwise_fma(x, x, tq[c][i]);
                        }
                    }
#ifdef MTFJSP_STAMP_TILE
                    STAMP(1);
#endif
                    // (out == NULL: the statistics-only pass in front of k_gemm_x6f, mtfjsp_gemm_pair.h — the product is formed, only its
                    // BatchNorm column sums leave the chip)
                    if (A.out) {
#pragma unroll
                    for (int i = 0; i < 2; i++) {
                        const int r8 = 8 * i + (lane >> 3);
                        const float4 v = *reinterpret_cast<const float4 *>(trb + r8 * 144 + 16 * (lane & 7));
                        *reinterpret_cast<float4 *>(A.out + ((size_t)ptile * 16 + r8) * HD + 32 * cg + 4 * (lane & 7)) = v;
                    }
                    }
#ifdef MTFJSP_STAMP_TILE
                    STAMP(6);
#endif
                }
            };
            if (tb + 4 <= last && ((A.rev ? PT(tb) : tb + 3) + 1) * 16 <= A.N) tiles4(std::true_type{});
            else tiles4(std::false_type{});
            STAMP(5);
            // column sums: 16 rows (lanes m) -> one value per column, f64 from there on; every 4th step (f32 partial sums of
            // <= 16 values per lane) and after the last one
            if ((s & 3) == 0 || s == nsteps) {
#pragma unroll
                for (int c = 0; c < 2; c++)
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const float a = row_sum16(ts[c][i >> 1][i & 1]), b = row_sum16(tq[c][i >> 1][i & 1]);
                        if (m == 0) { atomicAdd(&s_stat[32 * cg + 16 * c + 4 * q + i], (double)a); atomicAdd(&s_stat[HD + 32 * cg + 16 * c + 4 * q + i], (double)b); }
                        ts[c][i >> 1][i & 1] = 0.f; tq[c][i >> 1][i & 1] = 0.f;
                    }
            }
            STAMP(6);
            LDS_BARRIER();                                        // buffer (s-1)&1 may be overwritten; buffer s&1 is complete
            STAMP(7);
        }
        if (pool && nsteps > 0) pool_flush();
    }
    if (A.epi_stats && tid < 2 * HD) atomicAdd(&A.epi_stats[(blockIdx.x % STAT_REP) * 256 + tid], s_stat[tid]);
#ifdef MTFJSP_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_last)::"memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1)::"memory");
    ph[2] = t_last - t_first; ph[3] = rt1 - rt0;
    if (A.stamps && lane == 0)
        for (int i = 0; i < 8; i++) A.stamps[((size_t)blockIdx.x * 8 + wave) * 8 + i] = ph[i];
#endif
}
static size_t gemm_x6_lds_bytes() { return (size_t)X6_TR_OFF + 4 * 2 * X6_TRB; }

#include "mtfjsp_gemm_pair.h"
#define MTFJSP_GIN_RES_DECL_ONLY          // (the kernel itself: mtfjsp_gin_res.hip, a translation unit with its own scheduling strategy)
#include "mtfjsp_gin_resident.h"
// the grouped environment step as a device function (k_headsx_envstep below)
#define MTFJSP_ENV_GRP_NO_KERNELS
#include "mtfjsp_env_dev.h"
#include "mtfjsp_env_grp.h"

// ---------------------------------------------------------------------------------------------
// Machine path of the machine actor / global critic (ac:383-434, gat:82-159) in ONE launch.  The three applications of
// the single shared GATLayer on the fixed 2-node graph [[1,1],[0,1]] couple only rows (2u, 2u+1) = (node 0, node 1) of
// machine u, which lie in the same 16-row tile, so a wave keeps its tile in LDS across all three passes:
//   rows   : node 0 = m_fea_1_fcl(m_fea1[u]) (6 -> 128), node 1 = m_fea_2_fcl(m_fea2[u]) (8 -> 128), no bias (ac:383-384)
//   pass   : z = rows x W (matrix cores);  e0v = LeakyReLU_0.2(a_src.z0 + a_dst.zv);  (al0, al1) = softmax(e00, e01);
//            n0' = al0 z0 + al1 z1 ; n1' = z1 ;  ELU after passes 1 and 2, written back into the LDS tile (ac:409-413)
//   end    : mean of the two nodes (ac:420) -> node[u] and the column sums of the BatchNorm that follows (ac:434)
// W^T is staged once and no intermediate ever leaves the CU.  In the C layout accumulator registers (0,1) and (2,3) of a
// lane are (z0, z1) of machines 2q and 2q+1 of the tile, and the 128-column dot products reduce over the 16 lanes of a
// DPP row.
struct GatArgs {
    int R;                  // machine rows = B*M ; tile rows = 2R
    const void *f1, *f2;    // m_fea1 [R,6], m_fea2 [R,8] (obs dtype)
    int feat_f64;
    const float *W1, *W2;   // (m_fea_1_fcl.weight^T . gat W)^T [128,6], (m_fea_2_fcl.weight^T . gat W)^T [128,8]: input projection and
                            // the first pass' h W fused on the host (one 14 x 128 x 128 product per weight load)
    const float *Wq;        // the two of them as the f32 matrix instruction's B operands (k_gat3x): [c 8][lane 64][ks 4] = W_cat[4 ks + (lane >> 4)][16 c + (lane & 15)],
                            // W_cat = [W1^T (6 rows) ; 0 0 ; W2^T (8 rows)] (16 x 128)
    const float *Wt;        // gat_layer.W [in,out]
    const void *Wx6;        // the same (scaled by a power of two) as f16 x 2-plane operand fragments [c 8][plane 2][ks 4][lane 64][8] (k_gat3x)
    float w_sinv;           // 1 / that scale
    const float *gat_a;     // [256] a_src | a_dst (gat:68-79)
    float *node;            // [R,128] (padded) pre-BatchNorm node mean
    double *epi_stats;
    unsigned long long *stamps;
};
__global__ __launch_bounds__(512) void k_gat3(GatArgs A)
{
    extern __shared__ __align__(16) unsigned char smem[];
    float *s_w = reinterpret_cast<float *>(smem);                 // 128*128, swizzled
    float *s_a = s_w + HD * HD;                                   // 8 * 16 * LDA16
    float *s_bn = s_a + 8 * 16 * LDA16;
    float *s_feat = s_bn + 2 * HD + 16;                           // 8 waves * 16 rows * 8
    double *s_red = reinterpret_cast<double *>(s_a);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int j = lane & 31, h = lane >> 5, c4 = j * 4;
    const int m = lane & 15, q = lane >> 4, qo = q & 1;
#ifdef MTFJSP_STAMP
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_last)::"memory");
#endif
    stage_w16(s_w, A.Wt, tid);
    __syncthreads();
    STAMP(0);
    float *my_a = s_a + wave * 16 * LDA16;
    float *my_f = s_feat + wave * 128;
    const int N = 2 * A.R;
    const int ntiles = (N + 15) / 16;
    const int per = (ntiles + gridDim.x - 1) / gridDim.x;
    const int first = blockIdx.x * per;
    const int last = first + per < ntiles ? first + per : ntiles;
    double st_sum[8], st_sq[8];
    for (int c = 0; c < 8; c++) { st_sum[c] = 0; st_sq[c] = 0; }
    float wf[4][8];                                               // this lane's 4 output columns of W1 (h = 0) or W2 (h = 1)
    for (int x = 0; x < 4; x++)
        for (int k = 0; k < 8; k++) wf[x][k] = h == 0 ? (k < 6 ? A.W1[(c4 + x) * 6 + k] : 0.f) : A.W2[(c4 + x) * 8 + k];
    float asrc[8], adst[8];
    for (int c = 0; c < 8; c++) { asrc[c] = A.gat_a[c * 16 + m]; adst[c] = A.gat_a[HD + c * 16 + m]; }
    // lane L < 32 fetches 4 of the 128 feature words of a tile: row L/2 = (machine, node), words 4*(L&1)..+3
    auto fetch_feat = [&](int tile) __attribute__((always_inline)) -> float4 {
        const int r = tile * 16 + (lane >> 1), k0 = (lane & 1) * 4;
        float x[4] = {0.f, 0.f, 0.f, 0.f};
        if (lane < 32 && r < N) {
            const int u = r >> 1, node = r & 1, width = node ? 8 : 6;
            for (int k = 0; k < 4; k++)
                if (k0 + k < width) {
                    const size_t idx = (size_t)u * width + k0 + k;
                    x[k] = A.feat_f64 ? (float)reinterpret_cast<const double *>(node ? A.f2 : A.f1)[idx]
                                      : reinterpret_cast<const float *>(node ? A.f2 : A.f1)[idx];
                }
        }
        return make_float4(x[0], x[1], x[2], x[3]);
    };
    // static round-robin inside the CU's contiguous range: waves w and w+4 share a SIMD, so a remainder of <= 4 tiles
    // lands on four different SIMDs
    int t_cur = first + wave, t_n1 = t_cur + 8;
    float4 fpre = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t_cur < last) fpre = fetch_feat(t_cur);
    const float *ap = my_a + m * LDA16 + q;
    const float *bp = s_w + q * HD + m;
    int bo[8];
    for (int c = 0; c < 8; c++) bo[c] = (c ^ qo) * 16;
    STAMP(1);
    while (t_cur < last) {
        const int row0 = t_cur * 16;
        // ---- input rows: tile rows 2p+h are node h of machine (row0/2 + p)
        if (lane < 32) *reinterpret_cast<float4 *>(my_f + lane * 4) = fpre;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int p = 0; p < 8; p++) {
            const int r = 2 * p + h;
            const float4 fa = *reinterpret_cast<const float4 *>(my_f + r * 8), fb = *reinterpret_cast<const float4 *>(my_f + r * 8 + 4);
            const float ff[8] = {fa.x, fa.y, fa.z, fa.w, fb.x, fb.y, fb.z, fb.w};
            float o[4];
#pragma unroll
            for (int x = 0; x < 4; x++) {
                float a = 0.f;
#pragma unroll
                for (int k = 0; k < 8; k++) a = fmaf(ff[k], wf[x][k], a);
                o[x] = a;
            }
            float *d = my_a + r * LDA16 + c4;
            *reinterpret_cast<float2 *>(d) = make_float2(o[0], o[1]);
            *reinterpret_cast<float2 *>(d + 2) = make_float2(o[2], o[3]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (t_n1 < last) fpre = fetch_feat(t_n1);
        STAMP(2);
#pragma unroll 1
        for (int pass = 0; pass < 3; pass++) {
            f32x4 acc[8];
            if (pass == 0) {
                // the rows just written ARE z of the first pass: W1/W2 arrive pre-multiplied with the GAT weight (GatArgs), so
                // the 6/8 -> 128 projection and the first h W product are one K = 8 product — read it back in the C layout
#pragma unroll
                for (int c = 0; c < 8; c++)
#pragma unroll
                    for (int i = 0; i < 4; i++) acc[c][i] = my_a[(4 * q + i) * LDA16 + c * 16 + m];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else {
#pragma unroll
                for (int c = 0; c < 8; c++)
#pragma unroll
                    for (int i = 0; i < 4; i++) acc[c][i] = 0.f;
                mfma_tile16(ap, bp, bo, acc);
                asm volatile("" ::: "memory");
            }
            STAMP(3);
#pragma unroll
            for (int u = 0; u < 2; u++) {                                     // the lane's two machines: tile rows 4q+2u (node 0), +1 (node 1)
                const int i = 2 * u;
                float s0 = 0.f, d0 = 0.f, d1 = 0.f;
#pragma unroll
                for (int c = 0; c < 8; c++) { const float z0 = acc[c][i], z1 = acc[c][i + 1]; s0 += asrc[c] * z0; d0 += adst[c] * z0; d1 += adst[c] * z1; }
                s0 = row_sum16(s0); d0 = row_sum16(d0); d1 = row_sum16(d1);
                float e00 = s0 + d0, e01 = s0 + d1;
                e00 = e00 > 0.f ? e00 : 0.2f * e00;
                e01 = e01 > 0.f ? e01 : 0.2f * e01;
                const float mx = fmaxf(e00, e01);
                const float x0 = __expf(e00 - mx), x1 = __expf(e01 - mx);
                const float inv = 1.0f / (x0 + x1);
                const float al0 = x0 * inv, al1 = x1 * inv;
                const int r = 4 * q + i;
                if (pass < 2) {
#pragma unroll
                    for (int c = 0; c < 8; c++) {
                        const float z0 = acc[c][i], z1 = acc[c][i + 1];
                        float n0 = al0 * z0 + al1 * z1, n1 = z1;
                        n0 = n0 > 0.f ? n0 : __expf(n0) - 1.0f;               // ELU after passes 1 and 2 (ac:409-413); |err| < 2e-7
                        n1 = n1 > 0.f ? n1 : __expf(n1) - 1.0f;
                        my_a[r * LDA16 + c * 16 + m] = n0;
                        my_a[(r + 1) * LDA16 + c * 16 + m] = n1;
                    }
                } else {
                    const bool valid = row0 + r < N;
                    float *nd = A.node + (size_t)((row0 + r) >> 1) * HD + m;
#pragma unroll
                    for (int c = 0; c < 8; c++) {
                        const float z0 = acc[c][i], z1 = acc[c][i + 1];
                        float mv = (al0 * z0 + al1 * z1 + z1) * 0.5f;        // mean over the 2 nodes (ac:420)
                        nd[c * 16] = mv;
                        if (!valid) mv = 0.f;
                        st_sum[c] += (double)mv; st_sq[c] += (double)mv * (double)mv;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the rewritten tile is complete before the next pass reads it
            STAMP(4);
        }
        t_cur = t_n1; t_n1 += 8;
    }
    flush_stats16(s_red, A.epi_stats, st_sum, st_sq, tid, wave, m, q);
#ifdef MTFJSP_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(6);
    if (A.stamps && lane == 0)
        for (int i = 0; i < 8; i++) A.stamps[((size_t)blockIdx.x * 8 + wave) * 8 + i] = ph[i];
#endif
}

// ---------------------------------------------------------------------------------------------
// k_gat3x — k_gat3 with passes 2 and 3 on the 16-bit matrix cores at f32 accuracy (2-way f16 split, 3 piece products, weight
// pre-scaled by a power of two: see k_gemm_x6 / DESIGN.md §4).  Eight waves per workgroup, each keeping its own 16-row tile
// through all three passes:
//   * the GAT weight lives in LDS as f16 fragments in operand order [column block 8][plane 2][k-step 4][lane 64][16 B]
//     (64 KiB, conflict-free 16-byte reads), streamed two column blocks at a time with the next pair in flight;
//   * the tile's activations are the A operand, held in registers: after each pass's epilogue wrote the new rows to the
//     wave's f32 LDS tile (the C layout -> operand layout transpose, XOR-swizzled), they are read back as 8 x 16 bytes
//     per lane and split into 2 planes x 4 k-steps of fragments (32 registers);
//   * the accumulators stay in the ordinary C layout (rows 4q+i in the lane, columns across the 16 lanes of a DPP row), so
//     the attention epilogue is k_gat3's.
// 2 passes x 8 column blocks x 12 products x 16 cycles = 3.1 k cycles per tile instead of 16.4 k with the f32 instruction.
// f32 transpose tile without padding: 16-byte chunk index XOR (row & 7) makes the row-wise 16-byte operand reads, the
// C-layout word accesses and the row writes conflict-free
__device__ __forceinline__ int gx_off(int row, int col) { return row * HD + ((((col >> 2) ^ (row & 7)) << 2) | (col & 3)); }
#ifndef MTFJSP_BODY_FUNCS
#define MTFJSP_BODY_FUNCS 0   // diagnostic builds: bit 0 = the GAT statements, bit 1 = the heads statements as __forceinline__ functions
#endif
// Grid-wide exchange of the machine nodes' BatchNorm sums INSIDE a launch (k_headsx_gat3x_headsx below): the body files take these
// switches from the kernel that includes them — GAT_XCHG: the GAT statements publish their column sums as count-carrying fixed-point
// words instead of f64 atomics; HX_XCHG: the heads statements collect them from those words instead of reading finished sums.
#define GAT_XCHG 0
#define HX_XCHG 0
#define GAT_PRESTAGED 0
#define BODY_TID threadIdx.x              // (k_headsx_gat3x_headsx gives each of its three parts an opaque copy: left alone, hipcc keeps the thread-index
                                          // expressions common to the parts alive across all of them — through scratch memory where registers run out)
#ifndef GAT_PAIRED
#define GAT_PAIRED 1                        // the three-in-one launch takes a wave's two GAT tiles together (mtfjsp_gat3x_body.h); 0: one after the other
#endif
struct XchgArgs {
    unsigned long long *words;       // this forward's words (zero on entry): [fine | wide][8 dispatch groups][sum | sumsq][128 columns]
    unsigned long long *words_next;  // the next forward's set: zeroed by this launch
    unsigned nblk;                   // workgroups of the launch (all co-resident: the host checked the grid against the census)
    double fine_limit;               // contributions below this magnitude travel in the fine words (2^31; lower only in tests: MTFJSP_XCHG_FINE_LIMIT)
    unsigned *fail, *range_flag;     // host-mapped words: a wait timed out / a contribution was not a number
    unsigned long long *stamps;      // diagnostic build only (-DMTFJSP_STAMP3): [workgroups][8 waves][8] s_memrealtime
};
#define XW_SET (2 * 8 * 256)
#ifdef MTFJSP_STAMP3
#define X3_RT(i) do { if (XA.stamps && (threadIdx.x & 63) == 0) { __builtin_amdgcn_sched_barrier(0); XA.stamps[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define X3_RT(i) do { } while (0)
#endif
#define H3S_RT(i) do { } while (0)
#define H3T_RT(i) do { } while (0)
#define H3_RT(i) do { } while (0)                                  // (the heads statements' phase boundaries; live only inside k_headsx_gat3x_headsx of the -DMTFJSP_STAMP3 build)
#if MTFJSP_BODY_FUNCS & 1
__device__ __forceinline__ void gat3x_body(const GatArgs &A, unsigned char *smem)
{
#include "mtfjsp_gat3x_body.h"
}
#endif
__global__ __launch_bounds__(512) void k_gat3x(GatArgs A)
{
    extern __shared__ __align__(16) unsigned char smem[];
#if MTFJSP_BODY_FUNCS & 1
    gat3x_body(A, smem);
#else
#include "mtfjsp_gat3x_body.h"
#endif
}
static size_t gat3x_lds_bytes() { return (size_t)8 * 2 * 4 * 64 * 16 + (size_t)8 * 16 * HD * 4; }

// ---------------------------------------------------------------------------------------------
// Fused actor heads (ac:205-293 / ac:444-495): for a group of 16 instances (= R 16-row tiles of scorer rows, since an
// instance has R candidates / machines) one 8-wave workgroup computes
//   u   = Wb pooled + Wc other + b0                      (the per-instance thirds of the 384-wide scorer input)
//   s1  = tanh(Wa x_row + u[instance]) ; s2 = tanh(W1 s1 + b1) ; score = scale * (w2 . s2 + b2) ; masked softmax
//   c1  = tanh(Wc0 pooled + bc0) ; c2 = tanh(Wc1 c1 + bc1) ; value = Wc2 c2 + bc2
// Work is split by OUTPUT COLUMN: wave w owns columns 16w..16w+15 of every product, so its B operand — one 128x16 column
// block of a weight, 32 registers — is loaded from global memory straight into registers (no LDS staging, no commit
// barrier, the next weight is requested while the current one is multiplied), is reused for every tile, and the eight
// waves load the MFMA pipes evenly (512 products each).  Only activations live in LDS: every A tile is read by all waves
// and every intermediate is written back as the next product's A tile (three barriers per chunk of 8 tiles).
// Philox4x32-10 (counter-based; Salmon et al. 2011) and the categorical / greedy pick shared by k_sample and the fused
// selection at the end of k_heads (agent:22-72): p[0..n) with stride 1.
__device__ __forceinline__ void philox4x32(uint32_t c[4], uint32_t k0, uint32_t k1)
{
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
// the uniform number of instance b's draw (the part of pick_action that does not depend on the probabilities: k_headsx forms it early)
__device__ __forceinline__ float pick_uniform(int b, uint64_t seed, uint64_t counter)
{
    uint32_t c[4] = {(uint32_t)b, (uint32_t)counter, (uint32_t)(counter >> 32), 0x73616d70u};
    philox4x32(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    return (float)(c[0] >> 8) * (1.0f / 16777216.0f);      // [0,1)
}
// pick_action with the uniform number given (the same comparisons in the same order)
__device__ __forceinline__ int pick_action_u(const float *p, int n, int greedy, float u)
{
    int pick = 0;
    if (greedy) {
        float best = p[0];
        for (int i = 1; i < n; i++) if (p[i] > best) { best = p[i]; pick = i; }
    } else {
        float tot = 0.f;
        for (int i = 0; i < n; i++) tot += p[i];
        float acc = 0.f;
        const float thr = u * tot;
        pick = -1;
        int last = 0;
        for (int i = 0; i < n; i++) {
            if (p[i] > 0.f) {
                last = i;
                acc += p[i];
                if (pick < 0 && thr < acc) pick = i;
            }
        }
        if (pick < 0) pick = last;
    }
    return pick;
}
__device__ __forceinline__ int pick_action(const float *p, int n, int b, int greedy, uint64_t seed, uint64_t counter)
{
    int pick = 0;
    if (greedy) {
        float best = p[0];
        for (int i = 1; i < n; i++) if (p[i] > best) { best = p[i]; pick = i; }
    } else {
        uint32_t c[4] = {(uint32_t)b, (uint32_t)counter, (uint32_t)(counter >> 32), 0x73616d70u};
        philox4x32(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        const float u = (float)(c[0] >> 8) * (1.0f / 16777216.0f);      // [0,1)
        float tot = 0.f;
        for (int i = 0; i < n; i++) tot += p[i];
        float acc = 0.f;
        const float thr = u * tot;
        pick = -1;
        int last = 0;
        for (int i = 0; i < n; i++) {
            if (p[i] > 0.f) {
                last = i;
                acc += p[i];
                if (pick < 0 && thr < acc) pick = i;
            }
        }
        if (pick < 0) pick = last;
    }
    return pick;
}

struct HeadArgs {
    int B, R;
    int hg;                              // instances per workgroup of k_headsx: HG (16), or 8 where 16 would leave half the CUs without a workgroup (the LDS layout is HG's either way)
    const float *X;                      // [B*R,128] candidate / machine node embeddings
    const float *pooled, *other;         // [B,128]
    const float *W0i;                    // register images (mtfjsp_encoder::wimg) of the 3 blocks of linears.0: X | pooled | other
    const float *b0, *W1i, *b1, *w2, *b2;
    const float *Wc0i, *bc0, *Wc1i, *bc1, *wc2, *bc2;
    const void *W0x, *W1x, *Wc0x, *Wc1x; // the same weights as f16 x 2-plane register images (mtfjsp_encoder::wx6), k_headsx
    float sW0, sW1, sWc0, sWc1;          // 1 / the power-of-two scale folded into each of those images
    const uint8_t *mask;                 // [B,R]
    float scale;
    float *prob, *value;                 // [B,R], [B,2]
    // machine actor: X arrives pre-BatchNorm (k_gat3 output); normalise it while staging, pool it over the R rows of an
    // instance here (ac:434-444) and publish the pooled embedding — no separate normalisation pass over `node`
    const double *xbn_stats; const float *xbn_gamma, *xbn_beta; double xbn_inv_rows; float *pooled_out;
    // job actor: X is the last GIN product z [B*xT,128] (pre-BatchNorm); row (instance, r) of the scorer input is
    // relu(bn(z[instance*xT + xgather[instance*R + r]])) (candidate gather, ac:197-207) and the pooled embedding the mean of
    // relu(bn(.)) over all xT rows of the instance (gcn:192) — k_job_pool_gather folded in.  xgather == NULL: rows instance*R + r.
    const int *xgather; int xT, xrelu;
    // optional fused action selection (agent:22-72), same Philox stream as k_sample: 0 = off, 1 = sample, 2 = greedy
    int sample_mode; unsigned long long seed, counter; int *idx_out; float *logp_out; const int *gather_from; int *gathered_out;
    mtfjsp_mfea1_ctx_t mf; int mf_on;    // optional: m_fea1 / machine mask of the selected task (pe:152-214), see include/mtfjsp.h
    double *zero_stats; int zero_count;  // BatchNorm accumulators no kernel reads any more: zeroed here for the next forward
    double *zero_stats2; int zero_count2;
    unsigned long long *stamps;
    unsigned *range_flag;                  // host-mapped word, set when an output is not a number (an f16 operand piece overflowed somewhere upstream); NULL on the f32-instruction path
};
#define HG 16                            // instances per group
#define HCH 6                            // scorer tiles per chunk (even; X tiles and s1 tiles are separate LDS buffers)
#define HX_NPART 32                      // k_headsx: partial scores per (wave, row quarter) — 32 — or per wave with a cross-lane reduction per tile — 8 (HCH 10: LDS)

__global__ __launch_bounds__(512) void k_heads(HeadArgs A)
{
    extern __shared__ __align__(16) unsigned char smem[];
    float *s_x = reinterpret_cast<float *>(smem);                  // HCH tiles of 16 x LDA16: X rows
    float *s_s = s_x + HCH * 16 * LDA16;                           // HCH tiles: s1
    float *s_p = s_s + HCH * 16 * LDA16;                           // pooled tile
    float *s_o = s_p + 16 * LDA16;                                 // other tile
    float *s_c1 = s_o + 16 * LDA16;                                // c1 tile
    float *s_c2 = s_c1 + 16 * LDA16;                               // c2 tile
    float *s_u = s_c2 + 16 * LDA16;                                // [16][128]
    float *s_part = s_u + HG * HD;                                 // 8 waves * (HCH*16) rows
    float *s_score = s_part + 8 * HCH * 16;                        // HG * 64
    float *s_wc2 = s_score + HG * 64;                              // 2 * 128
    unsigned char *s_mask = reinterpret_cast<unsigned char *>(s_wc2 + 2 * HD);   // HG * 64
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int m = lane & 15, q = lane >> 4;
    const int col = 16 * wave + m;
#ifdef MTFJSP_STAMP
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_last)::"memory");
#endif
    const int R = A.R;
    const unsigned invR = (unsigned)((0x100000000ull + (unsigned)R - 1) / (unsigned)R);
    // this wave's column block of a weight, b[s] = W^T[k = 4s+q][col], from its register image: 8 coalesced 16-byte loads
#define WCOL(dst, Wi)                                                                              \
    do {                                                                                           \
        const float4 *w_ = reinterpret_cast<const float4 *>(Wi) + (size_t)wave * 8 * 64 + lane;    \
        _Pragma("unroll") for (int g_ = 0; g_ < 8; g_++) {                                         \
            const float4 v_ = w_[g_ * 64];                                                         \
            dst[4 * g_] = v_.x; dst[4 * g_ + 1] = v_.y; dst[4 * g_ + 2] = v_.z; dst[4 * g_ + 3] = v_.w; \
        }                                                                                          \
    } while (0)
    // two 16-row tiles x this wave's column block: two independent chains of 32 products (fully unrolled: b[] stays in registers)
#define TILE2(acc0, acc1, t0p, t1p, bw)                                                                          \
    do {                                                                                                         \
        _Pragma("unroll") for (int s_ = 0; s_ < 32; s_++) {                                                      \
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32((t0p)[4 * s_], bw[s_], acc0, 0, 0, 0);                   \
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32((t1p)[4 * s_], bw[s_], acc1, 0, 0, 0);                   \
        }                                                                                                        \
        MFMA_SETTLE2(acc0, acc1);                                                                                \
    } while (0)
    const int sr = tid >> 5, sc4 = (tid & 31) * 4;                  // staging: thread -> (row sr of 16, 4 columns)
    const int aoff = m * LDA16 + q;                                 // A operand: tile[row m][k = 4s+q]
    const float b0c = A.b0[col], bc0c = A.bc0[col], bc1c = A.bc1[col], b1c = A.b1[col], w2c = A.w2[col], b2 = A.b2[0];
    if (tid < 2 * HD) s_wc2[tid] = A.wc2[tid];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    if (A.zero_stats && blockIdx.x == 0) for (int i = tid; i < A.zero_count; i += 512) A.zero_stats[i] = 0.0;
    if (A.zero_stats2 && blockIdx.x == 0) for (int i = tid; i < A.zero_count2; i += 512) A.zero_stats2[i] = 0.0;
    float xs0 = 1.f, xs1 = 1.f, xs2 = 1.f, xs3 = 1.f, xh0 = 0.f, xh1 = 0.f, xh2 = 0.f, xh3 = 0.f;   // X scale / shift of this thread's 4 columns
    if (A.xbn_stats) {
        stage_bn(s_u, A.xbn_stats, A.xbn_inv_rows, A.xbn_gamma, A.xbn_beta, tid);      // s_u is free until phase A
        LDS_BARRIER();
        xs0 = s_u[sc4]; xs1 = s_u[sc4 + 1]; xs2 = s_u[sc4 + 2]; xs3 = s_u[sc4 + 3];
        xh0 = s_u[HD + sc4]; xh1 = s_u[HD + sc4 + 1]; xh2 = s_u[HD + sc4 + 2]; xh3 = s_u[HD + sc4 + 3];
        LDS_BARRIER();
    }
    STAMP(6);
    {   // one workgroup per group of 16 instances (a persistent loop here makes the compiler hoist ~200 loop-invariant
        // 64-bit weight addresses into registers and spill them)
        const int g0 = blockIdx.x * HG;
        const int ng = (A.B - g0) < HG ? (A.B - g0) : HG;
        const int nrows = ng * R;
        // ---- requests first: weights of phase A, the instance tiles, the masks
        float wA[32], wB[32], wC[32];
        WCOL(wA, A.W0i + 1 * HD * HD);                              // Wb
        WCOL(wB, A.W0i + 2 * HD * HD);                              // Wc
        WCOL(wC, A.Wc0i);
        {
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            float4 xp = z;
            if (A.xbn_stats) {                                      // pooled = mean over the instance's normalised rows (ac:444 / gcn:192)
                if (sr < ng) {
                    const int nr = A.xgather ? A.xT : R;
                    const float *src = A.X + (size_t)(g0 + sr) * nr * HD + sc4;
                    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 12
                    for (int r = 0; r < nr; r++) {                   // 12 rows in flight per thread
                        const float4 v = *reinterpret_cast<const float4 *>(src + (size_t)r * HD);
                        float y0 = fmaf(v.x, xs0, xh0), y1 = fmaf(v.y, xs1, xh1), y2 = fmaf(v.z, xs2, xh2), y3 = fmaf(v.w, xs3, xh3);
                        if (A.xrelu) { y0 = fmaxf(y0, 0.f); y1 = fmaxf(y1, 0.f); y2 = fmaxf(y2, 0.f); y3 = fmaxf(y3, 0.f); }
                        a0 += y0; a1 += y1; a2 += y2; a3 += y3;
                    }
                    const float ir = 1.0f / (float)nr;
                    xp = make_float4(a0 * ir, a1 * ir, a2 * ir, a3 * ir);
                    *reinterpret_cast<float4 *>(A.pooled_out + (size_t)(g0 + sr) * HD + sc4) = xp;
                }
            } else if (sr < ng) xp = *reinterpret_cast<const float4 *>(A.pooled + (size_t)(g0 + sr) * HD + sc4);
            STAMP(7);
            const float4 xo = sr < ng ? *reinterpret_cast<const float4 *>(A.other + (size_t)(g0 + sr) * HD + sc4) : z;
            float *dp = s_p + sr * LDA16 + sc4, *dq = s_o + sr * LDA16 + sc4;
            *reinterpret_cast<float2 *>(dp) = make_float2(xp.x, xp.y); *reinterpret_cast<float2 *>(dp + 2) = make_float2(xp.z, xp.w);
            *reinterpret_cast<float2 *>(dq) = make_float2(xo.x, xo.y); *reinterpret_cast<float2 *>(dq + 2) = make_float2(xo.z, xo.w);
        }
        for (int i = tid; i < nrows; i += 512) s_mask[i] = A.mask[(size_t)g0 * R + i];
        // scorer row `grow` of the group (instance grow / R, candidate / machine grow % R) -> its source row in X
        auto xrow = [&](int grow) __attribute__((always_inline)) -> const float * {
            if (!A.xgather) return A.X + ((size_t)g0 * R + grow) * HD + sc4;
            const int il = (int)__umulhi((unsigned)grow, invR);
            return A.X + ((size_t)(g0 + il) * A.xT + A.xgather[(size_t)g0 * R + grow]) * HD + sc4;
        };
        float4 xr[HCH];                                             // X rows of the first chunk: requested now, committed after phase A
#pragma unroll
        for (int t = 0; t < HCH; t++) {
            const int grow = t * 16 + sr;
            xr[t] = (t < R && grow < nrows) ? *reinterpret_cast<const float4 *>(xrow(grow)) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        LDS_BARRIER();
        STAMP(0);
        auto xnorm = [&](float4 v, bool valid) __attribute__((always_inline)) {
            if (!valid) return make_float4(0.f, 0.f, 0.f, 0.f);
            float y0 = fmaf(v.x, xs0, xh0), y1 = fmaf(v.y, xs1, xh1), y2 = fmaf(v.z, xs2, xh2), y3 = fmaf(v.w, xs3, xh3);
            if (A.xrelu) { y0 = fmaxf(y0, 0.f); y1 = fmaxf(y1, 0.f); y2 = fmaxf(y2, 0.f); y3 = fmaxf(y3, 0.f); }
            return make_float4(y0, y1, y2, y3);
        };
        // ---- phase A: u = Wb pooled + Wc other + b0 ; c1 = tanh(Wc0 pooled + bc0)
        {
            f32x4 au = zero4, au2 = zero4, ac = zero4;
            const float *pp = s_p + aoff, *po = s_o + aoff;
#pragma unroll
            for (int s = 0; s < 32; s++) {
                const float a_p = pp[4 * s], a_o = po[4 * s];
                au = __builtin_amdgcn_mfma_f32_16x16x4f32(a_p, wA[s], au, 0, 0, 0);
                ac = __builtin_amdgcn_mfma_f32_16x16x4f32(a_p, wC[s], ac, 0, 0, 0);
                au2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_o, wB[s], au2, 0, 0, 0);
            }
            MFMA_SETTLE3(au, ac, au2);
            WCOL(wA, A.Wc1i);                                       // requested now, used in phase B
            WCOL(wB, A.W0i);                                        // Wa
            WCOL(wC, A.W1i);                                        // phase C
#pragma unroll
            for (int i = 0; i < 4; i++) {
                s_u[(4 * q + i) * HD + col] = au[i] + au2[i] + b0c;
                s_c1[(4 * q + i) * LDA16 + col] = fast_tanh(ac[i] + bc0c);
            }
        }
        STAMP(1);
        for (int tb = 0; tb < R; tb += HCH) {
            const int nt = (R - tb) < HCH ? (R - tb) : HCH;
            // ---- X rows of this chunk -> LDS tiles (rows beyond the group's are zero)
#pragma unroll
            for (int t = 0; t < HCH; t++) {
                if (tb > 0) {
                    const int grow = (tb + t) * 16 + sr;
                    xr[t] = (tb + t < R && grow < nrows) ? *reinterpret_cast<const float4 *>(xrow(grow)) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                const float4 xv = xnorm(xr[t], (tb + t) * 16 + sr < nrows);
                float *d = s_x + (t * 16 + sr) * LDA16 + sc4;
                *reinterpret_cast<float2 *>(d) = make_float2(xv.x, xv.y); *reinterpret_cast<float2 *>(d + 2) = make_float2(xv.z, xv.w);
            }
            LDS_BARRIER();                                          // X tiles, u and c1 are complete
            STAMP(2);
            // ---- phase B: s1 = tanh(Wa x + u[instance]) -> s_s ; first chunk: c2 = tanh(Wc1 c1 + bc1)
#pragma unroll 1
            for (int t = 0; t < nt; t += 2) {
                f32x4 a0 = zero4, a1 = zero4;
                const float *p0 = s_x + t * 16 * LDA16 + aoff, *p1 = p0 + 16 * LDA16;
                TILE2(a0, a1, p0, p1, wB);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int grow = (tb + t) * 16 + 4 * q + i;
                    const int i0 = grow < nrows ? (int)__umulhi((unsigned)grow, invR) : 0;
                    const int i1 = grow + 16 < nrows ? (int)__umulhi((unsigned)(grow + 16), invR) : 0;
                    s_s[(t * 16 + 4 * q + i) * LDA16 + col] = fast_tanh(a0[i] + s_u[i0 * HD + col]);
                    s_s[((t + 1) * 16 + 4 * q + i) * LDA16 + col] = fast_tanh(a1[i] + s_u[i1 * HD + col]);
                }
            }
            if (tb == 0) {
                f32x4 a0 = zero4;
                const float *pc = s_c1 + aoff;
#pragma unroll
                for (int s = 0; s < 32; s++) a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pc[4 * s], wA[s], a0, 0, 0, 0);
                MFMA_SETTLE1(a0);
#pragma unroll
                for (int i = 0; i < 4; i++) s_c2[(4 * q + i) * LDA16 + col] = fast_tanh(a0[i] + bc1c);
            }
            LDS_BARRIER();                                          // s1 tiles and c2 are complete
            STAMP(3);
            // ---- phase C: s2 = tanh(W1 s1 + b1) ; partial scores of this wave's 16 columns
#pragma unroll 1
            for (int t = 0; t < nt; t += 2) {
                f32x4 a0 = zero4, a1 = zero4;
                const float *p0 = s_s + t * 16 * LDA16 + aoff, *p1 = p0 + 16 * LDA16;
                TILE2(a0, a1, p0, p1, wC);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const float v0 = row_sum16(fast_tanh(a0[i] + b1c) * w2c), v1 = row_sum16(fast_tanh(a1[i] + b1c) * w2c);
                    if (m == 0) { s_part[wave * (HCH * 16) + t * 16 + 4 * q + i] = v0; s_part[wave * (HCH * 16) + (t + 1) * 16 + 4 * q + i] = v1; }
                }
            }
            if (tb == 0) {   // value head: 32 threads per instance row, 4 columns each, both outputs
                const int r = tid >> 5, part = tid & 31;
                float p0 = 0.f, p1 = 0.f;
                for (int k = 0; k < 4; k++) { const float x = s_c2[r * LDA16 + part * 4 + k]; p0 = fmaf(x, s_wc2[part * 4 + k], p0); p1 = fmaf(x, s_wc2[HD + part * 4 + k], p1); }
                for (int o = 16; o > 0; o >>= 1) { p0 += __shfl_xor(p0, o); p1 += __shfl_xor(p1, o); }
                if (part == 0 && r < ng) { A.value[(size_t)(g0 + r) * 2] = p0 + A.bc2[0]; A.value[(size_t)(g0 + r) * 2 + 1] = p1 + A.bc2[1]; }
            }
            LDS_BARRIER();
            if (tid < nt * 16) {
                const int grow = tb * 16 + tid;
                float v = b2;
                for (int w = 0; w < 8; w++) v += s_part[w * (HCH * 16) + tid];
                if (grow < nrows) s_score[grow] = v * A.scale;
            }
            LDS_BARRIER();                                          // tiles / s_part are reused by the next chunk
            STAMP(4);
        }
        // ---- masked softmax per instance (ac:266-278 / ac:487-491): 16 lanes per instance; optional action selection
        {
            const int r0 = tid >> 4, l = tid & 15;
            if (r0 < ng) {
                float mx = -INFINITY;
                for (int r = l; r < R; r += 16) if (!s_mask[r0 * R + r]) mx = fmaxf(mx, s_score[r0 * R + r]);
                for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
                float sum = 0.f;
                for (int r = l; r < R; r += 16) if (!s_mask[r0 * R + r]) sum += __expf(s_score[r0 * R + r] - mx);
                for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
                for (int r = l; r < R; r += 16) {
                    const float pr = s_mask[r0 * R + r] ? 0.f : __expf(s_score[r0 * R + r] - mx) / sum;
                    A.prob[(size_t)(g0 + r0) * R + r] = pr;
                    s_score[r0 * R + r] = pr;                           // lanes of one wave: visible to lane l == 0 below
                }
                if (A.sample_mode) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (l == 0) {
                        const int b = g0 + r0;
                        const int pick = pick_action(s_score + r0 * R, R, b, A.sample_mode == 2, A.seed, A.counter);
                        A.idx_out[b] = pick;
                        if (A.logp_out) A.logp_out[b] = logf(s_score[r0 * R + pick]);
                        const int gsel = A.gather_from ? A.gather_from[(size_t)b * R + pick] : pick;
                        if (A.gather_from && A.gathered_out) A.gathered_out[b] = gsel;
                        s_part[r0] = __int_as_float(gsel);                     // hand the selected task to the instance's 16 lanes (s_part is free now)
                    }
                    if (A.mf_on) {
                        // = k_mfea1 (pe:152-214) for the task just selected: the 16 lanes of the instance take the machines
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        const int b = g0 + r0, T_ = A.mf.T, M_ = A.mf.M;
                        int a = __float_as_int(s_part[r0]);
                        if (a < 0 || a >= T_) a = 0;
                        const size_t row = (size_t)b * T_ + a;
                        int pm = 0;
                        if (a % M_ != 0) {
                            pm = reinterpret_cast<const short *>(A.mf.link)[(row - 1) * 8 + 4];   // machine of the job predecessor
                            if (pm < 0) pm += M_;                                             // python negative index (pe:206)
                        }
                        for (int mm = l; mm < M_; mm += 16) {
                            const double tv = A.mf.t[row * M_ + mm], pv = A.mf.p[row * M_ + mm];
                            const double ptv = tv * fabs(pv);
                            const unsigned char mk = (unsigned char)!(tv >= 0);              // run:258-259 ~(t >= 0)
                            const double x = (a % M_ != 0) ? A.mf.tt[((size_t)b * M_ + pm) * M_ + mm] : 0.0;
                            const double f0 = tv > 0 ? tv : A.mf.mean3[row * 3 + 0], f1 = ptv > 0 ? ptv : A.mf.mean3[row * 3 + 1];
                            const double f4 = pv > 0 ? pv : A.mf.mean3[row * 3 + 2];
                            const size_t o = ((size_t)b * M_ + mm) * 6;
                            const double f3 = (double)(1 - (int)mk), f5 = (double)(A.mf.shop[(size_t)b * M_ + mm] + 1);
                            if (A.mf.obs_f32) {
                                float *of = reinterpret_cast<float *>(A.mf.m_fea1_out) + o;
                                of[0] = (float)f0; of[1] = (float)f1; of[2] = (float)x; of[3] = (float)f3; of[4] = (float)f4; of[5] = (float)f5;
                            } else {
                                double *od = reinterpret_cast<double *>(A.mf.m_fea1_out) + o;
                                od[0] = f0; od[1] = f1; od[2] = x; od[3] = f3; od[4] = f4; od[5] = f5;
                            }
                            A.mf.mmask_out[(size_t)b * M_ + mm] = mk;
                        }
                    }
                }
            }
        }
        STAMP(5);
    }
#ifdef MTFJSP_STAMP
    if (A.stamps && lane == 0) for (int i = 0; i < 8; i++) A.stamps[((size_t)blockIdx.x * 8 + wave) * 8 + i] = ph[i];
#endif
}
static size_t heads_lds_bytes() { return (size_t)((2 * HCH + 4) * 16 * LDA16 + HG * HD + 8 * HCH * 16 + HG * 64 + 2 * HD) * 4 + HG * 64; }

// ---------------------------------------------------------------------------------------------
// k_headsx — k_heads with every 128x128 product on the 16-bit matrix cores at f32 accuracy (2-way f16 split, 3 piece
// products, weights pre-scaled by a power of two: see k_gemm_x6 / DESIGN.md §4).  Same work split (wave w owns output columns
// 16w..16w+15 of every product; its weight fragments — 2 planes x 4 k-steps = 32 registers per weight — come from the f16
// register images), but
//   * the operands are swapped (A := weight, B := activation rows), so a lane ends up with 4 consecutive columns of row m:
//     the next product's operand planes are written with 8-byte stores, u is read as float4, and a scorer row's partial
//     score is 4 FMAs + two cross-quarter shuffles instead of four 16-lane DPP reductions;
//   * activations live in LDS as two f16 planes per 16-row tile (pitch 272 B); each value is split once by its
//     producer: X rows by the staging threads, c1 / s1 in the epilogue of the product that makes them (all of them are
//     BatchNorm outputs, means of them or tanh values: far inside the f16 range);
//   * the s1 planes overwrite the X planes (all six accumulators of a chunk are held across one barrier).
// 192 matrix instructions x 16 cycles per wave instead of 512 x 32.
#define WCOLX(dst, Wx, blk)                                                                                     \
    do {                                                                                                         \
        const float4 *w_ = reinterpret_cast<const float4 *>(Wx) + ((size_t)(blk) * 8 + wave) * (2 * 4 * 64) + lane; \
        _Pragma("unroll") for (int p_ = 0; p_ < 2; p_++)                                                         \
            _Pragma("unroll") for (int k_ = 0; k_ < 4; k_++) dst[p_][k_] = __builtin_bit_cast(h16x8, w_[(p_ * 4 + k_) * 64]); \
    } while (0)
// the three piece products of one k-step on two accumulator chains (large term | small terms)
#define X6_STEP(accA, accB, wv, xv, ks)                                                          \
    do {                                                                                          \
        accB = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[0][ks], xv[1], accB, 0, 0, 0);           \
        accA = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[0][ks], xv[0], accA, 0, 0, 0);           \
        accB = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[1][ks], xv[0], accB, 0, 0, 0);           \
    } while (0)
#define HX_CLDA 132
#if MTFJSP_BODY_FUNCS & 2
__device__ __forceinline__ void headsx_body(const HeadArgs &A, unsigned char *smem)
{
#include "mtfjsp_headsx_body.h"
}
#endif
__global__ __launch_bounds__(512) void k_headsx(HeadArgs A)
{
    extern __shared__ __align__(16) unsigned char smem[];
#if MTFJSP_BODY_FUNCS & 2
    headsx_body(A, smem);
#else
#include "mtfjsp_headsx_body.h"
#endif
}
// The job actor's heads and the machine path's three GAT passes in ONE launch.  Both kernels give workgroup g the instances
// 16g .. 16g+15 (heads: HG = 16; GAT: 2 M row tiles per workgroup when the grids agree), and the only thing the GAT needs from
// the job actor is m_fea1 of the task each instance has just selected — written a few lines above by the same workgroup
// (mtfjsp_encoder_arm_mfea1).  So the GAT part starts as soon as its own workgroup's selections are made: one launch boundary and
// one launch ramp less per rollout step, no grid-wide dependency (the GAT's BatchNorm sums are consumed by the NEXT launch, the
// machine actor's heads).  __syncthreads() orders the m_fea1 stores before the loads (same workgroup, same CU).
__global__ __launch_bounds__(512) void k_headsx_gat3x(HeadArgs HA, GatArgs GA)
{
    extern __shared__ __align__(16) unsigned char smem[];
#if MTFJSP_BODY_FUNCS & 2
    headsx_body(HA, smem);
#else
    // (requesting the GAT part's 64 KB weight image under the heads part's softmax — eight float4 per thread held across the barrier —
    // was measured in round 4: 48.7 us against 46.2 us per launch; not kept)
    {
        const HeadArgs &A = HA;
#include "mtfjsp_headsx_body.h"
    }
#endif
    __syncthreads();
#if MTFJSP_BODY_FUNCS & 1
    gat3x_body(GA, smem);
#else
    {
        const GatArgs &A = GA;
#include "mtfjsp_gat3x_body.h"
    }
#endif
}
// Round 5: the machine actor's heads in the SAME launch as well — three launches per rollout step instead of four.  What separates
// the GAT passes from the machine heads is one BatchNorm over all B*M machine nodes (ac:434): a grid-wide dependency, but a small one
// (256 column sums).  It is exchanged the way k_gin_res exchanges its statistics (count-carrying integer atomics, no barrier, ~3 us)
// with everything the machine heads request on their own — weights, vectors, their node rows (written by this very workgroup), the
// job actor's pooled embedding — in flight before the wait; the launch boundary, its cold first phase (3.9 us of requests with
// nothing to overlap) and a second dispatch ramp go.  Needs the whole grid co-resident (one workgroup per CU: the host enables it
// only where the single-launch GIN kernel's census passed) and bounds its wait like that kernel (time-out -> MTFJSP_ERR_RETRY).
// What the GAT statements need in LDS before their first tile — the 64 KB of weight fragments (s_wf), the projection's operand image
// and the attention vectors — copied by the 256 threads of waves 4-7 while waves 0-3 run the job selection (HX_IDLE_HOOK in
// mtfjsp_headsx_body.h): [0, 64 KB) lies in the scorer's dead planes, the images beyond everything the heads part uses.
#define GAT_IMG_OFF (128 * 1024)                                  // behind s_wf (64 KB) and s_a (64 KB)
#define GAT_IMG_BYTES (8 * 64 * 4 * 4 + 256 * 4)
#define GAT_F1_OFF (GAT_IMG_OFF + GAT_IMG_BYTES)                  // m_fea1 of the workgroup's 16 x M machines as f32 [16 M][6] (M <= 8), written by the job selection
#define GAT_F2_OFF (GAT_F1_OFF + 16 * 8 * 6 * 4)                  // m_fea2 likewise [16 M][8], copied here
#define GAT_PRE_END (GAT_F2_OFF + 16 * 8 * 8 * 4)
// this workgroup's m_fea2 rows (written by the environment step a rollout step ago: the memory-side cache by now) are requested at the top of
// the launch — the last of the first phase's requests, by every thread (waves 0-3 drop theirs) — and wait in four registers
__device__ __forceinline__ void gat_f2_load(const GatArgs &G, int t, int M, float (&x)[4])
{
    const size_t base = (size_t)blockIdx.x * 16 * M * 8, total = (size_t)G.R * 8;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const size_t idx = base + (size_t)(i * 256 + t), ic = idx < total ? idx : total - 1;
        x[i] = G.feat_f64 ? (float)reinterpret_cast<const double *>(G.f2)[ic] : reinterpret_cast<const float *>(G.f2)[ic];
        if (!(i * 256 + t < 16 * M * 8 && idx < total)) x[i] = 0.f;
    }
}
__device__ __forceinline__ void gat_prestage(const GatArgs &G, unsigned char *smem, int t, const float (&x)[4])
{
    // every request first, none behind a branch; then the stores in request order
    const float4 *src = reinterpret_cast<const float4 *>(G.Wx6);
    float4 v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = src[i * 256 + t];
    const float4 q0 = reinterpret_cast<const float4 *>(G.Wq)[t], q1 = reinterpret_cast<const float4 *>(G.Wq)[256 + t];
    const float ga = G.gat_a[t];
    float4 *dst = reinterpret_cast<float4 *>(smem);
#pragma unroll
    for (int i = 0; i < 16; i++) dst[i * 256 + t] = v[i];
    float4 *img = reinterpret_cast<float4 *>(smem + GAT_IMG_OFF);
    img[t] = q0; img[256 + t] = q1;
    reinterpret_cast<float *>(smem + GAT_IMG_OFF + 8192)[t] = ga;
    float *d = reinterpret_cast<float *>(smem + GAT_F2_OFF);
#pragma unroll
    for (int i = 0; i < 4; i++) d[i * 256 + t] = x[i];
}
// (Requesting one word of every line gat_prestage() reads at the top of the launch, so that they wait in this XCD's L2 — they were last
// read a rollout step ago — shortened it by 2 us and lengthened the first phase of the job heads by 0.5 us: the launch ended 0.7-1.2 us
// later in alternating runs on one box, because the selection on waves 0-3, not this copy, ends the heads part.  Not kept.)
#if !MTFJSP_BODY_FUNCS
// ENV (round 6): 0 = the three parts; 1 / 2 = + the environment step of the workgroup's 16 instances as the launch's tail (f32 / f64
// observations; mtfjsp_env_grp.h's own code with the instances dealt to the 8 waves: bit-identical to k_env_grp16).  Two launches per
// rollout step.  Round 3's k_headsx_envstep lost to the separate launch because the tail's first loads were cold (the state was last
// touched a rollout step ago); here one word of every line of the workgroup's state records is requested in front of the machine
// heads' part, 13 us ahead of the tail, and waits in this XCD's L2.
// NLDS (round 6): the GAT part's node rows stay in LDS for the machine part (M <= 6: at most 12 tiles, each wave's tiles in buffers of their
// own, and the machine part stages its rows by instance) — 12.6 MB per launch less written and 12.6 MB less read back.
template <int ENV, bool NLDS>
__global__ __launch_bounds__(512) void k_headsx_gat3x_headsx(HeadArgs HA, GatArgs GA, HeadArgs HM, XchgArgs XG, EnvParams EP)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const XchgArgs &XA = XG;
    X3_RT(0);
    for (int i = blockIdx.x * 512 + threadIdx.x; i < XW_SET; i += (int)gridDim.x * 512) XA.words_next[i] = 0ull;
#ifdef MTFJSP_STAMP3
#undef H3_RT
#define H3_RT(i) do { if (XA.stamps && (threadIdx.x & 63) == 0) { __builtin_amdgcn_sched_barrier(0); XA.stamps[(size_t)H3_BASE * 64 + ((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#define H3_BASE 512
#undef H3T_RT
#define H3T_RT(i) do { if (XA.stamps && (threadIdx.x & 63) == 0) { __builtin_amdgcn_sched_barrier(0); XA.stamps[(size_t)(H3_BASE == 512 ? 1536 : 1792) * 64 + ((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#undef H3S_RT
#define H3S_RT(i) do { if (XA.stamps && (threadIdx.x & 63) == 0) { __builtin_amdgcn_sched_barrier(0); XA.stamps[(size_t)(H3_BASE + 256) * 64 + ((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#endif
    {
        const HeadArgs &A = HA;
#ifndef GAT_F2_LATE
#define HX_TOP_HOOK float gf2[4]; gat_f2_load(GA, tid & 255, A.mf.M, gf2);
#define HX_IDLE_HOOK gat_prestage(GA, smem, tid - 256, gf2);
#else                                                              // (diagnostic builds: the rows requested where they are stored, as before)
#define HX_IDLE_HOOK float gf2[4]; gat_f2_load(GA, tid - 256, A.mf.M, gf2); gat_prestage(GA, smem, tid - 256, gf2);
#endif
#define HX_MF1_LDS reinterpret_cast<float *>(smem + GAT_F1_OFF)
#include "mtfjsp_headsx_body.h"
#undef HX_IDLE_HOOK
#undef HX_TOP_HOOK
#undef HX_MF1_LDS
    }
    X3_RT(1);
    __syncthreads();                                               // m_fea1 / the machine mask of this workgroup's instances are written; the GAT part's LDS images are staged
    int tid_part = threadIdx.x;
    asm volatile("" : "+v"(tid_part));
#undef BODY_TID
#define BODY_TID tid_part
    {
        const GatArgs &A = GA;
#undef GAT_XCHG
#define GAT_XCHG 1
#undef GAT_PRESTAGED
#define GAT_PRESTAGED 1
#undef GAT_NLDS
#define GAT_NLDS NLDS
#include "mtfjsp_gat3x_body.h"
#undef GAT_NLDS
#define GAT_NLDS false
#undef GAT_PRESTAGED
#define GAT_PRESTAGED 0
#undef GAT_XCHG
#define GAT_XCHG 0
    }
    X3_RT(3);
    tid_part = threadIdx.x;
    asm volatile("" : "+v"(tid_part));
    unsigned env_warm = 0;
    if (ENV) {                                                     // one word per 128-byte line of this workgroup's state records (task records 2 x 16 T x 16 B, job / machine records, scalar rows)
        const int b0 = blockIdx.x * EG_SMALL, nb = EP.B - b0 < EG_SMALL ? EP.B - b0 : EG_SMALL;
        const unsigned l_sd = (unsigned)(nb * EP.T * 16 + 127) / 128, l_jr = (unsigned)(nb * EP.J * 16 + 127) / 128, l_mj = (unsigned)(nb * EP.MJ * 8 + 127) / 128, l_sc = (unsigned)(nb * SCAL_N * 8 + 127) / 128;
        unsigned l = (unsigned)tid_part;
        const unsigned char *wp = reinterpret_cast<const unsigned char *>(EP.scal + (size_t)b0 * SCAL_N);
        if (l < l_sd) wp = reinterpret_cast<const unsigned char *>(EP.sd + (size_t)b0 * EP.T) + (size_t)l * 128;
        else if ((l -= l_sd) < l_sd) wp = reinterpret_cast<const unsigned char *>(EP.pl + (size_t)b0 * EP.T) + (size_t)l * 128;
        else if ((l -= l_sd) < l_jr) wp = reinterpret_cast<const unsigned char *>(EP.jr + (size_t)b0 * EP.J) + (size_t)l * 128;
        else if ((l -= l_jr) < l_mj) wp = reinterpret_cast<const unsigned char *>(EP.mj + (size_t)b0 * EP.MJ) + (size_t)l * 128;
        else if ((l -= l_mj) < l_sc) wp += (size_t)l * 128;
        env_warm = *reinterpret_cast<const unsigned *>(wp);
    }
    LDS_BARRIER();                                                 // (not __syncthreads(): that would wait for the statistics' atomics to be acknowledged)
#ifdef MTFJSP_STAMP3
#undef H3_BASE
#define H3_BASE 1024
#endif
    {
        const HeadArgs &A = HM;
#undef HX_XCHG
#define HX_XCHG 1
#undef HX_NLDS
#define HX_NLDS NLDS
#include "mtfjsp_headsx_body.h"
#undef HX_NLDS
#define HX_NLDS false
#undef HX_XCHG
#define HX_XCHG 0
    }
#ifdef MTFJSP_STAMP3
#undef H3_RT
#define H3_RT(i) do { } while (0)
#undef H3S_RT
#define H3S_RT(i) do { } while (0)
#undef H3T_RT
#define H3T_RT(i) do { } while (0)
#endif
#undef BODY_TID
#define BODY_TID threadIdx.x
    X3_RT(7);
    if (ENV) {
        asm volatile("" :: "v"(env_warm));
        __syncthreads();                                           // the machine selection of this workgroup's instances is stored; the heads' LDS is free
        if (ENV == 1) env_grp_body_dyn<float, 1, 8>(EP, smem); else env_grp_body_dyn<double, 1, 8>(EP, smem);
    }
}
#endif
// The machine actor's heads and the environment step of the same 16 instances in ONE launch (round-2 review, item 3): the heads end
// with the machine selection of exactly the instances k_env_grp16 would give this blockIdx, and nothing else the step reads is written
// by this launch.  __syncthreads() orders the selected indices' stores before the step's loads (same workgroup, same CU); the step's
// arrays take the heads' LDS; 8 waves take the 16 instances in two rounds.  The step is mtfjsp_env_grp.h's own code: bit-identical.
template <typename OBS>
__global__ __launch_bounds__(512) void k_headsx_envstep(HeadArgs HA, EnvParams EP)
{
    extern __shared__ __align__(16) unsigned char smem[];
    {
        const HeadArgs &A = HA;
#include "mtfjsp_headsx_body.h"
    }
    __syncthreads();
    env_grp_body_dyn<OBS, 1, 8>(EP, smem);
}
static size_t headsx_lds_bytes() { return (size_t)(HCH + 3) * X2_TILE + (size_t)(16 * HX_CLDA + HG * HD + HX_NPART * HCH * 16 + HG * 64 + 2 * HD + 5 * HD) * 4 + HG * 64 + 2 * 512 * 4; }
static size_t fused3_lds_bytes() { const size_t g = GAT_PAIRED ? (size_t)(8 * 2 * 4 * 64 * 16 + 12 * 16 * HD * 4) : (size_t)GAT_PRE_END; return headsx_lds_bytes() > g ? headsx_lds_bytes() : g; }   // k_headsx_gat3x_headsx: + the prestaged GAT images and feature rows (GAT_PRE_END)
// The same kernel with TEN scorer tiles per chunk: a group of 16 instances with 7..10 candidates / machines each (J10M10: R = 10)
// goes through the product phases once instead of twice (6 + 4 tiles, each chunk with its own staging, four barriers and latency chain)
#undef HCH
#define HCH 10
#undef HX_NPART
#define HX_NPART 8
__global__ __launch_bounds__(512) void k_headsx10(HeadArgs A)
{
    extern __shared__ __align__(16) unsigned char smem[];
#include "mtfjsp_headsx_body.h"
}
static size_t headsx10_lds_bytes() { return (size_t)(HCH + 3) * X2_TILE + (size_t)(16 * HX_CLDA + HG * HD + HX_NPART * HCH * 16 + HG * 64 + 2 * HD + 5 * HD) * 4 + HG * 64 + 2 * 512 * 4; }
#undef HCH
#define HCH 6
#undef HX_NPART
#define HX_NPART 32
// The critic values alone (round 6): the post-terminal forward pair of an episode (Run.py:455-475) keeps only job_v and mach_v, so its
// two heads launches run the pooled / other staging, phase A, c2 and the value head — the same statements, the same bits — and none of
// the scorer (X rows, phases B and C on them, scores, softmax, selection).  mtfjsp_encoder_arm_values_only.
#undef HX_VALUES_ONLY
#define HX_VALUES_ONLY 1
__global__ __launch_bounds__(512) void k_headsx_values(HeadArgs A)
{
    extern __shared__ __align__(16) unsigned char smem[];
#include "mtfjsp_headsx_body.h"
}
#undef HX_VALUES_ONLY
#define HX_VALUES_ONLY 0

// ---------------------------------------------------------------------------------------------
// GIN layer 0, first Linear (12 -> 128) fused with the neighbour aggregation of the raw task features
// (gcn:125-153).  thread = (column c, row group); persistent blocks, column stats in registers.
template <typename OBS>
__global__ __launch_bounds__(256) void k_gin0(int N, int T, const OBS *tfea, const int *ell_col, const float *ell_val,
                                              const float *W /*[128,12]*/, const float *bias, float *out, double *stats)
{
    stats += (blockIdx.x % STAT_REP) * 256;
    __shared__ float s_p[32 * 12];
    const int tid = threadIdx.x, c = tid & 127, half = tid >> 7;
    float w[12];
    for (int k = 0; k < 12; k++) w[k] = W[c * 12 + k];
    const float bc = bias[c];
    double ssum = 0, ssq = 0;
    const int ntiles = (N + 31) / 32;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * 32;
        LDS_BARRIER();
        for (int i = tid; i < 32 * 12; i += 256) {
            const int r = i / 12, k = i % 12, g = row0 + r;
            float v = 0.f;
            if (g < N) {
                const int base = (g / T) * T;
                double acc = (double)(float)tfea[(size_t)g * 12 + k];            // ac:143 .float(), gcn:125 .double()
                int deg = 1;
                const int c0 = ell_col[(size_t)g * 2], c1 = ell_col[(size_t)g * 2 + 1];
                if (c0 >= 0) { acc += (double)ell_val[(size_t)g * 2] * (double)(float)tfea[(size_t)(base + c0) * 12 + k]; deg++; }
                if (c1 >= 0) { acc += (double)ell_val[(size_t)g * 2 + 1] * (double)(float)tfea[(size_t)(base + c1) * 12 + k]; deg++; }
                v = (float)(acc / (double)deg);
            }
            s_p[i] = v;
        }
        LDS_BARRIER();
        for (int rr = 0; rr < 16; rr++) {
            const int r = half * 16 + rr, g = row0 + r;
            if (g < N) {
                float a = 0.f;
                for (int k = 0; k < 12; k++) a = fmaf(s_p[r * 12 + k], w[k], a);
                a += bc;
                out[(size_t)g * HD + c] = a;
                ssum += (double)a; ssq += (double)a * (double)a;
            }
        }
    }
    // one atomic pair per column per block: fold the two row-halves through LDS first
    __shared__ double s_half[256];
    if (half == 1) { s_half[c] = ssum; s_half[HD + c] = ssq; }
    __syncthreads();
    if (half == 0) {
        atomicAdd(&stats[c], ssum + s_half[c]);
        atomicAdd(&stats[HD + c], ssq + s_half[HD + c]);
    }
}

// ---------------------------------------------------------------------------------------------
// After the last GIN BatchNorm: h = relu(bn(z)); graph mean pool (gcn:192) and candidate gather (ac:197-207).
// One 128-thread block per instance, thread = column.
#ifndef POOL_INFLIGHT
#define POOL_INFLIGHT 8                        // rows in flight per thread; 13 / 16 measured the same (114 / 108 us at J10M10 x 8192 / J20M20 x 2048), 25 slower (133 / 117)
#endif
// NT (non-temporal row loads) is a template parameter: as a run-time flag — `nt ? __builtin_nontemporal_load(p) : *p` — the two arms
// were merged into ONE plain load (rounds 3 and 4 shipped that: no `nt` load in the kernel's ISA), and a just-written 419 MB matrix
// reads at 3.6-3.8 TB/s with plain loads against 5.2-6.3 TB/s with non-temporal ones (tools/ubench/read_after_write.hip).
//
// A 256-thread block takes the instances of its sequence one after the other; thread = (row group rg = tid / 32 of 8, 4 columns).
// The rows are streamed in batches of 64 (8 per thread, 16-byte loads) through ONE software pipeline that runs across the instance
// boundaries: two batches are always in flight, so the memory pipe does not drain while an instance's 8 partial sums are folded
// through LDS (double-buffered: one barrier per instance).  The candidate rows (ac:197-207) are picked out of the stream — row v of
// job v / M is written to cand_feat when cand[b][v / M] == v — instead of being read a second time behind a dependent index load.
// The summation order per thread (rows rg, rg+8, ... ascending) and of the fold (row groups 0..7) is the earlier kernel's: same bits.
#ifndef POOL_INFLIGHT
#define POOL_INFLIGHT 8                        // rows per batch and thread (two batches in flight)
#endif
// Pooling epilogue of the last streaming product (k_gemm_x6, GemmArgs::pooled): that launch takes candidate j of an instance from job j's block of rows
// — where the environment's candidates always lie (ppo:306-309) — and its output is not stored.  A caller-made candidate outside its job's block is served
// here, one thread per such slot, by forming that one row again with f32 instructions: h4 = relu(bn4(z4[row])), z5 = W5 h4 + b5, relu(bn5(z5)).  Never on the
// rollout path (the launch then only reads the candidate array).
__global__ __launch_bounds__(128) void k_cand_fixup(int B, int T, int J, const int *cand, const float *z4, const double *st4, const float *g4, const float *b4,
                                                    const float *Wt5 /*[k][n]*/, const float *bias5, const double *st5, const float *g5, const float *b5,
                                                    double inv_rows, float *cand_feat)
{
    __shared__ int s_need[128];
    __shared__ float s_h4[HD];
    const int tid = threadIdx.x, slot = blockIdx.x * 128 + tid, M = T / J;
    int need = -1;
    if (slot < B * J) {
        const int b = slot / J, j = slot - b * J, c = cand[slot];
        if (c >= 0 && c < T && c / M != j) need = b * T + c;      // the row this slot wants
    }
    s_need[tid] = need;
    if (__syncthreads_or(need >= 0) == 0) return;                 // (the rollout path: nothing to do)
    // column tid of both BatchNorms
    float sc4, sh4, sc5, sh5;
    {
        double su = 0, sq = 0;
        for (int r = 0; r < STAT_REP; r++) { su += st4[r * 256 + tid]; sq += st4[r * 256 + HD + tid]; }
        double mean = su * inv_rows, var = sq * inv_rows - mean * mean;
        if (var < 0) var = 0;
        sc4 = (1.0f / sqrtf((float)(var + BN_EPS))) * g4[tid]; sh4 = b4[tid] - (float)mean * sc4;
        su = 0; sq = 0;
        for (int r = 0; r < STAT_REP; r++) { su += st5[r * 256 + tid]; sq += st5[r * 256 + HD + tid]; }
        mean = su * inv_rows; var = sq * inv_rows - mean * mean;
        if (var < 0) var = 0;
        sc5 = (1.0f / sqrtf((float)(var + BN_EPS))) * g5[tid]; sh5 = b5[tid] - (float)mean * sc5;
    }
    for (int i = 0; i < 128; i++) {
        const int row = s_need[i];                                // (uniform)
        if (row < 0) continue;
        __syncthreads();
        s_h4[tid] = fmaxf(__builtin_fmaf(z4[(size_t)row * HD + tid], sc4, sh4), 0.f);
        __syncthreads();
        float z = bias5 ? bias5[tid] : 0.f;
        for (int k = 0; k < HD; k++) z = __builtin_fmaf(s_h4[k], Wt5[k * HD + tid], z);
        cand_feat[((size_t)blockIdx.x * 128 + i) * HD + tid] = fmaxf(__builtin_fmaf(z, sc5, sh5), 0.f);
    }
}

// Sums and second moments of the aggregated raw features x (gcn:125-153 on the [N,12] task features, the first Linear's input) over all rows: the
// BatchNorm statistics of z0 = W0 x + b follow from them exactly (k_gemm_x6<PRO_GIN0BN>, moments mode), so no launch has to form z0 for its sums alone.
// A row is a quad of lanes: lane j < 3 of it reads features 4j..4j+3 of the row and of its <= 2 neighbours and aggregates them (the same f64 operations
// and the same rounding to f32 as the product kernels' producers: the moments are those of the values the Linear multiplies); the quad exchanges the 12
// values (DPP) and lane j accumulates its own rows 4j..4j+3 of x x' and of the sum in f64 (lane 3 of a quad adds zeros: no per-lane register choice,
// no branch).  out: one [256]-double replica per workgroup % STAT_REP (MOM_*).
__global__ __launch_bounds__(256) void k_gin0_moments(int N, int T, const float *tfea, const int *ell_col, const float *ell_val, double *out)
{
    __shared__ double s_part[16][3][52];                            // [wave x DPP row][j][4 sums | 4 x 12 products]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 3, rs = lane >> 2, jf = j < 3 ? j : 0;
    double S[4][12], sx[4];
#pragma unroll
    for (int a = 0; a < 4; a++) {
        sx[a] = 0;
#pragma unroll
        for (int k = 0; k < 12; k++) S[a][k] = 0;
    }
    const int stride = (int)gridDim.x * 64;
    const int niter = (N + stride - 1) / stride;
    const float invT = 1.0f / (float)T;
    // requests run ahead of their use, unconditional and clamped (a request behind an `if` drains the queue at the join): the feature rows two
    // iterations ahead, the ELL entries their addresses need another four (an iteration is ~0.25 us of arithmetic, a memory round trip 1-2 us)
    struct Ell { int2 cc; float2 vv; } e[4];
    struct Feat { float4 fo, fx, fy; int2 cc; float2 vv; int g; } f[2];
    auto req_ell = [&](Ell &el, int it) __attribute__((always_inline)) {
        const int g = it * stride + (int)blockIdx.x * 64 + wave * 16 + rs, gc = g < N ? g : N - 1;
        el.cc = *reinterpret_cast<const int2 *>(ell_col + (size_t)gc * 2);
        el.vv = *reinterpret_cast<const float2 *>(ell_val + (size_t)gc * 2);
    };
    auto req_feat = [&](Feat &ft, const Ell &el, int it) __attribute__((always_inline)) {
        const int g = it * stride + (int)blockIdx.x * 64 + wave * 16 + rs, gc = g < N ? g : N - 1;
        int qi = (int)((float)gc * invT);                         // gc / T for gc < 2^24: the f32 quotient is off by at most one
        const int rem = gc - qi * T;
        qi += rem >= T ? 1 : rem < 0 ? -1 : 0;
        const int base = qi * T;
        ft.cc = el.cc; ft.vv = el.vv; ft.g = g;
        ft.fo = *reinterpret_cast<const float4 *>(tfea + (size_t)gc * 12 + 4 * jf);
        ft.fx = *reinterpret_cast<const float4 *>(tfea + (size_t)(el.cc.x >= 0 ? base + el.cc.x : gc) * 12 + 4 * jf);
        ft.fy = *reinterpret_cast<const float4 *>(tfea + (size_t)(el.cc.y >= 0 ? base + el.cc.y : gc) * 12 + 4 * jf);
    };
    auto body = [&](Feat &ft, Ell &el, int it) __attribute__((always_inline)) {
        const int deg = 1 + (ft.cc.x >= 0) + (ft.cc.y >= 0);
        const double inv = deg == 1 ? 1.0 : deg == 2 ? 0.5 : (1.0 / 3.0);
        const float fo[4] = {ft.fo.x, ft.fo.y, ft.fo.z, ft.fo.w}, fx[4] = {ft.fx.x, ft.fx.y, ft.fx.z, ft.fx.w}, fy[4] = {ft.fy.x, ft.fy.y, ft.fy.z, ft.fy.w};
        float x4[4];
        const int keep = (ft.g < N && j < 3) ? -1 : 0;                  // (a mask, not a select: hipcc turns the select into a branch around the f64 work)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            double acc = (double)fo[i];
            acc += (double)ft.vv.x * (double)fx[i];
            acc += (double)ft.vv.y * (double)fy[i];
            x4[i] = __builtin_bit_cast(float, __builtin_bit_cast(int, (float)(acc * inv)) & keep);
        }
        float xa[12];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int xi = __builtin_bit_cast(int, x4[i]);
            xa[i] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, xi, 0x00, 0xF, 0xF, true));        // quad_perm [0,0,0,0]
            xa[4 + i] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, xi, 0x55, 0xF, 0xF, true));    // [1,1,1,1]
            xa[8 + i] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, xi, 0xAA, 0xF, 0xF, true));    // [2,2,2,2]
        }
        double xd[12];
#pragma unroll
        for (int k = 0; k < 12; k++) xd[k] = (double)xa[k];
#pragma unroll
        for (int a = 0; a < 4; a++) {
            const double xi = (double)x4[a];
            sx[a] += xi;
#pragma unroll
            for (int k = 0; k < 12; k++) S[a][k] = __builtin_fma(xi, xd[k], S[a][k]);
        }
        req_feat(ft, el, it + 2);
        req_ell(el, it + 6);
    };
#pragma unroll
    for (int i = 0; i < 4; i++) req_ell(e[i], i);
    req_feat(f[0], e[0], 0); req_ell(e[0], 4);
    req_feat(f[1], e[1], 1); req_ell(e[1], 5);
    for (int it = 0; it < niter; it += 4) {                       // (iterations past the last one see rows >= N only: zeros)
        body(f[0], e[2], it);
        body(f[1], e[3], it + 1);
        body(f[0], e[0], it + 2);
        body(f[1], e[1], it + 3);
    }
    // The 4 rows of a 16-lane DPP row (lane bits 2, 3) are folded in registers — row_ror:4, row_ror:8: partners with the same j —, the 4 DPP rows x 4
    // waves through LDS, then one atomic per entry and workgroup.  (As four __shfl_xor steps per value — 416 ds_bpermute with a wait each — this fold
    // was a quarter of the kernel's time.)
    auto dpp_d = [&](double x, auto Ctrl) __attribute__((always_inline)) {
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), decltype(Ctrl)::value, 0xF, 0xF, true);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), decltype(Ctrl)::value, 0xF, 0xF, true);
        return __hiloint2double(hi, lo);
    };
    auto fold4 = [&](double x) __attribute__((always_inline)) {
        x += dpp_d(x, std::integral_constant<int, 0x124>{});
        x += dpp_d(x, std::integral_constant<int, 0x128>{});
        return x;
    };
    const int drow = lane >> 4;
#pragma unroll
    for (int a = 0; a < 4; a++) {
        const double t = fold4(sx[a]);
        if ((lane & 15) < 3) s_part[wave * 4 + drow][j][a] = t;
#pragma unroll
        for (int k = 0; k < 12; k++) {
            const double u = fold4(S[a][k]);
            if ((lane & 15) < 3) s_part[wave * 4 + drow][j][4 + a * 12 + k] = u;
        }
    }
    __syncthreads();
    if (tid < 3 * 52) {
        const int jj = tid / 52, idx = tid % 52;
        double v = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) v += s_part[w][jj][idx];
        const int slot = idx < 4 ? MOM_SX + 4 * jj + idx : MOM_SXX + (4 * jj + (idx - 4) / 12) * 12 + (idx - 4) % 12;
        atomicAdd(&out[(blockIdx.x % STAT_REP) * 256 + slot], v);
    }
}

template <int NT>
__global__ __launch_bounds__(256) void k_job_pool_gather(unsigned *range_flag, int rpr, int S, int B, int T, int J, const float *z, const double *stats, double inv_rows,
                                                        const float *gamma, const float *beta, const int *cand,
                                                        float *h_pooled, float *cand_feat, float *h_nodes)
{
    constexpr int NB = POOL_INFLIGHT, RB = 8 * NB;            // rows per batch
    __shared__ float s_part[2][8][HD];
    const int tid = threadIdx.x, rg = tid >> 5, c4 = (tid & 31) * 4;
    float mean[4], rstd[4], g[4], be[4];
    for (int q = 0; q < 4; q++) {
        const int c = c4 + q;
        double su = 0, sq = 0;
        for (int r = 0; r < STAT_REP; r++) { su += stats[r * 256 + c]; sq += stats[r * 256 + HD + c]; }
        if (range_flag && (su != su || sq != sq)) __hip_atomic_store(range_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const double mean_d = su * inv_rows;
        double var = sq * inv_rows - mean_d * mean_d;
        if (var < 0) var = 0;
        mean[q] = (float)mean_d; rstd[q] = 1.0f / sqrtf((float)(var + BN_EPS)); g[q] = gamma[c]; be[q] = beta[c];
    }
    // instance sequence b(i), i < n_inst: rpr == 0: blockIdx.x, + gridDim.x, ...; else (matrix beyond the memory-side cache) the instances
    // of the row range [g rpr, (g+1) rpr) that ONE workgroup of the last product wrote front to back are taken from the back — what is
    // still cached first — by the S blocks (g, s), s = blockIdx.x / ranges
    int i_hi = 0, sub = 0, n_inst;
    if (rpr) {
        const int nranges = gridDim.x / S, gg = blockIdx.x % nranges;
        sub = blockIdx.x / nranges;
        const long long r0 = (long long)gg * rpr, r1 = r0 + rpr;
        const int i_lo = (int)((r0 + T - 1) / T);
        i_hi = (int)((r1 + T - 1) / T); if (i_hi > B) i_hi = B;
        const int n_g = i_hi - i_lo;
        n_inst = n_g > sub ? (n_g - sub + S - 1) / S : 0;
    } else n_inst = B > (int)blockIdx.x ? (B - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    auto inst = [&](int i) __attribute__((always_inline)) { return rpr ? i_hi - 1 - (sub + i * S) : (int)blockIdx.x + i * (int)gridDim.x; };
    const int nb = (T + RB - 1) / RB;                         // batches per instance
    const int nq = n_inst * nb;                               // batches of this block
    const int M = J > 0 ? T / J : 1;
    const unsigned invM = (unsigned)((0x100000000ull + (unsigned)M - 1) / (unsigned)M);   // v / M = umulhi(v, invM) for v < 65536
    // straight-line requests (a load behind an `if` would make every later wait drain the queue): rows beyond the instance / batches
    // beyond the sequence are clamped to a valid row and discarded where they are used
    auto request = [&](int q, f32x4 (&x)[NB], int (&cv)[NB]) __attribute__((always_inline)) {
        const int qq = q < nq ? q : (nq > 0 ? nq - 1 : 0);
        const int i = qq / nb, k = qq - i * nb;
        const int b = nq > 0 ? inst(i) : 0;
#pragma unroll
        for (int u = 0; u < NB; u++) {
            int v = k * RB + rg + 8 * u;
            v = v < T ? v : T - 1;
            const f32x4 *src = reinterpret_cast<const f32x4 *>(z + ((size_t)b * T + v) * HD + c4);
            x[u] = NT ? __builtin_nontemporal_load(src) : *src;
            cv[u] = J > 0 ? cand[(size_t)b * J + (int)__umulhi((unsigned)v, invM)] : -1;
        }
    };
    f32x4 xa[NB], xb[NB];
    int ca[NB], cb[NB];
    request(0, xa, ca);
    request(1, xb, cb);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    auto consume = [&](int q, const f32x4 (&x)[NB], const int (&cv)[NB]) __attribute__((always_inline)) {
        const int i = q / nb, k = q - i * nb;
        const int b = inst(i);
#pragma unroll
        for (int u = 0; u < NB; u++) {
            const int v = k * RB + rg + 8 * u;
            if (v < T) {
                const float hv[4] = {bn_relu(x[u][0], mean[0], rstd[0], g[0], be[0]), bn_relu(x[u][1], mean[1], rstd[1], g[1], be[1]),
                                     bn_relu(x[u][2], mean[2], rstd[2], g[2], be[2]), bn_relu(x[u][3], mean[3], rstd[3], g[3], be[3])};
                for (int qq = 0; qq < 4; qq++) acc[qq] += hv[qq];
                if (h_nodes) *reinterpret_cast<float4 *>(h_nodes + ((size_t)b * T + v) * HD + c4) = make_float4(hv[0], hv[1], hv[2], hv[3]);
                if (cv[u] == v)                                                         // candidate gather (ac:197-207); J = 0: cv = -1
                    *reinterpret_cast<float4 *>(cand_feat + ((size_t)b * J + (int)__umulhi((unsigned)v, invM)) * HD + c4) = make_float4(hv[0], hv[1], hv[2], hv[3]);
            }
        }
        if (k == nb - 1) {                                                              // the instance is complete: fold the 8 row groups
            float *sp = &s_part[i & 1][0][0];
            *reinterpret_cast<float4 *>(sp + rg * HD + c4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            for (int qq = 0; qq < 4; qq++) acc[qq] = 0.f;
            __syncthreads();                                                            // (buffer i & 1 is written again two instances later, behind the next barrier)
            if (tid < HD) {
                float t = 0.f;
                for (int r = 0; r < 8; r++) t += sp[r * HD + tid];
                h_pooled[(size_t)b * HD + tid] = t * (1.0f / (float)T);                 // sparse mm with 1/T entries (gcn:192)
            }
            // The stream above picks candidate j's row out of job j's block of rows [j M, (j+1) M) — where the environment's candidates
            // always lie (ppo:306-309: the next operation of job j).  A caller-made candidate outside its job's block (the other
            // forward paths accept any row < T) takes a dependent gather here instead of leaving a stale row: never on the rollout path.
            for (int j = rg; j < J; j += 8) {
                const int c = cand[(size_t)b * J + j];
                if (c >= 0 && c < T && (int)__umulhi((unsigned)c, invM) != j) {
                    const f32x4 xv = *reinterpret_cast<const f32x4 *>(z + ((size_t)b * T + c) * HD + c4);
                    *reinterpret_cast<float4 *>(cand_feat + ((size_t)b * J + j) * HD + c4) =
                        make_float4(bn_relu(xv[0], mean[0], rstd[0], g[0], be[0]), bn_relu(xv[1], mean[1], rstd[1], g[1], be[1]),
                                    bn_relu(xv[2], mean[2], rstd[2], g[2], be[2]), bn_relu(xv[3], mean[3], rstd[3], g[3], be[3]));
                }
            }
        }
    };
    for (int q = 0; q < nq; q += 2) {
        consume(q, xa, ca);
        request(q + 2, xa, ca);
        if (q + 1 < nq) consume(q + 1, xb, cb);
        request(q + 3, xb, cb);
    }
}

// (Measured on the way, J10M10 x 8192 / J20M20 x 2048, us per launch, all on one box unless said otherwise: the earlier kernel — one
// instance at a time, two barriers, candidate rows re-read behind their index, plain loads by accident — 116 / 108; the same with
// real non-temporal loads 105 / 105; this kernel 108 / 101 on that box and 86-92 / 86 on faster ones (the launch is bimodal from run to
// run, 92 or 107 on the same box: placement of the 419 MB matrix); without the candidate index loads 103 / 99; 4 or 8 rows per batch
// the same, 13 slower; plain instance order instead of back-to-front per writer range 8-20 slower.  A bare non-temporal read of the
// matrix in this order takes 69-76 us there (tools/ubench/read_after_write.hip).  Round 4's earlier attempts: ONE WAVE per instance,
// 122 / 131 against 116 / 107; a register cap for four resident blocks per CU, no change — both still with the accidental plain loads.)
// broadcast a [128] vector to [B,128] (first step: learned `_input` instead of h_m_prev, ac:229-233)
__global__ void k_bcast128(int B, const float *v, float *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B * HD) out[i] = v[i & 127];
}

// machine nodes: BatchNorm over all B*M rows (ac:434) and mean over M (ac:444). block = instance, thread = column
__global__ __launch_bounds__(128) void k_mach_bn_pool(unsigned *range_flag, int B, int M, float *node /*in: pre-BN, out: normalised*/, const double *stats, double inv_rows,
                                                     const float *gamma, const float *beta, float *h_pooled)
{
    const int b = blockIdx.x, c = threadIdx.x;
    double su = 0, sq = 0;
    for (int r = 0; r < STAT_REP; r++) { su += stats[r * 256 + c]; sq += stats[r * 256 + HD + c]; }
    if (range_flag && (su != su || sq != sq)) __hip_atomic_store(range_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const double mean_d = su * inv_rows;
    double var = sq * inv_rows - mean_d * mean_d;
    if (var < 0) var = 0;
    const float mean = (float)mean_d, rstd = 1.0f / sqrtf((float)(var + BN_EPS)), g = gamma[c], be = beta[c];
    float acc = 0.f;
    for (int m = 0; m < M; m++) {
        const size_t i = ((size_t)b * M + m) * HD + c;
        const float y = (node[i] - mean) * rstd * g + be;
        node[i] = y;
        acc += y;
    }
    h_pooled[(size_t)b * HD + c] = acc / (float)M;
}

// ---------------------------------------------------------------------------------------------
// categorical sampling / argmax (agent:22-72). thread = instance.
__global__ void k_sample(int B, int n, const float *prob, int greedy, uint64_t seed, uint64_t counter, int *idx_out, float *logp_out,
                         const int *gather_from, int *gathered_out)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float *p = prob + (size_t)b * n;
    const int pick = pick_action(p, n, b, greedy, seed, counter);
    idx_out[b] = pick;
    if (logp_out) logp_out[b] = logf(p[pick]);
    if (gather_from && gathered_out) gathered_out[b] = gather_from[(size_t)b * n + pick];
}

// =================================================================================================
// Per-instance BatchNorm.  The reference evaluates greedily with env_batch = 1 (validate.py:60-297): every
// BatchNorm then normalises over the rows of ONE instance (T task rows, M machine rows).  Batching those 100 serial
// episodes means one statistic set per instance, and since all rows of an instance can be given to one workgroup, every
// BatchNorm boundary is a workgroup-local reduction: no atomics, no grid-wide dependency, the whole GIN encoder (gin0 +
// five products + six BatchNorms + graph pool + candidate gather) is ONE launch with one workgroup per instance, the
// machine path (projections + 3 GAT passes + BatchNorm + pool) another.  Activations ping-pong through this instance's
// rows of the global workspaces; the heads / selection kernels are the batched ones (they contain no BatchNorm).
struct GinInstArgs {
    int B, T, J;
    const void *tfea; int feat_f64;          // [B*T,12] obs dtype
    const int *ell_col; const float *ell_val;
    const float *W0, *b0;                    // mlps.0.linears.0 [128,12], [128]
    const float *Wt[5]; const float *bias[5];   // the five 128x128 products (transposed weights), in execution order
    const float *gamma[6], *beta[6];         // BatchNorms in execution order: mlps.0.bn0, mlps.0.bn1, outer bn0, mlps.1.bn0, mlps.1.bn1, outer bn1
    float *zA, *zB;                          // [B*T(+pad),128] workspaces
    const int *cand;                         // [B,J] or NULL
    float *h_pooled, *cand_feat, *h_nodes;   // [B,128], [B*J,128], optional [B*T,128]
};
template <typename OBS>
__global__ __launch_bounds__(512) void k_gin_inst(GinInstArgs A)
{
    extern __shared__ __align__(16) unsigned char smem[];
    float *s_w = reinterpret_cast<float *>(smem);                 // 128*128, swizzled
    float *s_a = s_w + HD * HD;                                   // 8 * 16 * LDA16
    float *s_bn = s_a + 8 * 16 * LDA16;                           // scale | shift of the producer's BatchNorm
    double *s_st = reinterpret_cast<double *>(s_bn + 2 * HD);     // [2][256] column sum | sum of squares (ping-pong per layer)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int j = lane & 31, h = lane >> 5, c4 = j * 4;
    const int m = lane & 15, q = lane >> 4, qo = q & 1;
    const int b = blockIdx.x, T = A.T;
    const size_t r0 = (size_t)b * T;                              // first row of this instance
    const int ntile = (T + 15) / 16;
    const double invT = 1.0 / (double)T;
    float *my_a = s_a + wave * 16 * LDA16;
    const float *ap = my_a + m * LDA16 + q;
    const float *bp = s_w + q * HD + m;
    int bo[8];
    for (int c = 0; c < 8; c++) bo[c] = (c ^ qo) * 16;
    for (int i = tid; i < 512; i += 512) s_st[i] = 0.0;
    __syncthreads();
    // ---- layer 0: z0 = W0 agg(x) + b0 (gcn:125-153) -> zA, column sums -> s_st[0]
    {
        const int c = tid & 127, rg = tid >> 7;                  // column, row group (4)
        float w[12];
        for (int k = 0; k < 12; k++) w[k] = A.W0[c * 12 + k];
        const float bc = A.b0[c];
        double su = 0, sq = 0;
        for (int v = rg; v < T; v += 4) {
            const size_t g = r0 + v;
            const int c0 = A.ell_col[g * 2], c1 = A.ell_col[g * 2 + 1];
            const double v0 = (double)A.ell_val[g * 2], v1 = (double)A.ell_val[g * 2 + 1];
            const int deg = 1 + (c0 >= 0) + (c1 >= 0);
            float a = 0.f;
            for (int k = 0; k < 12; k++) {
                auto fe = [&](size_t row) { return A.feat_f64 ? (double)(float)reinterpret_cast<const double *>(A.tfea)[row * 12 + k]
                                                              : (double)reinterpret_cast<const float *>(A.tfea)[row * 12 + k]; };
                double acc = fe(g);
                if (c0 >= 0) acc += v0 * fe(r0 + c0);
                if (c1 >= 0) acc += v1 * fe(r0 + c1);
                a = fmaf((float)(acc / (double)deg), w[k], a);
            }
            a += bc;
            A.zA[g * HD + c] = a;
            su += (double)a; sq += (double)a * (double)a;
        }
        atomicAdd(&s_st[c], su); atomicAdd(&s_st[HD + c], sq);
    }
    __syncthreads();
    const float *in = A.zA;
    float *out = A.zB;
    for (int L = 0; L < 5; L++) {
        double *st_in = s_st + (L & 1) * 256, *st_out = s_st + ((L + 1) & 1) * 256;
        stage_w16(s_w, A.Wt[L], tid);
        // BatchNorm feeding this product: L=0: mlps.0.bn0, 1: mlps.0.bn1, 2: outer bn0 (then aggregation), 3: mlps.1.bn0, 4: mlps.1.bn1
        stage_bn_local(s_bn, st_in, invT, A.gamma[L], A.beta[L], tid);
        if (tid < 256) st_out[tid] = 0.0;
        __syncthreads();
        const float sc0 = s_bn[c4], sc1 = s_bn[c4 + 1], sc2 = s_bn[c4 + 2], sc3 = s_bn[c4 + 3];
        const float sh0 = s_bn[HD + c4], sh1 = s_bn[HD + c4 + 1], sh2 = s_bn[HD + c4 + 2], sh3 = s_bn[HD + c4 + 3];
        float bias8[8];
        for (int c = 0; c < 8; c++) bias8[c] = A.bias[L][c * 16 + m];
        double su[8], sq[8];
        for (int c = 0; c < 8; c++) { su[c] = 0; sq[c] = 0; }
        for (int t = wave; t < ntile; t += 8) {
#pragma unroll
            for (int p = 0; p < 8; p++) {
                const int v = t * 16 + 2 * p + h;
                float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
                if (v < T) {
                    const float4 x = *reinterpret_cast<const float4 *>(in + (r0 + v) * HD + c4);
                    v0 = bn_relu_ss(x.x, sc0, sh0); v1 = bn_relu_ss(x.y, sc1, sh1); v2 = bn_relu_ss(x.z, sc2, sh2); v3 = bn_relu_ss(x.w, sc3, sh3);
                    if (L == 2) {                                 // neighbour aggregation of h = relu(bn_outer0(z)) (gcn:125-149)
                        const int c0 = A.ell_col[(r0 + v) * 2], c1 = A.ell_col[(r0 + v) * 2 + 1];
                        double a0 = v0, a1 = v1, a2 = v2, a3 = v3;
                        int deg = 1;
                        if (c0 >= 0) {
                            const double w = (double)A.ell_val[(r0 + v) * 2];
                            const float4 y = *reinterpret_cast<const float4 *>(in + (r0 + c0) * HD + c4);
                            a0 += w * (double)bn_relu_ss(y.x, sc0, sh0); a1 += w * (double)bn_relu_ss(y.y, sc1, sh1);
                            a2 += w * (double)bn_relu_ss(y.z, sc2, sh2); a3 += w * (double)bn_relu_ss(y.w, sc3, sh3);
                            deg++;
                        }
                        if (c1 >= 0) {
                            const double w = (double)A.ell_val[(r0 + v) * 2 + 1];
                            const float4 y = *reinterpret_cast<const float4 *>(in + (r0 + c1) * HD + c4);
                            a0 += w * (double)bn_relu_ss(y.x, sc0, sh0); a1 += w * (double)bn_relu_ss(y.y, sc1, sh1);
                            a2 += w * (double)bn_relu_ss(y.z, sc2, sh2); a3 += w * (double)bn_relu_ss(y.w, sc3, sh3);
                            deg++;
                        }
                        const double inv = 1.0 / (double)deg;
                        v0 = (float)(a0 * inv); v1 = (float)(a1 * inv); v2 = (float)(a2 * inv); v3 = (float)(a3 * inv);
                    }
                }
                float *d = my_a + (2 * p + h) * LDA16 + c4;
                *reinterpret_cast<float2 *>(d) = make_float2(v0, v1);
                *reinterpret_cast<float2 *>(d + 2) = make_float2(v2, v3);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            f32x4 acc[8];
#pragma unroll
            for (int c = 0; c < 8; c++)
#pragma unroll
                for (int i = 0; i < 4; i++) acc[c][i] = bias8[c];
            mfma_tile16(ap, bp, bo, acc);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int c = 0; c < 8; c++)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int v = t * 16 + 4 * q + i;
                    if (v < T) {                                  // rows >= T belong to the next instance
                        const float x = acc[c][i];
                        out[(r0 + v) * HD + c * 16 + m] = x;
                        su[c] += (double)x; sq[c] += (double)x * (double)x;
                    }
                }
        }
        for (int c = 0; c < 8; c++) {
            double a = su[c], bq = sq[c];
            a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
            bq += __shfl_xor(bq, 16); bq += __shfl_xor(bq, 32);
            if (q == 0) { atomicAdd(&st_out[c * 16 + m], a); atomicAdd(&st_out[HD + c * 16 + m], bq); }
        }
        __syncthreads();                                          // rows of `out` and st_out complete (vmcnt drained by the barrier)
        const float *tmp = in; in = out; out = const_cast<float *>(tmp);
    }
    // ---- h = relu(bn_outer1(z5)); graph mean pool (gcn:192) and candidate gather (ac:197-207)
    stage_bn_local(s_bn, s_st + 256, invT, A.gamma[5], A.beta[5], tid);     // five products: the last sums are in buffer 1
    __syncthreads();
    {
        const int rg = tid >> 5, cc = (tid & 31) * 4;             // 16 row groups x 4 columns
        const float sc0 = s_bn[cc], sc1 = s_bn[cc + 1], sc2 = s_bn[cc + 2], sc3 = s_bn[cc + 3];
        const float sh0 = s_bn[HD + cc], sh1 = s_bn[HD + cc + 1], sh2 = s_bn[HD + cc + 2], sh3 = s_bn[HD + cc + 3];
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int v = rg; v < T; v += 16) {
            const float4 x = *reinterpret_cast<const float4 *>(in + (r0 + v) * HD + cc);
            const float4 y = make_float4(bn_relu_ss(x.x, sc0, sh0), bn_relu_ss(x.y, sc1, sh1), bn_relu_ss(x.z, sc2, sh2), bn_relu_ss(x.w, sc3, sh3));
            a0 += y.x; a1 += y.y; a2 += y.z; a3 += y.w;
            if (A.h_nodes) *reinterpret_cast<float4 *>(A.h_nodes + (r0 + v) * HD + cc) = y;
        }
        float *part = s_a + rg * HD + cc;                         // [16][128] partial sums (tiles are free now)
        part[0] = a0; part[1] = a1; part[2] = a2; part[3] = a3;
        for (int jj = rg; jj < A.J; jj += 16) {
            if (!A.cand) break;
            const int v = A.cand[(size_t)b * A.J + jj];
            const float4 x = *reinterpret_cast<const float4 *>(in + (r0 + v) * HD + cc);
            *reinterpret_cast<float4 *>(A.cand_feat + ((size_t)b * A.J + jj) * HD + cc) =
                make_float4(bn_relu_ss(x.x, sc0, sh0), bn_relu_ss(x.y, sc1, sh1), bn_relu_ss(x.z, sc2, sh2), bn_relu_ss(x.w, sc3, sh3));
        }
    }
    __syncthreads();
    if (tid < HD) {
        float t = 0.f;
        for (int r = 0; r < 16; r++) t += s_a[r * HD + tid];
        A.h_pooled[(size_t)b * HD + tid] = t * (1.0f / (float)T);
    }
}

struct GatInstArgs {
    int B, M;
    const void *f1, *f2; int feat_f64;       // m_fea1 [B*M,6], m_fea2 [B*M,8]
    const float *W1, *W2, *Wt, *gat_a, *gamma, *beta;
    float *node;                             // [B*M,128] normalised machine nodes (ac:434)
    float *h_pooled;                         // [B,128] (ac:444)
};
__global__ __launch_bounds__(512) void k_gat_inst(GatInstArgs A)
{
    extern __shared__ __align__(16) unsigned char smem[];
    float *s_w = reinterpret_cast<float *>(smem);
    float *s_a = s_w + HD * HD;                                   // 8 tiles
    float *s_bn = s_a + 8 * 16 * LDA16;
    double *s_st = reinterpret_cast<double *>(s_bn + 2 * HD);     // [256]
    float *s_pool = reinterpret_cast<float *>(s_st + 512);        // [128]
    float *s_feat = s_pool + HD;                                  // 8 waves * 16 rows * 8
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int j = lane & 31, h = lane >> 5, c4 = j * 4;
    const int m = lane & 15, q = lane >> 4, qo = q & 1;
    const int b = blockIdx.x, M = A.M, N = 2 * M;                 // tile rows of this instance: (machine, node) interleaved
    stage_w16(s_w, A.Wt, tid);
    if (tid < 256) s_st[tid] = 0.0;
    if (tid < HD) s_pool[tid] = 0.f;
    __syncthreads();
    float *my_a = s_a + wave * 16 * LDA16;
    float *my_f = s_feat + wave * 128;
    const float *ap = my_a + m * LDA16 + q;
    const float *bp = s_w + q * HD + m;
    int bo[8];
    for (int c = 0; c < 8; c++) bo[c] = (c ^ qo) * 16;
    float asrc[8], adst[8];
    for (int c = 0; c < 8; c++) { asrc[c] = A.gat_a[c * 16 + m]; adst[c] = A.gat_a[HD + c * 16 + m]; }
    const int t = wave;                                           // 2M <= 128 rows: at most one tile per wave
    const bool active = t * 16 < N;
    float mv[8][2];                                               // node means of this lane's two machines, 8 column blocks
    for (int c = 0; c < 8; c++) { mv[c][0] = 0.f; mv[c][1] = 0.f; }
    if (active) {
        {   // feature words of the tile: lane L < 32 -> row L/2, words 4*(L&1)..+3
            const int r = t * 16 + (lane >> 1), k0 = (lane & 1) * 4;
            float x[4] = {0.f, 0.f, 0.f, 0.f};
            if (lane < 32 && r < N) {
                const int u = r >> 1, node = r & 1, width = node ? 8 : 6;
                for (int k = 0; k < 4; k++)
                    if (k0 + k < width) {
                        const size_t idx = ((size_t)b * M + u) * width + k0 + k;
                        x[k] = A.feat_f64 ? (float)reinterpret_cast<const double *>(node ? A.f2 : A.f1)[idx]
                                          : reinterpret_cast<const float *>(node ? A.f2 : A.f1)[idx];
                    }
            }
            if (lane < 32) *reinterpret_cast<float4 *>(my_f + lane * 4) = make_float4(x[0], x[1], x[2], x[3]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int p = 0; p < 8; p++) {
            const int r = 2 * p + h;
            const float4 fa = *reinterpret_cast<const float4 *>(my_f + r * 8), fb = *reinterpret_cast<const float4 *>(my_f + r * 8 + 4);
            const float ff[8] = {fa.x, fa.y, fa.z, fa.w, fb.x, fb.y, fb.z, fb.w};
            float o[4];
            for (int x = 0; x < 4; x++) {
                float a = 0.f;
                for (int k = 0; k < 8; k++) {
                    const float w = h == 0 ? (k < 6 ? A.W1[(c4 + x) * 6 + k] : 0.f) : A.W2[(c4 + x) * 8 + k];
                    a = fmaf(ff[k], w, a);
                }
                o[x] = a;
            }
            float *d = my_a + r * LDA16 + c4;
            *reinterpret_cast<float2 *>(d) = make_float2(o[0], o[1]);
            *reinterpret_cast<float2 *>(d + 2) = make_float2(o[2], o[3]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
        for (int pass = 0; pass < 3; pass++) {
            f32x4 acc[8];
            for (int c = 0; c < 8; c++) for (int i = 0; i < 4; i++) acc[c][i] = 0.f;
            mfma_tile16(ap, bp, bo, acc);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int i = 2 * u;
                float s0 = 0.f, d0 = 0.f, d1 = 0.f;
                for (int c = 0; c < 8; c++) { const float z0 = acc[c][i], z1 = acc[c][i + 1]; s0 += asrc[c] * z0; d0 += adst[c] * z0; d1 += adst[c] * z1; }
                s0 = row_sum16(s0); d0 = row_sum16(d0); d1 = row_sum16(d1);
                float e00 = s0 + d0, e01 = s0 + d1;
                e00 = e00 > 0.f ? e00 : 0.2f * e00;
                e01 = e01 > 0.f ? e01 : 0.2f * e01;
                const float mx = fmaxf(e00, e01);
                const float x0 = __expf(e00 - mx), x1 = __expf(e01 - mx);
                const float inv = 1.0f / (x0 + x1);
                const float al0 = x0 * inv, al1 = x1 * inv;
                const int r = 4 * q + i;
                for (int c = 0; c < 8; c++) {
                    const float z0 = acc[c][i], z1 = acc[c][i + 1];
                    float n0 = al0 * z0 + al1 * z1, n1 = z1;
                    if (pass < 2) {
                        n0 = n0 > 0.f ? n0 : __expf(n0) - 1.0f;
                        n1 = n1 > 0.f ? n1 : __expf(n1) - 1.0f;
                        my_a[r * LDA16 + c * 16 + m] = n0;
                        my_a[(r + 1) * LDA16 + c * 16 + m] = n1;
                    } else mv[c][u] = (n0 + n1) * 0.5f;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        for (int c = 0; c < 8; c++) {                             // BatchNorm sums over the instance's M machines
            double a = 0, bq = 0;
            for (int u = 0; u < 2; u++)
                if (t * 16 + 4 * q + 2 * u < N) { a += (double)mv[c][u]; bq += (double)mv[c][u] * (double)mv[c][u]; }
            a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
            bq += __shfl_xor(bq, 16); bq += __shfl_xor(bq, 32);
            if (q == 0) { atomicAdd(&s_st[c * 16 + m], a); atomicAdd(&s_st[HD + c * 16 + m], bq); }
        }
    }
    __syncthreads();
    stage_bn_local(s_bn, s_st, 1.0 / (double)M, A.gamma, A.beta, tid);
    __syncthreads();
    if (active) {
        for (int c = 0; c < 8; c++) {
            const int col = c * 16 + m;
            float pl = 0.f;
            for (int u = 0; u < 2; u++) {
                const int r = t * 16 + 4 * q + 2 * u;
                if (r < N) {
                    const float y = fmaf(mv[c][u], s_bn[col], s_bn[HD + col]);            // ac:434 (no ReLU)
                    A.node[((size_t)b * M + (r >> 1)) * HD + col] = y;
                    pl += y;
                }
            }
            pl += __shfl_xor(pl, 16); pl += __shfl_xor(pl, 32);
            if (q == 0) atomicAdd(&s_pool[col], pl);
        }
    }
    __syncthreads();
    if (tid < HD) A.h_pooled[(size_t)b * HD + tid] = s_pool[tid] / (float)M;
}
static size_t inst_lds_bytes() { return (size_t)(HD * HD + 8 * 16 * LDA16 + 2 * HD) * 4 + 512 * 8 + (HD + 8 * 128) * 4 + 64; }

// =================================================================================================
// host side
struct mtfjsp_encoder {
    mtfjsp_encoder_config_t cfg;
    int T = 0;
    hipStream_t stream = nullptr;
    std::string err;
    std::map<std::string, float *> w;       // device copies, torch layout
    std::map<std::string, float *> wt;      // transposed [in,out] copies of the 128-wide Linear weights (split per 128-block of `in`)
    std::map<std::string, std::vector<float>> hostw;   // host copies of gat_layer.W / m_fea_*_fcl.weight (inputs of the fused projections)
    std::map<std::string, float *> wfused;  // per prefix + "1"/"2": (m_fea_k_fcl.weight^T . gat_layer.W)^T, [128,6] / [128,8]
    std::map<std::string, size_t> wx6_bytes;   // size of each wx6 image
    std::map<std::string, size_t> wx32_bytes;  // size of each wx32 image (a GIN Linear has both: one map keyed by name held whichever was uploaded last)
    std::map<std::string, void *> wx6;      // 128x128 Linear weights as 2 f16 planes (scaled) in k_gemm_x6's register-image order; GAT W / first Linear: 3 bf16 planes
    std::map<std::string, void *> wx32;     // GIN Linear weights as operand-piece planes in k_gin_res's (32x32x16) register-image order
    std::map<std::string, float> wx32_sinv; // ... 1 / the power-of-two scale folded into that image
    std::map<std::string, float> wx6_sinv;  // the same for the f16 images in wx6
    std::map<std::string, float *> wimg;    // the same blocks as per-wave register images for k_heads: [block][wave 8][g 8][lane 64][4]
    std::vector<void *> owned;
    int num_cu = 256;
    // workspaces
    float *zA = nullptr, *zB = nullptr;     // [B*T,128] ping-pong
    float *cand_feat = nullptr;             // [B*max(J,M),128]
    float *u = nullptr, *c1 = nullptr, *c2 = nullptr, *hm_b = nullptr, *pooled_int = nullptr;   // [B,128]
    bool hm_b_valid = false;                // hm_b holds the broadcast of the current job_actor._input (rebuilt when that weight is loaded again)
    float *node = nullptr;                  // [B*M,128]
    double *stats = nullptr;                // [8][STAT_REP][256]: slots 0..5 GIN layers, 6/7 machine path (alternating)
    bool gin_slot5_dirty = false;           // slots 0..4 zeroed by the job heads, slot 5 still holds sums (zeroed by the machine heads)
    bool gin_stats_clean = false;           // slots 0..5 are zero (the job-actor heads kernel zeroes them after their last reader)
    bool gat_stats_clean[2] = {false, false};
    int bn_mode = 0;                        // 0: BatchNorm statistics over the whole device batch; 1: per instance (validate.py semantics)
    bool defer_poll = false;                // forward entries do not poll the asynchronous failure words: only mtfjsp_encoder_check reports them (mtfjsp_encoder_set_deferred_poll)
    mtfjsp_stats_reduce_fn reduce_fn = nullptr; void *reduce_user = nullptr;   // exact multi-shard BatchNorm: sums of every BN summed over the shards
    double reduce_scale = 1.0;              // global rows / local rows
    // which products run with the f32 matrix instruction instead of the exact bf16 split (A/B reference; bits: 1 GIN products,
    // 2 GAT passes, 4 heads, 8 first GIN Linear on the VALU); default from MTFJSP_GEMM_F32MFMA / _GAT_ / _HEADS_ / MTFJSP_GIN0_VALU
    int f32_products = (getenv("MTFJSP_GEMM_F32MFMA") ? 1 : 0) | (getenv("MTFJSP_GAT_F32MFMA") ? 2 : 0) |
                       (getenv("MTFJSP_HEADS_F32MFMA") ? 4 : 0) | (getenv("MTFJSP_GIN0_VALU") ? 8 : 0);
    // resident GIN kernel (mtfjsp_gin_resident.h): eligibility decided at create time, then verified by a census launch
    bool res_ok = false; int res_ipc = 0, res_grid = 0;
    double *res_stats = nullptr;            // [2 sets][GR_STATS_SET] 64-bit count-carrying fixed-point words (mtfjsp_gin_resident.h); forward n uses set n & 1 and zeroes the other one
    unsigned long long *res_bar = nullptr;  // [17 * 16] barrier words
    unsigned *res_fail = nullptr;           // device address of *res_fail_host
    volatile unsigned *res_fail_host = nullptr;   // host-mapped word the kernel sets when a grid barrier times out: polled at every forward entry (no synchronisation)
    unsigned *range_flag = nullptr;         // device address of the next word of the same host-mapped block (flags_host[1]): an output was not a number
    long long range_fallbacks = 0;          // times the handle switched to the f32-instruction kernels because of it
    bool res_eligible = false;              // the shape can use the single-launch kernel (res_ok: and the census passed / no failure since)
    long long res_failures = 0, res_launches = 0;
    long long res_fail_at = getenv("MTFJSP_GIN_RES_FAIL_AT") ? atoll(getenv("MTFJSP_GIN_RES_FAIL_AT")) : 0;   // diagnostic: this launch's barriers time out
    long long range_fail_at = getenv("MTFJSP_RANGE_FAIL_AT") ? atoll(getenv("MTFJSP_RANGE_FAIL_AT")) : 0, job_forwards = 0;   // diagnostic: the n-th job-actor forward raises the range word (as a NaN output would)
    float *res_zspill = nullptr;            // [grid][4][2][1024] f32: the two row tiles per workgroup that do not fit the registers
    unsigned long long res_epoch = 0;
    int gat_slot = 0;                       // the machine-path slot of the NEXT forward; the other one is zeroed by that forward's heads kernel
    struct { bool valid = false; const void *f1 = nullptr, *f2 = nullptr; int slot = 0;
             bool heads = false; const float *h_pooled_o = nullptr; const uint8_t *mmask = nullptr; float *prob = nullptr, *h_pooled = nullptr, *machine_v = nullptr;
           } prefused;   // the GAT passes of the coming machine forward already ran inside the job actor's heads launch (k_headsx_gat3x); heads: the WHOLE machine forward did (k_headsx_gat3x_headsx)
    bool fuse_gat = !getenv("MTFJSP_NO_FUSED_GAT");
    bool fuse_mheads = !getenv("MTFJSP_NO_FUSED_MHEADS");       // the machine heads in the job heads + GAT launch behind an in-launch exchange of the node statistics (three launches per rollout step)
    struct { bool armed = false; float *prob = nullptr, *h_pooled = nullptr, *machine_v = nullptr; } mh;   // mtfjsp_encoder_arm_machine_heads
    unsigned long long *xw = nullptr;       // [2 sets][XW_SET] count-carrying words of that exchange; forward n uses set n & 1 and zeroes the other
    unsigned long long xw_epoch = 0;
    long long fused3_launches = 0;
    double xchg_fine_limit = getenv("MTFJSP_XCHG_FINE_LIMIT") ? atof(getenv("MTFJSP_XCHG_FINE_LIMIT")) : 2147483648.0;   // diagnostic: a lower limit sends ordinary contributions through the wide-range words
    long long fused3_fail_at = getenv("MTFJSP_FUSED3_FAIL_AT") ? atoll(getenv("MTFJSP_FUSED3_FAIL_AT")) : 0;   // diagnostic: this three-in-one launch waits for a workgroup that does not exist
    bool heads_hg8 = !getenv("MTFJSP_NO_HEADS_HG8");
    bool heads10 = !getenv("MTFJSP_NO_HEADS10");              // k_headsx10 (ten tiles per chunk) for groups of 7..10 tiles
    // the environment step as the tail of the machine heads' launch (mtfjsp_encoder_arm_env_step).  OFF unless MTFJSP_FUSED_ENV is set:
    // bit-identical (tests/test_fused_env_step_gpu.py) and one launch less per step, but measured SLOWER at the headline shape —
    // 220.4 against 217.6 us per step, three alternating runs on one box — because the heads' 8 waves take the 16 instances in two
    // rounds where k_env_grp16's 16 waves take them in one: the second round costs more than the launch boundary saves
    bool fuse_env = getenv("MTFJSP_FUSED_ENV") != nullptr;
    // Round 6: the step as the tail of the THREE-in-one launch (k_headsx_gat3x_headsx<1|2>) with its state lines requested 13 us ahead:
    // two launches per rollout step, bit-identical (tests/test_fused_env_step_gpu.py).  OFF unless MTFJSP_FUSED_ENV3 is set: measured
    // SLOWER again — 0.1772-0.1786 against 0.1750-0.1751 ms per step, three alternating runs on one box (tools/ab_bench_env3.sh,
    // profiles/r06_ab_env3.txt): the launch's 8 waves take the 16 instances in two rounds behind one another, which costs about 3 us
    // more than the launch boundary and the separate kernel's dispatch ramp together, warm state lines or not.
    bool fuse_env3 = getenv("MTFJSP_FUSED_ENV3") != nullptr;
    bool nodes_lds = !getenv("MTFJSP_FUSED3_NODES_HBM");         // three-in-one launch: the GAT part's node rows stay in LDS for the machine part (round 6)
    bool warm_heads = !getenv("MTFJSP_NO_WARM_HEADS");          // k_gin_res requests the heads launch's weight lines in its last phase
    struct { bool armed = false, done = false; EnvParams P; } env_step;          // groups of 8 instances in k_headsx when groups of 16 fill at most half the CUs
    // streaming GIN launches: the two inner Linears of an MLP in one launch behind a statistics-only pass (mtfjsp_gemm_pair.h).
    // MTFJSP_FUSE_PAIR=1: always; -1: where the activation matrix exceeds the 256 MB memory-side cache; 0 (default): never —
    // measured (round 5, J10M10 x 8192 / J20M20 x 2048): statistics-only pass 102 us + pair launch 203 us against 2 x 148 us for the two
    // launches: a consumer wave runs BOTH products of a step's four tiles back to back (8 tile products, ~10 k cycles per step) and
    // is the critical path; the 25 % of HBM traffic saved does not show because these launches are bound by the SIMDs' instruction
    // issue (~100 us per launch even with no output at all), not by memory
    int fuse_pair = getenv("MTFJSP_FUSE_PAIR") ? atoi(getenv("MTFJSP_FUSE_PAIR")) : 0;
    int stream_order = getenv("MTFJSP_NO_STREAM_ORDER") ? 0 : 1;   // streaming GIN launches: alternating row direction + non-temporal input reads (A/B switch)
    int fuse_pool = getenv("MTFJSP_FUSE_POOL") ? atoi(getenv("MTFJSP_FUSE_POOL")) : 1;   // run_gin: the last Linear's output pooled / gathered in a second pass' epilogue instead of stored
    int fuse_gin0 = getenv("MTFJSP_FUSE_GIN0") ? atoi(getenv("MTFJSP_FUSE_GIN0")) : 1;   // run_gin: the first Linear's output formed again by the second launch's producers instead of stored
    int pool_s = getenv("MTFJSP_POOL_S") ? atoi(getenv("MTFJSP_POOL_S")) : 4;      // k_job_pool_gather: blocks per row range of the last product (0: plain instance order)
    int stream_nt = getenv("MTFJSP_STREAM_NT") ? atoi(getenv("MTFJSP_STREAM_NT")) : 5;   // which readers use non-temporal loads: 1 BatchNorm+ReLU products, 2 aggregation product, 4 pool / gather
    mtfjsp_mfea1_ctx_t mf_ctx{}; bool mf_armed = false;
    struct FusedSample { bool armed = false; int greedy = 0; uint64_t seed = 0, counter = 0; int32_t *idx = nullptr; float *logp = nullptr;
                         const int32_t *gather_from = nullptr; int32_t *gathered = nullptr; } fs[2];   // [0] job actor, [1] machine actor
    int values_only = 0;                    // mtfjsp_encoder_arm_values_only: the next (job, machine) forward pair produces the critic values only
    bool vo_now = false;                    // (the forward in progress is such a forward)
    FusedSample fs1_consumed;               // the machine selection the last three-in-one launch consumed (restored when the machine forward that follows is NOT the one it ran)
    // timing
    bool timing = false;
    std::map<std::string, std::vector<std::pair<hipEvent_t, hipEvent_t>>> ev;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_free;
};
static thread_local std::string g_enc_err;

#define HIPCHK(e, call)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            (e)->err = std::string(#call) + ": " + hipGetErrorString(e_);                       \
            return MTFJSP_ERR_HIP;                                                              \
        }                                                                                       \
    } while (0)

template <typename Tp>
static int dalloc(mtfjsp_encoder *e, Tp **ptr, size_t n)
{
    HIPCHK(e, hipMalloc((void **)ptr, n * sizeof(Tp)));
    e->owned.push_back(*ptr);
    return 0;
}

static const char *REQUIRED[] = {
    "job_actor._input",
    "job_actor.encoder.feature_extract.mlps.0.linears.0.weight", "job_actor.encoder.feature_extract.mlps.0.linears.0.bias",
    "job_actor.encoder.feature_extract.mlps.0.linears.1.weight", "job_actor.encoder.feature_extract.mlps.0.linears.1.bias",
    "job_actor.encoder.feature_extract.mlps.0.linears.2.weight", "job_actor.encoder.feature_extract.mlps.0.linears.2.bias",
    "job_actor.encoder.feature_extract.mlps.0.batch_norms.0.weight", "job_actor.encoder.feature_extract.mlps.0.batch_norms.0.bias",
    "job_actor.encoder.feature_extract.mlps.0.batch_norms.1.weight", "job_actor.encoder.feature_extract.mlps.0.batch_norms.1.bias",
    "job_actor.encoder.feature_extract.mlps.1.linears.0.weight", "job_actor.encoder.feature_extract.mlps.1.linears.0.bias",
    "job_actor.encoder.feature_extract.mlps.1.linears.1.weight", "job_actor.encoder.feature_extract.mlps.1.linears.1.bias",
    "job_actor.encoder.feature_extract.mlps.1.linears.2.weight", "job_actor.encoder.feature_extract.mlps.1.linears.2.bias",
    "job_actor.encoder.feature_extract.mlps.1.batch_norms.0.weight", "job_actor.encoder.feature_extract.mlps.1.batch_norms.0.bias",
    "job_actor.encoder.feature_extract.mlps.1.batch_norms.1.weight", "job_actor.encoder.feature_extract.mlps.1.batch_norms.1.bias",
    "job_actor.encoder.feature_extract.batch_norms.0.weight", "job_actor.encoder.feature_extract.batch_norms.0.bias",
    "job_actor.encoder.feature_extract.batch_norms.1.weight", "job_actor.encoder.feature_extract.batch_norms.1.bias",
    "job_actor.o_policy.linears.0.weight", "job_actor.o_policy.linears.0.bias",
    "job_actor.o_policy.linears.1.weight", "job_actor.o_policy.linears.1.bias",
    "job_actor.o_policy.linears.2.weight", "job_actor.o_policy.linears.2.bias",
    "job_actor.job_critic.linears.0.weight", "job_actor.job_critic.linears.0.bias",
    "job_actor.job_critic.linears.1.weight", "job_actor.job_critic.linears.1.bias",
    "job_actor.job_critic.linears.2.weight", "job_actor.job_critic.linears.2.bias",
    "machine_actor.bn.weight", "machine_actor.bn.bias", "machine_actor.m_fea_1_fcl.weight", "machine_actor.m_fea_2_fcl.weight",
    "machine_actor.gat_layer.W", "machine_actor.gat_layer.a",
    "machine_actor.m_policy.linears.0.weight", "machine_actor.m_policy.linears.0.bias",
    "machine_actor.m_policy.linears.1.weight", "machine_actor.m_policy.linears.1.bias",
    "machine_actor.m_policy.linears.2.weight", "machine_actor.m_policy.linears.2.bias",
    "machine_actor.machine_critic.linears.0.weight", "machine_actor.machine_critic.linears.0.bias",
    "machine_actor.machine_critic.linears.1.weight", "machine_actor.machine_critic.linears.1.bias",
    "machine_actor.machine_critic.linears.2.weight", "machine_actor.machine_critic.linears.2.bias",
};

// the host-mapped failure word of the single-launch GIN kernel
static int res_alloc_fail_word(mtfjsp_encoder *e)
{
    void *h = nullptr, *d = nullptr;
    if (hipHostMalloc(&h, 64, hipHostMallocMapped) != hipSuccess) return 1;
    if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipHostFree(h); return 1; }
    e->res_fail_host = (volatile unsigned *)h; e->res_fail = (unsigned *)d; e->range_flag = (unsigned *)d + 1;
    e->res_fail_host[0] = 0u; e->res_fail_host[1] = 0u;
    return 0;
}
// census: every workgroup must be resident at once for the grid barriers to complete (bounded spins report it).  Synchronises the device.
static bool res_census(mtfjsp_encoder *e)
{
    if (hipDeviceSynchronize() != hipSuccess) return false;
    if (hipMemset(e->res_bar, 0, (size_t)17 * 16 * 8) != hipSuccess) return false;      // fresh barrier words (a timed-out launch leaves them inconsistent)
    e->res_epoch = 0;
    *e->res_fail_host = 0u;
    GinResArgs a{};
    a.bar = e->res_bar; a.epoch = e->res_epoch++; a.fail = e->res_fail; a.barrier_only = 1;
    hipLaunchKernelGGL(k_gin_res, dim3(e->res_grid), dim3(256), gin_res_lds_bytes(), nullptr, a);
    if (hipDeviceSynchronize() != hipSuccess || *e->res_fail_host) {
        (void)hipMemset(e->res_bar, 0, (size_t)17 * 16 * 8); e->res_epoch = 0; *e->res_fail_host = 0u;
        return false;
    }
    return true;
}
// Polled at the entry of every forward that may use the single-launch kernel (a plain read of host memory) and, after a stream
// synchronisation, by mtfjsp_encoder_check: a grid barrier of an earlier launch timed out (its workgroups were not all resident,
// e.g. another process or stream held compute units).  Everything enqueued on this handle since that launch is invalid.  The
// handle falls back to the six streaming launches (mtfjsp_encoder_check re-runs the census and re-enables the single launch
// when the device can hold the grid again) and the caller is told to repeat the work: MTFJSP_ERR_RETRY.
static bool split_products_in_use(const mtfjsp_encoder *e) { return (e->f32_products & 15) != 15; }
static int res_poll_failure(mtfjsp_encoder *e)
{
    if (!e->res_fail_host) return MTFJSP_OK;
    if (e->res_fail_host[0]) {
        // (first: a launch whose statistics never completed computes on garbage and may well have raised the range word too — that
        // says nothing about the operands)
        (void)hipStreamSynchronize(e->stream);
        e->res_ok = false; e->res_failures++;
        e->res_fail_host[0] = 0u; e->res_fail_host[1] = 0u;
        (void)hipMemset(e->res_bar, 0, (size_t)17 * 16 * 8); e->res_epoch = 0;
        if (e->res_stats) (void)hipMemset(e->res_stats, 0, (size_t)2 * GR_STATS_SET * 8);
        if (e->xw) (void)hipMemset(e->xw, 0, (size_t)2 * XW_SET * 8);
        e->xw_epoch = 0; e->prefused.valid = false; e->prefused.heads = false;
        e->gin_stats_clean = false; e->gat_stats_clean[0] = e->gat_stats_clean[1] = false;   // (the heads launches behind it may have summed NaNs)
        e->err = "single-launch GIN kernel / fused heads launch: a grid-wide statistics exchange timed out (the workgroups were not co-resident); every output "
                 "enqueued since that launch is invalid and must be recomputed; the handle now uses the streaming launches";
        return MTFJSP_ERR_RETRY;
    }
    if (e->res_fail_host[1]) {
        // An output of an earlier forward was not a number.  On the split-product kernels that is what an activation beyond the
        // f16 range (65 504) turns into — BatchNorm outputs scaled by a large gamma, neighbour sums with large edge weights, GAT
        // mixtures — (inf | -inf) operand pieces whose products cancel to NaN and spread through the BatchNorm statistics to every
        // output.  The reference computes in f32 / f64 and has no such limit: the handle switches to the f32-instruction kernels
        // (product mode 15; mtfjsp_encoder_set_product_mode(0) goes back) and the caller repeats the forward.
        (void)hipStreamSynchronize(e->stream);
        e->res_fail_host[1] = 0u;
        if (split_products_in_use(e)) {
            e->f32_products |= 15; e->range_fallbacks++;
            e->gin_stats_clean = false; e->gat_stats_clean[0] = e->gat_stats_clean[1] = false;   // the accumulators hold NaNs
            e->err = "an activation left the f16 range of the split products (outputs were not numbers): the handle now uses the "
                     "f32-instruction kernels; repeat the forward(s) enqueued since";
            return MTFJSP_ERR_RETRY;
        }
    }
    return MTFJSP_OK;
}

extern "C" const char *mtfjsp_encoder_last_error(mtfjsp_encoder_t e) { return e ? e->err.c_str() : g_enc_err.c_str(); }

extern "C" int mtfjsp_encoder_create(const mtfjsp_encoder_config_t *cfg, mtfjsp_encoder_t *out)
{
    if (!cfg || !out) { g_enc_err = "null argument"; return MTFJSP_ERR_ARG; }
    if (cfg->hidden != HD || cfg->n_job < 1 || cfg->n_job > 64 || cfg->n_machine < 2 || cfg->n_machine > 64 || cfg->batch < 1) {
        g_enc_err = "bad encoder configuration (hidden must be 128, n_job<=64, 2<=n_machine<=64)"; return MTFJSP_ERR_ARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { g_enc_err = "no HIP device available"; return MTFJSP_ERR_HIP; }
    if (cfg->device_id < 0 || cfg->device_id >= ndev) { g_enc_err = "device_id out of range"; return MTFJSP_ERR_ARG; }
    if (hipSetDevice(cfg->device_id) != hipSuccess) { g_enc_err = "hipSetDevice failed"; return MTFJSP_ERR_HIP; }
    mtfjsp_encoder *e = new mtfjsp_encoder();
    e->cfg = *cfg;
    e->T = cfg->n_job * cfg->n_machine;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device_id) == hipSuccess) e->num_cu = prop.multiProcessorCount;
    const size_t B = cfg->batch, T = e->T, M = cfg->n_machine, J = cfg->n_job, R = J > M ? J : M;
    int rc = 0;
    // every [rows,128] workspace is padded to whole 32-row tiles (+1 tile) and zero-filled once: the matrix-core kernels
    // load and store whole tiles without bounds checks
    auto rows_padded = [](size_t rows) { return ((rows + 31) / 32 + 1) * 32; };
    auto dalloc_rows = [&](float **ptr, size_t rows) {
        const size_t n = rows_padded(rows) * HD;
        if (dalloc(e, ptr, n)) return 1;
        return hipMemset(*ptr, 0, n * sizeof(float)) == hipSuccess ? 0 : 1;
    };
    rc |= dalloc_rows(&e->zA, B * T); rc |= dalloc_rows(&e->zB, B * T);
    rc |= dalloc_rows(&e->cand_feat, B * R);
    rc |= dalloc_rows(&e->u, B); rc |= dalloc_rows(&e->c1, B); rc |= dalloc_rows(&e->c2, B); rc |= dalloc_rows(&e->hm_b, B);
    rc |= dalloc_rows(&e->pooled_int, B);
    rc |= dalloc_rows(&e->node, B * M);
    rc |= dalloc(e, &e->stats, 8 * STAT_REP * 256);
    rc |= res_alloc_fail_word(e);
    rc |= dalloc(e, &e->xw, (size_t)2 * XW_SET);
    if (!rc && hipMemset(e->xw, 0, (size_t)2 * XW_SET * 8) != hipSuccess) rc = 1;
    if (!rc && hipMemset(e->stats, 0, 8 * STAT_REP * 256 * sizeof(double)) == hipSuccess) { e->gin_stats_clean = true; e->gat_stats_clean[0] = e->gat_stats_clean[1] = true; }
    if (rc) { g_enc_err = e->err; mtfjsp_encoder_destroy(e); return MTFJSP_ERR_HIP; }
    const int lds16 = (int)gemm16_lds_bytes();
    (void)hipFuncSetAttribute((const void *)k_gemm16<EPI_PLAIN, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds16);
    (void)hipFuncSetAttribute((const void *)k_gemm16<EPI_TANH, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds16);
    (void)hipFuncSetAttribute((const void *)k_gemm16<EPI_TANH, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds16);
    (void)hipFuncSetAttribute((const void *)k_gemm16p<PRO_BNRELU>, hipFuncAttributeMaxDynamicSharedMemorySize, lds16);
    (void)hipFuncSetAttribute((const void *)k_gemm16p<PRO_AGG>, hipFuncAttributeMaxDynamicSharedMemorySize, lds16);
    (void)hipFuncSetAttribute((const void *)k_gat3, hipFuncAttributeMaxDynamicSharedMemorySize, lds16);
    (void)hipFuncSetAttribute((const void *)k_gat3x, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gat3x_lds_bytes());
    (void)hipFuncSetAttribute((const void *)k_gemm_x6<PRO_BNRELU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_x6_lds_bytes());
    (void)hipFuncSetAttribute((const void *)k_gemm_x6<PRO_AGG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_x6_lds_bytes());
    (void)hipFuncSetAttribute((const void *)k_gemm_x6<PRO_GIN0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_x6_lds_bytes());
    (void)hipFuncSetAttribute((const void *)k_gemm_x6<PRO_GIN0BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_x6_lds_bytes());
    if (hipFuncSetAttribute((const void *)k_gemm_x6f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_x6f_lds_bytes()) != hipSuccess) e->fuse_pair = 0;
    (void)hipFuncSetAttribute((const void *)k_gin_inst<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)inst_lds_bytes());
    (void)hipFuncSetAttribute((const void *)k_gin_inst<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)inst_lds_bytes());
    (void)hipFuncSetAttribute((const void *)k_gat_inst, hipFuncAttributeMaxDynamicSharedMemorySize, (int)inst_lds_bytes());
    (void)hipFuncSetAttribute((const void *)k_heads, hipFuncAttributeMaxDynamicSharedMemorySize, (int)heads_lds_bytes());
    (void)hipFuncSetAttribute((const void *)k_headsx, hipFuncAttributeMaxDynamicSharedMemorySize, (int)headsx_lds_bytes());
    if (hipFuncSetAttribute((const void *)k_headsx10, hipFuncAttributeMaxDynamicSharedMemorySize, (int)headsx10_lds_bytes()) != hipSuccess) e->heads10 = false;
    (void)hipFuncSetAttribute((const void *)k_headsx_envstep<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(headsx_lds_bytes() > EnvGrpDynLds<1>::bytes ? headsx_lds_bytes() : EnvGrpDynLds<1>::bytes));
    (void)hipFuncSetAttribute((const void *)k_headsx_envstep<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(headsx_lds_bytes() > EnvGrpDynLds<1>::bytes ? headsx_lds_bytes() : EnvGrpDynLds<1>::bytes));
    (void)hipFuncSetAttribute((const void *)k_headsx_gat3x, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(headsx_lds_bytes() > gat3x_lds_bytes() ? headsx_lds_bytes() : gat3x_lds_bytes()));
#if !MTFJSP_BODY_FUNCS
    (void)hipFuncSetAttribute((const void *)k_headsx_values, hipFuncAttributeMaxDynamicSharedMemorySize, (int)headsx_lds_bytes());
    (void)hipFuncSetAttribute((const void *)k_headsx_gat3x_headsx<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fused3_lds_bytes());
    (void)hipFuncSetAttribute((const void *)k_headsx_gat3x_headsx<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fused3_lds_bytes());
    (void)hipFuncSetAttribute((const void *)k_headsx_gat3x_headsx<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fused3_lds_bytes());
    (void)hipFuncSetAttribute((const void *)k_headsx_gat3x_headsx<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fused3_lds_bytes());
    (void)hipFuncSetAttribute((const void *)k_headsx_gat3x_headsx<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fused3_lds_bytes());
    if (hipFuncSetAttribute((const void *)k_headsx_gat3x_headsx<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)fused3_lds_bytes()) != hipSuccess) e->fuse_mheads = false;
#else
    e->fuse_mheads = false;
#endif
    {   // resident GIN kernel: whole instances per workgroup, at most 576 rows, one workgroup per CU
        const int T = e->T, B = cfg->batch;
        if (!getenv("MTFJSP_NO_RESIDENT_GIN") && T >= GR_MINT && T <= GR_MAXT) {
            const int ipc = (B + e->num_cu - 1) / e->num_cu;
            const int grid = (B + ipc - 1) / ipc;
            // (grid + 7) / 8 <= 63: a count-carrying statistics word holds the arrivals of one dispatch group in 6 bits (gr_fix_encode)
            if (ipc * T <= GR_ROWS && ipc <= GR_MAXIPC && grid <= e->num_cu && (grid + 7) / 8 <= 63 && ipc * cfg->n_job <= GR_MAXCAND &&
                hipFuncSetAttribute((const void *)k_gin_res, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gin_res_lds_bytes()) == hipSuccess) {
                int rc = dalloc(e, &e->res_stats, (size_t)2 * GR_STATS_SET) | dalloc(e, &e->res_bar, (size_t)17 * 16) |
                         dalloc(e, &e->res_zspill, (size_t)grid * 4 * (GR_NT - GR_NRES) * 1024);
                if (!rc && hipMemset(e->res_stats, 0, (size_t)2 * GR_STATS_SET * 8) == hipSuccess &&
                    hipMemset(e->res_bar, 0, (size_t)17 * 16 * 8) == hipSuccess) {
                    e->res_ipc = ipc; e->res_grid = grid; e->res_eligible = true;
                    e->res_ok = res_census(e);
                }
            }
        }
    }
    *out = e;
    return MTFJSP_OK;
}

extern "C" int mtfjsp_encoder_destroy(mtfjsp_encoder_t e)
{
    if (!e) return MTFJSP_OK;
    (void)hipSetDevice(e->cfg.device_id);
    (void)hipDeviceSynchronize();
    for (void *p : e->owned) (void)hipFree(p);
    if (e->res_fail_host) (void)hipHostFree((void *)e->res_fail_host);
    for (auto &kv : e->ev) for (auto &p : kv.second) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (auto &p : e->ev_free) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    delete e;
    return MTFJSP_OK;
}
extern "C" int mtfjsp_encoder_set_stream(mtfjsp_encoder_t e, void *s) { if (!e) return MTFJSP_ERR_ARG; e->stream = (hipStream_t)s; return MTFJSP_OK; }

// host-side IEEE binary16 conversion (round to nearest even, subnormals kept) for the f16 operand-piece images
static uint16_t f32_to_f16_bits(float x)
{
    uint32_t u; memcpy(&u, &x, 4);
    const uint32_t sign = (u >> 16) & 0x8000u; u &= 0x7fffffffu;
    uint32_t o;
    if (u >= ((127u + 16u) << 23)) o = u > (255u << 23) ? 0x7e00u : 0x7c00u;
    else if (u < (113u << 23)) { float f, magic; const uint32_t mb = ((127u - 15u) + (23u - 10u) + 1u) << 23; memcpy(&f, &u, 4); memcpy(&magic, &mb, 4);
                                 f += magic; uint32_t fu; memcpy(&fu, &f, 4); o = fu - mb; }
    else { const uint32_t odd = (u >> 13) & 1u; u += ((15u - 127u) << 23) + 0xfffu; u += odd; o = u >> 13; }
    return (uint16_t)(o | sign);
}
static float f16_bits_to_f32(uint16_t hbits)
{
    const uint32_t sgn = (uint32_t)(hbits & 0x8000u) << 16, ex = (hbits >> 10) & 31u, m = hbits & 0x3ffu;
    float x;
    if (ex == 0) { x = ldexpf((float)m, -24); return sgn ? -x : x; }
    const uint32_t u = sgn | ((ex == 31 ? 255u : ex + 112u) << 23) | (m << 13);
    memcpy(&x, &u, 4); return x;
}
// the power of two that puts max |w| into [2^13, 2^14): the low f16 piece of every weight then stays a normal number
static float f16_image_scale(const float *data, int64_t numel, bool *finite)
{
    float mx = 0.f;
    for (int64_t i = 0; i < numel; i++) mx = fmaxf(mx, fabsf(data[i]));
    *finite = mx < INFINITY;
    int ex = 0;
    if (mx > 0.f && *finite) (void)frexpf(mx, &ex);               // mx = f * 2^ex, f in [0.5, 1)
    int k = 14 - ex; k = k > 60 ? 60 : k < -60 ? -60 : k;
    return ldexpf(1.0f, k);
}

// element count of every tensor the forwards read, by the reference's state_dict key (after the "job_actor." / "machine_actor."
// / "global_critic." prefix): gcn:60-107 (GraphCNN / MLP), gcn:322-433 (MLPActor / MLPCritic), ac:60-100, 330-357, 540-585
static int64_t expected_numel(const std::string &key)
{
    const size_t dot = key.find('.');
    if (dot == std::string::npos) return -1;
    const std::string pre = key.substr(0, dot), k = key.substr(dot + 1);
    if (pre != "job_actor" && pre != "machine_actor" && pre != "global_critic") return -1;
    auto ends = [&](const char *suf) { const size_t n = strlen(suf); return k.size() >= n && k.compare(k.size() - n, n, suf) == 0; };
    auto has = [&](const char *sub) { return k.find(sub) != std::string::npos; };
    if (k == "_input") return HD;
    if (has("batch_norms") || k == "bn.weight" || k == "bn.bias" || k == "encoder.feature_extract.bn.weight" || k == "encoder.feature_extract.bn.bias")
        return has("feature_extract.bn.") ? 12 : HD;                           // GraphCNN.bn(input_dim) is defined but unused (gcn:66)
    if (has("feature_extract.mlps.")) {
        if (ends(".bias")) return HD;
        return has("mlps.0.linears.0.weight") ? (int64_t)HD * 12 : (int64_t)HD * HD;
    }
    if (has("feature_extract.eps")) return 2;
    if (k == "m_fea_1_fcl.weight") return (int64_t)HD * 6;
    if (k == "m_fea_2_fcl.weight") return (int64_t)HD * 8;
    if (k == "gat_layer.W") return (int64_t)HD * HD;
    if (k == "gat_layer.a") return 2 * HD;
    if (has("fcl_pooling")) return ends(".bias") ? HD : (int64_t)HD * HD;      // defined, unused (ac:357)
    for (const char *head : {"o_policy", "m_policy", "job_critic", "machine_critic", "critic"}) {
        const std::string h = std::string(head) + ".linears.";
        if (k.compare(0, h.size(), h) != 0) continue;
        const int layer = k[h.size()] - '0';
        const bool policy = strstr(head, "policy") != nullptr;
        const int first_in = policy ? 3 * HD : (strcmp(head, "critic") == 0 ? 2 * HD : HD);
        const int out_last = policy ? 1 : (strcmp(head, "critic") == 0 ? 4 : 2);
        const int in = layer == 0 ? first_in : HD, out = layer == 2 ? out_last : HD;
        if (layer < 0 || layer > 2) return -1;
        return ends(".bias") ? out : (int64_t)out * in;
    }
    return -1;
}

extern "C" int mtfjsp_encoder_load_weight_host(mtfjsp_encoder_t e, const char *name, const float *data, int64_t numel)
{
    if (!e || !name || !data || numel <= 0) return MTFJSP_ERR_ARG;
    std::string key(name);
    {   // the kernels assume 128-wide layers of the reference's architecture: reject a tensor of any other size (a checkpoint
        // with another hidden size or layer count would otherwise be read out of bounds)
        const int64_t want = expected_numel(key);
        if (want < 0) { e->err = "unknown weight name: " + key; return MTFJSP_ERR_ARG; }
        if (want != numel) {
            e->err = "weight " + key + ": expected " + std::to_string(want) + " elements, got " + std::to_string(numel);
            return MTFJSP_ERR_ARG;
        }
    }
    HIPCHK(e, hipSetDevice(e->cfg.device_id));
    float *d = nullptr;
    auto it = e->w.find(key);
    if (it != e->w.end()) d = it->second;
    else { if (dalloc(e, &d, (size_t)numel)) return MTFJSP_ERR_HIP; e->w[key] = d; }
    HIPCHK(e, hipMemcpy(d, data, (size_t)numel * 4, hipMemcpyHostToDevice));
    if (key == "job_actor._input") e->hm_b_valid = false;
    if ((key.size() >= 11 && key.compare(key.size() - 11, 11, "gat_layer.W") == 0) || key.find("m_fea_1_fcl.weight") != std::string::npos ||
        key.find("m_fea_2_fcl.weight") != std::string::npos) {
        e->hostw[key].assign(data, data + numel);
        const std::string prefix = key.substr(0, key.find('.') + 1);
        e->wfused.erase(prefix + "1"); e->wfused.erase(prefix + "2"); e->wfused.erase(prefix + "q");   // rebuilt at the next forward (buffers stay owned)
    }
    if (numel == (int64_t)HD * 12 && key.find("mlps.0.linears.0.weight") != std::string::npos) {
        // k_gemm_x6<PRO_GIN0>: the 12 -> 128 first Linear as ONE k-step of 32 (k >= 12 zero), exact 3-way bf16 split:
        // img[cg 4][c 2][plane 3][lane 64][i 8] = plane(W[n = 32cg + 16c + (lane & 15)][k = 8(lane >> 4) + i])
        auto to_bf16 = [](float x) { uint32_t u; memcpy(&u, &x, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); };
        auto from_bf16 = [](uint16_t b) { uint32_t u = (uint32_t)b << 16; float x; memcpy(&x, &u, 4); return x; };
        std::vector<uint16_t> im((size_t)4 * 2 * 3 * 64 * 8, 0);
        for (int cgi = 0; cgi < 4; cgi++)
            for (int c = 0; c < 2; c++)
                for (int lane = 0; lane < 64; lane++)
                    for (int i = 0; i < 8; i++) {
                        const int n = 32 * cgi + 16 * c + (lane & 15), k = 8 * (lane >> 4) + i;
                        if (k >= 12) continue;
                        const float w = data[(size_t)n * 12 + k];
                        const uint16_t p0 = to_bf16(w); const float r1 = w - from_bf16(p0);
                        const uint16_t p1 = to_bf16(r1); const float r2 = r1 - from_bf16(p1);
                        const uint16_t pl[3] = {p0, p1, to_bf16(r2)};
                        for (int p = 0; p < 3; p++) im[(((((size_t)cgi * 2 + c) * 3 + p) * 64 + lane) * 8) + i] = pl[p];
                    }
        void *dx = nullptr;
        auto kt = e->wx6.find(key);
        if (kt != e->wx6.end()) dx = kt->second;
        else { float *tmp = nullptr; if (dalloc(e, &tmp, im.size() / 2)) return MTFJSP_ERR_HIP; dx = tmp; e->wx6[key] = dx; }
        HIPCHK(e, hipMemcpy(dx, im.data(), im.size() * 2, hipMemcpyHostToDevice));
            e->wx6_bytes[key] = im.size() * 2;
    }
    if (key.find("feature_extract.mlps.") != std::string::npos && key.size() > 7 && key.compare(key.size() - 7, 7, ".weight") == 0 &&
        key.find("linears") != std::string::npos) {
        // k_gin_res (v_mfma_f32_32x32x16_*, A := weight): img[w 4][plane P][ks KS][lane 64][j 8] =
        // plane(W[n = 32w + (lane & 31)][k = 16ks + 8(lane >> 5) + j]).  The 12 -> 128 Linear: one k-step (k >= 12 zero), P = 3
        // bf16 planes (exact split).  The 128 -> 128 ones: P = 2 f16 planes (high = f16(s W), low = f16(s W - high), round to
        // nearest) of the weights scaled by the power of two s that puts max |W| in [2^13, 2^14): the low piece keeps its full 11
        // bits well clear of the f16 subnormals.  1/s goes into the BatchNorm that follows (exact).
        const int in = (int)(numel / HD), KS = in == 12 ? 1 : in / 16, P = in == 12 ? 3 : 2;
        auto to_bf16 = [](float x) { uint32_t u; memcpy(&u, &x, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); };
        auto from_bf16 = [](uint16_t b) { uint32_t u = (uint32_t)b << 16; float x; memcpy(&x, &u, 4); return x; };
        float scale = 1.0f;
        if (P == 2) {
            bool finite = true;
            scale = f16_image_scale(data, numel, &finite);
            if (!finite) { e->err = "load_weight: non-finite value in " + key; return MTFJSP_ERR_ARG; }
        }
        std::vector<uint16_t> im((size_t)4 * P * KS * 64 * 8, 0);
        for (int w = 0; w < 4; w++)
            for (int ks = 0; ks < KS; ks++)
                for (int lane = 0; lane < 64; lane++)
                    for (int j = 0; j < 8; j++) {
                        const int n = 32 * w + (lane & 31), k = 16 * ks + 8 * (lane >> 5) + j;
                        if (k >= in) continue;
                        const float x = data[(size_t)n * in + k] * scale;
                        uint16_t pl[3] = {0, 0, 0};
                        if (P == 3) {
                            pl[0] = to_bf16(x); const float r1 = x - from_bf16(pl[0]);
                            pl[1] = to_bf16(r1); pl[2] = to_bf16(r1 - from_bf16(pl[1]));
                        } else {
                            pl[0] = f32_to_f16_bits(x); pl[1] = f32_to_f16_bits(x - f16_bits_to_f32(pl[0]));
                        }
                        for (int p = 0; p < P; p++) im[(((((size_t)w * P + p) * KS + ks) * 64 + lane) * 8) + j] = pl[p];
                    }
        void *dx = nullptr;
        auto kt = e->wx32.find(key);
        if (kt != e->wx32.end()) dx = kt->second;
        else { float *tmp = nullptr; if (dalloc(e, &tmp, im.size() / 2)) return MTFJSP_ERR_HIP; dx = tmp; e->wx32[key] = dx; }
        HIPCHK(e, hipMemcpy(dx, im.data(), im.size() * 2, hipMemcpyHostToDevice));
        e->wx32_bytes[key] = im.size() * 2;
        e->wx32_sinv[key] = 1.0f / scale;
    }
    // 128-wide Linear weights [out=128, in=128*k] and gat W [in,out]: keep GEMM-ready [in-block][k][n] copies
    const bool is_w = key.size() > 7 && key.compare(key.size() - 7, 7, ".weight") == 0 && key.find("linears") != std::string::npos;
    const bool is_gat_w = key.size() >= 11 && key.compare(key.size() - 11, 11, "gat_layer.W") == 0;
    if ((is_w && numel % (HD * HD) == 0) || is_gat_w) {
        const int blocks = (int)(numel / (HD * HD));
        std::vector<float> t((size_t)numel);
        if (is_gat_w) memcpy(t.data(), data, (size_t)numel * 4);     // already [in,out] (gat:82 h @ W)
        else {
            const int in = blocks * HD;
            for (int blk = 0; blk < blocks; blk++)
                for (int k = 0; k < HD; k++)
                    for (int n = 0; n < HD; n++) t[((size_t)blk * HD + k) * HD + n] = data[(size_t)n * in + blk * HD + k];
        }
        float *dt = nullptr;
        auto jt = e->wt.find(key);
        if (jt != e->wt.end()) dt = jt->second;
        else { if (dalloc(e, &dt, (size_t)numel)) return MTFJSP_ERR_HIP; e->wt[key] = dt; }
        HIPCHK(e, hipMemcpy(dt, t.data(), (size_t)numel * 4, hipMemcpyHostToDevice));
        if (!is_gat_w) {
            // k_heads: wave w (output columns 16w..16w+15), lane (m = lane & 15, q = lane >> 4) holds b[s] = W^T[k = 4s+q][16w+m],
            // s = 0..31, fetched as 8 coalesced 16-byte loads: img[blk][w][g][lane][x] = W^T[4(4g+x)+q][16w+m]
            std::vector<float> im((size_t)numel);
            for (int blk = 0; blk < blocks; blk++)
                for (int w = 0; w < 8; w++)
                    for (int g = 0; g < 8; g++)
                        for (int lane = 0; lane < 64; lane++)
                            for (int x = 0; x < 4; x++) {
                                const int m = lane & 15, q = lane >> 4, k = 4 * (4 * g + x) + q;
                                im[((((size_t)blk * 8 + w) * 8 + g) * 64 + lane) * 4 + x] = t[((size_t)blk * HD + k) * HD + 16 * w + m];
                            }
            float *di = nullptr;
            auto kt = e->wimg.find(key);
            if (kt != e->wimg.end()) di = kt->second;
            else { if (dalloc(e, &di, (size_t)numel)) return MTFJSP_ERR_HIP; e->wimg[key] = di; }
            HIPCHK(e, hipMemcpy(di, im.data(), (size_t)numel * 4, hipMemcpyHostToDevice));
        }
        if (is_gat_w) {
            // k_gat3x: B-operand fragments of s W [in k][out n] as two f16 planes: img[c 8][plane 2][ks 4][lane 64][i 8] =
            // plane(s W[32ks + 8(lane >> 4) + i][16c + (lane & 15)]), s = power of two (1/s: wx6_sinv)
            bool finite = true;
            const float scale = f16_image_scale(data, numel, &finite);
            if (!finite) { e->err = "load_weight: non-finite value in " + key; return MTFJSP_ERR_ARG; }
            std::vector<uint16_t> im((size_t)2 * HD * HD);
            for (int c = 0; c < 8; c++)
                for (int ks = 0; ks < 4; ks++)
                    for (int lane = 0; lane < 64; lane++)
                        for (int i = 0; i < 8; i++) {
                            const int k = 32 * ks + 8 * (lane >> 4) + i, n = 16 * c + (lane & 15);
                            const float w = data[(size_t)k * HD + n] * scale;
                            const uint16_t p0 = f32_to_f16_bits(w), p1 = f32_to_f16_bits(w - f16_bits_to_f32(p0));
                            const uint16_t pl[2] = {p0, p1};
                            for (int p = 0; p < 2; p++) im[(((((size_t)c * 2 + p) * 4 + ks) * 64 + lane) * 8) + i] = pl[p];
                        }
            e->wx6_sinv[key] = 1.0f / scale;
            void *dx = nullptr;
            auto kt = e->wx6.find(key);
            if (kt != e->wx6.end()) dx = kt->second;
            else { float *tmp = nullptr; if (dalloc(e, &tmp, im.size() / 2)) return MTFJSP_ERR_HIP; dx = tmp; e->wx6[key] = dx; }
            HIPCHK(e, hipMemcpy(dx, im.data(), im.size() * 2, hipMemcpyHostToDevice));
            e->wx6_bytes[key] = im.size() * 2;
        }
        if (!is_gat_w) {
            // k_gemm_x6 / k_headsx: 2-way f16 split (round to nearest) of the weight scaled by a power of two, per 128-wide input block
            // img[blk][cg 4][c 2][plane 2][ks 4][lane 64][i 8] = plane(s W[n = 32cg + 16c + (lane & 15)][blk*128 + 32ks + 8(lane >> 4) + i]);
            // W is the torch layout [out n][in].  (cg, c) = column block w = 2cg + c of k_headsx's wave w.  1/s: wx6_sinv.
            bool finite = true;
            const float scale = f16_image_scale(data, numel, &finite);
            if (!finite) { e->err = "load_weight: non-finite value in " + key; return MTFJSP_ERR_ARG; }
            std::vector<uint16_t> im((size_t)2 * numel);
            const int in = blocks * HD;
            for (int blk = 0; blk < blocks; blk++)
                for (int cgi = 0; cgi < 4; cgi++)
                    for (int c = 0; c < 2; c++)
                        for (int ks = 0; ks < 4; ks++)
                            for (int lane = 0; lane < 64; lane++)
                                for (int i = 0; i < 8; i++) {
                                    const int n = 32 * cgi + 16 * c + (lane & 15), k = 32 * ks + 8 * (lane >> 4) + i;
                                    const float w = data[(size_t)n * in + blk * HD + k] * scale;
                                    const uint16_t p0 = f32_to_f16_bits(w), p1 = f32_to_f16_bits(w - f16_bits_to_f32(p0));
                                    const uint16_t pl[2] = {p0, p1};
                                    for (int p = 0; p < 2; p++) im[(((((((size_t)blk * 4 + cgi) * 2 + c) * 2 + p) * 4 + ks) * 64 + lane) * 8) + i] = pl[p];
                                }
            e->wx6_sinv[key] = 1.0f / scale;
            void *dx = nullptr;
            auto kt = e->wx6.find(key);
            if (kt != e->wx6.end()) dx = kt->second;
            else { float *tmp = nullptr; if (dalloc(e, &tmp, im.size() / 2)) return MTFJSP_ERR_HIP; dx = tmp; e->wx6[key] = dx; }
            HIPCHK(e, hipMemcpy(dx, im.data(), im.size() * 2, hipMemcpyHostToDevice));
            e->wx6_bytes[key] = im.size() * 2;
        }
    }
    return MTFJSP_OK;
}

extern "C" int mtfjsp_encoder_weights_ready(mtfjsp_encoder_t e)
{
    if (!e) return MTFJSP_ERR_ARG;
    for (const char *k : REQUIRED)
        if (!e->w.count(k)) { e->err = std::string("missing weight: ") + k; return MTFJSP_ERR_STATE; }
    return MTFJSP_OK;
}

// ---- timed launch helper
struct Timed {
    mtfjsp_encoder *e; std::pair<hipEvent_t, hipEvent_t> ev; bool on;
    Timed(mtfjsp_encoder *e_, const char *name) : e(e_), on(e_->timing)
    {
        if (!on) return;
        if (e->ev_free.empty()) { hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b); ev = {a, b}; }
        else { ev = e->ev_free.back(); e->ev_free.pop_back(); }
        (void)hipEventRecord(ev.first, e->stream);
        e->ev[name].push_back(ev);
    }
    ~Timed() { if (on) (void)hipEventRecord(ev.second, e->stream); }
};

template <int PRO, int EPI, bool ACC = false>
static void launch_gemm(mtfjsp_encoder *e, const GemmArgs &a, const char *name)
{
    Timed t(e, name);
    const int ntiles = (a.N + 15) / 16;
    int grid = (ntiles + 7) / 8;
    if (grid > e->num_cu) grid = e->num_cu;
    GemmArgs b = a;
    b.range_flag = (b.Wx6 && !(e->f32_products & 1)) ? e->range_flag : nullptr;
    static const int dbg_env = getenv("MTFJSP_GEMM_DBG") ? atoi(getenv("MTFJSP_GEMM_DBG")) : 0;
    b.dbg = dbg_env;
#ifdef MTFJSP_STAMP
    static unsigned long long *d_st = nullptr;
    if (!d_st) (void)hipMalloc((void **)&d_st, 2048 * 8 * 8);
    (void)hipMemsetAsync(d_st, 0, 2048 * 8 * 8, e->stream);
    b.stamps = d_st;
#endif
    if constexpr (PRO == PRO_PLAIN) hipLaunchKernelGGL((k_gemm16<EPI, ACC>), dim3(grid), dim3(512), gemm16_lds_bytes(), e->stream, b);
    else if constexpr (PRO == PRO_GIN0 || PRO == PRO_GIN0BN) hipLaunchKernelGGL((k_gemm_x6<PRO>), dim3(grid), dim3(512), gemm_x6_lds_bytes(), e->stream, b);
    else {
        if (b.Wx6 && !(e->f32_products & 1)) hipLaunchKernelGGL((k_gemm_x6<PRO>), dim3(grid), dim3(512), gemm_x6_lds_bytes(), e->stream, b);
        else hipLaunchKernelGGL((k_gemm16p<PRO>), dim3(grid), dim3(512), gemm16_lds_bytes(), e->stream, b);
    }
#ifdef MTFJSP_STAMP
    static int printed = 0;
    if (PRO != PRO_PLAIN && printed < 40 && getenv("MTFJSP_STAMP_PRINT")) {
        (void)hipStreamSynchronize(e->stream);
        std::vector<unsigned long long> hst(2048 * 8);
        (void)hipMemcpy(hst.data(), d_st, 2048 * 8 * 8, hipMemcpyDeviceToHost);
        for (int half = 0; half < 2; half++) {
            double m[8] = {0}; int n = 0;
            for (int w = 0; w < grid * 8; w++) {
                if (((w % 8) >= 4) != (half == 1)) continue;
                n++;
                for (int i = 0; i < 8; i++) m[i] += (double)hst[w * 8 + i];
            }
            for (int i = 0; i < 8; i++) m[i] /= n;
            printf("STAMP %-18s N=%d grid=%d half=%d  tiles|x6:barrier2 %.2f  W+bn+sync %.0f  first-tile|x6:T %.0f  tile-loop|x6:barrier1 %.0f  x6:M %.0f  x6:fold %.0f | kernel %.0f shader ticks = %.2f us (s_memrealtime)\n",
                   name, a.N, grid, half, m[7], m[0], m[1], m[4], m[5], m[6], m[2], m[3] / 100.0);
        }
        printed++;
    }
#endif
}

static GemmArgs gemm_args(const float *in, int N, const float *Wt, const float *bias, float *out)
{
    GemmArgs a{};
    a.in = in; a.N = N; a.Wt = Wt; a.bias = bias; a.out = out;
    return a;
}

// GIN encoder (gcn:109-197) with the weights under `pre` ("job_actor." / "global_critic."), followed by the graph mean
// pool and (optionally) the candidate gather.  Uses BatchNorm accumulator slots 0..5.
static int reduce_stats(mtfjsp_encoder *e, double *st);
static int run_gin(mtfjsp_encoder *e, const std::string &pre, const void *tasks_fea, const int32_t *ell_col, const float *ell_val,
                   const int32_t *candidate, int J, float *h_pooled, float *cand_feat, float *h_nodes)
{
    const int B = e->cfg.batch, T = e->T, N = B * T;
    const std::string P = pre + "encoder.feature_extract.";
    auto W = [&](const std::string &k) { return e->w.at(k); };
    auto WT = [&](const std::string &k) { return e->wt.at(k); };
    double *st = e->stats;
    if (!e->gin_stats_clean) HIPCHK(e, hipMemsetAsync(st, 0, 6 * STAT_REP * 256 * sizeof(double), e->stream));
    e->gin_stats_clean = false; e->gin_slot5_dirty = false;
    const double invN = 1.0 / ((double)N * e->reduce_scale);      // (exact multi-shard BatchNorm: rows of all shards)
    const int pgrid = e->num_cu * 8;
    int rrc = 0;
    // Two Linears per launch (mtfjsp_gemm_pair.h): split products only
    const bool pair = !(e->f32_products & 1) && (e->fuse_pair > 0 || (e->fuse_pair < 0 && (size_t)N * HD * 4 > ((size_t)256 << 20)));
    // Round 6 (e->fuse_gin0; MTFJSP_FUSE_GIN0=0 switches it off): z0 = Linear0(aggregated features) is not stored.  Its BatchNorm sums come from the second
    // moments of the 12 aggregated features (k_gin0_moments; MTFJSP_FUSE_GIN0=2: from a PRO_GIN0 launch with out == NULL, the two-launch form's bits), and
    // the producers of the second launch form z0 again from the raw features (PRO_GIN0BN): 838 MB of the forward's traffic at 819 200 rows replaced by
    // two reads of 52 MB.  f32 observations, split products (tests/test_encoder_sizes_gpu.py holds the forms against each other).
    const bool fuse0 = e->fuse_gin0 && !(e->f32_products & 9) && !pair && e->cfg.obs_dtype == MTFJSP_OBS_F32 &&
                       (size_t)N < ((size_t)1 << 24) && T >= 8;    // (row -> instance by an f32 reciprocal + one correction in k_gin0_moments and the PRO_GIN0BN producers: right for quotients below 2^21)
    // (2: the sums by a statistics-only PRO_GIN0 launch — the two-launch form's bits; A/B and tests.  With a statistics reduction over several ranks
    // slot 0 must hold the SAME quantities on every rank, and a rank that has left the split products after a range failure sums z0 itself: no moments then)
    const bool mom0 = fuse0 && e->fuse_gin0 != 2 && !e->reduce_fn;
    if (mom0) {
        Timed t(e, "gin0_moments");
        hipLaunchKernelGGL(k_gin0_moments, dim3(e->num_cu * 2), dim3(256), 0, e->stream, N, T, (const float *)tasks_fea, ell_col, ell_val, st + 0 * STAT_REP * 256);
        if ((rrc = reduce_stats(e, st + 0 * STAT_REP * 256))) return rrc;
    } else if (!(e->f32_products & 8)) {   // layer 0 / linear 0 with aggregation of the raw features, on the producer/consumer product kernel
        GemmArgs a = gemm_args(nullptr, N, nullptr, W(P + "mlps.0.linears.0.bias"), fuse0 ? nullptr : e->zA);
        a.tfea = tasks_fea; a.feat_f64 = e->cfg.obs_dtype == MTFJSP_OBS_F64; a.ell_col = ell_col; a.ell_val = ell_val; a.T = T;
        a.epi_stats = st + 0 * STAT_REP * 256;
        a.rev = e->stream_order;
        a.Wx6 = e->wx6.at(P + "mlps.0.linears.0.weight"); a.w_sinv = 1.0f;
        launch_gemm<PRO_GIN0, EPI_STATS>(e, a, fuse0 ? "gin0_stats_only" : "gin0_agg_linear12");
        if ((rrc = reduce_stats(e, st + 0 * STAT_REP * 256))) return rrc;
    } else {
        Timed t(e, "gin0_agg_linear12");
        if (e->cfg.obs_dtype == MTFJSP_OBS_F32)
            hipLaunchKernelGGL((k_gin0<float>), dim3(pgrid), dim3(256), 0, e->stream, N, T, (const float *)tasks_fea, ell_col, ell_val,
                               W(P + "mlps.0.linears.0.weight"), W(P + "mlps.0.linears.0.bias"), e->zA, st + 0 * STAT_REP * 256);
        else
            hipLaunchKernelGGL((k_gin0<double>), dim3(pgrid), dim3(256), 0, e->stream, N, T, (const double *)tasks_fea, ell_col, ell_val,
                               W(P + "mlps.0.linears.0.weight"), W(P + "mlps.0.linears.0.bias"), e->zA, st + 0 * STAT_REP * 256);
        if ((rrc = reduce_stats(e, st + 0 * STAT_REP * 256))) return rrc;      // (the same number of reductions per forward in every product mode)
    }
    // the launches alternate the direction in which a workgroup walks its rows (GemmArgs::rev): each starts on the part of its
    // input that is still in the memory-side cache.  The aggregation product walks FORWARD (a row's job predecessor is the row
    // before it: read a moment ago and still in L2; backward it was measured 251 against 234 us), which fixes the others
    const int so = e->stream_order;
    auto bn_gemm = [&](const float *in, float *out, int sin, const std::string &bn, const std::string &lin, int sout, int rev) {
        GemmArgs a = gemm_args(in, N, WT(P + lin + ".weight"), W(P + lin + ".bias"), out);
        a.rev = so ? rev : 0; a.nt = so && (e->stream_nt & 1);
        a.pro_stats = st + sin * STAT_REP * 256; a.pro_gamma = W(P + bn + ".weight"); a.pro_beta = W(P + bn + ".bias"); a.pro_inv_rows = invN;
        a.epi_stats = st + sout * STAT_REP * 256;
        a.Wx6 = e->wx6.at(P + lin + ".weight"); a.w_sinv = e->wx6_sinv.at(P + lin + ".weight");
        launch_gemm<PRO_BNRELU, EPI_STATS>(e, a, "gin_gemm_bn_relu");
        if (!rrc) rrc = reduce_stats(e, st + sout * STAT_REP * 256);
    };
    // Two Linears per launch (mtfjsp_gemm_pair.h): pass 1 = the first product with no output (its BatchNorm sums only, slot `smid`),
    // pass 2 = k_gemm_x6f: in -> bn(sin) -> lin1 -> bn(smid) -> lin2 -> out with the sums of slot `sout`.
    auto bn_gemm_pair = [&](const float *in, float *out, int sin, const std::string &bn1, const std::string &lin1, int smid,
                            const std::string &bn2, const std::string &lin2, int sout, int rev) {
        {   // pass 1: statistics of z_b
            GemmArgs a = gemm_args(in, N, WT(P + lin1 + ".weight"), W(P + lin1 + ".bias"), nullptr);
            a.rev = so ? rev : 0; a.nt = so && (e->stream_nt & 1);
            a.pro_stats = st + sin * STAT_REP * 256; a.pro_gamma = W(P + bn1 + ".weight"); a.pro_beta = W(P + bn1 + ".bias"); a.pro_inv_rows = invN;
            a.epi_stats = st + smid * STAT_REP * 256;
            a.Wx6 = e->wx6.at(P + lin1 + ".weight"); a.w_sinv = e->wx6_sinv.at(P + lin1 + ".weight");
            launch_gemm<PRO_BNRELU, EPI_STATS>(e, a, "gin_gemm_stats_only");
            if (!rrc) rrc = reduce_stats(e, st + smid * STAT_REP * 256);
        }
        {   // pass 2
            Timed t(e, "gin_gemm_pair");
            Gemm2Args g{};
            g.a = gemm_args(in, N, WT(P + lin1 + ".weight"), W(P + lin1 + ".bias"), out);
            g.a.rev = so ? rev ^ 1 : 0; g.a.nt = so && (e->stream_nt & 1);   // (opposite direction: it starts on what pass 1 read last)
            g.a.pro_stats = st + sin * STAT_REP * 256; g.a.pro_gamma = W(P + bn1 + ".weight"); g.a.pro_beta = W(P + bn1 + ".bias"); g.a.pro_inv_rows = invN;
            g.a.epi_stats = st + sout * STAT_REP * 256;
            g.a.Wx6 = e->wx6.at(P + lin1 + ".weight"); g.a.w_sinv = e->wx6_sinv.at(P + lin1 + ".weight");
            g.a.range_flag = e->range_flag;
            g.Wx6b = e->wx6.at(P + lin2 + ".weight"); g.w_sinvb = e->wx6_sinv.at(P + lin2 + ".weight"); g.biasb = W(P + lin2 + ".bias");
            g.mid_stats = st + smid * STAT_REP * 256; g.mid_gamma = W(P + bn2 + ".weight"); g.mid_beta = W(P + bn2 + ".bias"); g.mid_inv_rows = invN;
            const int ntiles = (N + 15) / 16;
            int grid = (ntiles + 7) / 8;
            if (grid > e->num_cu) grid = e->num_cu;
            hipLaunchKernelGGL(k_gemm_x6f, dim3(grid), dim3(512), gemm_x6f_lds_bytes(), e->stream, g);
        }
        if (!rrc) rrc = reduce_stats(e, st + sout * STAT_REP * 256);
    };
    if (pair) bn_gemm_pair(e->zA, e->zB, 0, "mlps.0.batch_norms.0", "mlps.0.linears.1", 1, "mlps.0.batch_norms.1", "mlps.0.linears.2", 2, 0);
    else {
    if (fuse0) {
        const std::string lin = "mlps.0.linears.1", bn = "mlps.0.batch_norms.0";
        GemmArgs a = gemm_args(nullptr, N, WT(P + lin + ".weight"), W(P + lin + ".bias"), e->zB);
        a.rev = 0; a.nt = 0;
        a.tfea = tasks_fea; a.feat_f64 = 0; a.ell_col = ell_col; a.ell_val = ell_val; a.T = T;
        a.Wx6_0 = e->wx6.at(P + "mlps.0.linears.0.weight"); a.bias0 = W(P + "mlps.0.linears.0.bias");
        a.W0 = mom0 ? W(P + "mlps.0.linears.0.weight") : nullptr;
        a.pro_stats = st + 0 * STAT_REP * 256; a.pro_gamma = W(P + bn + ".weight"); a.pro_beta = W(P + bn + ".bias"); a.pro_inv_rows = invN;
        a.epi_stats = st + 1 * STAT_REP * 256;
        a.Wx6 = e->wx6.at(P + lin + ".weight"); a.w_sinv = e->wx6_sinv.at(P + lin + ".weight");
        launch_gemm<PRO_GIN0BN, EPI_STATS>(e, a, "gin0_bn_gemm");
        if (!rrc) rrc = reduce_stats(e, st + 1 * STAT_REP * 256);
    } else
    bn_gemm(e->zA, e->zB, 0, "mlps.0.batch_norms.0", "mlps.0.linears.1", 1, 0);
    bn_gemm(e->zB, e->zA, 1, "mlps.0.batch_norms.1", "mlps.0.linears.2", 2, 1);
    }
    const float *z2 = pair ? e->zB : e->zA;                        // where the first MLP's output is
    float *z3 = pair ? e->zA : e->zB;
    {   // layer 1 / linear 0: aggregation of h = relu(bn_outer0(z)) over the ELL adjacency
        GemmArgs a = gemm_args(z2, N, WT(P + "mlps.1.linears.0.weight"), W(P + "mlps.1.linears.0.bias"), z3);
        a.pro_stats = st + 2 * STAT_REP * 256; a.pro_gamma = W(P + "batch_norms.0.weight"); a.pro_beta = W(P + "batch_norms.0.bias"); a.pro_inv_rows = invN;
        a.ell_col = ell_col; a.ell_val = ell_val; a.T = T;
        a.rev = 0; a.nt = so && (e->stream_nt & 2);
        a.epi_stats = st + 3 * STAT_REP * 256;
        a.Wx6 = e->wx6.at(P + "mlps.1.linears.0.weight"); a.w_sinv = e->wx6_sinv.at(P + "mlps.1.linears.0.weight");
        launch_gemm<PRO_AGG, EPI_STATS>(e, a, "gin_gemm_agg");
        if (!rrc) rrc = reduce_stats(e, st + 3 * STAT_REP * 256);
    }
    if (pair) bn_gemm_pair(z3, e->zB, 3, "mlps.1.batch_norms.0", "mlps.1.linears.1", 4, "mlps.1.batch_norms.1", "mlps.1.linears.2", 5, 1);
    else {
    bn_gemm(e->zB, e->zA, 3, "mlps.1.batch_norms.0", "mlps.1.linears.1", 4, 1);
    // Round 6 (e->fuse_pool; MTFJSP_FUSE_POOL=0 switches it off): the LAST Linear's output is not stored either.  A statistics-only pass of the product
    // leaves its BatchNorm sums, a second pass forms it again and keeps only what the heads read — the per-instance means and the candidates' rows — in
    // its epilogue: two passes without output instead of a product with output + k_job_pool_gather's read of it
    const bool fuse_pool = e->fuse_pool && h_pooled && !h_nodes && T >= 16 && T < 65536 && (!candidate || (J > 0 && T % J == 0)) && (size_t)N < ((size_t)1 << 24) &&
                           e->wx6.count(P + "mlps.1.linears.2.weight") && !(e->f32_products & 1);
    if (fuse_pool) {
        const std::string lin = "mlps.1.linears.2", bn = "mlps.1.batch_norms.1";
        GemmArgs a = gemm_args(e->zA, N, WT(P + lin + ".weight"), W(P + lin + ".bias"), nullptr);
        a.rev = 0; a.nt = 0;                                      // (read twice: plain loads; the second pass walks back from the end)
        a.pro_stats = st + 4 * STAT_REP * 256; a.pro_gamma = W(P + bn + ".weight"); a.pro_beta = W(P + bn + ".bias"); a.pro_inv_rows = invN;
        a.epi_stats = st + 5 * STAT_REP * 256;
        a.Wx6 = e->wx6.at(P + lin + ".weight"); a.w_sinv = e->wx6_sinv.at(P + lin + ".weight");
        a.zero_f32 = h_pooled; a.zero_count = B * HD;              // (the second pass adds into it; no launch of its own for the clearing)
        launch_gemm<PRO_BNRELU, EPI_STATS>(e, a, "gin_gemm_stats_only");
        a.zero_f32 = nullptr; a.zero_count = 0;
        if (!rrc) rrc = reduce_stats(e, st + 5 * STAT_REP * 256);
        if (rrc) return rrc;
        a.rev = so ? 1 : 0; a.nt = so && (e->stream_nt & 1);
        a.epi_stats = nullptr;
        a.T = T; a.pooled = h_pooled; a.cand_feat = candidate ? cand_feat : nullptr; a.cand = candidate; a.pool_J = candidate ? J : 0;
        a.pool_stats = st + 5 * STAT_REP * 256; a.pool_gamma = W(P + "batch_norms.1.weight"); a.pool_beta = W(P + "batch_norms.1.bias"); a.pool_inv_rows = invN;
        launch_gemm<PRO_BNRELU, EPI_STATS>(e, a, "gin_gemm_pool");
        if (candidate) {
            Timed t(e, "cand_fixup");
            hipLaunchKernelGGL(k_cand_fixup, dim3((B * J + 127) / 128), dim3(128), 0, e->stream, B, T, J, candidate, e->zA, st + 4 * STAT_REP * 256, W(P + bn + ".weight"), W(P + bn + ".bias"),
                               WT(P + lin + ".weight"), W(P + lin + ".bias"), st + 5 * STAT_REP * 256, W(P + "batch_norms.1.weight"), W(P + "batch_norms.1.bias"), invN, cand_feat);
        }
        HIPCHK(e, hipGetLastError());
        return MTFJSP_OK;
    }
    bn_gemm(e->zA, e->zB, 4, "mlps.1.batch_norms.1", "mlps.1.linears.2", 5, 0);
    }
    if (rrc) return rrc;
    if (h_pooled) {                                               // h_pooled == NULL: the consumer (k_heads) normalises, pools and gathers itself
        Timed t(e, "job_pool_gather");
        // (the last product's partition: launch_gemm's grid and k_gemm_x6's tiles per workgroup)
        const int ntiles = (N + 15) / 16, ggrid = std::min((ntiles + 7) / 8, e->num_cu), rpr = so && e->pool_s > 0 ? ((ntiles + ggrid - 1) / ggrid) * 16 : 0;
        const int pool_grid = rpr ? ggrid * e->pool_s : (B < e->num_cu * 8 ? B : e->num_cu * 8);
        if (so && (e->stream_nt & 4))
            hipLaunchKernelGGL(k_job_pool_gather<1>, dim3(pool_grid), dim3(256), 0, e->stream, split_products_in_use(e) ? e->range_flag : nullptr, rpr, e->pool_s, B, T, candidate ? J : 0, e->zB, st + 5 * STAT_REP * 256, invN,
                               W(P + "batch_norms.1.weight"), W(P + "batch_norms.1.bias"), candidate, h_pooled, cand_feat, h_nodes);
        else
            hipLaunchKernelGGL(k_job_pool_gather<0>, dim3(pool_grid), dim3(256), 0, e->stream, split_products_in_use(e) ? e->range_flag : nullptr, rpr, e->pool_s, B, T, candidate ? J : 0, e->zB, st + 5 * STAT_REP * 256, invN,
                               W(P + "batch_norms.1.weight"), W(P + "batch_norms.1.bias"), candidate, h_pooled, cand_feat, h_nodes);
    }
    HIPCHK(e, hipGetLastError());
    return MTFJSP_OK;
}

// The same encoder as ONE launch with register-resident activations (mtfjsp_gin_resident.h); h_pooled and cand_feat (when
// `candidate` is given) are written normalised, as run_gin's k_job_pool_gather would.
static int run_gin_resident(mtfjsp_encoder *e, const std::string &pre, const void *tasks_fea, const int32_t *ell_col, const float *ell_val,
                            const int32_t *candidate, int J, float *h_pooled, float *cand_feat, float *h_nodes)
{
    const int B = e->cfg.batch, T = e->T;
    const std::string P = pre + "encoder.feature_extract.";
    auto W = [&](const std::string &k) { return e->w.at(k); };
    GinResArgs a{};
    a.B = B; a.T = T; a.J = candidate ? J : 0; a.ipc = e->res_ipc;
    a.tfea = tasks_fea; a.feat_f64 = e->cfg.obs_dtype == MTFJSP_OBS_F64; a.ell_col = ell_col; a.ell_val = ell_val;
    const char *lin[6] = {"mlps.0.linears.0", "mlps.0.linears.1", "mlps.0.linears.2", "mlps.1.linears.0", "mlps.1.linears.1", "mlps.1.linears.2"};
    const char *bn[6] = {"mlps.0.batch_norms.0", "mlps.0.batch_norms.1", "batch_norms.0", "mlps.1.batch_norms.0", "mlps.1.batch_norms.1", "batch_norms.1"};
    for (int i = 0; i < 6; i++) {
        a.Wx32[i] = e->wx32.at(P + lin[i] + ".weight");           // (the Linear biases cancel in the BatchNorms that follow them)
        a.wsinv[i] = e->wx32_sinv.at(P + lin[i] + ".weight");
        a.gamma[i] = W(P + bn[i] + ".weight"); a.beta[i] = W(P + bn[i] + ".bias");
    }
    for (int i = 0; i < 6; i++) {                                 // the image scales are powers of two: their exponents (the fixed-point statistics divide them out)
        int ex = 0;
        const float mant = frexpf(a.wsinv[i], &ex);                // wsinv = 0.5 * 2^ex
        if (mant != 0.5f) { e->err = "resident GIN: a weight image scale is not a power of two"; return MTFJSP_ERR_STATE; }
        a.wexp[i] = 1 - ex;                                        // log2(wscale) = -log2(wsinv)
        a.ffrac[i] = i == 0 ? GR_FIX_FRAC_FIRST : GR_FIX_FRAC_DEFAULT;
    }
    const int set = (int)(e->res_epoch & 1);
    a.stats = reinterpret_cast<unsigned long long *>(e->res_stats) + (size_t)set * GR_STATS_SET;
    a.stats_next = reinterpret_cast<unsigned long long *>(e->res_stats) + (size_t)(set ^ 1) * GR_STATS_SET;
    a.bar = e->res_bar; a.epoch = e->res_epoch++; a.fail = e->res_fail; a.range_flag = e->range_flag;
    a.candidate = candidate; a.pooled = h_pooled; a.cand_feat = cand_feat; a.h_nodes = h_nodes; a.zspill = e->res_zspill;
    a.inv_rows = 1.0 / ((double)B * (double)T);
    if (pre == "job_actor." && e->warm_heads) {
        // what the heads launch behind this one reads first — the weight images of both actors' heads and of the GAT passes, last read a
        // rollout step ago and evicted from the XCDs' L2 since — is requested (one word per 128-byte line, dropped) in this kernel's last phase
        const char *keys[9] = {"job_actor.o_policy.linears.0.weight", "job_actor.job_critic.linears.0.weight", "job_actor.o_policy.linears.1.weight",
                               "job_actor.job_critic.linears.1.weight", "machine_actor.gat_layer.W", "machine_actor.m_policy.linears.0.weight",
                               "machine_actor.machine_critic.linears.0.weight", "machine_actor.m_policy.linears.1.weight", "machine_actor.machine_critic.linears.1.weight"};
        for (const char *k : keys) {
            auto it = e->wx6.find(k);
            auto bt = e->wx6_bytes.find(k);
            if (it == e->wx6.end() || bt == e->wx6_bytes.end() || a.nwarm >= GR_MAXWARM) continue;
            a.warm[a.nwarm] = it->second; a.warm_lines[a.nwarm] = (unsigned)(bt->second / 128); a.nwarm++;
        }
        auto q = e->wfused.find("machine_actor.q");               // the GAT projection's operand image (8 KB), once it has been formed
        if (q != e->wfused.end() && a.nwarm < GR_MAXWARM) { a.warm[a.nwarm] = q->second; a.warm_lines[a.nwarm] = 64; a.nwarm++; }
    }
    if (++e->res_launches == e->res_fail_at) a.expect_extra = 1u;      // (diagnostic) this launch's barriers never complete
#ifdef GR_STAMP
    static unsigned long long *d_st = nullptr;
    if (!d_st) (void)hipMalloc((void **)&d_st, 256 * 64 * 8);
    a.stamps = d_st;
#endif
    {
        Timed t(e, "gin_resident");
        hipLaunchKernelGGL(k_gin_res, dim3(e->res_grid), dim3(256), gin_res_lds_bytes(), e->stream, a);
    }
#ifdef GR_STAMP
    static int printed = 0;
    if (printed++ % 50 == 20 && getenv("MTFJSP_STAMP_PRINT")) {
        (void)hipStreamSynchronize(e->stream);
        std::vector<unsigned long long> h((size_t)e->res_grid * 64);
        (void)hipMemcpy(h.data(), d_st, h.size() * 8, hipMemcpyDeviceToHost);
        // per stamp index: mean over blocks of (stamp - min over blocks of stamp 0), in us (s_memrealtime = 100 MHz)
        unsigned long long t0 = ~0ull;
        for (int b = 0; b < e->res_grid; b++) t0 = h[(size_t)b * 64 + 32] < t0 ? h[(size_t)b * 64 + 32] : t0;
        printf("GR_STAMP (us since first block start; mean / max over %d blocks):", e->res_grid);
        for (int i = 0; i < 33; i++) {
            double m = 0, mx = 0;
            for (int b = 0; b < e->res_grid; b++) { const double v = (double)(h[(size_t)b * 64 + i] - t0) / 100.0; m += v; mx = v > mx ? v : mx; }
            printf(" [%d] %.1f/%.1f", i, m / e->res_grid, mx);
        }
        printf("\n");
        {   // the clock inside the kernel: shader cycles (s_memtime) per 10 ns tick (s_memrealtime) over a span, median over the workgroups
            auto clk = [&](int a, int b) {
                std::vector<double> v;
                for (int blk = 0; blk < e->res_grid; blk++) {
                    const double dr = (double)(h[(size_t)blk * 64 + b] - h[(size_t)blk * 64 + a]), dc = (double)(h[(size_t)blk * 64 + 33 + b] - h[(size_t)blk * 64 + 33 + a]);
                    if (dr > 0) v.push_back(dc / dr * 0.1);                    // GHz
                }
                std::sort(v.begin(), v.end());
                return v.empty() ? 0.0 : v[v.size() / 2];
            };
            printf("GR_CLOCK launch %d (GHz, median over %d workgroups): five layers [7 -> 24] %.3f | layer 1 [7 -> 8] %.3f  layer 2 [11 -> 12] %.3f  layer 3 [15 -> 16] %.3f  "
                   "layer 4 [19 -> 20] %.3f  layer 5 [23 -> 24] %.3f | whole kernel [0 -> 30] %.3f\n", printed, e->res_grid, clk(7, 24), clk(7, 8), clk(11, 12), clk(15, 16), clk(19, 20), clk(23, 24), clk(0, 30));
        }
    }
#endif
    HIPCHK(e, hipGetLastError());
    return MTFJSP_OK;
}

// Machine path shared by the machine actor and the global critic (ac:383-444): input projections + 3x the same GATLayer
// + node mean (in-place GEMM passes), then BatchNorm over all B*M rows and the mean over M.  Uses accumulator slot 6.
// (m_fea_k_fcl.weight^T . gat_layer.W)^T on the host, cached per prefix until one of the three weights is loaded again
static std::vector<float> fused_projection_host(mtfjsp_encoder *e, const std::string &pre, int which)
{
    const int K = which == 1 ? 6 : 8;
    const std::vector<float> &P = e->hostw.at(pre + (which == 1 ? "m_fea_1_fcl.weight" : "m_fea_2_fcl.weight"));   // [128,K]
    const std::vector<float> &W = e->hostw.at(pre + "gat_layer.W");                                                  // [128(in),128(out)]
    std::vector<float> f((size_t)HD * K);
    for (int n = 0; n < HD; n++)
        for (int k = 0; k < K; k++) {
            double a = 0.0;
            for (int i = 0; i < HD; i++) a += (double)P[(size_t)i * K + k] * (double)W[(size_t)i * HD + n];
            f[(size_t)n * K + k] = (float)a;
        }
    return f;
}
static int fused_projection(mtfjsp_encoder *e, const std::string &pre, int which, const float **out)
{
    const std::string key = pre + (which == 1 ? "1" : "2");
    auto it = e->wfused.find(key);
    if (it != e->wfused.end()) { *out = it->second; return MTFJSP_OK; }
    const std::vector<float> f = fused_projection_host(e, pre, which);
    float *d = nullptr;
    if (dalloc(e, &d, f.size())) return MTFJSP_ERR_HIP;
    HIPCHK(e, hipMemcpy(d, f.data(), f.size() * 4, hipMemcpyHostToDevice));
    e->wfused[key] = d;
    *out = d;
    return MTFJSP_OK;
}
// both fused projections as the B operands of v_mfma_f32_16x16x4_f32 (GatArgs::Wq), cached like them
static int fused_projection_image(mtfjsp_encoder *e, const std::string &pre, const float **out)
{
    const std::string key = pre + "q";
    auto it = e->wfused.find(key);
    if (it != e->wfused.end()) { *out = it->second; return MTFJSP_OK; }
    const std::vector<float> f1 = fused_projection_host(e, pre, 1), f2 = fused_projection_host(e, pre, 2);
    std::vector<float> img((size_t)8 * 64 * 4);
    for (int c = 0; c < 8; c++)
        for (int lane = 0; lane < 64; lane++)
            for (int ks = 0; ks < 4; ks++) {
                const int kk = 4 * ks + (lane >> 4), col = 16 * c + (lane & 15);
                img[((size_t)c * 64 + lane) * 4 + ks] = kk < 6 ? f1[(size_t)col * 6 + kk] : kk < 8 ? 0.f : f2[(size_t)col * 8 + (kk - 8)];
            }
    float *d = nullptr;
    if (dalloc(e, &d, img.size())) return MTFJSP_ERR_HIP;
    HIPCHK(e, hipMemcpy(d, img.data(), img.size() * 4, hipMemcpyHostToDevice));
    e->wfused[key] = d;
    *out = d;
    return MTFJSP_OK;
}
// h_pooled == nullptr: leave `node` pre-BatchNorm for a consumer that normalises and pools it itself (k_heads); *slot_out = the
// accumulator slot holding the column sums.
// the arguments of k_gat3x for the machine path under `pre`, accumulating the node statistics into slot `slot` (which is made clean)
static int gat3x_args(mtfjsp_encoder *e, const std::string &pre, const void *m_fea1, const void *m_fea2, int slot, GatArgs *out)
{
    const int R = e->cfg.batch * e->cfg.n_machine;
    double *st = nullptr;                                                  // slot < 0: the launch exchanges the sums itself (k_headsx_gat3x_headsx)
    if (slot >= 0) {
        st = e->stats + (6 + slot) * STAT_REP * 256;
        if (!e->gat_stats_clean[slot]) HIPCHK(e, hipMemsetAsync(st, 0, STAT_REP * 256 * sizeof(double), e->stream));
        e->gat_stats_clean[slot] = false;
    }
    GatArgs a{};
    a.R = R; a.f1 = m_fea1; a.f2 = m_fea2; a.feat_f64 = e->cfg.obs_dtype == MTFJSP_OBS_F64;
    int frc = fused_projection(e, pre, 1, &a.W1);
    if (!frc) frc = fused_projection(e, pre, 2, &a.W2);
    if (!frc) frc = fused_projection_image(e, pre, &a.Wq);
    if (frc) return frc;
    a.Wt = e->wt.at(pre + "gat_layer.W"); a.gat_a = e->w.at(pre + "gat_layer.a"); a.node = e->node; a.epi_stats = st;
    a.Wx6 = e->wx6.at(pre + "gat_layer.W"); a.w_sinv = e->wx6_sinv.at(pre + "gat_layer.W");
    *out = a;
    return MTFJSP_OK;
}
// can the machine actor's GAT passes ride in the job actor's heads launch (k_headsx_gat3x)?  Same instances per workgroup in both
// parts (16), m_fea1 produced by that launch itself, whole-batch statistics without a cross-shard reduction, split products.
static bool gat_fusable(const mtfjsp_encoder *e)
{
    // Worth it when the heads' grid (B / 16 workgroups) is exactly as wide as the GAT launch would make its own — min(row tiles / 8,
    // CUs) — and fits the chip in one round: J6M6 x 4096: 47.7 us against 24.6 + 26.8.  Fewer workgroups than the GAT's own grid leave
    // CUs idle during the GAT part (J20M20 x 2048, 128 workgroups: 110 against 53 + 38), two rounds of workgroups gain nothing
    // (J10M10 x 8192: 130 against 128).
    const int hgrid = e->cfg.batch / HG, gtiles = (2 * e->cfg.batch * e->cfg.n_machine + 15) / 16, ggrid = (gtiles + 7) / 8 < e->num_cu ? (gtiles + 7) / 8 : e->num_cu;
    return e->fuse_gat && !e->bn_mode && !e->reduce_fn && !(e->f32_products & (2 | 4)) && e->cfg.batch % HG == 0 && hgrid <= e->num_cu && hgrid >= ggrid &&
           e->w.count("machine_actor.gat_layer.W") && e->wx6.count("machine_actor.gat_layer.W");
}
// ... and the machine actor's heads as well (k_headsx_gat3x_headsx)?  The in-launch exchange needs every workgroup resident at once:
// one per CU, on a device where the census of the single-launch GIN kernel found all of them available; a count field of the exchange
// words holds the arrivals of one dispatch group in 6 bits.
static bool mheads_fusable(const mtfjsp_encoder *e)
{
    const int hgrid = e->cfg.batch / HG;
    return e->fuse_mheads && e->res_ok && gat_fusable(e) && hgrid <= e->num_cu && (hgrid + 7) / 8 <= 63 && e->xw && e->cfg.n_machine <= 8 &&   // (<= 8: the pool rows are requested ahead of the exchange, mtfjsp_headsx_body.h)
           e->wx6.count("machine_actor.m_policy.linears.0.weight") && e->wx6.count("machine_actor.machine_critic.linears.0.weight");
}
static int run_gat(mtfjsp_encoder *e, const std::string &pre, const void *m_fea1, const void *m_fea2, float *h_pooled, int *slot_out = nullptr)
{
    const int B = e->cfg.batch, M = e->cfg.n_machine, R = B * M;
    auto W = [&](const std::string &k) { return e->w.at(k); };
    auto WT = [&](const std::string &k) { return e->wt.at(k); };
    const int slot = e->gat_slot;
    e->gat_slot ^= 1;
    if (slot_out) *slot_out = slot;
    double *st = e->stats + (6 + slot) * STAT_REP * 256;
    if (!e->gat_stats_clean[slot]) HIPCHK(e, hipMemsetAsync(st, 0, STAT_REP * 256 * sizeof(double), e->stream));
    e->gat_stats_clean[slot] = false;
    {                                                                          // the SAME GATLayer three times (ac:409-414), one launch
        Timed t(e, "gat3");
        GatArgs a{};
        a.R = R; a.f1 = m_fea1; a.f2 = m_fea2; a.feat_f64 = e->cfg.obs_dtype == MTFJSP_OBS_F64;
        int frc = fused_projection(e, pre, 1, &a.W1);
        if (!frc) frc = fused_projection(e, pre, 2, &a.W2);
        if (!frc) frc = fused_projection_image(e, pre, &a.Wq);
        if (frc) return frc;
        a.Wt = WT(pre + "gat_layer.W"); a.gat_a = W(pre + "gat_layer.a"); a.node = e->node; a.epi_stats = st;
        const int ntiles = (2 * R + 15) / 16;
        int grid = (ntiles + 7) / 8;
        if (grid > e->num_cu) grid = e->num_cu;
#ifdef MTFJSP_STAMP
        static unsigned long long *d_st = nullptr;
        if (!d_st) (void)hipMalloc((void **)&d_st, 2048 * 8 * 8);
        (void)hipMemsetAsync(d_st, 0, 2048 * 8 * 8, e->stream);
        a.stamps = d_st;
#endif
        if (e->f32_products & 2) hipLaunchKernelGGL(k_gat3, dim3(grid), dim3(512), gemm16_lds_bytes(), e->stream, a);
        else {
            a.Wx6 = e->wx6.at(pre + "gat_layer.W"); a.w_sinv = e->wx6_sinv.at(pre + "gat_layer.W");
            hipLaunchKernelGGL(k_gat3x, dim3(grid), dim3(512), gat3x_lds_bytes(), e->stream, a);
        }
#ifdef MTFJSP_STAMP
        static int printed = 0;
        if (printed++ % 40 == 30 && printed < 200 && getenv("MTFJSP_STAMP_PRINT")) {
            (void)hipStreamSynchronize(e->stream);
            std::vector<unsigned long long> hst(2048 * 8);
            (void)hipMemcpy(hst.data(), d_st, 2048 * 8 * 8, hipMemcpyDeviceToHost);
            if (!(e->f32_products & 2)) {                                   // k_gat3x: s_memrealtime stamps (100 MHz) per wave
                unsigned long long t0 = ~0ull;
                for (int w = 0; w < grid * 8; w++) t0 = hst[w * 8] < t0 ? hst[w * 8] : t0;
                double a1[5] = {0}, a2[5] = {0}; int n1 = 0, n2 = 0;
                for (int w = 0; w < grid * 8; w++) {
                    const bool two = hst[w * 8 + 3] != 0;
                    for (int i = 0; i < 5; i++) { const double v = hst[w * 8 + i] ? (double)(hst[w * 8 + i] - t0) / 100.0 : 0.0; (two ? a2 : a1)[i] += v; }
                    (two ? n2 : n1)++;
                }
                printf("STAMP k_gat3x R=%d grid=%d (us since the first wave's start)  waves with one tile (%d): entry %.2f staged %.2f tile done %.2f end %.2f | with two tiles (%d): entry %.2f staged %.2f first tile %.2f second tile %.2f end %.2f\n",
                       R, grid, n1, a1[0] / (n1 ? n1 : 1), a1[1] / (n1 ? n1 : 1), a1[2] / (n1 ? n1 : 1), a1[4] / (n1 ? n1 : 1), n2, a2[0] / (n2 ? n2 : 1), a2[1] / (n2 ? n2 : 1), a2[2] / (n2 ? n2 : 1), a2[3] / (n2 ? n2 : 1), a2[4] / (n2 ? n2 : 1));
            } else {
            double m[8] = {0};
            for (int w = 0; w < grid * 8; w++) for (int i = 0; i < 8; i++) m[i] += (double)hst[w * 8 + i] / (grid * 8);
            printf("STAMP k_gat3 R=%d grid=%d  Wload %.0f  feat0 %.0f  rows %.0f  mfma %.0f  gat-epilogue %.0f  tail %.0f (cycles/wave, summed)\n",
                   R, grid, m[0], m[1], m[2], m[3], m[4], m[6]);
            }
        }
#endif
    }
    {
        const int rrc = reduce_stats(e, st);                                     // exact multi-shard BatchNorm of the machine nodes
        if (rrc) return rrc;
    }
    if (h_pooled) {
        Timed t(e, "mach_bn_pool");
        hipLaunchKernelGGL(k_mach_bn_pool, dim3(B), dim3(128), 0, e->stream, split_products_in_use(e) ? e->range_flag : nullptr, B, M, e->node, st, 1.0 / ((double)R * e->reduce_scale), W(pre + "bn.weight"),
                           W(pre + "bn.bias"), h_pooled);
    }
    HIPCHK(e, hipGetLastError());
    return MTFJSP_OK;
}

// per-instance BatchNorm variants (bn_mode 1): one workgroup per instance, one launch per path
static int run_gin_inst(mtfjsp_encoder *e, const std::string &pre, const void *tasks_fea, const int32_t *ell_col, const float *ell_val,
                        const int32_t *candidate, int J, float *h_pooled, float *cand_feat, float *h_nodes)
{
    const std::string P = pre + "encoder.feature_extract.";
    auto W = [&](const std::string &k) { return e->w.at(k); };
    auto WT = [&](const std::string &k) { return e->wt.at(k); };
    GinInstArgs a{};
    a.B = e->cfg.batch; a.T = e->T; a.J = candidate ? J : 0;
    a.tfea = tasks_fea; a.feat_f64 = e->cfg.obs_dtype == MTFJSP_OBS_F64; a.ell_col = ell_col; a.ell_val = ell_val;
    a.W0 = W(P + "mlps.0.linears.0.weight"); a.b0 = W(P + "mlps.0.linears.0.bias");
    const char *lin[5] = {"mlps.0.linears.1", "mlps.0.linears.2", "mlps.1.linears.0", "mlps.1.linears.1", "mlps.1.linears.2"};
    const char *bn[6] = {"mlps.0.batch_norms.0", "mlps.0.batch_norms.1", "batch_norms.0", "mlps.1.batch_norms.0", "mlps.1.batch_norms.1", "batch_norms.1"};
    for (int i = 0; i < 5; i++) { a.Wt[i] = WT(P + lin[i] + ".weight"); a.bias[i] = W(P + lin[i] + ".bias"); }
    for (int i = 0; i < 6; i++) { a.gamma[i] = W(P + bn[i] + ".weight"); a.beta[i] = W(P + bn[i] + ".bias"); }
    a.zA = e->zA; a.zB = e->zB; a.cand = candidate; a.h_pooled = h_pooled; a.cand_feat = cand_feat; a.h_nodes = h_nodes;
    Timed t(e, "gin_inst");
    if (a.feat_f64) hipLaunchKernelGGL((k_gin_inst<double>), dim3(a.B), dim3(512), inst_lds_bytes(), e->stream, a);
    else hipLaunchKernelGGL((k_gin_inst<float>), dim3(a.B), dim3(512), inst_lds_bytes(), e->stream, a);
    HIPCHK(e, hipGetLastError());
    return MTFJSP_OK;
}
static int run_gat_inst(mtfjsp_encoder *e, const std::string &pre, const void *m_fea1, const void *m_fea2, float *h_pooled)
{
    auto W = [&](const std::string &k) { return e->w.at(k); };
    GatInstArgs a{};
    a.B = e->cfg.batch; a.M = e->cfg.n_machine; a.f1 = m_fea1; a.f2 = m_fea2; a.feat_f64 = e->cfg.obs_dtype == MTFJSP_OBS_F64;
    a.W1 = W(pre + "m_fea_1_fcl.weight"); a.W2 = W(pre + "m_fea_2_fcl.weight"); a.Wt = e->wt.at(pre + "gat_layer.W");
    a.gat_a = W(pre + "gat_layer.a"); a.gamma = W(pre + "bn.weight"); a.beta = W(pre + "bn.bias");
    a.node = e->node; a.h_pooled = h_pooled;
    Timed t(e, "gat_inst");
    hipLaunchKernelGGL(k_gat_inst, dim3(a.B), dim3(512), inst_lds_bytes(), e->stream, a);
    HIPCHK(e, hipGetLastError());
    return MTFJSP_OK;
}
// = evaluating with env_batch 1 per instance (validate.py:60-297): mode 1 makes every BatchNorm of the two
// actor forwards normalise over the rows of ONE instance; mode 0 (default) over the whole device batch (training rollout).
extern "C" int mtfjsp_encoder_set_product_mode(mtfjsp_encoder_t e, int32_t f32_instruction_mask)
{
    if (!e || f32_instruction_mask < 0 || f32_instruction_mask > 31) return MTFJSP_ERR_ARG;
    e->f32_products = f32_instruction_mask;
    e->prefused.valid = false; e->prefused.heads = false;   // (work done ahead by another kernel family is not reused across a mode change)
    return MTFJSP_OK;
}
// exact multi-shard BatchNorm: after the launch that completes a BatchNorm's column sums, hand them to the caller's reduction
static int reduce_stats(mtfjsp_encoder *e, double *st)
{
    if (!e->reduce_fn) return MTFJSP_OK;
    HIPCHK(e, hipStreamSynchronize(e->stream));
    if (e->reduce_fn(e->reduce_user, st, STAT_REP * 256) != 0) { e->err = "the BatchNorm statistics reduction callback failed"; return MTFJSP_ERR_STATE; }
    return MTFJSP_OK;
}
extern "C" int mtfjsp_encoder_set_stats_reduce(mtfjsp_encoder_t e, mtfjsp_stats_reduce_fn fn, void *user, int64_t global_batch)
{
    if (!e || (fn && global_batch < e->cfg.batch)) return MTFJSP_ERR_ARG;
    e->reduce_fn = fn; e->reduce_user = user;
    e->prefused.valid = false; e->prefused.heads = false;
    e->reduce_scale = fn ? (double)global_batch / (double)e->cfg.batch : 1.0;
    return MTFJSP_OK;
}
extern "C" int mtfjsp_encoder_set_bn_mode(mtfjsp_encoder_t e, int32_t per_instance)
{
    if (!e || per_instance < 0 || per_instance > 1) return MTFJSP_ERR_ARG;
    e->bn_mode = per_instance;
    e->prefused.valid = false; e->prefused.heads = false;
    return MTFJSP_OK;
}
// deferred = 1: the forward entries stop polling the asynchronous failure words; failures surface at mtfjsp_encoder_check only.  For
// callers whose forwards contain collectives (exact multi-shard BatchNorm): every rank must issue the same sequence of forwards, so a
// rank-local failure may only be acted upon at a point all ranks agree on (rollout.py: once per step, after a synchronised check).
extern "C" int mtfjsp_encoder_set_deferred_poll(mtfjsp_encoder_t e, int32_t deferred)
{
    if (!e || deferred < 0 || deferred > 1) return MTFJSP_ERR_ARG;
    e->defer_poll = deferred != 0;
    return MTFJSP_OK;
}

static void set_head_images(mtfjsp_encoder *e, HeadArgs &ha, const std::string &policy, const std::string &critic)
{
    ha.range_flag = e->range_flag;
    ha.W0x = e->wx6.at(policy + ".linears.0.weight"); ha.W1x = e->wx6.at(policy + ".linears.1.weight");
    ha.Wc0x = e->wx6.at(critic + ".linears.0.weight"); ha.Wc1x = e->wx6.at(critic + ".linears.1.weight");
    ha.sW0 = e->wx6_sinv.at(policy + ".linears.0.weight"); ha.sW1 = e->wx6_sinv.at(policy + ".linears.1.weight");
    ha.sWc0 = e->wx6_sinv.at(critic + ".linears.0.weight"); ha.sWc1 = e->wx6_sinv.at(critic + ".linears.1.weight");
}
static void launch_heads(mtfjsp_encoder *e, HeadArgs &ha, const std::string &policy, const std::string &critic, const GatArgs *fused_gat = nullptr,
                         const EnvParams *env_tail = nullptr, const HeadArgs *fused_mheads = nullptr)
{
    if (e->f32_products & 4) { hipLaunchKernelGGL(k_heads, dim3((ha.B + HG - 1) / HG), dim3(512), heads_lds_bytes(), e->stream, ha); return; }
    ha.hg = (!fused_gat && !env_tail && e->heads_hg8 && 2 * ((ha.B + HG - 1) / HG) <= e->num_cu) ? HG / 2 : HG;
    const int grid = (ha.B + ha.hg - 1) / ha.hg;
    set_head_images(e, ha, policy, critic);
    if (fused_gat) {                                                // + the machine path's GAT passes of the same 16 instances
        const size_t lds = headsx_lds_bytes() > gat3x_lds_bytes() ? headsx_lds_bytes() : gat3x_lds_bytes();
#if !MTFJSP_BODY_FUNCS
        if (fused_mheads) {                                         // + the machine actor's heads behind the in-launch exchange of the node statistics
            XchgArgs xg{};
            const int set = (int)(e->xw_epoch++ & 1);
            xg.words = e->xw + (size_t)set * XW_SET; xg.words_next = e->xw + (size_t)(set ^ 1) * XW_SET;
            xg.nblk = (unsigned)grid; xg.fail = e->res_fail; xg.range_flag = e->range_flag;
            xg.fine_limit = e->xchg_fine_limit > 0 && e->xchg_fine_limit <= 2147483648.0 ? e->xchg_fine_limit : 2147483648.0;
            if (++e->fused3_launches == e->fused3_fail_at) xg.nblk += 8;      // (diagnostic) every group's count stays one short: the time-out path
#ifdef MTFJSP_STAMP3
            static unsigned long long *d_st3 = nullptr;
            if (!d_st3) { (void)hipMalloc((void **)&d_st3, (size_t)2048 * 64 * 8); (void)hipMemset(d_st3, 0, (size_t)2048 * 64 * 8); }
            xg.stamps = d_st3;
#endif
            // the node rows stay in LDS where every tile has a buffer of its own (<= 12 tiles per workgroup: M <= 6) and the machine part stages
            // its rows by instance (R <= HCH); MTFJSP_FUSED3_NODES_HBM=1: the round-5 form (rows written to e->node and read back)
            const int gtiles = (2 * ha.B * fused_mheads->R + 15) / 16, gper = (gtiles + grid - 1) / grid;
            const bool nlds = e->nodes_lds && gper <= 12 && fused_mheads->R <= HCH && (fused3_lds_bytes() >= (size_t)(8 * 2 * 4 * 64 * 16 + gper * 16 * HD * 4));
            const EnvParams ep0 = env_tail ? *env_tail : EnvParams{};
            const int envm = !env_tail ? 0 : env_tail->obs_f32 ? 1 : 2;
#define L3(EV, NL) hipLaunchKernelGGL((k_headsx_gat3x_headsx<EV, NL>), dim3(grid), dim3(512), fused3_lds_bytes(), e->stream, ha, *fused_gat, *fused_mheads, xg, ep0)
            if (nlds) { if (envm == 0) L3(0, true); else if (envm == 1) L3(1, true); else L3(2, true); }
            else { if (envm == 0) L3(0, false); else if (envm == 1) L3(1, false); else L3(2, false); }
#undef L3
#ifdef MTFJSP_STAMP3
            if (e->fused3_launches % 50 == 20 && getenv("MTFJSP_STAMP_PRINT")) {
                (void)hipStreamSynchronize(e->stream);
                std::vector<unsigned long long> h((size_t)grid * 64);
                (void)hipMemcpy(h.data(), d_st3, h.size() * 8, hipMemcpyDeviceToHost);
                std::vector<unsigned long long> h2((size_t)grid * 64);
                (void)hipMemcpy(h2.data(), d_st3 + (size_t)256 * 64, h2.size() * 8, hipMemcpyDeviceToHost);
                unsigned long long t0 = ~0ull;
                for (int w = 0; w < grid * 8; w++) t0 = h[(size_t)w * 8] < t0 ? h[(size_t)w * 8] : t0;
                printf("STAMP3 (us since the first wave's start; mean / max over %d waves): ", grid * 8);
                const char *nm[8] = {"start", "job heads done", "gat tiles done", "gat stats out", "poll done", "bn staged", "mheads products done", "end"};
                for (int i = 0; i < 8; i++) {
                    double m = 0, mx = 0; int n = 0;
                    for (int w = 0; w < grid * 8; w++) { if (!h[(size_t)w * 8 + i]) continue; const double v = (double)(h[(size_t)w * 8 + i] - t0) / 100.0; m += v; mx = v > mx ? v : mx; n++; }
                    printf(" [%s] %.2f/%.2f", nm[i], n ? m / n : 0.0, mx);
                }
                printf("\n");
                for (int i = 1; i <= 3; i++) {                          // per wave index: who finishes when
                    printf("STAMP3 [%s] by wave:", nm[i]);
                    for (int wv = 0; wv < 8; wv++) {
                        double m = 0; int n = 0;
                        for (int b = 0; b < grid; b++) { const unsigned long long x = h[((size_t)b * 8 + wv) * 8 + i]; if (x) { m += (double)(x - t0) / 100.0; n++; } }
                        printf(" %.2f", n ? m / n : 0.0);
                    }
                    printf("\n");
                }
                for (int part = 0; part < 2; part++) {                  // the heads statements' phase boundaries (H3_RT): job part, machine part
                    std::vector<unsigned long long> h3((size_t)grid * 64);
                    (void)hipMemcpy(h3.data(), d_st3 + (size_t)(part ? 1024 : 512) * 64, h3.size() * 8, hipMemcpyDeviceToHost);
                    const char *hn[8] = {"requests waited for (6)", "weights + pool (7)", "stage done (0)", "phase A (1)", "X planes (2)", "phase B + s1 (3)", "phase C + scores (4)", "softmax + selection (5)"};
                    const int order[8] = {6, 7, 0, 1, 2, 3, 4, 5};
                    printf("STAMP3 %s heads, mean over waves 0-3 / 4-7:", part ? "machine" : "job");
                    for (int oi = 0; oi < 8; oi++) {
                        double m[2] = {0, 0}; int n[2] = {0, 0};
                        for (int w = 0; w < grid * 8; w++) { const unsigned long long x = h3[(size_t)w * 8 + order[oi]]; if (x) { m[(w & 7) >> 2] += (double)(x - t0) / 100.0; n[(w & 7) >> 2]++; } }
                        printf(" [%s] %.2f/%.2f", hn[oi], n[0] ? m[0] / n[0] : 0.0, n[1] ? m[1] / n[1] : 0.0);
                    }
                    printf("\n");
                }
                for (int part = 0; part < 2; part++) {                  // ... and inside phases B, C and the selection (H3S_RT)
                    std::vector<unsigned long long> h3((size_t)grid * 64);
                    (void)hipMemcpy(h3.data(), d_st3 + (size_t)((part ? 1024 : 512) + 256) * 64, h3.size() * 8, hipMemcpyDeviceToHost);
                    const char *hn[8] = {"B products", "c2", "barrier", "s1 planes", "C products + partial scores", "value head", "barrier + scores", "probabilities"};
                    printf("STAMP3 %s heads detail, mean over waves 0-3 / 4-7:", part ? "machine" : "job");
                    for (int oi = 0; oi < 8; oi++) {
                        double m[2] = {0, 0}; int n[2] = {0, 0};
                        for (int w = 0; w < grid * 8; w++) { const unsigned long long x = h3[(size_t)w * 8 + oi]; if (x) { m[(w & 7) >> 2] += (double)(x - t0) / 100.0; n[(w & 7) >> 2]++; } }
                        printf(" [%s] %.2f/%.2f", hn[oi], n[0] ? m[0] / n[0] : 0.0, n[1] ? m[1] / n[1] : 0.0);
                    }
                    printf("\n");
                }
                for (int part = 0; part < 2; part++) {                  // ... and inside the selection (H3T_RT; waves 0-3)
                    std::vector<unsigned long long> h3((size_t)grid * 64);
                    (void)hipMemcpy(h3.data(), d_st3 + (size_t)(part ? 1792 : 1536) * 64, h3.size() * 8, hipMemcpyDeviceToHost);
                    const char *hj[5] = {"picked, index handed over", "predecessor's machine", "t / p / transport requested", "t / p arrived", "means arrived"};
                    const char *hm[5] = {"picked", "poll loop entered", "first poll back", "second poll back", "third poll back"};
                    const char **hn = part ? hm : hj;
                    printf("STAMP3 %s, mean over waves 0-3:", part ? "machine exchange / selection" : "job selection");
                    for (int oi = 0; oi < 5; oi++) {
                        double m = 0; int n = 0;
                        for (int w = 0; w < grid * 8; w++) { const unsigned long long x = h3[(size_t)w * 8 + oi]; if (x) { m += (double)(x - t0) / 100.0; n++; } }
                        printf(" [%s] %.2f", hn[oi], n ? m / n : 0.0);
                    }
                    printf("\n");
                }
                const char *gn[8] = {"gat entry", "staged", "projection", "pass 1", "pass 2", "pass 3 = first tile", "second tile", "all tiles done (barrier)"};
                for (int i = 0; i < 8; i++) {                           // inside the GAT statements (mtfjsp_gat3x_body.h: G3_RT)
                    printf("STAMP3 gat [%s] by wave:", gn[i]);
                    for (int wv = 0; wv < 8; wv++) {
                        double m = 0; int n = 0;
                        for (int b = 0; b < grid; b++) { const unsigned long long x = h2[((size_t)b * 8 + wv) * 8 + i]; if (x) { m += (double)(x - t0) / 100.0; n++; } }
                        printf(" %.2f", n ? m / n : 0.0);
                    }
                    printf("\n");
                }
            }
#endif
            return;
        }
#endif
        hipLaunchKernelGGL(k_headsx_gat3x, dim3(grid), dim3(512), lds, e->stream, ha, *fused_gat);
        return;
    }
    if (env_tail) {                                                 // + the environment step of the same 16 instances (k_env_grp16's partition)
        const size_t lds = headsx_lds_bytes() > EnvGrpDynLds<1>::bytes ? headsx_lds_bytes() : EnvGrpDynLds<1>::bytes;
        if (env_tail->obs_f32) hipLaunchKernelGGL(k_headsx_envstep<float>, dim3(grid), dim3(512), lds, e->stream, ha, *env_tail);
        else hipLaunchKernelGGL(k_headsx_envstep<double>, dim3(grid), dim3(512), lds, e->stream, ha, *env_tail);
        return;
    }
    if (e->vo_now) { hipLaunchKernelGGL(k_headsx_values, dim3(grid), dim3(512), headsx_lds_bytes(), e->stream, ha); return; }
    const int ntl = (ha.hg * ha.R + 15) / 16;                       // tiles of a full group
    if (e->heads10 && ntl > 6 && ntl <= 10) { hipLaunchKernelGGL(k_headsx10, dim3(grid), dim3(512), headsx10_lds_bytes(), e->stream, ha); return; }
    hipLaunchKernelGGL(k_headsx, dim3(grid), dim3(512), headsx_lds_bytes(), e->stream, ha);
}
static void arm_sampling(mtfjsp_encoder *e, int which, HeadArgs &ha)
{
    mtfjsp_encoder::FusedSample &f = e->fs[which];
    if (!f.armed) return;
    ha.sample_mode = f.greedy ? 2 : 1; ha.seed = f.seed; ha.counter = f.counter;
    ha.idx_out = f.idx; ha.logp_out = f.logp; ha.gather_from = f.gather_from; ha.gathered_out = f.gathered;
    f.armed = false;                                                // one forward only
}
extern "C" int mtfjsp_encoder_arm_selection(mtfjsp_encoder_t e, int32_t which, int32_t greedy, uint64_t seed, uint64_t counter,
                                            int32_t *idx_out, float *logp_out, const int32_t *gather_from, int32_t *gathered_out)
{
    if (!e || which < 0 || which > 1 || !idx_out) return MTFJSP_ERR_ARG;
    mtfjsp_encoder::FusedSample &f = e->fs[which];
    f.armed = true; f.greedy = greedy; f.seed = seed; f.counter = counter; f.idx = idx_out; f.logp = logp_out;
    f.gather_from = gather_from; f.gathered = gathered_out;
    return MTFJSP_OK;
}

extern "C" int mtfjsp_encoder_arm_mfea1(mtfjsp_encoder_t e, const mtfjsp_mfea1_ctx_t *ctx)
{
    if (!e || !ctx || !ctx->t || !ctx->p || !ctx->tt || !ctx->mean3 || !ctx->shop || !ctx->link || !ctx->m_fea1_out || !ctx->mmask_out) return MTFJSP_ERR_ARG;
    if (ctx->M != e->cfg.n_machine || ctx->T != e->T) { e->err = "mfea1 context of a different problem size"; return MTFJSP_ERR_ARG; }
    e->mf_ctx = *ctx; e->mf_armed = true;
    return MTFJSP_OK;
}

extern "C" int mtfjsp_encoder_arm_machine_heads(mtfjsp_encoder_t e, float *prob, float *h_pooled, float *machine_v)
{
    if (!e || !prob || !h_pooled || !machine_v) return MTFJSP_ERR_ARG;
    e->mh.armed = true; e->mh.prob = prob; e->mh.h_pooled = h_pooled; e->mh.machine_v = machine_v;
    return MTFJSP_OK;
}
extern "C" int mtfjsp_encoder_arm_values_only(mtfjsp_encoder_t e)
{
    if (!e) return MTFJSP_ERR_ARG;
    e->values_only = 2;                                             // the next job forward and the next machine forward
    return MTFJSP_OK;
}
extern "C" int mtfjsp_encoder_fused_launches(mtfjsp_encoder_t e, int64_t *three_in_one_out)
{
    if (!e || !three_in_one_out) return MTFJSP_ERR_ARG;
    *three_in_one_out = e->fused3_launches;
    return MTFJSP_OK;
}

extern "C" int mtfjsp_encoder_arm_env_step(mtfjsp_encoder_t e, const void *params, int32_t bytes)
{
    if (!e || !params) return MTFJSP_ERR_ARG;
    if (bytes != (int32_t)sizeof(EnvParams)) { e->err = "mtfjsp_encoder_arm_env_step: parameter block of another library version"; return MTFJSP_ERR_ARG; }
    memcpy(&e->env_step.P, params, sizeof(EnvParams));
    e->env_step.armed = true; e->env_step.done = false;
    return MTFJSP_OK;
}
extern "C" int mtfjsp_encoder_env_step_fused(mtfjsp_encoder_t e) { return e && e->env_step.done ? 1 : 0; }

static HeadArgs machine_head_args(mtfjsp_encoder *e, const float *h_pooled_o, const uint8_t *mmask, float *prob, float *machine_v);
static int job_actor_forward_impl(mtfjsp_encoder_t e, const void *tasks_fea, const int32_t *ell_col, const float *ell_val,
                                  const int32_t *candidate, const uint8_t *job_mask, const float *h_m_prev,
                                  float *prob, float *h_pooled, float *job_v, float *h_nodes)
{
    if (!e || !tasks_fea || !ell_col || !ell_val || !candidate || !job_mask || !prob || !h_pooled || !job_v) return MTFJSP_ERR_ARG;
    int rc = mtfjsp_encoder_weights_ready(e);
    if (rc) return rc;
    HIPCHK(e, hipSetDevice(e->cfg.device_id));
    if (!e->defer_poll && (rc = res_poll_failure(e))) return rc;
    if (++e->job_forwards == e->range_fail_at && e->res_fail_host) e->res_fail_host[1] = 1u;   // (diagnostic, MTFJSP_RANGE_FAIL_AT)
    const int B = e->cfg.batch, J = e->cfg.n_job;
    auto W = [&](const std::string &k) { return e->w.at(k); };
    auto WI = [&](const std::string &k) { return e->wimg.at(k); };
    // the heads kernel does BatchNorm+ReLU, graph pool and candidate gather itself — while T is small: a workgroup pools 16
    // instances with 512 threads, which is too little parallelism for 100- or 400-row instances (measured: J20M20 x 2048 199 vs
    // 81+113 us; J10M10 x 8192 job+machine heads 264 us fused vs 110 + 130 us with the stand-alone pool/gather kernel)
    const bool resident = !e->bn_mode && e->res_ok && !e->reduce_fn && !(e->f32_products & (1 | 8 | 16));   // (a cross-shard reduction cannot happen inside the single launch)
    static const int fuse_maxT = getenv("MTFJSP_FUSE_POOL_MAXT") ? atoi(getenv("MTFJSP_FUSE_POOL_MAXT")) : 64;   // (J10M10 x 8192: 264 us fused, 240 us apart)
    const bool fuse_pool = !e->bn_mode && !resident && !h_nodes && e->T <= fuse_maxT;
    rc = e->bn_mode ? run_gin_inst(e, "job_actor.", tasks_fea, ell_col, ell_val, candidate, J, h_pooled, e->cand_feat, h_nodes)
         : resident ? run_gin_resident(e, "job_actor.", tasks_fea, ell_col, ell_val, candidate, J, h_pooled, e->cand_feat, h_nodes)
                    : run_gin(e, "job_actor.", tasks_fea, ell_col, ell_val, candidate, J, fuse_pool ? nullptr : h_pooled, e->cand_feat, h_nodes);
    if (rc) return rc;
    // ---- heads (ac:205-293): score = L2 tanh(L1 tanh(Wa cand + Wb pooled + Wc hm + b0))
    const float *hm = h_m_prev;
    if (!hm) {                                                     // first decision of an episode: the learned `_input` row for every instance (a constant
        if (!e->hm_b_valid) {                                      // of the weights: broadcast once per weight load, not once per episode)
            Timed t(e, "small");
            hipLaunchKernelGGL(k_bcast128, dim3((B * HD + 255) / 256), dim3(256), 0, e->stream, B, W("job_actor._input"), e->hm_b);
            e->hm_b_valid = true;
        }
        hm = e->hm_b;
    }
    {
        Timed t(e, (e->mf_armed && e->mf_ctx.m_fea2 && gat_fusable(e)) ? ((e->mh.armed && e->fs[1].armed && mheads_fusable(e)) ? "heads_gat3_heads" : "heads_gat3") : "heads");
        HeadArgs ha{};
        ha.B = B; ha.R = J; ha.X = e->cand_feat; ha.pooled = h_pooled; ha.other = hm;
        ha.W0i = WI("job_actor.o_policy.linears.0.weight"); ha.b0 = W("job_actor.o_policy.linears.0.bias");
        ha.W1i = WI("job_actor.o_policy.linears.1.weight"); ha.b1 = W("job_actor.o_policy.linears.1.bias");
        ha.w2 = W("job_actor.o_policy.linears.2.weight"); ha.b2 = W("job_actor.o_policy.linears.2.bias");
        ha.Wc0i = WI("job_actor.job_critic.linears.0.weight"); ha.bc0 = W("job_actor.job_critic.linears.0.bias");
        ha.Wc1i = WI("job_actor.job_critic.linears.1.weight"); ha.bc1 = W("job_actor.job_critic.linears.1.bias");
        ha.wc2 = W("job_actor.job_critic.linears.2.weight"); ha.bc2 = W("job_actor.job_critic.linears.2.bias");
        ha.mask = job_mask; ha.scale = 1.0f; ha.prob = prob; ha.value = job_v;
        if (fuse_pool) {
            const std::string P = "job_actor.encoder.feature_extract.";
            ha.X = e->zB; ha.pooled = nullptr; ha.pooled_out = h_pooled; ha.xgather = candidate; ha.xT = e->T; ha.xrelu = 1;
            ha.xbn_stats = e->stats + 5 * STAT_REP * 256; ha.xbn_gamma = W(P + "batch_norms.1.weight"); ha.xbn_beta = W(P + "batch_norms.1.bias");
            ha.xbn_inv_rows = 1.0 / ((double)B * (double)e->T * e->reduce_scale);
            ha.zero_stats = e->stats; ha.zero_count = 5 * STAT_REP * 256;    // slots 0..4 are consumed; slot 5 is being read by this very
            e->gin_slot5_dirty = true;                                       // kernel and is zeroed by the machine heads (or a memset)
        } else if (!e->bn_mode && !resident) {
            ha.zero_stats = e->stats; ha.zero_count = 6 * STAT_REP * 256;    // every GIN accumulator has been consumed by now
            e->gin_stats_clean = true;
        }
        arm_sampling(e, 0, ha);
        if (e->mf_armed) {
            e->mf_armed = false;
            if (!ha.sample_mode || !ha.gather_from) { e->err = "mtfjsp_encoder_arm_mfea1 needs mtfjsp_encoder_arm_selection(which = 0) with gather_from"; return MTFJSP_ERR_STATE; }
            ha.mf = e->mf_ctx; ha.mf_on = 1;
        }
#ifdef MTFJSP_STAMP
        static unsigned long long *d_st = nullptr;
        if (!d_st) (void)hipMalloc((void **)&d_st, 4096 * 8 * 8);
        ha.stamps = d_st;
#endif
        e->prefused.valid = false; e->prefused.heads = false;
        GatArgs ga{};
        HeadArgs hm_args{};
        // values only (armed pair): nothing rides in this launch, no selection, no m_fea1
        e->vo_now = e->values_only > 0 && !ha.sample_mode && !ha.mf_on && !(e->f32_products & 4) && !e->bn_mode;
        if (e->values_only > 0) e->values_only--;
        const bool with_gat = !e->vo_now && ha.mf_on && ha.mf.m_fea2 && gat_fusable(e);
        // the whole machine forward in this launch: armed outputs (mtfjsp_encoder_arm_machine_heads) and an armed machine selection
        const bool with_mheads = with_gat && e->mh.armed && e->fs[1].armed && mheads_fusable(e);
        e->mh.armed = false;
        if (with_gat && !with_mheads) {                                  // the machine forward that follows finds its GAT passes done
            const int slot = e->gat_slot;
            e->gat_slot ^= 1;
            const int grc = gat3x_args(e, "machine_actor.", ha.mf.m_fea1_out, ha.mf.m_fea2, slot, &ga);
            if (grc) return grc;
            e->prefused.valid = true; e->prefused.f1 = ha.mf.m_fea1_out; e->prefused.f2 = ha.mf.m_fea2; e->prefused.slot = slot;
        }
        if (with_mheads) {                                               // ... finds itself done (same pointers) and returns at once
            const int grc = gat3x_args(e, "machine_actor.", ha.mf.m_fea1_out, ha.mf.m_fea2, -1, &ga);
            if (grc) return grc;
            hm_args = machine_head_args(e, h_pooled, ha.mf.mmask_out, e->mh.prob, e->mh.machine_v);
            hm_args.pooled = nullptr; hm_args.pooled_out = e->mh.h_pooled;
            hm_args.xbn_stats = e->stats;                                // (non-NULL = "normalise X here"; the sums come from the in-launch exchange, these are not read)
            hm_args.xbn_gamma = W("machine_actor.bn.weight"); hm_args.xbn_beta = W("machine_actor.bn.bias");
            hm_args.xbn_inv_rows = 1.0 / ((double)B * (double)e->cfg.n_machine);
            hm_args.hg = HG;
            e->fs1_consumed = e->fs[1]; e->fs1_consumed.armed = true;
            arm_sampling(e, 1, hm_args);
            set_head_images(e, hm_args, "machine_actor.m_policy", "machine_actor.machine_critic");
            e->prefused.heads = true; e->prefused.f1 = ha.mf.m_fea1_out; e->prefused.f2 = ha.mf.m_fea2; e->prefused.h_pooled_o = h_pooled;
            e->prefused.mmask = ha.mf.mmask_out; e->prefused.prob = e->mh.prob; e->prefused.h_pooled = e->mh.h_pooled; e->prefused.machine_v = e->mh.machine_v;
        }
        // an environment step armed BEFORE this forward (mtfjsp_encoder_arm_env_step) rides in the three-in-one launch when the machine
        // selection made there is the one it reads and the shapes agree (round 6: two launches per rollout step)
        const EnvParams &EP = e->env_step.P;
        const bool env_tail = with_mheads && e->env_step.armed && e->fuse_env3 && !e->timing && hm_args.sample_mode && ha.sample_mode &&
                              (const void *)hm_args.idx_out == (const void *)EP.mach_idx && (const void *)ha.gathered_out == (const void *)EP.task_idx &&
                              EP.B == B && EP.M == e->cfg.n_machine && EP.T <= 64 && EP.M * EP.M <= 64 && EP.J <= 64 && fused3_lds_bytes() >= EnvGrpDynLds<1>::bytes;
        if (e->env_step.armed && with_mheads) { e->env_step.armed = false; e->env_step.done = env_tail; }
        launch_heads(e, ha, "job_actor.o_policy", "job_actor.job_critic", with_gat ? &ga : nullptr, env_tail ? &EP : nullptr, with_mheads ? &hm_args : nullptr);
        e->vo_now = false;
#ifdef MTFJSP_STAMP
        static int printed = 0;
        if (printed++ < 3 && getenv("MTFJSP_STAMP_PRINT")) {
            (void)hipStreamSynchronize(e->stream);
            const int nw = ((B + HG - 1) / HG) * 8;
            std::vector<unsigned long long> hst((size_t)nw * 8);
            (void)hipMemcpy(hst.data(), d_st, (size_t)nw * 64, hipMemcpyDeviceToHost);
            double m[8] = {0};
            for (int w = 0; w < nw; w++) for (int i = 0; i < 8; i++) m[i] += (double)hst[(size_t)w * 8 + i] / nw;
            printf("STAMP k_heads: bn-stage %.0f  weights+pool %.0f  rest-of-stage %.0f  phaseA %.0f  X-stage %.0f  phaseB %.0f  phaseC+score %.0f  softmax %.0f (cycles/wave)\n", m[6], m[7], m[0], m[1], m[2], m[3], m[4], m[5]);
        }
#endif
    }
    HIPCHK(e, hipGetLastError());
    return MTFJSP_OK;
}

// scorer / critic weights and in/out pointers of the machine actor's heads (ac:444-495); the caller adds the BatchNorm source
static HeadArgs machine_head_args(mtfjsp_encoder *e, const float *h_pooled_o, const uint8_t *mmask, float *prob, float *machine_v)
{
    auto W = [&](const std::string &k) { return e->w.at(k); };
    auto WI = [&](const std::string &k) { return e->wimg.at(k); };
    HeadArgs ha{};
    ha.B = e->cfg.batch; ha.R = e->cfg.n_machine; ha.X = e->node; ha.other = h_pooled_o;
    ha.W0i = WI("machine_actor.m_policy.linears.0.weight"); ha.b0 = W("machine_actor.m_policy.linears.0.bias");
    ha.W1i = WI("machine_actor.m_policy.linears.1.weight"); ha.b1 = W("machine_actor.m_policy.linears.1.bias");
    ha.w2 = W("machine_actor.m_policy.linears.2.weight"); ha.b2 = W("machine_actor.m_policy.linears.2.bias");
    ha.Wc0i = WI("machine_actor.machine_critic.linears.0.weight"); ha.bc0 = W("machine_actor.machine_critic.linears.0.bias");
    ha.Wc1i = WI("machine_actor.machine_critic.linears.1.weight"); ha.bc1 = W("machine_actor.machine_critic.linears.1.bias");
    ha.wc2 = W("machine_actor.machine_critic.linears.2.weight"); ha.bc2 = W("machine_actor.machine_critic.linears.2.bias");
    ha.mask = mmask; ha.scale = 10.0f; ha.prob = prob; ha.value = machine_v;
    return ha;
}
static int machine_actor_forward_impl(mtfjsp_encoder_t e, const void *m_fea1, const void *m_fea2, const float *h_pooled_o,
                                      const uint8_t *mmask, float *prob, float *h_pooled, float *machine_v)
{
    if (!e || !m_fea1 || !m_fea2 || !h_pooled_o || !mmask || !prob || !h_pooled || !machine_v) return MTFJSP_ERR_ARG;
    int rc = mtfjsp_encoder_weights_ready(e);
    if (rc) return rc;
    HIPCHK(e, hipSetDevice(e->cfg.device_id));
    if (!e->defer_poll && (rc = res_poll_failure(e))) return rc;
    const int B = e->cfg.batch, M = e->cfg.n_machine, R = B * M;
    auto W = [&](const std::string &k) { return e->w.at(k); };
    int slot = 0;
    if (e->prefused.heads && !e->bn_mode && m_fea1 == e->prefused.f1 && m_fea2 == e->prefused.f2 && h_pooled_o == e->prefused.h_pooled_o &&
        mmask == e->prefused.mmask && prob == e->prefused.prob && h_pooled == e->prefused.h_pooled && machine_v == e->prefused.machine_v) {
        // k_headsx_gat3x_headsx ran this whole forward (and its armed selection) inside the job actor's heads launch.  The match is by
        // POINTER: a caller that rewrites m_fea1 / the mask in place between the two forwards must not arm the machine heads
        // (include/mtfjsp.h, mtfjsp_encoder_arm_machine_heads).  A time-out of that launch's exchange is reported HERE, at the machine
        // forward of the same step, as the separate launch would (advisor r5) — unless the caller polls once per step itself.
        if (!e->defer_poll && (rc = res_poll_failure(e))) { e->prefused.heads = false; e->prefused.valid = false; return rc; }
        e->prefused.heads = false; e->prefused.valid = false;
        e->fs1_consumed = mtfjsp_encoder::FusedSample{};
        e->env_step.armed = false;                                    // (env_step.done: set by the job forward whose launch ran this forward — and the armed step, if it could)
        e->fs[1].armed = false;                                       // (a selection armed again for this call has been made already)
        return MTFJSP_OK;
    }
    if (e->prefused.heads && e->fs1_consumed.idx && !e->fs[1].armed) e->fs[1] = e->fs1_consumed;   // other pointers: this forward is computed here and selects again with the selection the fused launch consumed
    e->fs1_consumed = mtfjsp_encoder::FusedSample{};
    e->prefused.heads = false;
    if (e->prefused.valid && !e->bn_mode && m_fea1 == e->prefused.f1 && m_fea2 == e->prefused.f2) slot = e->prefused.slot;   // k_headsx_gat3x did it
    else rc = e->bn_mode ? run_gat_inst(e, "machine_actor.", m_fea1, m_fea2, h_pooled) : run_gat(e, "machine_actor.", m_fea1, m_fea2, nullptr, &slot);
    e->prefused.valid = false;
    if (rc) return rc;
    {
        Timed t(e, "heads");
        HeadArgs ha = machine_head_args(e, h_pooled_o, mmask, prob, machine_v);
        if (e->bn_mode) ha.pooled = h_pooled;                     // k_gat_inst already normalised and pooled the nodes per instance
        else {
            ha.pooled = nullptr;
            ha.xbn_stats = e->stats + (6 + slot) * STAT_REP * 256; ha.xbn_gamma = W("machine_actor.bn.weight"); ha.xbn_beta = W("machine_actor.bn.bias");
            ha.xbn_inv_rows = 1.0 / ((double)R * e->reduce_scale); ha.pooled_out = h_pooled;
            ha.zero_stats = e->stats + (6 + (slot ^ 1)) * STAT_REP * 256; ha.zero_count = STAT_REP * 256;   // the slot of the next machine forward
            e->gat_stats_clean[slot ^ 1] = true;
            if (e->gin_slot5_dirty) {                                 // left by a job forward that pooled inside its heads kernel
                ha.zero_stats2 = e->stats + 5 * STAT_REP * 256; ha.zero_count2 = STAT_REP * 256;
                e->gin_slot5_dirty = false; e->gin_stats_clean = true;
            }
        }
        arm_sampling(e, 1, ha);
        e->vo_now = e->values_only > 0 && !ha.sample_mode && !(e->f32_products & 4) && !e->bn_mode;
        if (e->values_only > 0) e->values_only--;
        // an armed environment step (mtfjsp_encoder_arm_env_step) rides in this launch when the selection made here is the one it
        // reads, the shapes agree and the launch is the split-product heads kernel; otherwise the caller steps the environment itself
        const EnvParams &EP = e->env_step.P;
        const bool env_tail = e->env_step.armed && e->fuse_env && !e->timing && !e->bn_mode && !(e->f32_products & 4) && ha.sample_mode &&
                              (const void *)ha.idx_out == (const void *)EP.mach_idx && EP.B == B && EP.M == M && EP.T <= 64 && EP.M * EP.M <= 64 && EP.J <= 64;
        e->env_step.armed = false; e->env_step.done = env_tail;
        launch_heads(e, ha, "machine_actor.m_policy", "machine_actor.machine_critic", nullptr, (env_tail && !e->vo_now) ? &EP : nullptr);
        e->vo_now = false;
    }
    HIPCHK(e, hipGetLastError());
    return MTFJSP_OK;
}

// out[b][o] = W[o,:] . x[b,:] + bias[o]   (the last 128 -> O linear of a critic head), thread = (b, o)
__global__ void k_linear_small(int B, int O, const float *x, const float *W, const float *bias, float *out, unsigned *range_flag)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * O) return;
    const int b = i / O, o = i % O;
    float p = 0.f;
    for (int k = 0; k < HD; k++) p = fmaf(x[(size_t)b * HD + k], W[o * HD + k], p);
    p += bias[o];
    out[i] = p;
    if (range_flag && p != p) __hip_atomic_store(range_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// = Global_Critic_JointAction_GAT.forward (ac:587-750): own GIN encoder -> graph pool, own GAT machine path
// -> pool, MLPCritic(256 -> 128 -> 128 -> 4) on [pooled_m, pooled_o].  Weights under "global_critic.".
static int global_critic_forward_impl(mtfjsp_encoder_t e, const void *tasks_fea, const int32_t *ell_col, const float *ell_val,
                                      const void *m_fea1, const void *m_fea2, float *value4)
{
    if (!e || !tasks_fea || !ell_col || !ell_val || !m_fea1 || !m_fea2 || !value4) return MTFJSP_ERR_ARG;
    static const char *need[] = {"global_critic.critic.linears.0.weight", "global_critic.critic.linears.2.bias", "global_critic.gat_layer.W",
                                 "global_critic.encoder.feature_extract.mlps.1.linears.2.weight", "global_critic.bn.weight"};
    for (const char *k : need)
        if (!e->w.count(k)) { e->err = std::string("missing weight: ") + k; return MTFJSP_ERR_STATE; }
    HIPCHK(e, hipSetDevice(e->cfg.device_id));
    // this forward's GAT passes overwrite e->node and move on to another statistics slot: GAT passes the job actor's heads launch
    // ran ahead for a coming machine-actor forward (k_headsx_gat3x) are gone, that forward has to redo them — and so is a whole
    // machine forward done ahead (k_headsx_gat3x_headsx): its node rows are overwritten
    e->prefused.valid = false; e->prefused.heads = false;
    if (!e->defer_poll) { const int prc = res_poll_failure(e); if (prc) return prc; }
    const int B = e->cfg.batch;
    auto W = [&](const std::string &k) { return e->w.at(k); };
    auto WT = [&](const std::string &k) { return e->wt.at(k); };
    const bool resident = e->res_ok && !e->reduce_fn && !(e->f32_products & (1 | 8 | 16));
    int rc = resident ? run_gin_resident(e, "global_critic.", tasks_fea, ell_col, ell_val, nullptr, 0, e->pooled_int, e->cand_feat, nullptr)
                      : run_gin(e, "global_critic.", tasks_fea, ell_col, ell_val, nullptr, 0, e->pooled_int, e->cand_feat, nullptr);
    if (rc) return rc;
    rc = run_gat(e, "global_critic.", m_fea1, m_fea2, e->u);
    if (rc) return rc;
    const float *W0t = WT("global_critic.critic.linears.0.weight");             // [pooled_m | pooled_o] (ac: torch.cat((h_g_m_pooled, h_g_o_pooled)))
    GemmArgs a = gemm_args(e->u, B, W0t, W("global_critic.critic.linears.0.bias"), e->c1);
    launch_gemm<PRO_PLAIN, EPI_PLAIN>(e, a, "head_gemm");
    GemmArgs b = gemm_args(e->pooled_int, B, W0t + HD * HD, nullptr, e->c1);
    launch_gemm<PRO_PLAIN, EPI_TANH, true>(e, b, "head_gemm");
    GemmArgs c = gemm_args(e->c1, B, WT("global_critic.critic.linears.1.weight"), W("global_critic.critic.linears.1.bias"), e->c2);
    launch_gemm<PRO_PLAIN, EPI_TANH>(e, c, "head_gemm");
    {
        Timed t(e, "small");
        hipLaunchKernelGGL(k_linear_small, dim3((B * 4 + 255) / 256), dim3(256), 0, e->stream, B, 4, e->c2, W("global_critic.critic.linears.2.weight"),
                           W("global_critic.critic.linears.2.bias"), value4, split_products_in_use(e) ? e->range_flag : nullptr);
    }
    HIPCHK(e, hipGetLastError());
    return MTFJSP_OK;
}

// C ABI wrappers: no C++ exception may cross the boundary (a weight missing from the maps surfaces as MTFJSP_ERR_STATE)
#define GUARDED(e, call)                                                                               \
    try { return call; }                                                                               \
    catch (const std::exception &ex) { if (e) (e)->err = std::string("missing weight or internal error: ") + ex.what(); return MTFJSP_ERR_STATE; }
extern "C" int mtfjsp_job_actor_forward(mtfjsp_encoder_t e, const void *tasks_fea, const int32_t *ell_col, const float *ell_val,
                                        const int32_t *candidate, const uint8_t *job_mask, const float *h_m_prev,
                                        float *prob, float *h_pooled, float *job_v, float *h_nodes)
{
    GUARDED(e, job_actor_forward_impl(e, tasks_fea, ell_col, ell_val, candidate, job_mask, h_m_prev, prob, h_pooled, job_v, h_nodes))
}
extern "C" int mtfjsp_machine_actor_forward(mtfjsp_encoder_t e, const void *m_fea1, const void *m_fea2, const float *h_pooled_o,
                                            const uint8_t *mmask, float *prob, float *h_pooled, float *machine_v)
{
    GUARDED(e, machine_actor_forward_impl(e, m_fea1, m_fea2, h_pooled_o, mmask, prob, h_pooled, machine_v))
}
extern "C" int mtfjsp_global_critic_forward(mtfjsp_encoder_t e, const void *tasks_fea, const int32_t *ell_col, const float *ell_val,
                                            const void *m_fea1, const void *m_fea2, float *value4)
{
    GUARDED(e, global_critic_forward_impl(e, tasks_fea, ell_col, ell_val, m_fea1, m_fea2, value4))
}

extern "C" int mtfjsp_sample_categorical(mtfjsp_encoder_t e, const float *prob, int32_t n, int32_t greedy, uint64_t seed, uint64_t counter,
                                         int32_t *idx_out, float *logp_out, const int32_t *gather_from, int32_t *gathered_out)
{
    if (!e || !prob || !idx_out || n < 1) return MTFJSP_ERR_ARG;
    HIPCHK(e, hipSetDevice(e->cfg.device_id));
    Timed t(e, "sample");
    const int B = e->cfg.batch;
    hipLaunchKernelGGL(k_sample, dim3((B + 127) / 128), dim3(128), 0, e->stream, B, n, prob, greedy, seed, counter, idx_out, logp_out, gather_from, gathered_out);
    HIPCHK(e, hipGetLastError());
    return MTFJSP_OK;
}

extern "C" int mtfjsp_encoder_check(mtfjsp_encoder_t e, int32_t *gin_resident_out)
{
    if (!e) return MTFJSP_ERR_ARG;
    HIPCHK(e, hipSetDevice(e->cfg.device_id));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    int rc = res_poll_failure(e);                                  // (the stream is idle: the word is final)
    if (!rc && e->res_eligible && !e->res_ok && e->res_failures > 0 && !getenv("MTFJSP_NO_RESIDENT_GIN")) {
        // a failure was reported earlier: the stream is idle now, so the census can run again; when the device holds the whole
        // grid again the single launch comes back (one transient overlap does not leave the handle on the slow path for good)
        e->res_ok = res_census(e);
    }
    if (gin_resident_out) *gin_resident_out = (e->res_ok && !e->reduce_fn && !(e->f32_products & (1 | 8 | 16))) ? 1 : 0;
    return rc;
}
extern "C" int mtfjsp_encoder_resident_failures(mtfjsp_encoder_t e, int64_t *count_out)
{
    if (!e || !count_out) return MTFJSP_ERR_ARG;
    *count_out = e->res_failures;
    return MTFJSP_OK;
}

// diagnostic: the machine path's node rows as the GAT passes left them ([B*M,128] f32, pre-BatchNorm unless a pooling kernel has
// normalised them in place) copied to host memory — what tools/first_launch/ compares between two builds of the same kernel
extern "C" int mtfjsp_encoder_peek_nodes_host(mtfjsp_encoder_t e, float *out_host, int64_t count)
{
    if (!e || !out_host || count < 0 || count > (int64_t)e->cfg.batch * e->cfg.n_machine * HD) return MTFJSP_ERR_ARG;
    HIPCHK(e, hipSetDevice(e->cfg.device_id));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    HIPCHK(e, hipMemcpy(out_host, e->node, (size_t)count * sizeof(float), hipMemcpyDeviceToHost));
    return MTFJSP_OK;
}
extern "C" int mtfjsp_encoder_range_fallbacks(mtfjsp_encoder_t e, int64_t *count_out, int32_t *product_mode_out)
{
    if (!e || !count_out) return MTFJSP_ERR_ARG;
    *count_out = e->range_fallbacks;
    if (product_mode_out) *product_mode_out = e->f32_products;
    return MTFJSP_OK;
}

extern "C" int mtfjsp_encoder_timing_begin(mtfjsp_encoder_t e)
{
    if (!e) return MTFJSP_ERR_ARG;
    for (auto &kv : e->ev) { for (auto &p : kv.second) e->ev_free.push_back(p); kv.second.clear(); }
    e->timing = true;
    return MTFJSP_OK;
}
// total over all kernels; per-kernel-family numbers via mtfjsp_encoder_timing_query
extern "C" int mtfjsp_encoder_timing_end(mtfjsp_encoder_t e, double *ms_total, int64_t *launches)
{
    if (!e) return MTFJSP_ERR_ARG;
    HIPCHK(e, hipSetDevice(e->cfg.device_id));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    double tot = 0; int64_t n = 0;
    for (auto &kv : e->ev)
        for (auto &p : kv.second) { float ms = 0; HIPCHK(e, hipEventElapsedTime(&ms, p.first, p.second)); tot += ms; n++; }
    if (ms_total) *ms_total = tot;
    if (launches) *launches = n;
    e->timing = false;
    return MTFJSP_OK;
}
extern "C" int mtfjsp_encoder_timing_query(mtfjsp_encoder_t e, const char *family, double *ms_total, int64_t *launches)
{
    if (!e || !family) return MTFJSP_ERR_ARG;
    HIPCHK(e, hipSetDevice(e->cfg.device_id));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    double tot = 0; int64_t n = 0;
    auto it = e->ev.find(family);
    if (it != e->ev.end())
        for (auto &p : it->second) { float ms = 0; HIPCHK(e, hipEventElapsedTime(&ms, p.first, p.second)); tot += ms; n++; }
    if (ms_total) *ms_total = tot;
    if (launches) *launches = n;
    return MTFJSP_OK;
}
