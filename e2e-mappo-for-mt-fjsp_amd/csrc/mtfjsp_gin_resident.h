// mtfjsp_gin_resident.h — the whole GIN encoder (gcn:109-197) of one forward as ONE launch whose activations never leave
// the chip.  Compiled in mtfjsp_gin_res.hip; mtfjsp_encoder.hip includes it with MTFJSP_GIN_RES_DECL_ONLY (arguments, sizes, helpers, the kernel's
// declaration).  Uses mtfjsp_enc_shared.h (split3x4, row_sum16, bn_relu_ss, LDS_BARRIER, STAT_REP, BN_EPS).
//
// Why: with training-mode BatchNorm every one of the six Linear products needs the batch statistics of its output before the
// next layer can start, so the streaming design (k_gemm_x6) writes and re-reads the [rows,128] f32 activations at every
// boundary: 6 launches, 983 MB per forward at 147 456 rows, each launch HBM-bound.  At the headline batch (4096 J6M6
// instances = 576 rows per CU) the activations fit in the register file instead: 256 workgroups (one per CU, 4 waves, one per
// SIMD with 512 registers each) keep z[576 rows][128] as 288 accumulator registers per lane; a layer is
//     registers --BatchNorm+ReLU(+aggregation), 2-way f16 split--> LDS planes --ds_read_b128--> matrix cores --> registers
// and the only global traffic between layers is the BatchNorm column sums (2 KB per workgroup) and a grid-wide barrier.
//
// Products.  An f32 operand x is split into high = f16(x) and low = f16(x - high) (round to nearest both times: x = high +
// low up to 2^-22 |x|), the weights likewise on the host after scaling by a power of two (so that their low pieces stay
// normal f16 numbers; 1/scale goes into the following BatchNorm, exactly), and a product is the three significant piece
// products w_hi x_lo + w_lo x_hi + w_hi x_hi accumulated in f32 by v_mfma_f32_32x32x16_f16 — the dropped w_lo x_lo term is
// <= 2^-22 |w x| too.  The rounding of a 128-term f32 FMA chain is of the same size, and the tests hold the kernel to the same
// bounds as the exact 6-product bf16 split of the streaming kernels (which it replaced here: half the matrix work and about
// half the vector work per element; measured 16.5 us instead of 23.4 us per layer).  f16 overflows at 65504: the operands are
// BatchNorm outputs (|.| <= sqrt(rows) * |gamma| + |beta|) and, in the second GIN layer, their neighbour sums WEIGHTED BY THE EDGE
// WEIGHTS (values to 309 at J6M6, larger on larger instances: (h + w0 h0 + w1 h1) / nnz can reach ~100 |h|) — comfortably inside
// the range for trained checkpoints, but not guaranteed: an operand beyond it becomes (inf | -inf) pieces whose products cancel
// to NaN; the NaN reaches the BatchNorm sums of the layer and through them every output of the forward, where the heads kernel
// raises the handle's range flag and the host repeats the forward on the f32-instruction kernels (mtfjsp_encoder.hip:
// res_poll_failure) — never a silent inf or a saturated value.  The first Linear (12 raw features, unbounded) keeps the exact
// bf16 split.  Matrix work: 14.5 GFLOP of piece products per layer = 6.4 us at the
// measured 2.27 PFLOP/s.
//
// Layout.  v_mfma_f32_32x32x16_f16 with the operands swapped (A := weight fragments, resident in registers; B := activation
// rows from LDS): wave w owns output columns 32w..32w+31 of ALL rows of the workgroup.  Lane l = (n = l & 31, h = l >> 5) holds,
// for row tile rt (32 rows), row 32rt+n, columns 32w + 8g + 4h + r (g, r = 0..3) in acc[rt][4g+r] — 16 registers per tile, 18
// tiles.  The 32x32 shape is chosen over 16x16x32 because one wave per SIMD cannot overlap its own vector and matrix
// instructions beyond the issue slots a matrix instruction leaves free: 24 of 32 cycles here against 8 of 16 (measured: the
// 16x16x32 form of this kernel took 28 us per layer, of which 11 us were vector work that did not overlap).
// A tile's planes (2 x 32 rows x 272 B) are written by all four waves (each its 32 columns) into one of two LDS buffers while
// the previous tile is being multiplied: one LDS barrier per tile.
// The second GIN layer's neighbour aggregation (gcn:125-149) reads rows of the same instance: h = relu(bn(z)) is staged as
// f32 in a ring of 6 row tiles (192 rows, XOR-swizzled); with T <= 65 rows per instance the neighbours of tile rt lie in tiles
// rt-2..rt+2.  Eligibility (host): 16 <= T <= 65 and ceil(B / 256) instances * T <= 576 rows per workgroup.
// The bias of a Linear that feeds a BatchNorm cancels exactly (BN(z + b) = BN(z)), so no bias is added anywhere here.
#pragma once
#include <type_traits>
#include <utility>

#ifndef GR_ABL                            // diagnostic timing ablations (wrong results): 1 no plane production, 2 no matrix products, 4 no statistics,
                                          // 8 no plane writes, 16 no pooling, 32 no candidate gather
#define GR_ABL 0
#endif
#define GR_NT 18                          // row tiles per workgroup (576 rows)
#ifndef GR_NRES
#define GR_NRES 16                        // ... of which this many stay in registers; the others go through zspill (see below)
#endif
#ifndef GR_VRES
#define GR_VRES 1                         // 1: row tiles GR_NRES.. stay on chip too — in the VECTOR registers that used to stage their round trip through global
#endif                                    // memory (the matrix instructions on them are written out with vector-register accumulators); 0: the round trip
#define GR_ROWS (32 * GR_NT)
#define GR_RING 6                         // row tiles in the f32 ring of the aggregation layer
#define GR_NBAR 6                         // grid barriers per launch
#ifndef GR_GRP
#define GR_GRP 2                          // row tiles per workgroup barrier in the layers without aggregation (2 * GR_GRP plane buffers; measured 1: 142, 2: 135, 3: 138 us per launch)
#endif
#ifndef GR_LOOK
#define GR_LOOK 2                         // layers without aggregation: the planes of tile RT + GR_LOOK are produced while tile RT is multiplied; 2 * GR_LOOK
#endif                                    // plane buffers.  GR_LOOK == GR_GRP: one workgroup barrier per group of tiles (4 buffers); GR_LOOK == 2 * GR_GRP: the four
                                          // waves synchronise through one LDS counter per group with a whole group of slack, no barrier inside a layer —
                                          // correct (same tests), measured SLOWER: 144 vs 135 us per launch (the four waves run in step anyway, and the
                                          // deeper look-ahead costs 20 more spilled registers); kept as a build option
#define GR_ROWB 272                       // plane row pitch in bytes (256 + 16: conflict-free 16-byte operand reads)
#define GR_PLANE (32 * GR_ROWB)
#define GR_TILE (2 * GR_PLANE)                // a tile buffer: the (high | low) f16 planes of 32 rows
#define GR_MAXCAND 384
#define GR_MAXWARM 10
#define GR_MAXT 65                        // rows per instance: the in-edge sources of a tile lie within two tiles of it
#define GR_MINT 16                        // ... and a 16-row run spans at most two instances (pooling)
#define GR_MAXIPC 64                      // instances per workgroup (u8 instance ids; pool accumulators in the ring area)

#ifndef GR_MIX
#define GR_MIX 2                          // 1: the low operand piece as fma(f16 high piece, -1, x) = v_fma_mix_f32 (no separate f16 -> f32 conversion); same bits
#endif                                    // 2: ... and rounded to f16 by the same instruction (v_fma_mixlo_f16 / v_fma_mixhi_f16: the remainder is exact in f32,
                                          //    so this is the one rounding the separate conversion performed): a packed conversion less per value pair; same bits
#ifndef GR_SETTLE
#define GR_SETTLE 1                       // two extra wait states between a tile's last matrix instruction and the first vector read of its result
#endif
#ifndef GR_FOLD_DPP
#define GR_FOLD_DPP 1                     // 1: the per-lane column sums of a layer are folded over the 32 row lanes in registers (DPP reduce-scatter); 0: through LDS
#endif
#ifndef GR_STAT_ASM
#define GR_STAT_ASM 0                     // (see stat2)
#endif
#ifndef GR_PK
#define GR_PK 1                           // 1: two-wide f32 vector arithmetic (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32); 0: one instruction per element
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 gr_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 gr_h2 __attribute__((ext_vector_type(2)));

struct GinResArgs {
    int B, T, J, ipc;                     // instances, rows per instance, candidates per instance, instances per workgroup
    const void *tfea; int feat_f64;       // [B*T,12] raw task features
    const int *ell_col; const float *ell_val;   // [B*T,2]
    const void *Wx32[6];                  // register images of the six Linear weights for the 32x32x16 form (first: the 12 -> 128 one-k-step bf16 image;
                                          // the others: two f16 planes of W * wscale)
    float wsinv[6];                       // 1 / wscale of each image (a power of two; 1 for the first)
    const float *gamma[6], *beta[6];      // BatchNorm after each Linear (mlps.0.bn0, mlps.0.bn1, outer 0, mlps.1.bn0, mlps.1.bn1, outer 1)
    unsigned long long *stats;            // this forward's accumulators (zero on entry): [6 layers][8 groups][128 columns][sum | sumsq], see gr_fix_encode
    unsigned long long *stats_next;       // the set of the next forward: zeroed here
    int wexp[6];                          // log2 of each weight image's scale (wsinv = 2^-wexp)
    int ffrac[6];                         // fractional bits of each layer's fixed-point statistics (gr_fix_encode): 20, and 12 for the first Linear, whose
                                          // outputs inherit the range of the raw features (start / finish times in the thousands)
    unsigned long long *bar;              // barrier words (monotonic counters, never reset)
    unsigned long long epoch;             // launches on `bar` so far
    unsigned *fail;                       // set when a barrier timed out (the outputs are then garbage)
    unsigned *range_flag;                 // set when a layer's BatchNorm sums are not numbers (an operand beyond the f16 range upstream)
    const int *candidate;                 // [B,J] or NULL
    float *pooled;                        // [B,128] graph mean pool of h (gcn:192)
    float *cand_feat;                     // [B*J,128] h rows of the candidates (ac:197-207)
    float *h_nodes;                       // optional [B*T,128]
    float *zspill;                        // [blocks][4 waves][2 tiles][4][64 lanes][4] f32: the pre-BatchNorm values of row tiles 16, 17
    double inv_rows;                      // 1 / (B*T)
    int barrier_only;                     // census launch: barriers only
    unsigned expect_extra;                // diagnostic (MTFJSP_GIN_RES_FAIL_AT): the barriers wait for this many workgroups that do not exist -> time-out path
    unsigned long long *stamps;           // diagnostic build only (-DGR_STAMP): [blocks][64] s_memrealtime at the phase boundaries
    const void *warm[GR_MAXWARM];         // buffers the NEXT launch reads first: one word of every 128-byte line is requested in the last phase, so that
    unsigned warm_lines[GR_MAXWARM];      // the lines wait in this XCD's L2 (workgroups blockIdx % 8 share an XCD; each requests its share)
    int nwarm;
};
#define GR_STATS_PART (6 * 8 * HD * 2)                            // 64-bit words: per layer, per dispatch group: (sum, sumsq) per column
#define GR_STATS_SET GR_STATS_PART
#ifdef GR_STAMP
// slot i: s_memrealtime (100 MHz); slot 33 + i (i <= 30): s_memtime (shader clock) at the same point — the clock the chip holds inside the
// kernel is d(s_memtime) / d(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS item 6; round-5 review item 2)
#define GR_STAMP_AT(i) do { if (tid == 0 && A.stamps) { A.stamps[(size_t)blockIdx.x * 64 + (i)] = __builtin_amdgcn_s_memrealtime(); \
                                                         if ((i) <= 30) A.stamps[(size_t)blockIdx.x * 64 + 33 + (i)] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define GR_STAMP_AT(i) do { } while (0)
#endif

// LDS map (bytes)
#define GR_OFF_PLANES 0                                           // [2][2 planes][32 rows][272] (first Linear: 3 bf16 planes of 8 tiles side by side)
#define GR_OFF_RING (2 * GR_TILE)                                 // f32 [6 tiles * 32 rows][128], swizzled (first phase: features [576][12])
#define GR_RING_BYTES ((2 * GR_LOOK - 2) * GR_TILE > GR_RING * 32 * HD * 4 ? (2 * GR_LOOK - 2) * GR_TILE : GR_RING * 32 * HD * 4)   // (also plane buffers 2 .. 2 * GR_LOOK - 1)
#define GR_OFF_ELLC (GR_OFF_RING + GR_RING_BYTES)                 // u32 [576]: workgroup-relative rows of the <= 2 in-edges, 0xffff = none
#define GR_OFF_ELLV0 (GR_OFF_ELLC + GR_ROWS * 4)
#define GR_OFF_ELLV1 (GR_OFF_ELLV0 + GR_ROWS * 4)
#define GR_OFF_BN (GR_OFF_ELLV1 + GR_ROWS * 4)                    // f32 [256] scale | shift
#define GR_OFF_CAND (GR_OFF_BN + 2 * HD * 4)                      // i32 [576]: candidate slot of a row (instance-local slot + J * local instance), -1 = none
#define GR_OFF_ZERO (GR_OFF_CAND + GR_ROWS * 4)                   // f32 [256] zeros: the "scale | shift" of rows >= nrows
#define GR_OFF_FLAG (GR_OFF_ZERO + 2 * HD * 4)
#define GR_OFF_CNT (GR_OFF_FLAG + 64)                             // u32 [5 layers][GR_NT / GR_GRP]: waves that finished a group of tiles (GR_LOOK > GR_GRP)
#define GR_LDS_BYTES (GR_OFF_CNT + 5 * (GR_NT / GR_GRP) * 4 + 12)
static size_t gin_res_lds_bytes() { return (size_t)GR_LDS_BYTES; }
static_assert(GR_LDS_BYTES <= 160 * 1024, "resident GIN kernel: LDS budget");

#if GR_PK == 1
__device__ __forceinline__ f32x2 gr_fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 gr_add2(f32x2 a, f32x2 b) { return a + b; }
__device__ __forceinline__ f32x2 gr_sub2(f32x2 a, f32x2 b) { return a - b; }
__device__ __forceinline__ f32x2 gr_mul2(f32x2 a, f32x2 b) { return a * b; }
#else
// (element by element, each result pinned so that no later pass pairs them up again)
#if GR_PK == 2
__device__ __forceinline__ float gr_pin(float x) { return x; }     // (with -fno-slp-vectorize)
#else
__device__ __forceinline__ float gr_pin(float x) { asm("" : "+v"(x)); return x; }
#endif
__device__ __forceinline__ f32x2 gr_fma2(f32x2 a, f32x2 b, f32x2 c) { return f32x2{gr_pin(__builtin_fmaf(a[0], b[0], c[0])), gr_pin(__builtin_fmaf(a[1], b[1], c[1]))}; }
__device__ __forceinline__ f32x2 gr_add2(f32x2 a, f32x2 b) { return f32x2{gr_pin(a[0] + b[0]), gr_pin(a[1] + b[1])}; }
__device__ __forceinline__ f32x2 gr_sub2(f32x2 a, f32x2 b) { return f32x2{gr_pin(a[0] - b[0]), gr_pin(a[1] - b[1])}; }
__device__ __forceinline__ f32x2 gr_mul2(f32x2 a, f32x2 b) { return f32x2{gr_pin(a[0] * b[0]), gr_pin(a[1] * b[1])}; }
#endif
// x - (float)high piece, elementwise (exactly rounded either way: the product by -1 is exact)
__device__ __forceinline__ f32x2 gr_rem2(f32x2 v, gr_h2 p)
{
#if GR_MIX
    // (written as instructions: the optimiser turns fma(x, -1, v) back into a subtraction behind a conversion)
    float r0, r1;
    const unsigned pp = __builtin_bit_cast(unsigned, p);
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(pp), "v"(v[0]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(pp), "v"(v[1]));
    return f32x2{r0, r1};
#else
    return gr_sub2(v, __builtin_convertvector(p, f32x2));
#endif
}
// f16(x - (float)high piece), both elements, packed
__device__ __forceinline__ gr_h2 gr_rem16(f32x2 v, gr_h2 p)
{
    unsigned r;
    const unsigned pp = __builtin_bit_cast(unsigned, p);
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pp), "v"(v[0]));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(pp), "v"(v[1]));
    return __builtin_bit_cast(gr_h2, r);
}
template <typename F, int... I>
__device__ __forceinline__ void gr_static_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void gr_static_for(F &&f) { gr_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// sum over the 32 lanes of a half wave, result in every lane (row_sum16 + the other 16-lane row)
__device__ __forceinline__ float gr_sum32(float x)
{
    x = row_sum16(x);
    return x + __shfl_xor(x, 16);
}

// ---- BatchNorm column sums across the grid WITHOUT a separate barrier: every 64-bit accumulator word carries its own arrival count.
// A workgroup's f32 column sum x (of the true z: the power-of-two scale of the weight image is divided out of the exponent) goes
// out as ONE integer atomic add of (1 << 58) | (round(x * 2^frac) + 2^51), frac = 20: after n contributions a word holds n
// in its top 6 bits and, below, the exact sum of the fixed-point values (n <= 63 workgroups per dispatch group, |x| < 2^31).  A
// reader that finds the expected count in a word HAS that word's complete sum: no data atomics to wait for before signalling, no
// counter atomics, no release flag, no second read.  Integer sums are exact and order-independent: the statistics (and with them
// the whole forward) are bit-reproducible run to run, which f64 atomics are not.  Resolution: 2^-20 per contribution, i.e. an
// absolute error below 256 * 2^-20 / rows = 1.7e-9 on a mean or a mean square over the 147 456 rows of the headline batch — four
// orders of magnitude under the BatchNorm epsilon (1e-5) the variance is added to.  Range: a workgroup's sum of squares must stay
// below 2^31 (|z| < 1900 rms over its 576 rows); beyond it — or for a NaN — the contribution is flagged (range word) and the host
// repeats the forward on the streaming f32-instruction kernels, exactly as for an operand beyond the f16 range.  The first Linear's
// outputs inherit the range of the raw features (times in the thousands): its words carry 12 fractional bits (range 2^39).
#define GR_FIX_FRAC_DEFAULT 20
#define GR_FIX_FRAC_FIRST 12
#define GR_FIX_BIAS (1ull << 51)
#define GR_FIX_ONE (1ull << 58)
#define GR_FIX_PAYLOAD (GR_FIX_ONE - 1ull)
__device__ __forceinline__ unsigned long long gr_fix_encode(float x, int exp2, int frac)
{
    // integer arithmetic only (an f64 form — ldexp, floor, conversion to a 64-bit integer — took 0.8 us of every boundary):
    // x = (-1)^s * mant * 2^(e - 150)  ->  V = (-1)^s * (mant << sh), sh = e - 150 + FRAC - exp2 <= 27 by the caller's range check
    // (mant < 2^24: |V| < 2^51); for sh < 0 the magnitude is rounded to the nearest multiple of 2^-frac
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const int e = (int)((u >> 23) & 255u);
    const unsigned long long mant = (unsigned long long)((u & 0x7fffffu) | (e ? 0x800000u : 0u));
    const int sh = (e ? e : 1) - 150 + frac - exp2;
    long long v = sh >= 0 ? (long long)(mant << (sh > 39 ? 39 : sh)) : (sh > -25 ? (long long)((mant + (1ull << (-sh - 1))) >> (-sh)) : 0ll);   // (round half up)
    if (u >> 31) v = -v;
    return GR_FIX_ONE | (unsigned long long)(v + (long long)GR_FIX_BIAS);
}
// total of `n` contributions from the summed payloads
__device__ __forceinline__ double gr_fix_decode(unsigned long long sum, unsigned n, int frac)
{
    const long long v = (long long)sum - (long long)n * (long long)GR_FIX_BIAS;
    return __builtin_ldexp((double)v, -frac);
}

// Grid-wide barrier over `nblk` co-resident workgroups that also completes a reduction: hierarchical over the 8 dispatch
// groups blockIdx % 8 (observed to share an XCD; a different placement changes speed only).  Monotonic counters: generation
// `gen` (1, 2, ...) is complete for a group when its counter reaches gen * group size.  Before the call every thread has
// added its (sum | sumsq) word to its group's accumulators with a device-scope atomic and waited for it (s_waitcnt vmcnt(0)).
// The last workgroup of a group to arrive reports the group; the last group releases everybody.  On return
// (workgroup-wide) the 8 groups' accumulators are complete and may be read with sc1 loads — they are written by memory-side
// atomics only and no line of them has been touched by this launch before — and are added up by every workgroup itself, in
// a fixed order: two round trips less on the critical path than folding them into one total behind a second set of atomics.
__device__ __forceinline__ void gr_grid_barrier(unsigned long long *bar, unsigned long long gen, unsigned nblk, unsigned *fail, unsigned *s_flag)
{
    const int tid = threadIdx.x;
    const unsigned g = blockIdx.x & 7u;
    const unsigned ngroups = nblk < 8u ? nblk : 8u;
    (void)s_flag;
    __syncthreads();                                              // every thread's data atomics have completed
    if (tid == 0) {
        // (two levels: a single counter for all 256 workgroups was measured at 3.8 us per barrier against 2.1 us)
        const unsigned long long gsize = (nblk >> 3) + (g < (nblk & 7u) ? 1u : 0u);
        const unsigned long long old = __hip_atomic_fetch_add(&bar[16 * g], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == gen * gsize) {                             // this workgroup completes its group
            const unsigned long long old2 = __hip_atomic_fetch_add(&bar[16 * 8], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old2 + 1 == gen * ngroups)
                for (unsigned j = 0; j < ngroups; j++) __hip_atomic_store(&bar[16 * (9 + j)], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(&bar[16 * (9 + g)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) {
            __builtin_amdgcn_s_sleep(1);
            if (__builtin_amdgcn_s_memrealtime() - t0 > 400000ull) {                           // 4 ms at 100 MHz: not all workgroups are resident
                __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);       // (host-mapped word: the host polls it without synchronising)
                break;
            }
        }
    }
    __syncthreads();
}

#ifdef MTFJSP_GIN_RES_DECL_ONLY
__global__ __launch_bounds__(256) void k_gin_res(GinResArgs A);
#else
__global__ __launch_bounds__(256) void k_gin_res(GinResArgs A)
{
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char *s_planes = smem + GR_OFF_PLANES;
    float *s_ring = reinterpret_cast<float *>(smem + GR_OFF_RING);
    unsigned *s_ellc = reinterpret_cast<unsigned *>(smem + GR_OFF_ELLC);
    float *s_ellv0 = reinterpret_cast<float *>(smem + GR_OFF_ELLV0);
    float *s_ellv1 = reinterpret_cast<float *>(smem + GR_OFF_ELLV1);
    float *s_bn = reinterpret_cast<float *>(smem + GR_OFF_BN);
    int *s_rowcand = reinterpret_cast<int *>(smem + GR_OFF_CAND);
    float *s_zero = reinterpret_cast<float *>(smem + GR_OFF_ZERO);
    unsigned *s_flag = reinterpret_cast<unsigned *>(smem + GR_OFF_FLAG);
    unsigned *s_cnt = reinterpret_cast<unsigned *>(smem + GR_OFF_CNT);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int n = lane & 31, h = lane >> 5;
    const unsigned nblk = gridDim.x + A.expect_extra;
    const unsigned long long gen0 = A.epoch * GR_NBAR;
    if (A.barrier_only) {
        for (int k = 0; k < GR_NBAR; k++) gr_grid_barrier(A.bar, gen0 + k + 1, nblk, A.fail, s_flag);
        return;
    }
    const int T = A.T;
    const int inst0 = blockIdx.x * A.ipc;
    const int ninst = A.B - inst0 < A.ipc ? A.B - inst0 : A.ipc;
    const int nrows = ninst * T;
    const size_t grow0 = (size_t)inst0 * T;                       // first global row of this workgroup
    // (every workgroup computes all GR_NT tiles: rows >= nrows are zero inputs, masked out of the BatchNorm sums and never
    // written — no tile-count branches, which would split the fully unrolled body into blocks the register allocator has to
    // reconcile through scratch memory)

    // ---------------------------------------------------------------- prologue: features + adjacency to LDS, zero the planes
    GR_STAMP_AT(32);
    f32x4 w0pre[3];                                               // the first Linear's weight image (requested in the prologue)
    for (int i = blockIdx.x * 256 + tid; i < GR_STATS_SET; i += (int)gridDim.x * 256) A.stats_next[i] = 0ull;
    {
        // (the candidate indices are requested with everything else: their use below would otherwise be a second memory round trip)
        int cand_pre[(GR_MAXCAND + 255) / 256];
#pragma unroll
        for (int k = 0; k < (GR_MAXCAND + 255) / 256; k++) {
            const int i = tid + 256 * k;
            cand_pre[k] = (A.candidate && i < ninst * A.J) ? A.candidate[(size_t)inst0 * A.J + i] : -1;
        }
        // Every request of the prologue goes out before the first of them is waited for (round 5: as `for (i = tid; i < n; i += 256)` loops
        // with a run-time bound the feature and adjacency copies were compiled into load -> s_waitcnt vmcnt(0) -> LDS store per iteration:
        // ten dependent round trips, ~2.5 of the prologue's 4.1 us).  Indices are clamped into the workgroup's rows, not branched around.
        float *s_feat = s_ring;                                   // [nrows][12]
        constexpr int NF4 = (GR_ROWS * 3 + 255) / 256, NER = (GR_ROWS + 255) / 256;
        f32x4 fpre[NF4];
        {   // (unconditional; f64 features: the same bytes lie inside the buffer and are overwritten below)
            const f32x4 *src = reinterpret_cast<const f32x4 *>(reinterpret_cast<const float *>(A.tfea) + grow0 * 12);
            const int last = nrows * 3 - 1;
#pragma unroll
            for (int k = 0; k < NF4; k++) { const int i = tid + 256 * k; fpre[k] = src[i < last ? i : last]; }
        }
        int2 ecc[NER]; float2 evv[NER];
#pragma unroll
        for (int k = 0; k < NER; k++) {
            const int r = tid + 256 * k, rc = r < nrows ? r : nrows - 1;
            ecc[k] = *reinterpret_cast<const int2 *>(A.ell_col + (grow0 + rc) * 2);
            evv[k] = *reinterpret_cast<const float2 *>(A.ell_val + (grow0 + rc) * 2);
        }
        // (stored unconditionally too: a store behind `if (i < nrows * 3)` or behind `if (!A.feat_f64)` pulls its request down to itself or sends
        // the seven registers through scratch memory; words beyond the workgroup's rows are copies nobody uses)
#pragma unroll
        for (int k = 0; k < NF4; k++) reinterpret_cast<f32x4 *>(s_feat)[tid + 256 * k] = fpre[k];
        if (A.feat_f64) {                                         // f64 observations: converted here, over what the lines above left
            const double *src = reinterpret_cast<const double *>(A.tfea) + grow0 * 12;
            const int n12 = nrows * 12;
            __syncthreads();
#pragma unroll 1
            for (int i0 = 0; i0 < n12; i0 += 256 * 9) {            // 9 requests in flight per thread
                double d[9];
#pragma unroll
                for (int k = 0; k < 9; k++) { const int i = i0 + tid + 256 * k; d[k] = src[i < n12 ? i : n12 - 1]; }
#pragma unroll
                for (int k = 0; k < 9; k++) { const int i = i0 + tid + 256 * k; if (i < n12) s_feat[i] = (float)d[k]; }
            }
        }
#pragma unroll
        for (int k = 0; k < NER; k++) {
            const int r = tid + 256 * k;
            if (r < GR_ROWS) {
                unsigned cp = 0xffffffffu; float v0 = 0.f, v1 = 0.f;
                if (r < nrows) {
                    const int2 cc = ecc[k];
                    const float2 vv = evv[k];
                    const int base = (r / T) * T;
                    const unsigned r0 = cc.x >= 0 ? (unsigned)(base + cc.x) : 0xffffu, r1 = cc.y >= 0 ? (unsigned)(base + cc.y) : 0xffffu;
                    cp = r0 | (r1 << 16); v0 = cc.x >= 0 ? vv.x : 0.f; v1 = cc.y >= 0 ? vv.y : 0.f;
                }
                s_ellc[r] = cp; s_ellv0[r] = v0; s_ellv1[r] = v1; s_rowcand[r] = -1;
            }
        }
        {   // the first Linear's weight image: last read a rollout step ago (memory-side cache by now), needed ~2 us from here.  Behind the
            // prologue's other requests and their waits: loads return in order, and those (written by the environment step a moment ago) come from L2
            const f32x4 *wi = reinterpret_cast<const f32x4 *>(A.Wx32[0]) + (size_t)wave * (3 * 64) + lane;
            __builtin_amdgcn_sched_barrier(0);                    // (hipcc would slip them in front of the last adjacency request)
#pragma unroll
            for (int p = 0; p < 3; p++) w0pre[p] = wi[p * 64];
            __builtin_amdgcn_sched_barrier(0);
        }
        if (tid == 0) s_flag[1] = 0u;
        if (tid < 5 * (GR_NT / GR_GRP)) s_cnt[tid] = 0u;
        for (int i = tid; i < 2 * GR_TILE / 16; i += 256) reinterpret_cast<float4 *>(s_planes)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (A.candidate) {                                        // row -> candidate slot (ac:197-207 gathers h of one row per job)
            LDS_BARRIER();                                        // (s_rowcand is LDS; __syncthreads() would also wait for the weight request above)
#pragma unroll
            for (int k = 0; k < (GR_MAXCAND + 255) / 256; k++) {
                const int i = tid + 256 * k, c = cand_pre[k];
                if (i < ninst * A.J && c >= 0 && c < T && atomicExch(&s_rowcand[(i / A.J) * T + c], i) != -1) s_flag[1] = 1u;   // two slots on one row: fixed up at the end
            }
        }
        s_zero[tid] = 0.f;
    }
    // 18 tiles x 16 accumulators = 288 values per lane, but only 256 accumulation registers exist and the matrix instructions
    // need one 16-register tuple of them to work in: left to itself hipcc keeps two values per tile in scratch memory and the
    // reloads stall the wave.  So tiles 0..15 are resident and tiles 16, 17 make an explicit round trip per layer through 32 KB
    // of global memory per workgroup (L2-resident, each wave its own 8 KB, 16-byte coalesced, requested a tile ahead of use).
    f32x16 acc[GR_NRES];
    f32x16 zs[2];                                                 // ring: spilled tile j uses zs[j & 1]
    float *zsp = A.zspill + ((size_t)(blockIdx.x * 4 + wave) * (GR_NT - GR_NRES)) * 1024 + lane * 4;
    auto zload = [&](auto Jc) __attribute__((always_inline)) {
        constexpr int j = decltype(Jc)::value;
        if constexpr (GR_VRES) return;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const float4 v = *reinterpret_cast<const float4 *>(zsp + j * 1024 + g * 256);
            zs[j & 1][4 * g] = v.x; zs[j & 1][4 * g + 1] = v.y; zs[j & 1][4 * g + 2] = v.z; zs[j & 1][4 * g + 3] = v.w;
        }
    };
    auto zstore = [&](auto Jc, const f32x16 &a) __attribute__((always_inline)) {
        constexpr int j = decltype(Jc)::value;
        if constexpr (GR_VRES) return;
#pragma unroll
        for (int g = 0; g < 4; g++) *reinterpret_cast<float4 *>(zsp + j * 1024 + g * 256) = make_float4(a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]);
    };
#define GR_TILEVAL(rt) ((rt) < GR_NRES ? acc[(rt) < GR_NRES ? (rt) : 0] : zs[(rt) >= GR_NRES ? ((rt) - GR_NRES) & 1 : 0])
    // v_mfma with its accumulator in VECTOR registers, written out: the compiler's intrinsic would allocate it in the accumulation
    // file (full) and move it back and forth.  ZERO: the first product of a tile (C = 0).
    auto mfma_v = [&](f32x16 &c, const gr_h8 &a, const gr_h8 &b, bool zero) __attribute__((always_inline)) {
        if (zero) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(c) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    };
    // The per-tile "is this lane's row valid" masks and the scale / shift addresses selected by them are the same expressions in every
    // layer: left alone, hipcc forms all 18 tiles' once and keeps them for five layers — through scratch memory (128 reloads, 47
    // stores per wave) and SGPR spill lanes (~570 v_readlane / v_writelane).  nrows_l is the same number made opaque at the top of each
    // layer (and so is the lane's row index n_l), so each layer forms its own (an or, a compare, two selects and two adds per tile).
    int nrows_l = nrows, n_l = n;
    bool timed_out = false;                                       // a statistics wait of this launch gave up: everything after it is garbage
    gr_h8 wf[2][8];
    bf16x8 w0f[3];                                                // (first Linear only)
    f32x2 ts[8], tq[8];                                           // per-lane column (sum, sumsq) of this layer, two columns per register pair
#pragma unroll
    for (int i = 0; i < 8; i++) { ts[i] = f32x2{0.f, 0.f}; tq[i] = f32x2{0.f, 0.f}; }

    // Weight fragments of a 128 -> 128 Linear from their register image: 2 planes x 8 k-steps x 4 registers = 64 registers.
    auto load_weights = [&](int layer) __attribute__((always_inline)) {
        const float4 *wi = reinterpret_cast<const float4 *>(A.Wx32[layer]) + (size_t)wave * (2 * 8 * 64) + lane;
#pragma unroll
        for (int p = 0; p < 2; p++)
#pragma unroll
            for (int ks = 0; ks < 8; ks++) wf[p][ks] = __builtin_bit_cast(gr_h8, wi[(p * 8 + ks) * 64]);
    };
    // the two statistics updates of a column pair, pinned where they are written (hipcc would sink a whole layer's sums to their use)
    auto stat2 = [&](int i, float x, float y) __attribute__((always_inline)) {
        const f32x2 v = {x, y};
#if GR_STAT_ASM == 1                        // the two updates written out as packed instructions (hipcc splits about half of them into four scalar ones):
        asm volatile("v_pk_add_f32 %0, %2, %0\n\tv_pk_fma_f32 %1, %2, %2, %1" : "+v"(ts[i]), "+v"(tq[i]) : "v"(v));      // measured SLOWER, 110.7 against 108.6 us per launch (A/B, one box)
#elif GR_STAT_ASM == 2                      // ... all four as scalar instructions: 137.9 us (the pairs are taken apart and put together again)
        float s0 = ts[i][0], s1 = ts[i][1], q0 = tq[i][0], q1 = tq[i][1];
        asm volatile("v_add_f32 %0, %4, %0\n\tv_add_f32 %1, %5, %1\n\tv_fma_f32 %2, %4, %4, %2\n\tv_fma_f32 %3, %5, %5, %3" : "+v"(s0), "+v"(s1), "+v"(q0), "+v"(q1) : "v"(x), "v"(y));
        ts[i] = f32x2{s0, s1}; tq[i] = f32x2{q0, q1};
#else
        ts[i] = gr_add2(ts[i], v); tq[i] = gr_fma2(v, v, tq[i]);
        asm volatile("" : "+v"(ts[i]), "+v"(tq[i]));
#endif
    };
    auto stats_all = [&](const f32x16 &a) __attribute__((always_inline)) {
        if (GR_ABL & 4) return;
#pragma unroll
        for (int i = 0; i < 8; i++) stat2(i, a[2 * i], a[2 * i + 1]);
    };
    // exact-to-2^-22 operand split of 4 values: high = f16(v), low = f16(v - high) (both round-to-nearest) -> packed pairs
    auto split2x4 = [&](f32x2 v01, f32x2 v23, uint2 &p0, uint2 &p1) __attribute__((always_inline)) {
        const gr_h2 a = __builtin_convertvector(v01, gr_h2), b = __builtin_convertvector(v23, gr_h2);
#if GR_MIX == 2
        const gr_h2 c = gr_rem16(v01, a), d = gr_rem16(v23, b);
#else
        const f32x2 r01 = gr_rem2(v01, a), r23 = gr_rem2(v23, b);
        const gr_h2 c = __builtin_convertvector(r01, gr_h2), d = __builtin_convertvector(r23, gr_h2);
#endif
        p0 = make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b));
        p1 = make_uint2(__builtin_bit_cast(unsigned, c), __builtin_bit_cast(unsigned, d));
    };
    // BatchNorm statistics.  stats_tile: this lane's row of tile RT into the per-lane column sums (one value per tile and lane:
    // f32 is ample) — called one tile late, so that it never waits for the matrix pipe.  Rows >= nrows need no mask: their
    // operand planes are written as zeros (scale = shift = 0 below), and without a bias a zero row stays exactly zero.
    auto stats_tile = [&](auto Tc) __attribute__((always_inline)) {
        constexpr int RT = decltype(Tc)::value;
        stats_all(GR_TILEVAL(RT));
    };
    // Once per layer: the per-lane sums -> this workgroup's (sum | sumsq) of column tid >> 1, one value per thread.  Every lane
    // parks its 32 values in LDS (ring area, free at the boundaries; 144-byte lane pitch: conflict-free 16-byte writes) and
    // thread (column, kind) adds up the 32 row-lanes of its column — 8 writes + 32 reads + 31 adds per thread instead of 32
    // cross-lane reductions.
    auto fold_stats = [&]() __attribute__((always_inline)) {
        float *red = s_ring;
        float *mine = red + (wave * 64 + lane) * 36;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            *reinterpret_cast<float4 *>(mine + 4 * j) = make_float4(ts[2 * j][0], ts[2 * j][1], ts[2 * j + 1][0], ts[2 * j + 1][1]);
            *reinterpret_cast<float4 *>(mine + 16 + 4 * j) = make_float4(tq[2 * j][0], tq[2 * j][1], tq[2 * j + 1][0], tq[2 * j + 1][1]);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) { ts[i] = f32x2{0.f, 0.f}; tq[i] = f32x2{0.f, 0.f}; }
        LDS_BARRIER();
        const int col = tid >> 1, kind = tid & 1, cw = col & 31;
        const float *src = red + ((col >> 5) * 64 + 32 * ((cw >> 2) & 1)) * 36 + 4 * (cw >> 3) + (cw & 3) + 16 * kind;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int i = 0; i < 32; i += 4) { s0 += src[i * 36]; s1 += src[(i + 1) * 36]; s2 += src[(i + 2) * 36]; s3 += src[(i + 3) * 36]; }
        return (s0 + s1) + (s2 + s3);
    };
    // The same fold without LDS and without a workgroup barrier (round 4; GR_FOLD_DPP): a reduce-scatter over the 32 row lanes of a
    // half wave in five halving steps — partner = lane ^ 16 (ds_bpermute), ^ 15 (row mirror), ^ 7 (half-row mirror), ^ 2, ^ 1 (quad
    // permutes): at each step a lane keeps the half of its values selected by one of its own lane bits (bit 4, 3, 2, 1, 0), hands the
    // other half to its partner and adds what it receives, the exchange riding on the add as a DPP operand.  {16, 15, 7, 2, 1} span all
    // 32 lanes, so after the fifth step lane n holds the total of value n: kind = n >> 4 (sum | sumsq), column slot c = n & 15, i.e.
    // column 32 wave + 8 (c >> 2) + 4 h + (c & 3).  ~110 vector instructions instead of 8 stores + barrier + 32 loads + 31 adds.
    auto fold_stats_dpp = [&]() __attribute__((always_inline)) {
        const bool b4 = (lane & 16) != 0, b3 = (lane & 8) != 0, b2 = (lane & 4) != 0, b1 = (lane & 2) != 0, b0 = (lane & 1) != 0;
        auto dpp = [&](float x, auto Ctrl) __attribute__((always_inline)) {
            return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(Ctrl)::value, 0xF, 0xF, true));
        };
        float v16[16];
#pragma unroll
        for (int c = 0; c < 16; c++) {                                // bit 4: lanes 0..15 keep the sums, lanes 16..31 the sums of squares
            const float a = ts[c >> 1][c & 1], q_ = tq[c >> 1][c & 1];
            v16[c] = (b4 ? q_ : a) + __shfl_xor(b4 ? a : q_, 16);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) { ts[i] = f32x2{0.f, 0.f}; tq[i] = f32x2{0.f, 0.f}; }
        float v8[8];
#pragma unroll
        for (int c = 0; c < 8; c++) v8[c] = (b3 ? v16[c + 8] : v16[c]) + dpp(b3 ? v16[c] : v16[c + 8], std::integral_constant<int, 0x140>{});    // row mirror: lane ^ 15
        float v4[4];
#pragma unroll
        for (int c = 0; c < 4; c++) v4[c] = (b2 ? v8[c + 4] : v8[c]) + dpp(b2 ? v8[c] : v8[c + 4], std::integral_constant<int, 0x141>{});        // half-row mirror: lane ^ 7
        float v2[2];
#pragma unroll
        for (int c = 0; c < 2; c++) v2[c] = (b1 ? v4[c + 2] : v4[c]) + dpp(b1 ? v4[c] : v4[c + 2], std::integral_constant<int, 0x4E>{});         // quad_perm [2,3,0,1]: lane ^ 2
        return (b0 ? v2[1] : v2[0]) + dpp(b0 ? v2[0] : v2[1], std::integral_constant<int, 0xB1>{});                                               // quad_perm [1,0,3,2]: lane ^ 1
    };
    // BatchNorm + ReLU of columns 8g+4h..+3 of tile rt's resident values -> operand split -> planes of buffer buf
    // Four plane buffers in the layers without aggregation (two of them in the ring area, free there): the tiles go in pairs —
    // tiles 2p+2, 2p+3 are produced while tiles 2p, 2p+1 are multiplied — with ONE workgroup barrier per pair: half the
    // barriers (each costs the wait for the slowest of the four waves, ~450 cycles measured).
    unsigned char *const pl0 = s_planes, *const pl1 = s_planes + GR_TILE, *const plr = reinterpret_cast<unsigned char *>(s_ring);
    auto plane_buf = [&](int i) __attribute__((always_inline)) { return i == 0 ? pl0 : i == 1 ? pl1 : plr + (i - 2) * GR_TILE; };
    auto produce_quarter = [&](auto Tc, auto Gc, unsigned char *pbuf) __attribute__((always_inline)) {
        constexpr int rt = decltype(Tc)::value, g = decltype(Gc)::value;
        unsigned char *dst = pbuf + n * GR_ROWB + (32 * wave + 8 * g + 4 * h) * 2;
        const float *bn = rt * 32 + n_l < nrows_l ? s_bn : s_zero;  // rows >= nrows: scale = shift = 0 -> zero planes
        const float4 s4 = *reinterpret_cast<const float4 *>(bn + 32 * wave + 8 * g + 4 * h);        // scale | shift of these 4 columns
        const float4 h4 = *reinterpret_cast<const float4 *>(bn + HD + 32 * wave + 8 * g + 4 * h);
        const f32x16 &a = GR_TILEVAL(rt);
        uint2 p0, p1;
        split2x4(f32x2{bn_relu_ss(a[4 * g], s4.x, h4.x), bn_relu_ss(a[4 * g + 1], s4.y, h4.y)},
                 f32x2{bn_relu_ss(a[4 * g + 2], s4.z, h4.z), bn_relu_ss(a[4 * g + 3], s4.w, h4.w)}, p0, p1);
        *reinterpret_cast<uint2 *>(dst) = p0;
        *reinterpret_cast<uint2 *>(dst + GR_PLANE) = p1;
    };
    auto produce_tile = [&](auto Tc, unsigned char *pbuf) __attribute__((always_inline)) {
        gr_static_for<4>([&](auto Gc) __attribute__((always_inline)) { produce_quarter(Tc, Gc, pbuf); });
        __builtin_amdgcn_sched_barrier(0);
    };
    // multiply tile RT (planes in buffer `buf`) by the resident weight fragments -> acc[RT]: per k-step the three significant
    // products of the (high | low) operand pieces, smallest first: w_hi x_lo, w_lo x_hi, w_hi x_hi.  One wave per SIMD is
    // in-order: a vector instruction placed behind a run of matrix instructions waits for all of them to issue, and the matrix
    // pipe then idles behind a run of vector instructions.  hipcc emits each kind as one clump (and sched_group_barrier did not
    // change that here), so the stream is hand-placed: 24 slices of ONE matrix instruction + 4..6 vector / LDS instructions, each
    // closed by a scheduling barrier.  The vector work riding along is a quarter (per 6 slices) of the NEXT tile's plane
    // production (BatchNorm + ReLU + split of its resident values) and of the PREVIOUS tile's BatchNorm sums, plus the operand
    // reads of the following k-steps.
    const unsigned char *xa0 = s_planes + n * GR_ROWB + 16 * h;
#define GR_FENCE() __builtin_amdgcn_sched_barrier(0)
#define GR_PIN_V() do { } while (0)
    auto consume_tile = [&](auto Tc) __attribute__((always_inline)) {
        constexpr int RT = decltype(Tc)::value;
        constexpr int PT = RT + GR_LOOK;                                  // the tile whose planes are produced here
        constexpr bool NEXT = (PT < GR_NT) && !(GR_ABL & 1);
        constexpr bool STATS = RT > 0 && (GR_VRES || RT - 1 < GR_NRES) && !(GR_ABL & 4);     // (a spilled tile's sums are taken when it is stored)
        const unsigned char *xa = plane_buf(RT % (2 * GR_LOOK)) + n * GR_ROWB + 16 * h;
        unsigned char *pnext = plane_buf(PT % (2 * GR_LOOK));
        const float *bn = PT * 32 + n_l < nrows_l ? s_bn : s_zero;          // rows >= nrows: scale = shift = 0 -> zero planes
        gr_h8 xf[2][2];
#pragma unroll
        for (int p = 0; p < 2; p++) xf[0][p] = *reinterpret_cast<const gr_h8 *>(xa + p * GR_PLANE);
        float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f), h4 = s4;
        if constexpr (NEXT) {
            s4 = *reinterpret_cast<const float4 *>(bn + 32 * wave + 4 * h);
            h4 = *reinterpret_cast<const float4 *>(bn + HD + 32 * wave + 4 * h);
        }
        constexpr bool VT = GR_VRES && RT >= GR_NRES;                    // this tile lives in vector registers
        f32x16 atmp;
        f32x16 &a = RT < GR_NRES ? acc[RT < GR_NRES ? RT : 0] : VT ? zs[RT >= GR_NRES ? (RT - GR_NRES) & 1 : 0] : atmp;     // (its previous-layer values went into the planes one tile ago)
        // (the tile's first matrix instruction takes a literal zero as its accumulator: no 16 writes into the accumulation file)
        // the spilled tiles' old values: requested a tile before the slices that turn them into planes
        if constexpr (RT + GR_LOOK + 1 >= GR_NRES && RT + GR_LOOK + 1 < GR_NT) zload(std::integral_constant<int, RT + GR_LOOK + 1 - GR_NRES>{});
        gr_static_for<4>([&](auto Pc) __attribute__((always_inline)) {
            constexpr int g = decltype(Pc)::value;                       // region = k-steps 2g, 2g+1 = column quarter g of the next tile
            const gr_h8 *x0 = xf[0], *x1 = xf[1];
            f32x2 v01 = {0.f, 0.f}, v23 = {0.f, 0.f}, e01 = v01, e23 = v01;
            gr_h2 p01 = {0, 0}, p23 = p01, q01 = p01, q23 = p01;
            unsigned char *dst = pnext + n * GR_ROWB + (32 * wave + 8 * g + 4 * h) * 2;
            const f32x16 &nx = GR_TILEVAL(PT < GR_NT ? PT : RT);
            const f32x16 &pv = GR_TILEVAL(RT > 0 ? RT - 1 : 0);
            auto M = [&](int ks, int wp, int xp) __attribute__((always_inline)) {
                if (GR_ABL & 2) return;
                if constexpr (VT) mfma_v(a, wf[wp][ks], (ks & 1) ? x1[xp] : x0[xp], ks == 0 && wp == 0 && xp == 1);
                else if (ks == 0 && wp == 0 && xp == 1)
                    a = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[wp][ks], x0[xp], f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                else a = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[wp][ks], (ks & 1) ? x1[xp] : x0[xp], a, 0, 0, 0);
            };
            // slice 0
            M(2 * g, 0, 1);
            xf[1][0] = *reinterpret_cast<const gr_h8 *>(xa + 32 * (2 * g + 1));
            if constexpr (NEXT) { v01 = gr_fma2(f32x2{nx[4 * g], nx[4 * g + 1]}, f32x2{s4.x, s4.y}, f32x2{h4.x, h4.y});
                                  v23 = gr_fma2(f32x2{nx[4 * g + 2], nx[4 * g + 3]}, f32x2{s4.z, s4.w}, f32x2{h4.z, h4.w}); }
            GR_FENCE();
            // slice 1
            M(2 * g, 1, 0);
            xf[1][1] = *reinterpret_cast<const gr_h8 *>(xa + GR_PLANE + 32 * (2 * g + 1));
            if constexpr (NEXT) { v01 = __builtin_elementwise_max(v01, f32x2{0.f, 0.f}); v23 = __builtin_elementwise_max(v23, f32x2{0.f, 0.f}); }
            GR_FENCE();
            // slice 2
            M(2 * g, 0, 0);
            if constexpr (NEXT) { p01 = __builtin_convertvector(v01, gr_h2); p23 = __builtin_convertvector(v23, gr_h2); }
            if constexpr (NEXT && g < 3) s4 = *reinterpret_cast<const float4 *>(bn + 32 * wave + 8 * (g + 1) + 4 * h);   // scale | shift of the next quarter (this one's are used up)
            GR_FENCE();
            // slice 3: the k-step 2g operands are dead once its three instructions have issued
            M(2 * g + 1, 0, 1);
            if constexpr (g < 3) xf[0][0] = *reinterpret_cast<const gr_h8 *>(xa + 32 * (2 * g + 2));
            if constexpr (NEXT && !GR_MIX) { e01 = __builtin_convertvector(p01, f32x2); e23 = __builtin_convertvector(p23, f32x2); }
            if constexpr (NEXT && g < 3) h4 = *reinterpret_cast<const float4 *>(bn + HD + 32 * wave + 8 * (g + 1) + 4 * h);
            GR_FENCE();
            // slice 4
            M(2 * g + 1, 1, 0);
            if constexpr (g < 3) xf[0][1] = *reinterpret_cast<const gr_h8 *>(xa + GR_PLANE + 32 * (2 * g + 2));
            if constexpr (NEXT) { if constexpr (GR_MIX == 2) { q01 = gr_rem16(v01, p01); q23 = gr_rem16(v23, p23); }
                                  else if constexpr (GR_MIX == 1) { v01 = gr_rem2(v01, p01); v23 = gr_rem2(v23, p23); } else { v01 = gr_sub2(v01, e01); v23 = gr_sub2(v23, e23); }
                                  if (!(GR_ABL & 8)) *reinterpret_cast<uint2 *>(dst) = make_uint2(__builtin_bit_cast(unsigned, p01), __builtin_bit_cast(unsigned, p23));
                                  if constexpr (GR_MIX != 2) { q01 = __builtin_convertvector(v01, gr_h2); q23 = __builtin_convertvector(v23, gr_h2); } }
            GR_FENCE();
            // slice 5
            M(2 * g + 1, 0, 0);
            if constexpr (NEXT) { if (!(GR_ABL & 8)) *reinterpret_cast<uint2 *>(dst + GR_PLANE) = make_uint2(__builtin_bit_cast(unsigned, q01), __builtin_bit_cast(unsigned, q23));
                                  else asm volatile("" :: "v"(p01), "v"(p23), "v"(q01), "v"(q23)); }
            if constexpr (STATS) { stat2(2 * g, pv[4 * g], pv[4 * g + 1]); stat2(2 * g + 1, pv[4 * g + 2], pv[4 * g + 3]); }
            GR_FENCE();
        });
        if constexpr (RT >= GR_NRES && !GR_VRES) {                    // not resident: BatchNorm sums now, then out to its spill slot
            stats_all(a);
            zstore(std::integral_constant<int, RT - GR_NRES>{}, a);
            GR_FENCE();
        }
        // margin behind the matrix pipe's write-back (MFMA_SETTLE in mtfjsp_encoder.hip): hipcc copies the finished tile out of the
        // accumulation file right here, at exactly the distance its hazard table asks for
        if constexpr (!VT && GR_SETTLE) { asm volatile("s_nop 1"); GR_FENCE(); }          // (no operands: nothing for the register allocator to reconcile)
    };
    // first Linear: tile RT's 16-wide operand sits at byte offset 32*(RT & 7) of buffer 0's rows
    auto consume_tile0 = [&](auto Tc) __attribute__((always_inline)) {
        constexpr int RT = decltype(Tc)::value;
        const unsigned char *xa = xa0 + 32 * (RT & 7);
        bf16x8 x[3];
#pragma unroll
        for (int p = 0; p < 3; p++) x[p] = *reinterpret_cast<const bf16x8 *>(xa + p * GR_PLANE);
        f32x16 a = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0f[0], x[2], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0f[2], x[0], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0f[1], x[1], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0f[0], x[1], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0f[1], x[0], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0f[0], x[0], a, 0, 0, 0);
        if constexpr (RT < GR_NRES) acc[RT] = a;
        else if constexpr (GR_VRES) zs[(RT - GR_NRES) & 1] = a;
        else {
            stats_all(a);
            zstore(std::integral_constant<int, RT - GR_NRES>{}, a);
        }
        if constexpr (RT > 0 && (GR_VRES || RT - 1 < GR_NRES)) stats_tile(std::integral_constant<int, RT - 1>{});
        __builtin_amdgcn_sched_barrier(0);
    };
    // this workgroup's column sums -> its dispatch group's accumulators, grid barrier `k` (which folds the groups into the totals),
    // then the BatchNorm scale / shift of the NEXT layer's input from the complete sums
    auto layer_boundary = [&](auto Kc) __attribute__((always_inline)) {
        constexpr int k = decltype(Kc)::value;
        GR_STAMP_AT(4 + 4 * k);
        if constexpr (GR_VRES && GR_NT > GR_NRES) {               // the last tile's sums (it has no successor to take them beside)
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // (its matrix instructions were written as text: no compiler-inserted wait before the reads)
            stats_all(zs[(GR_NT - 1 - GR_NRES) & 1]);
        }
#if GR_FOLD_DPP
        (void)fold_stats;
        const float colsum = fold_stats_dpp();                    // lane n of a half wave: value n of its 32 (see above)
        const int kind = n >> 4, scol = 32 * wave + 8 * ((n & 15) >> 2) + 4 * h + (n & 3);
        constexpr int PAIR = 16;                                  // the lane holding the other kind of the same column
#else
        const float colsum = fold_stats();                        // (GR_VRES 0: tile 17's sums were taken when it was stored)
        const int kind = tid & 1, scol = tid >> 1;
        constexpr int PAIR = 1;
#endif
        const int sidx = 2 * scol + kind;                         // this thread's (column, sum | sumsq) word pair
        if (k == 1) GR_STAMP_AT(2);
        // one integer atomic, fire and forget
        const unsigned grp = blockIdx.x & 7u;
        {
            float x = colsum;
            if (!(__builtin_fabsf(x) * (kind ? A.wsinv[k] * A.wsinv[k] : A.wsinv[k]) < __builtin_ldexpf(1.0f, 51 - A.ffrac[k]))) {
                // not a number (an operand beyond the f16 range upstream: (inf | -inf) pieces) or beyond the fixed-point range: the host
                // repeats the forward on the f32-instruction kernels; the contribution still goes out so that nobody waits for it
                if (A.range_flag && !timed_out) __hip_atomic_store(A.range_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (after a time-out the values mean nothing)
                x = 0.f;
            }
            (void)__hip_atomic_fetch_add(A.stats + ((size_t)k * 8 + grp) * 256 + sidx, gr_fix_encode(x, kind ? 2 * A.wexp[k] : A.wexp[k], A.ffrac[k]),
                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (k == 1) GR_STAMP_AT(3);
        // requests that do not depend on the other workgroups go out behind it: this BatchNorm's affine parameters and the next
        // Linear's weight fragments.  (Measured, round 4: with the weight requests at the TOP of the boundary — under the fold — the
        // statistics' atomics queue behind 64 KB of loads in the in-order memory pipeline and every boundary gets 0.4 to 1.9 us longer.)
        const float ga = A.gamma[k][scol], be = A.beta[k][scol];
        if constexpr (k < 5) load_weights(k + 1);
        GR_STAMP_AT(5 + 4 * k);
        // every thread collects its own (column, kind) from the 8 dispatch groups: a word is complete when it carries its group's size
        // in the count field
        const unsigned long long *w0 = A.stats + (size_t)k * 8 * 256 + sidx;
        unsigned long long vw[8], want[8], ssum = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) want[j] = (unsigned long long)((nblk >> 3) + ((unsigned)j < (nblk & 7u) ? 1u : 0u));
        {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            for (;;) {
#pragma unroll
                for (int j = 0; j < 8; j++) vw[j] = __hip_atomic_load(w0 + (size_t)j * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bool done = true;
                ssum = 0;
#pragma unroll
                for (int j = 0; j < 8; j++) { done = done && (vw[j] >> 58) == want[j]; ssum += vw[j] & GR_FIX_PAYLOAD; }
                if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;                  // (wave-uniform exit: the pair exchange below is cross-lane)
                if (__builtin_amdgcn_s_memrealtime() - t0 > 400000ull) {                 // 4 ms at 100 MHz: not all workgroups are resident
                    __hip_atomic_store(A.fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (host-mapped word: the host polls it without synchronising)
                    timed_out = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            GR_STAMP_AT(6 + 4 * k);
            const double tot = gr_fix_decode(ssum, nblk, A.ffrac[k]);   // the true-z (sum | sumsq) of this thread's column over all rows
            const double oth = __shfl_xor(tot, PAIR);
            if (!kind) {
                const double mean = tot * A.inv_rows;
                double var = oth * A.inv_rows - mean * mean;       // biased variance (training-mode BN)
                if (var < 0) var = 0;
                const float rstd = 1.0f / sqrtf((float)(var + BN_EPS));
                const float s = rstd * ga;
                // the accumulators hold z * wscale (power of two, exact): statistics of z, scale applied to the stored value
                s_bn[scol] = s * A.wsinv[k];
                s_bn[HD + scol] = be - (float)mean * s;
            }
        }
        LDS_BARRIER();
        GR_STAMP_AT(7 + 4 * k);
    };

    // ---------------------------------------------------------------- layer 0 / Linear 0: aggregated raw features (12 -> 128)
    GR_STAMP_AT(0);
#pragma unroll
    for (int p = 0; p < 3; p++) w0f[p] = __builtin_bit_cast(bf16x8, w0pre[p]);
    LDS_BARRIER();
    GR_STAMP_AT(1);
    {
        const float *s_feat = s_ring;
        // The operand of this Linear is 16 wide (12 features + 4 zeros) = 32 bytes of a plane row, so one plane buffer takes the
        // operands of 8 tiles side by side (tile t at byte offset 32*(t & 7)): 3 barriers for the whole layer.
        // wave w fills rows 8w..8w+7 of a tile: lane = (row r = lane >> 3, feature pair fk = lane & 7 < 6)
        auto produce0 = [&](int rt) __attribute__((always_inline)) {
            const int r = 8 * wave + (lane >> 3), fk = lane & 7, row = rt * 32 + r;
            const int fo = fk < 6 ? 2 * fk : 0;
            const unsigned cp = s_ellc[row];
            const unsigned r0 = cp & 0xffffu, r1 = cp >> 16;
            const int deg = 1 + (r0 != 0xffffu) + (r1 != 0xffffu);
            const float inv = deg == 1 ? 1.0f : deg == 2 ? 0.5f : (1.0f / 3.0f);
            const float w0 = s_ellv0[row], w1 = s_ellv1[row];            // 0 where there is no edge (and for rows >= nrows)
            const int n0 = r0 != 0xffffu ? (int)r0 : row, n1 = r1 != 0xffffu ? (int)r1 : row;
            const float2 o = *reinterpret_cast<const float2 *>(s_feat + row * 12 + fo);
            const float2 x = *reinterpret_cast<const float2 *>(s_feat + n0 * 12 + fo);
            const float2 y = *reinterpret_cast<const float2 *>(s_feat + n1 * 12 + fo);
            // gcn:125-153 (A_w @ h) / nnz_row on the raw features; small-integer weights, <= 3 terms: f32 FMA chain
            float v[4] = {__builtin_fmaf(w1, y.x, __builtin_fmaf(w0, x.x, o.x)) * inv, __builtin_fmaf(w1, y.y, __builtin_fmaf(w0, x.y, o.y)) * inv, 0.f, 0.f};
            if (row >= nrows) { v[0] = 0.f; v[1] = 0.f; }
            uint2 p0, p1, p2;
            split3x4(v, p0, p1, p2);                              // elements 2, 3 are padding
            unsigned char *d = s_planes + r * GR_ROWB + 32 * (rt & 7) + 4 * fk;
            if (fk < 6) {
                *reinterpret_cast<unsigned *>(d) = p0.x;
                *reinterpret_cast<unsigned *>(d + GR_PLANE) = p1.x;
                *reinterpret_cast<unsigned *>(d + 2 * GR_PLANE) = p2.x;
            }
        };
        // (producing chunk C + 1's operands between chunk C's matrix instructions — a second plane buffer — was measured in round 4: 6.4 us
        // for this Linear against 6.5 us; not kept)
        gr_static_for<(GR_NT + 7) / 8>([&](auto Cc) __attribute__((always_inline)) {
            constexpr int C = decltype(Cc)::value;
            constexpr int NTC = GR_NT - 8 * C < 8 ? GR_NT - 8 * C : 8;
#pragma unroll
            for (int i = 0; i < NTC; i++) produce0(8 * C + i);
            LDS_BARRIER();
            __builtin_amdgcn_sched_barrier(0);
            gr_static_for<NTC>([&](auto Ic) __attribute__((always_inline)) {
                constexpr int RT = 8 * C + decltype(Ic)::value;
                consume_tile0(std::integral_constant<int, RT>{});
            });
            LDS_BARRIER();
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    layer_boundary(std::integral_constant<int, 0>{});

    // ---------------------------------------------------------------- Linears 1..5 (the fourth, layer 3, aggregates over the graph first)
    // (straight-line over the layers: a run-time loop would carry all 288 accumulators through its back edge, which the register
    // allocator resolves through scratch memory)
    gr_static_for<5>([&](auto Lc) __attribute__((always_inline)) {
        constexpr int layer = decltype(Lc)::value + 1;
#ifndef GR_KEEP_MASKS                     // (A/B: -DGR_KEEP_MASKS is the code before this change)
        asm volatile("" : "+s"(nrows_l), "+v"(n_l));
#endif
        if constexpr (layer != 3) {
            gr_static_for<GR_LOOK>([&](auto Ic) __attribute__((always_inline)) { produce_tile(Ic, plane_buf(decltype(Ic)::value)); });
            LDS_BARRIER();
            unsigned *cnt = s_cnt + (layer - 1) * (GR_NT / GR_GRP);
            gr_static_for<GR_NT>([&](auto Tc) __attribute__((always_inline)) {
                constexpr int RT = decltype(Tc)::value;
                constexpr int step = RT / GR_GRP, back = GR_LOOK / GR_GRP;
                if constexpr (GR_LOOK > GR_GRP && RT % GR_GRP == 0 && step >= back) {
                    // this group's planes were written during group step - back, by all four waves (each its 32 columns), after they had
                    // finished reading the group that held these buffers before: one counter says both
                    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&cnt[step - back], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < 4u)
                        __builtin_amdgcn_s_sleep(1);
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
                consume_tile(Tc);                                 // produces tile RT + GR_LOOK between its matrix instructions
                GR_PIN_V();
                if constexpr (RT % GR_GRP == GR_GRP - 1) {
                    if constexpr (GR_LOOK > GR_GRP) {             // (LDS executes a wave's operations in issue order: the add lands after its plane writes and reads)
                        asm volatile("" ::: "memory");
                        if (lane == 0) __hip_atomic_fetch_add(&cnt[step], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    } else LDS_BARRIER();                         // one barrier per group of tiles
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (GR_LOOK > GR_GRP) LDS_BARRIER();        // (the boundary parks its sums in the ring area, i.e. in plane buffers)
            static_assert(GR_NT % GR_GRP == 0 && GR_GRP >= 1 && (GR_LOOK == GR_GRP || GR_LOOK == 2 * GR_GRP) && GR_LOOK <= GR_NRES, "tile groups");
        } else {
            // gcn:125-149: (A_w @ h) / nnz_row with h = relu(bn_outer0(z)), A_w incl. the self loop.  h tiles go through a ring of
            // GR_RING tiles in LDS; tile rt's rows and their in-edge sources (same instance, T <= 65 rows) lie in tiles rt-2..rt+2
            auto write_h = [&](auto Tc) __attribute__((always_inline)) {
                constexpr int rt = decltype(Tc)::value;
                float *base = s_ring + ((rt % GR_RING) * 32 + n) * HD;
                gr_static_for<4>([&](auto Gc) __attribute__((always_inline)) {
                    constexpr int g = decltype(Gc)::value;
                    const float *bn = rt * 32 + n_l < nrows_l ? s_bn : s_zero;
                    const float4 s4 = *reinterpret_cast<const float4 *>(bn + 32 * wave + 8 * g + 4 * h);
                    const float4 h4 = *reinterpret_cast<const float4 *>(bn + HD + 32 * wave + 8 * g + 4 * h);
                    const f32x16 &a = GR_TILEVAL(rt);
                    const int ch = (8 * wave + 2 * g + h) ^ (n & 7);
                    *reinterpret_cast<float4 *>(base + 4 * ch) = make_float4(bn_relu_ss(a[4 * g], s4.x, h4.x), bn_relu_ss(a[4 * g + 1], s4.y, h4.y),
                                                                             bn_relu_ss(a[4 * g + 2], s4.z, h4.z), bn_relu_ss(a[4 * g + 3], s4.w, h4.w));
                });
                __builtin_amdgcn_sched_barrier(0);
            };
            auto ring_row = [&](int R) __attribute__((always_inline)) {      // workgroup-relative row -> float offset of its ring row
                const int t = R >> 5;
                return ((t - GR_RING * ((t * 43) >> 8)) * 32 + (R & 31)) * HD;   // t % 6 for t < 128
            };
            // per-row constants of tile RT's aggregation: ring pointers of the row and of its (<= 2) in-edge sources, edge weights, 1 / nnz
            struct AggRow { const float *po, *px, *py; float w0, w1, inv; int s0, s1, s2; };
            auto agg_row = [&](int RT) __attribute__((always_inline)) {
                AggRow r;
                const int row = RT * 32 + n_l;
                const unsigned cp = s_ellc[row];
                const unsigned r0 = cp & 0xffffu, r1 = cp >> 16;
                r.w0 = s_ellv0[row]; r.w1 = s_ellv1[row];
                const int deg = 1 + (r0 != 0xffffu) + (r1 != 0xffffu);
                r.inv = deg == 1 ? 1.0f : deg == 2 ? 0.5f : (1.0f / 3.0f);
                const int n0 = r0 != 0xffffu ? (int)r0 : row, n1 = r1 != 0xffffu ? (int)r1 : row;
                r.po = s_ring + ((RT % GR_RING) * 32 + n) * HD; r.px = s_ring + ring_row(n0); r.py = s_ring + ring_row(n1);
                r.s0 = n & 7; r.s1 = n0 & 7; r.s2 = n1 & 7;
                return r;
            };
            // small-integer edge weights, <= 3 terms: an f32 FMA chain is within 2 ulp of the reference's f64-then-cast
            auto agg_quarter_standalone = [&](const AggRow &r, int g, unsigned char *dst) __attribute__((always_inline)) {
                const int chl = 8 * wave + 2 * g + h;
                const float4 o = *reinterpret_cast<const float4 *>(r.po + 4 * (chl ^ r.s0));
                const float4 x = *reinterpret_cast<const float4 *>(r.px + 4 * (chl ^ r.s1));
                const float4 y = *reinterpret_cast<const float4 *>(r.py + 4 * (chl ^ r.s2));
                const f32x2 W0 = {r.w0, r.w0}, W1 = {r.w1, r.w1}, IV = {r.inv, r.inv};
                const f32x2 v01 = gr_mul2(gr_fma2(W1, f32x2{y.x, y.y}, gr_fma2(W0, f32x2{x.x, x.y}, f32x2{o.x, o.y})), IV);
                const f32x2 v23 = gr_mul2(gr_fma2(W1, f32x2{y.z, y.w}, gr_fma2(W0, f32x2{x.z, x.w}, f32x2{o.z, o.w})), IV);
                uint2 p0, p1;
                split2x4(v01, v23, p0, p1);
                *reinterpret_cast<uint2 *>(dst + 16 * g) = p0;
                *reinterpret_cast<uint2 *>(dst + 16 * g + GR_PLANE) = p1;
            };
            // tile RT: matrix products of its planes, with the aggregation + split of tile RT+1 (h of tiles RT-1..RT+3 from the ring)
            // and the h = relu(bn(z)) of tile RT+4 (into the ring slot of tile RT-2) riding between the matrix instructions
            auto consume_agg = [&](auto Tc) __attribute__((always_inline)) {
                constexpr int RT = decltype(Tc)::value;
                constexpr bool NEXT = RT + 1 < GR_NT && !(GR_ABL & 1);
                constexpr bool WH = RT + 4 < GR_NT && !(GR_ABL & 1);
                constexpr int WT = WH ? RT + 4 : RT;
                constexpr bool STATS = RT > 0 && (GR_VRES || RT - 1 < GR_NRES) && !(GR_ABL & 4);
                constexpr bool VT = GR_VRES && RT >= GR_NRES;
                const unsigned char *xa = xa0 + (RT & 1) * GR_TILE;
                gr_h8 xf[2][2];
#pragma unroll
                for (int p = 0; p < 2; p++) xf[0][p] = *reinterpret_cast<const gr_h8 *>(xa + p * GR_PLANE);
                AggRow r = agg_row(NEXT ? RT + 1 : RT);
                unsigned char *dst = s_planes + ((RT + 1) & 1) * GR_TILE + n * GR_ROWB + (32 * wave + 4 * h) * 2;
                const float *bnw = WT * 32 + n_l < nrows_l ? s_bn : s_zero;
                float *hbase = s_ring + ((WT % GR_RING) * 32 + n) * HD;
                float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f), h4 = s4;
                if constexpr (WH) {
                    s4 = *reinterpret_cast<const float4 *>(bnw + 32 * wave + 4 * h);
                    h4 = *reinterpret_cast<const float4 *>(bnw + HD + 32 * wave + 4 * h);
                }
                f32x16 atmp;
                f32x16 &a = RT < GR_NRES ? acc[RT < GR_NRES ? RT : 0] : VT ? zs[RT >= GR_NRES ? (RT - GR_NRES) & 1 : 0] : atmp;
                // the spilled tiles' old values: requested two tiles before they are turned into h
                if constexpr (RT + 6 >= GR_NRES && RT + 6 < GR_NT) zload(std::integral_constant<int, RT + 6 - GR_NRES>{});
                gr_static_for<4>([&](auto Pc) __attribute__((always_inline)) {
                    constexpr int g = decltype(Pc)::value;
                    const gr_h8 *x0 = xf[0], *x1 = xf[1];
                    const f32x16 &wv = GR_TILEVAL(WT);
                    const f32x16 &pv = GR_TILEVAL(RT > 0 ? RT - 1 : 0);
                    const int chl = 8 * wave + 2 * g + h;
                    float4 o = make_float4(0.f, 0.f, 0.f, 0.f), x = o, y = o;
                    f32x2 v01 = {0.f, 0.f}, v23 = v01, e01 = v01, e23 = v01, u01 = v01, u23 = v01;
                    gr_h2 p01 = {0, 0}, p23 = p01, q01 = p01, q23 = p01;
                    const f32x2 W0 = {r.w0, r.w0}, W1 = {r.w1, r.w1}, IV = {r.inv, r.inv};
                    auto M = [&](int ks, int wp, int xp) __attribute__((always_inline)) {
                        if (!(GR_ABL & 2)) {
                            if constexpr (VT) mfma_v(a, wf[wp][ks], (ks & 1) ? x1[xp] : x0[xp], g == 0 && ks == 0 && wp == 0 && xp == 1);
                            else if (g == 0 && ks == 0 && wp == 0 && xp == 1)
                                a = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[wp][ks], x0[xp], f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                            else a = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[wp][ks], (ks & 1) ? x1[xp] : x0[xp], a, 0, 0, 0);
                        }
                    };
                    // slice 0
                    M(2 * g, 0, 1);
                    xf[1][0] = *reinterpret_cast<const gr_h8 *>(xa + 32 * (2 * g + 1));
                    if constexpr (NEXT) { o = *reinterpret_cast<const float4 *>(r.po + 4 * (chl ^ r.s0)); x = *reinterpret_cast<const float4 *>(r.px + 4 * (chl ^ r.s1));
                                          y = *reinterpret_cast<const float4 *>(r.py + 4 * (chl ^ r.s2)); }
                    if constexpr (STATS) { stat2(2 * g, pv[4 * g], pv[4 * g + 1]); stat2(2 * g + 1, pv[4 * g + 2], pv[4 * g + 3]); }
                    GR_FENCE();
                    // slice 1
                    M(2 * g, 1, 0);
                    xf[1][1] = *reinterpret_cast<const gr_h8 *>(xa + GR_PLANE + 32 * (2 * g + 1));
                    if constexpr (WH) { u01 = __builtin_elementwise_max(gr_fma2(f32x2{wv[4 * g], wv[4 * g + 1]}, f32x2{s4.x, s4.y}, f32x2{h4.x, h4.y}), f32x2{0.f, 0.f});
                                        u23 = __builtin_elementwise_max(gr_fma2(f32x2{wv[4 * g + 2], wv[4 * g + 3]}, f32x2{s4.z, s4.w}, f32x2{h4.z, h4.w}), f32x2{0.f, 0.f}); }
                    GR_FENCE();
                    // slice 2
                    M(2 * g, 0, 0);
                    if constexpr (WH) *reinterpret_cast<float4 *>(hbase + 4 * (chl ^ (n & 7))) = make_float4(u01[0], u01[1], u23[0], u23[1]);
                    if constexpr (NEXT) { v01 = gr_fma2(W0, f32x2{x.x, x.y}, f32x2{o.x, o.y}); v23 = gr_fma2(W0, f32x2{x.z, x.w}, f32x2{o.z, o.w}); }
                    if constexpr (WH && g < 3) s4 = *reinterpret_cast<const float4 *>(bnw + 32 * wave + 8 * (g + 1) + 4 * h);
                    GR_FENCE();
                    // slice 3
                    M(2 * g + 1, 0, 1);
                    if constexpr (g < 3) xf[0][0] = *reinterpret_cast<const gr_h8 *>(xa + 32 * (2 * g + 2));
                    if constexpr (NEXT) { v01 = gr_fma2(W1, f32x2{y.x, y.y}, v01); v23 = gr_fma2(W1, f32x2{y.z, y.w}, v23); }
                    if constexpr (WH && g < 3) h4 = *reinterpret_cast<const float4 *>(bnw + HD + 32 * wave + 8 * (g + 1) + 4 * h);
                    GR_FENCE();
                    // slice 4
                    M(2 * g + 1, 1, 0);
                    if constexpr (g < 3) xf[0][1] = *reinterpret_cast<const gr_h8 *>(xa + GR_PLANE + 32 * (2 * g + 2));
                    if constexpr (NEXT) { v01 = gr_mul2(v01, IV); v23 = gr_mul2(v23, IV); p01 = __builtin_convertvector(v01, gr_h2); p23 = __builtin_convertvector(v23, gr_h2);
                                          if constexpr (!GR_MIX) { e01 = __builtin_convertvector(p01, f32x2); e23 = __builtin_convertvector(p23, f32x2); } }
                    GR_FENCE();
                    // slice 5
                    M(2 * g + 1, 0, 0);
                    if constexpr (NEXT) { if constexpr (GR_MIX == 2) { q01 = gr_rem16(v01, p01); q23 = gr_rem16(v23, p23); }
                                          else { if constexpr (GR_MIX) { v01 = gr_rem2(v01, p01); v23 = gr_rem2(v23, p23); } else { v01 = gr_sub2(v01, e01); v23 = gr_sub2(v23, e23); } q01 = __builtin_convertvector(v01, gr_h2); q23 = __builtin_convertvector(v23, gr_h2); }
                                          *reinterpret_cast<uint2 *>(dst + 16 * g) = make_uint2(__builtin_bit_cast(unsigned, p01), __builtin_bit_cast(unsigned, p23));
                                          *reinterpret_cast<uint2 *>(dst + 16 * g + GR_PLANE) = make_uint2(__builtin_bit_cast(unsigned, q01), __builtin_bit_cast(unsigned, q23)); }
                    GR_FENCE();
                });
                if constexpr (RT >= GR_NRES && !GR_VRES) {
                    stats_all(a);
                    zstore(std::integral_constant<int, RT - GR_NRES>{}, a);
                    GR_FENCE();
                }
                if constexpr (!VT && GR_SETTLE) { asm volatile("s_nop 1"); GR_FENCE(); }          // (no operands: nothing for the register allocator to reconcile)
            };
            write_h(std::integral_constant<int, 0>{}); write_h(std::integral_constant<int, 1>{}); write_h(std::integral_constant<int, 2>{});
            write_h(std::integral_constant<int, 3>{});
            LDS_BARRIER();
            {
                const AggRow r = agg_row(0);
                unsigned char *dst = s_planes + n * GR_ROWB + (32 * wave + 4 * h) * 2;
#pragma unroll
                for (int g = 0; g < 4; g++) agg_quarter_standalone(r, g, dst);
            }
            LDS_BARRIER();
            __builtin_amdgcn_sched_barrier(0);
            gr_static_for<GR_NT>([&](auto Tc) __attribute__((always_inline)) {
                consume_agg(Tc);
                LDS_BARRIER();
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        layer_boundary(std::integral_constant<int, layer>{});
    });

    GR_STAMP_AT(30);
    // ---------------------------------------------------------------- h = relu(bn_outer1(z)): graph mean pool, candidate gather, node embeddings
    // Pooling is per column, so every wave works alone on its 32 columns (no workgroup barriers): it transposes a tile of h
    // through a private 32 x 36 f32 buffer (ring area) — lane (column c, half) then sums rows 16 half .. +15, split at the one
    // instance boundary a 16-row chunk can contain (T >= 16) — and parks the two partial sums per chunk; the chunks of an
    // instance are added in a fixed order at the end (deterministic, no atomics).  Candidate rows and the optional node
    // embeddings go out straight from the registers (16 bytes per lane and column group).
    {
        float *s_part = reinterpret_cast<float *>(s_planes);      // [2 * GR_NT chunks][2 segments][128] = 36 KB: the plane buffers and the first 2 KB of the ring area
        static_assert(2 * GR_NT * 2 * HD * 4 <= 2 * GR_TILE + 4096, "partial pool sums: plane buffers + 4 KB");
        float *tb = s_ring + 1024 + wave * (2 * 32 * 36);         // transposition buffers: ring area past those 4 KB
        float *s_wtab = s_ring + 1024 + 4 * (2 * 32 * 36);        // [17][16]: row j = 1 for the first j of a chunk's 16 rows, 0 for the others (pool_tile)
        for (int i = tid; i < 17 * 16; i += 256) s_wtab[i] = (i & 15) < (i >> 4) ? 1.0f : 0.0f;
        // ONE request per thread on behalf of the next launch (GinResArgs::warm): thread (blockIdx / 8) * 256 + tid of this XCD's workgroups
        // takes that line of the concatenated buffers (lines beyond the XCD's thread count are not requested); the word is dropped behind
        // the last tile, where it has long arrived — no wait of this phase can be held up by it
        unsigned warm_word = 0;
        if (A.nwarm > 0) {
            unsigned l = (blockIdx.x >> 3) * 256u + (unsigned)tid;
            const unsigned char *wp = reinterpret_cast<const unsigned char *>(A.warm[0]);
            bool found = false;
            for (int w = 0; w < A.nwarm; w++) {
                const unsigned nl = A.warm_lines[w];
                if (!found && l < nl) { wp = reinterpret_cast<const unsigned char *>(A.warm[w]) + (size_t)l * 128; found = true; }
                if (!found) l -= nl;
            }
            warm_word = *reinterpret_cast<const unsigned *>(wp);
        }
        float4 S[4], Hs[4];
#pragma unroll
        for (int g = 0; g < 4; g++) {
            S[g] = *reinterpret_cast<const float4 *>(s_bn + 32 * wave + 8 * g + 4 * h);
            Hs[g] = *reinterpret_cast<const float4 *>(s_bn + HD + 32 * wave + 8 * g + 4 * h);
        }
        if constexpr (GR_NT - GR_NRES >= 1) zload(std::integral_constant<int, 0>{});
        if constexpr (GR_NT - GR_NRES >= 2) zload(std::integral_constant<int, 1>{});
        const int c = lane & 31;
        int left = ((16 * h) / T + 1) * T - 16 * h;                   // rows of the instance of this lane's first row (16h of tile 0) still ahead
        int slot[GR_NT];                                              // candidate slot of this lane's row in every tile (-1: none)
#ifndef GR_KEEP_MASKS
        asm volatile("" : "+v"(n_l));                                 // (the row indices of this phase are formed here, not kept from the layers)
#endif
#pragma unroll
        for (int rt = 0; rt < GR_NT; rt++) slot[rt] = (GR_ABL & 32) ? -1 : s_rowcand[rt * 32 + n_l];
        LDS_BARRIER();                                                // the plane buffers are no longer read
        GR_STAMP_AT(28);
        // gcn:192 for one tile: rows 16h .. 16h+15 of column c from the transposition buffer, split at the instance boundary
        // pool_load: the requests (a chunk's 16 values of column c, its weights) — issued a tile's BatchNorm + ReLU ahead of pool_math, which
        // would otherwise meet an LDS round trip with nothing to cover it (one wave per SIMD)
        float x[16];
        float4 w0, w1, w2, w3;
        auto pool_load = [&](int rt) __attribute__((always_inline)) {
            const float *rb = tb + (rt & 1) * (32 * 36) + (16 * h) * 36 + c;
#pragma unroll
            for (int i = 0; i < 16; i++) x[i] = rb[i * 36];
            const float4 *wt = reinterpret_cast<const float4 *>(s_wtab + (left < 16 ? left : 16) * 16);
            w0 = wt[0]; w1 = wt[1]; w2 = wt[2]; w3 = wt[3];
        };
        auto pool_math = [&](int rt) __attribute__((always_inline)) {
            // the chunk's total by a pairwise tree of two-wide adds, the part in front of the instance boundary by two-wide FMAs against the
            // 0 / 1 weights of row `left` of s_wtab (round 5: the weights were formed per value — a subtraction, a clamp and an FMA each, 48
            // vector instructions per tile where the vector unit is what this phase waits for), the rest as their difference
            f32x2 sa2 = gr_fma2(f32x2{x[0], x[1]}, f32x2{w0.x, w0.y}, f32x2{0.f, 0.f});
            sa2 = gr_fma2(f32x2{x[2], x[3]}, f32x2{w0.z, w0.w}, sa2);
            sa2 = gr_fma2(f32x2{x[4], x[5]}, f32x2{w1.x, w1.y}, sa2);
            sa2 = gr_fma2(f32x2{x[6], x[7]}, f32x2{w1.z, w1.w}, sa2);
            sa2 = gr_fma2(f32x2{x[8], x[9]}, f32x2{w2.x, w2.y}, sa2);
            sa2 = gr_fma2(f32x2{x[10], x[11]}, f32x2{w2.z, w2.w}, sa2);
            sa2 = gr_fma2(f32x2{x[12], x[13]}, f32x2{w3.x, w3.y}, sa2);
            sa2 = gr_fma2(f32x2{x[14], x[15]}, f32x2{w3.z, w3.w}, sa2);
            const f32x2 t2 = gr_add2(gr_add2(gr_add2(f32x2{x[0], x[1]}, f32x2{x[2], x[3]}), gr_add2(f32x2{x[4], x[5]}, f32x2{x[6], x[7]})),
                                     gr_add2(gr_add2(f32x2{x[8], x[9]}, f32x2{x[10], x[11]}), gr_add2(f32x2{x[12], x[13]}, f32x2{x[14], x[15]})));
            const float tot = t2[0] + t2[1];
            const float sa = sa2[0] + sa2[1];
            float *pp = s_part + ((2 * rt + h) * 2) * HD + 32 * wave + c;
            pp[0] = left >= 16 ? tot : sa; pp[HD] = left >= 16 ? 0.f : tot - sa;
            left -= 32;                                               // on to this lane's rows of the next tile (T >= 16: at most two instances further)
            if (left <= 0) left += T;
            if (left <= 0) left += T;
        };
        gr_static_for<GR_NT>([&](auto Tc) __attribute__((always_inline)) {
            constexpr int rt = decltype(Tc)::value;
            const f32x16 &a = GR_TILEVAL(rt);
            if constexpr (rt > 0) { if (!(GR_ABL & 16)) pool_load(rt - 1); }     // one tile behind: its writes have landed (the other buffer)
            float4 v[4];
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const f32x2 lo = __builtin_elementwise_max(gr_fma2(f32x2{a[4 * g], a[4 * g + 1]}, f32x2{S[g].x, S[g].y}, f32x2{Hs[g].x, Hs[g].y}), f32x2{0.f, 0.f});
                const f32x2 hi = __builtin_elementwise_max(gr_fma2(f32x2{a[4 * g + 2], a[4 * g + 3]}, f32x2{S[g].z, S[g].w}, f32x2{Hs[g].z, Hs[g].w}), f32x2{0.f, 0.f});
                v[g] = make_float4(lo[0], lo[1], hi[0], hi[1]);                  // = bn_relu_ss: max(fma(a, scale, shift), 0)
            }
            float *wb = tb + (rt & 1) * (32 * 36);
#pragma unroll
            for (int g = 0; g < 4; g++) *reinterpret_cast<float4 *>(wb + n * 36 + 8 * g + 4 * h) = v[g];
            if (slot[rt] >= 0) {
                float *d = A.cand_feat + ((size_t)inst0 * A.J + slot[rt]) * HD + 32 * wave + 4 * h;
#pragma unroll
                for (int g = 0; g < 4; g++) *reinterpret_cast<float4 *>(d + 8 * g) = v[g];
            }
            if (A.h_nodes && rt * 32 + n_l < nrows) {
                float *d = A.h_nodes + (grow0 + rt * 32 + n_l) * HD + 32 * wave + 4 * h;
#pragma unroll
                for (int g = 0; g < 4; g++) *reinterpret_cast<float4 *>(d + 8 * g) = v[g];
            }
            if constexpr (rt > 0) { if (!(GR_ABL & 16)) pool_math(rt - 1); }
            __builtin_amdgcn_sched_barrier(0);
        });
        if (!(GR_ABL & 16)) { pool_load(GR_NT - 1); pool_math(GR_NT - 1); }
        asm volatile("" :: "v"(warm_word));
        GR_STAMP_AT(29);
        LDS_BARRIER();
        const float invT = 1.0f / (float)T;
        // (Round 6: an instance spans at most five 16-row chunks (T <= 65); all five partial sums are requested together — clamped, the ones beyond the
        // instance not added — instead of a loop with a run-time trip count whose reads each waited for the one before: 2.0 -> ~0.7 us of this phase's tail.
        // Same additions in the same order.)
        static_assert(GR_MAXT <= 65, "an instance spans at most five 16-row chunks");
        for (int item = tid; item < ninst * HD; item += 256) {
            const int inst = item >> 7, col = item & (HD - 1);
            const int r0 = inst * T, k0 = r0 >> 4, k1 = (r0 + T - 1) >> 4;
            float pv[5];
#pragma unroll
            for (int kk = 0; kk < 5; kk++) {
                const int k = k0 + kk < k1 ? k0 + kk : k1;
                pv[kk] = s_part[(2 * k + (16 * k < r0 ? 1 : 0)) * HD + col];     // a chunk that began in the previous instance: second segment
            }
            float sum = 0.f;
#pragma unroll
            for (int kk = 0; kk < 5; kk++) sum = k0 + kk <= k1 ? sum + pv[kk] : sum;
            A.pooled[(size_t)inst0 * HD + item] = sum * invT;
        }
        if (A.candidate && s_flag[1]) {                               // (malformed input) slots that share a row: copy from the slot that was written
            __threadfence();
            __syncthreads();
            for (int i = wave; i < ninst * A.J; i += 4) {
                const int cnd = A.candidate[(size_t)inst0 * A.J + i];
                if (cnd < 0 || cnd >= T) continue;
                const int win = s_rowcand[(i / A.J) * T + cnd];
                if (win != i) {
                    const float2 x = *reinterpret_cast<const float2 *>(A.cand_feat + ((size_t)inst0 * A.J + win) * HD + 2 * lane);
                    *reinterpret_cast<float2 *>(A.cand_feat + ((size_t)inst0 * A.J + i) * HD + 2 * lane) = x;
                }
            }
        }
    }
    GR_STAMP_AT(31);
}
#endif
