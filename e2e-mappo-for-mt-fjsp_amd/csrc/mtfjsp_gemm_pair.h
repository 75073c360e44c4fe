// mtfjsp_gemm_pair.h — TWO consecutive Linears of a GIN MLP (gcn:204-249: Linear -> BatchNorm -> ReLU -> Linear) in ONE streaming
// launch, for the shapes whose activations do not fit the chip (k_gin_res takes the others).  Included by mtfjsp_encoder.hip behind
// k_gemm_x6 (uses its helpers: split2x4m, X6_*, MFMA_SETTLE2, row_sum16, LDS_BARRIER, GemmArgs).
//
// Why.  With training-mode BatchNorm the streaming design writes and re-reads the [rows,128] f32 activations at every one of the six
// boundaries: at 819 200 rows (J10M10 x 8192, J20M20 x 2048) a 419 MB matrix each way per launch, every launch HBM-bound
// (k_gemm_x6<BNRELU>: 144 us for 838 MB).  The boundary between the two inner Linears of an MLP can go: with z_a the first Linear's
// output (stored, statistics a known),
//     pass 1 (k_gemm_x6<PRO_BNRELU> with out = NULL): z_b = W1 relu(bn_a(z_a)) is formed tile by tile and only its BatchNorm column
//             sums leave the chip — a read-only stream of z_a (non-temporal: the memory-side cache keeps what the next pass reads first);
//     pass 2 (this kernel): reads z_a again, forms z_b tile-locally, applies bn_b + ReLU + operand split in the accumulators' layout,
//             multiplies by W2 and writes z_c (and its sums): z_b never exists in memory.
// 1 257 MB instead of 1 676 MB per pair, and one product's worth of matrix work more — which the matrix pipes have to spare (a
// k_gemm_x6 launch keeps them ~25 % busy).  Bit-for-bit the same arithmetic per element as the two separate launches.
//
// Structure = k_gemm_x6's: 4 producer waves (rows two steps ahead, bn_a + ReLU + split -> LDS planes) + 4 consumer waves (wave cg owns
// output columns 32cg..32cg+31 of BOTH products: 2 x 64 registers of weight fragments), groups of four 16-row tiles, ONE barrier per
// step.  The consumers run the second product one group behind the first: in step s they multiply group s-2's intermediate planes
// (written in step s-1) by W2 and store, then multiply group s-1's input planes by W1 and write ITS intermediate planes — each wave
// its 32 columns, 8-byte stores, the layout k_headsx uses — into the other intermediate buffer.  LDS: 2 x 34 KB input planes + 2 x 34 KB
// intermediate planes + 18 KB of store transposition buffers = 158 KB.
struct Gemm2Args {
    GemmArgs a;                 // in = z_a, pro_* = BatchNorm a, Wx6 / w_sinv / bias = first Linear, out = z_c, epi_stats = sums of z_c
    const void *Wx6b;           // second weight's register image (f16 x 2 planes, scaled)
    float w_sinvb;
    const float *biasb;         // second Linear's bias
    const double *mid_stats;    // BatchNorm b: the sums of z_b from pass 1
    const float *mid_gamma, *mid_beta;
    double mid_inv_rows;
};
#define X6F_XT (2 * X6_PLANE)
#define X6F_OFF_MID (8 * X6F_XT)
#define X6F_OFF_STAT (16 * X6F_XT)
#define X6F_OFF_BN (X6F_OFF_STAT + 2 * HD * 8)
#define X6F_OFF_BN2 (X6F_OFF_BN + 2 * HD * 4)
#define X6F_OFF_TR (X6F_OFF_BN2 + 2 * HD * 4 + 64)
static size_t gemm_x6f_lds_bytes() { return (size_t)X6F_OFF_TR + 4 * 2 * X6_TRB; }
static_assert(X6F_OFF_TR + 4 * 2 * X6_TRB <= 160 * 1024, "k_gemm_x6f: LDS budget");

__global__ __launch_bounds__(512) void k_gemm_x6f(Gemm2Args G)
{
    const GemmArgs &A = G.a;
    constexpr int XT = X6F_XT;
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char *s_tiles = smem;                                // [2 buffers][4 tiles][2 planes][16 rows x 272 B]: relu(bn_a(z_a))
    unsigned char *s_mid = smem + X6F_OFF_MID;                    // the same for relu(bn_b(z_b))
    double *s_stat = reinterpret_cast<double *>(smem + X6F_OFF_STAT);
    float *s_bn = reinterpret_cast<float *>(smem + X6F_OFF_BN);   // scale | shift of BatchNorm a
    float *s_bn2 = reinterpret_cast<float *>(smem + X6F_OFF_BN2); // ... of BatchNorm b
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int ntiles = (A.N + 15) / 16;
    const int per = (ntiles + gridDim.x - 1) / gridDim.x;
    const int first = blockIdx.x * per;
    const int last = first + per < ntiles ? first + per : ntiles;
    const int nsteps = last > first ? (last - first + 3) >> 2 : 0;
    const int nsteps_c = (nsteps + 3) & ~3;
    const int rev_sum = A.rev ? first + last - 1 : 0;
    auto PT = [&](int t) __attribute__((always_inline)) { return A.rev ? rev_sum - t : t; };
    // requests in the order their data is needed (vmcnt retires in order): the BatchNorm sums first
    double bsu[STAT_REP], bsq[STAT_REP];
    float bga = 0.f, bbe = 0.f;
    if (tid < 2 * HD) {                                           // threads 0..127: BatchNorm a, 128..255: BatchNorm b
        const double *st = tid < HD ? A.pro_stats : G.mid_stats;
        const int c = tid & (HD - 1);
#pragma unroll
        for (int r = 0; r < STAT_REP; r++) { bsu[r] = st[r * 256 + c]; bsq[r] = st[r * 256 + HD + c]; }
        bga = (tid < HD ? A.pro_gamma : G.mid_gamma)[c]; bbe = (tid < HD ? A.pro_beta : G.mid_beta)[c];
    }
    auto stage_scale_shift = [&]() __attribute__((always_inline)) {
        if (tid < 2 * HD) {
            double su = 0, sq = 0;
#pragma unroll
            for (int r = 0; r < STAT_REP; r++) { su += bsu[r]; sq += bsq[r]; }
            if (A.range_flag && (su != su || sq != sq)) __hip_atomic_store(A.range_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const double ir = tid < HD ? A.pro_inv_rows : G.mid_inv_rows;
            const double mean = su * ir;
            double var = sq * ir - mean * mean;                   // biased variance (training-mode BN)
            if (var < 0) var = 0;
            const float rstd = 1.0f / sqrtf((float)(var + BN_EPS));
            const float sc = rstd * bga;
            float *d = tid < HD ? s_bn : s_bn2;
            d[tid & (HD - 1)] = sc;
            d[HD + (tid & (HD - 1))] = bbe - (float)mean * sc;
            s_stat[tid] = 0.0;
        }
    };
    if (wave >= 4) {
        // ================================ producer: k_gemm_x6<PRO_BNRELU>'s ================================
        const int pw = wave - 4;
        const int j = lane & 31, h = lane >> 5, c4 = j * 4;       // rows 2p+h of the tile, 4 columns
        const int lane_off = h * HD + c4;
        float4 preA[8], preB[8];
        const int base_row = first * 16;
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
        const int t0 = first + pw;
        // (unconditional requests, clamped tiles, steps padded to a multiple of four: see k_gemm_x6)
        if (nsteps == 0) { stage_scale_shift(); LDS_BARRIER(); LDS_BARRIER(); LDS_BARRIER(); }
        else {
            const int lastm1 = last - 1;
            auto CL = [&](int t) __attribute__((always_inline)) { return t < lastm1 ? t : lastm1; };
            request_rows(preA, CL(t0));
            request_rows(preB, CL(t0 + 4));
            stage_scale_shift();
            LDS_BARRIER();
            const float sc0 = s_bn[c4], sc1 = s_bn[c4 + 1], sc2 = s_bn[c4 + 2], sc3 = s_bn[c4 + 3];
            const float sh0 = s_bn[HD + c4], sh1 = s_bn[HD + c4 + 1], sh2 = s_bn[HD + c4 + 2], sh3 = s_bn[HD + c4 + 3];
            auto produce = [&](float4 (&pre)[8], int s) __attribute__((always_inline)) {
                const int tile = t0 + 4 * s;
                if (tile < last) {
                    unsigned char *dst = s_tiles + ((s & 1) * 4 + pw) * XT + h * X6_ROWB + j * 8;
#pragma unroll
                    for (int p = 0; p < 8; p++) {
                        const float4 &x = pre[p];
                        const f32x2 a = __builtin_elementwise_fma(f32x2{x.x, x.y}, f32x2{sc0, sc1}, f32x2{sh0, sh1});
                        const f32x2 b = __builtin_elementwise_fma(f32x2{x.z, x.w}, f32x2{sc2, sc3}, f32x2{sh2, sh3});
                        const float v[4] = {fmaxf(a[0], 0.f), fmaxf(a[1], 0.f), fmaxf(b[0], 0.f), fmaxf(b[1], 0.f)};
                        uint2 p0, p1;
                        split2x4m(v, p0, p1);
                        *reinterpret_cast<uint2 *>(dst + p * 2 * X6_ROWB) = p0;
                        *reinterpret_cast<uint2 *>(dst + p * 2 * X6_ROWB + X6_PLANE) = p1;
                    }
                }
                request_rows(pre, CL(tile + 8));
                LDS_BARRIER();
            };
            for (int s = 0; s < nsteps_c; s += 4) {
                produce(preA, s); produce(preB, s + 1); produce(preA, s + 2); produce(preB, s + 3);
            }
            LDS_BARRIER();                                        // the consumers' last first product ...
            LDS_BARRIER();                                        // ... and their last second product
        }
    } else {
        // ================================ consumer ================================
        const int m = lane & 15, q = lane >> 4;                   // operands swapped: A[col c0+m][k = 8q..], B[k = 8q..][row m], C[col c0+4q+i][row m]
        const int cg = wave;
        float4 wf1[2][2][4], wf2[2][2][4];                        // [column block][plane][k-step] of the two weights
        {
            const float4 *wi = reinterpret_cast<const float4 *>(A.Wx6) + (size_t)cg * (2 * 2 * 4 * 64) + lane;
            const float4 *wj = reinterpret_cast<const float4 *>(G.Wx6b) + (size_t)cg * (2 * 2 * 4 * 64) + lane;
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int p = 0; p < 2; p++)
#pragma unroll
                    for (int ks = 0; ks < 4; ks++) { wf1[c][p][ks] = wi[((c * 2 + p) * 4 + ks) * 64]; wf2[c][p][ks] = wj[((c * 2 + p) * 4 + ks) * 64]; }
        }
        const float wsinv1 = A.w_sinv, wsinv2 = G.w_sinvb;
        f32x4 bias1[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, biasv[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int c = 0; c < 2; c++) {
            if (A.bias) { const float4 b = *reinterpret_cast<const float4 *>(A.bias + 32 * cg + 16 * c + 4 * q); bias1[c] = f32x4{b.x, b.y, b.z, b.w}; }
            if (G.biasb) { const float4 b = *reinterpret_cast<const float4 *>(G.biasb + 32 * cg + 16 * c + 4 * q); biasv[c] = f32x4{b.x, b.y, b.z, b.w}; }
        }
        stage_scale_shift();
        LDS_BARRIER();
        // BatchNorm b of this lane's 2 x 4 columns, applied to z_b = acc * wsinv1 + bias exactly as k_gemm_x6 stores it and as its
        // producers normalise what they read back (same operations in the same order: z_c is bit-identical to the two launches')
        f32x4 msc[2], msh[2];
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const float4 s4 = *reinterpret_cast<const float4 *>(s_bn2 + 32 * cg + 16 * c + 4 * q), h4 = *reinterpret_cast<const float4 *>(s_bn2 + HD + 32 * cg + 16 * c + 4 * q);
            msc[c] = f32x4{s4.x, s4.y, s4.z, s4.w}; msh[c] = f32x4{h4.x, h4.y, h4.z, h4.w};
        }
        float ts[2][4], tq[2][4];                                 // per-lane column sums of z_c (row m of the tiles multiplied so far)
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int i = 0; i < 4; i++) { ts[c][i] = 0.f; tq[c][i] = 0.f; }
        const unsigned frag_off = m * X6_ROWB + 16 * q;           // operand fragment (slot t, plane p, k-step ks): + t*XT + p*X6_PLANE + 64*ks
        unsigned char *s_tr = smem + X6F_OFF_TR + cg * 2 * X6_TRB; // this wave's two output transposition buffers
        // the products of four tiles' planes at `xa` with the fragments `wf`: fin(t, acc) takes each tile's two accumulators
        auto tiles4 = [&](const unsigned char *xa, const float4 (&wf)[2][2][4], int ntl, auto fin) __attribute__((always_inline)) {
            float4 xf[3][2];
#pragma unroll
            for (int p = 0; p < 2; p++) xf[0][p] = *reinterpret_cast<const float4 *>(xa + p * X6_PLANE);
#pragma unroll
            for (int p = 0; p < 2; p++) xf[1][p] = *reinterpret_cast<const float4 *>(xa + p * X6_PLANE + 64);
#pragma unroll
            for (int t = 0; t < 4; t++) {
                if (t >= ntl) break;
                f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int ks = 0; ks < 4; ks++) {
                    const int u = t * 4 + ks;
                    if (u + 2 < 16) {                             // a stale slot beyond the last tile is read but never used
                        const int tn = (u + 2) / 4, kn = (u + 2) % 4;
#pragma unroll
                        for (int p = 0; p < 2; p++) xf[(u + 2) % 3][p] = *reinterpret_cast<const float4 *>(xa + tn * XT + p * X6_PLANE + 64 * kn);
                    }
                    const float4 *x = xf[u % 3];
                    auto MH = [&](int wp, int xp) __attribute__((always_inline)) {
#pragma unroll
                        for (int c = 0; c < 2; c++)
                            acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, wf[c][wp][ks]), __builtin_bit_cast(h16x8, x[xp]), acc[c], 0, 0, 0);
                    };
                    MH(0, 1); MH(1, 0); MH(0, 0);                 // smallest terms first
                }
                MFMA_SETTLE2(acc[0], acc[1]);
                fin(t, acc);
            }
        };
        LDS_BARRIER();                                            // step 0: the producers fill input buffer 0
        for (int s = 1; s <= nsteps_c + 1; s++) {
            // ---- second product: group s-2 (its intermediate planes were written in step s-1), stores + sums of z_c
            if (s >= 2 && s - 2 < nsteps) {
                const int tb = first + 4 * (s - 2);
                const int ntl = last - tb < 4 ? last - tb : 4;
                tiles4(s_mid + (s & 1) * 4 * XT + frag_off, wf2, ntl, [&](int t, f32x4 (&acc)[2]) __attribute__((always_inline)) {
#pragma unroll
                    for (int c = 0; c < 2; c++) acc[c] = acc[c] * wsinv2 + biasv[c];
                    const int ptile = PT(tb + t);
                    const bool ok = ptile * 16 + m < A.N;
                    unsigned char *trb = s_tr + (t & 1) * X6_TRB;  // whole 128-byte lines per store instruction (k_gemm_x6)
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        const f32x4 v = acc[c];
                        *reinterpret_cast<float4 *>(trb + m * 144 + 16 * (4 * c + q)) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
                        for (int i = 0; i < 4; i++) { const float x = ok ? v[i] : 0.f; ts[c][i] += x; tq[c][i] = __builtin_fmaf(x, x, tq[c][i]); }
                    }
#pragma unroll
                    for (int i = 0; i < 2; i++) {
                        const int r8 = 8 * i + (lane >> 3);
                        const float4 v = *reinterpret_cast<const float4 *>(trb + r8 * 144 + 16 * (lane & 7));
                        *reinterpret_cast<float4 *>(A.out + ((size_t)ptile * 16 + r8) * HD + 32 * cg + 4 * (lane & 7)) = v;
                    }
                });
                if (((s - 1) & 3) == 0 || s - 1 == nsteps) {      // column sums: f32 partial sums of <= 16 values per lane -> f64
#pragma unroll
                    for (int c = 0; c < 2; c++)
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const float a = row_sum16(ts[c][i]), b = row_sum16(tq[c][i]);
                            if (m == 0) { atomicAdd(&s_stat[32 * cg + 16 * c + 4 * q + i], (double)a); atomicAdd(&s_stat[HD + 32 * cg + 16 * c + 4 * q + i], (double)b); }
                            ts[c][i] = 0.f; tq[c][i] = 0.f;
                        }
                }
            }
            // ---- first product: group s-1 (input planes of buffer (s-1)&1) -> bn_b + ReLU + split -> this wave's 32 columns of the
            //      group's intermediate planes (buffer (s-1)&1)
            if (s - 1 < nsteps) {
                const int tb = first + 4 * (s - 1);
                const int ntl = last - tb < 4 ? last - tb : 4;
                unsigned char *mdst = s_mid + ((s - 1) & 1) * 4 * XT + m * X6_ROWB + (32 * cg + 4 * q) * 2;
                tiles4(s_tiles + ((s - 1) & 1) * 4 * XT + frag_off, wf1, ntl, [&](int t, f32x4 (&acc)[2]) __attribute__((always_inline)) {
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        const f32x4 zb = acc[c] * wsinv1 + bias1[c];
                        const f32x2 ya = __builtin_elementwise_fma(f32x2{zb[0], zb[1]}, f32x2{msc[c][0], msc[c][1]}, f32x2{msh[c][0], msh[c][1]});
                        const f32x2 yb = __builtin_elementwise_fma(f32x2{zb[2], zb[3]}, f32x2{msc[c][2], msc[c][3]}, f32x2{msh[c][2], msh[c][3]});
                        const float v[4] = {fmaxf(ya[0], 0.f), fmaxf(ya[1], 0.f), fmaxf(yb[0], 0.f), fmaxf(yb[1], 0.f)};
                        uint2 p0, p1;
                        split2x4m(v, p0, p1);
                        *reinterpret_cast<uint2 *>(mdst + t * XT + 32 * c) = p0;
                        *reinterpret_cast<uint2 *>(mdst + t * XT + 32 * c + X6_PLANE) = p1;
                    }
                });
            }
            LDS_BARRIER();
        }
    }
    if (tid < 2 * HD) atomicAdd(&A.epi_stats[(blockIdx.x % STAT_REP) * 256 + tid], s_stat[tid]);
}
