// mtfjsp_headsx_body.h — the statements of k_headsx (csrc/mtfjsp_encoder.hip), included inside a kernel with `A` (HeadArgs) and `smem`
// in scope: k_headsx and k_headsx_gat3x (see mtfjsp_gat3x_body.h for why this is textual).
#ifndef HX_NLDS
#define HX_NLDS false                      // k_headsx_gat3x_headsx<., true>, machine part: X (the node rows) is in LDS, left there by the GAT part of this launch
#endif
#ifndef HX_VALUES_ONLY
#define HX_VALUES_ONLY 0                   // 1 (k_headsx_values): the critic values only — no scorer rows, no probabilities, no selection (the post-terminal forward pair keeps nothing else: Run.py:455-475)
#endif
    unsigned char *s_xs = smem;                                    // HCH tiles of 2 planes: X rows, then s1
    unsigned char *s_pp = s_xs + HCH * X2_TILE;                    // pooled planes
    unsigned char *s_op = s_pp + X2_TILE;                          // other planes
    unsigned char *s_c1p = s_op + X2_TILE;                         // c1 planes
    float *s_c2 = reinterpret_cast<float *>(s_c1p + X2_TILE);      // [16][HX_CLDA] f32
    float *s_u = s_c2 + 16 * HX_CLDA;                              // [16][128]
    float *s_part = s_u + HG * HD;                                 // HX_NPART (wave[, row quarter]) partial scores of HCH*16 rows
    float *s_score = s_part + HX_NPART * HCH * 16;                 // HG * 64
    float *s_wc2 = s_score + HG * 64;                              // 2 * 128
    float *s_vec = s_wc2 + 2 * HD;                                 // b0 | bc0 | bc1 | b1 | w2
    unsigned char *s_mask = reinterpret_cast<unsigned char *>(s_vec + 5 * HD);   // HG * 64
    int *s_gf = reinterpret_cast<int *>(s_mask + HG * 64);         // [512] gather_from of the first 512 scorer rows (the selection's task per row)
    int *s_pm = s_gf + 512;                                        // [512] machine of that task's job predecessor (m_fea1)
    const int tid = BODY_TID, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int m = lane & 15, q = lane >> 4;
    const int col4 = 16 * wave + 4 * q;                            // this lane's 4 output columns
#ifdef MTFJSP_STAMP
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_last)::"memory");
#endif
    const int R = A.R;
    const unsigned invR = (unsigned)((0x100000000ull + (unsigned)R - 1) / (unsigned)R);
    const int sr = tid >> 5, sc4 = (tid & 31) * 4;                  // staging: thread -> (row sr of 16, 4 columns)
    const int xoff = m * X6_ROWB + 16 * q;                          // operand fragment of (plane p, k-step ks): + p*X6_PLANE + 64*ks
    const float b2 = A.b2[0];
    // EVERY request of the kernel's first phase goes out before anything is waited for, unconditionally (indices clamped or
    // masked into range, values discarded where they are used): the small vectors, the BatchNorm sums of X, the X rows of the
    // first chunk (gather indices first, rows behind them), the weights of phase A.  (Round 2's source staged the small vectors
    // through LDS first and put every request behind its own `if`: in the ISA that was a chain of s_waitcnt vmcnt(0) — four to
    // twelve round trips, one after the other, before the first product.)
    const float r_wc2 = A.wc2[tid & (2 * HD - 1)];
    const int tc = tid & (HD - 1);
    const float r_v0 = A.b0[tc], r_v1 = A.bc0[tc], r_v2 = A.bc1[tc], r_v3 = A.b1[tc], r_v4 = A.w2[tc];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    float xs0 = 1.f, xs1 = 1.f, xs2 = 1.f, xs3 = 1.f, xh0 = 0.f, xh1 = 0.f, xh2 = 0.f, xh3 = 0.f;   // X scale / shift of this thread's 4 columns
    double bsu[STAT_REP], bsq[STAT_REP];
    float bga = 0.f, bbe = 0.f;
    if (A.xbn_stats) {
#if !HX_XCHG
#pragma unroll
        for (int r = 0; r < STAT_REP; r++) { bsu[r] = A.xbn_stats[r * 256 + tc]; bsq[r] = A.xbn_stats[r * 256 + HD + tc]; }
#endif
        bga = A.xbn_gamma[tc]; bbe = A.xbn_beta[tc];
    }
    {   // one workgroup per group of 16 instances (a persistent loop here makes the compiler hoist ~200 loop-invariant
        // 64-bit weight addresses into registers and spill them)
        const int g0 = blockIdx.x * A.hg;
        const int ng = (A.B - g0) < A.hg ? (A.B - g0) : A.hg;
        const int nrows = ng * R;
        // scorer row `grow` of the group (instance grow / R, candidate / machine grow % R) -> its source row in X
        auto xrow = [&](int grow) __attribute__((always_inline)) -> const float * {
            if (!A.xgather) return A.X + ((size_t)g0 * R + grow) * HD + sc4;
            const int il = (int)__umulhi((unsigned)grow, invR);
            return A.X + ((size_t)(g0 + il) * A.xT + A.xgather[(size_t)g0 * R + grow]) * HD + sc4;
        };
#if HX_XCHG
        // Behind the in-launch exchange nothing may wait for memory that could have been requested before it.  X is this workgroup's
        // own (its GAT passes wrote it), the BatchNorm in front of the pool has no ReLU (ac:434-444), so the pooled embedding is
        // scale * (raw row mean) + shift: the raw mean of this thread's instance and the other actor's embedding are formed now.
        // R <= 8 (host: mheads_fusable); rows beyond R are clamped and weighted 0 — no request behind a branch.
        // R <= HCH (round 5, end): the scorer's X rows are staged by INSTANCE — thread (sr, 4 columns) takes rows sr R + t — so the rows it
        // stages are the rows it pools: no second set of requests, and no wait for them in front of the exchange (by_inst).
        const bool by_inst = R <= HCH;
        float4 praw = make_float4(0.f, 0.f, 0.f, 0.f), xo_pre;
        xo_pre = *reinterpret_cast<const float4 *>(A.other + (size_t)(g0 + (sr < ng ? sr : ng - 1)) * HD + sc4);
        if (!by_inst) {
            const int ic = sr < ng ? sr : ng - 1;
            const float *src = A.X + (size_t)(g0 + ic) * R * HD + sc4;
            float4 rv[8];
#pragma unroll
            for (int r = 0; r < 8; r++) rv[r] = *reinterpret_cast<const float4 *>(src + (size_t)(r < R ? r : R - 1) * HD);
            const float ir = 1.0f / (float)R;
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const float wgt = r < R ? ir : 0.f;
                praw.x = fmaf(rv[r].x, wgt, praw.x); praw.y = fmaf(rv[r].y, wgt, praw.y); praw.z = fmaf(rv[r].z, wgt, praw.z); praw.w = fmaf(rv[r].w, wgt, praw.w);
            }
            asm volatile("" : "+v"(praw.x), "+v"(praw.y), "+v"(praw.z), "+v"(praw.w));   // (formed here: the 8 row registers are free before the weight requests go out)
        }
#else
        const bool by_inst = false;
#endif
        const unsigned char r_mask = A.mask[(size_t)g0 * R + (tid < nrows ? tid : nrows - 1)];   // the group's action mask (rows beyond 512: fetched where they are stored)
#if !HX_XCHG
        // the two embedding rows of this thread's instance (the stage below used to request them where it needed them: a memory round trip
        // on every workgroup's critical path, round 5); without `pooled` (it is formed here from X) some valid row is fetched and ignored
        const float4 xp_top = *reinterpret_cast<const float4 *>((A.pooled ? A.pooled : A.other) + (size_t)(g0 + (sr < ng ? sr : ng - 1)) * HD + sc4);
        const float4 xo_top = *reinterpret_cast<const float4 *>(A.other + (size_t)(g0 + (sr < ng ? sr : ng - 1)) * HD + sc4);
#endif
        // The selection at the end walks a chain of dependent requests — the picked row's task (gather_from), that task's job predecessor's
        // machine (link), then the task's rows — behind one another, ~0.6 us each on every workgroup's critical path.  The first two links
        // are walked for EVERY scorer row instead, up here where requests are free: index now, predecessor's machine behind the first
        // barrier, both parked in LDS (rows beyond 512: the old way).  Pointers are chosen, not branched around (a request behind an `if`
        // makes the later waits drain everything): without gather_from / m_fea1 some valid word of X is fetched and ignored.
        const int r_gf = (A.gather_from ? A.gather_from + (size_t)g0 * R : reinterpret_cast<const int *>(A.X))[tid < nrows ? tid : nrows - 1];
        h16x8 wA[2][4], wB[2][4], wC[2][4];
        const float sW0 = A.sW0, sW1 = A.sW1, sWc0 = A.sWc0, sWc1 = A.sWc1;   // 1 / scale of the weight images
        WCOLX(wA, A.W0x, 1);                                        // Wb
        WCOLX(wB, A.W0x, 2);                                        // Wc
        WCOLX(wC, A.Wc0x, 0);
        // (the X rows behind the weights: phase A needs the weights first and covers the rows' arrival)
        float4 xr[HCH];                                             // X rows of the first chunk: requested now, committed after phase A
        if (HX_VALUES_ONLY) { for (int t = 0; t < HCH; t++) xr[t] = make_float4(0.f, 0.f, 0.f, 0.f); }
        else {   // (rows beyond the group's are clamped to its last one here and zeroed by xnorm() where they are used)
            int gi[HCH];
#pragma unroll
            for (int t = 0; t < HCH; t++) {
                const int grow = by_inst ? sr * R + t : t * 16 + sr, gc = (grow < nrows && (!by_inst || t < R)) ? grow : nrows - 1;
                gi[t] = A.xgather ? A.xgather[(size_t)g0 * R + gc] : gc;
            }
#pragma unroll
            for (int t = 0; t < HCH; t++) {
                const int grow = by_inst ? sr * R + t : t * 16 + sr, gc = (grow < nrows && (!by_inst || t < R)) ? grow : nrows - 1;
                const int il = (int)__umulhi((unsigned)gc, invR);
                if (HX_NLDS) {                                      // node row gc of the workgroup = row gc & 7 of GAT tile gc >> 3: the first half of that tile's buffer (mtfjsp_gat3x_body.h)
                    xr[t] = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(smem + 8 * 2 * 4 * 64 * 16) + ((gc >> 3) * 16 + (gc & 7)) * HD + sc4);
                    continue;
                }
                const float *src = A.xgather ? A.X + ((size_t)(g0 + il) * A.xT + gi[t]) * HD + sc4 : A.X + ((size_t)g0 * R + gc) * HD + sc4;
                xr[t] = *reinterpret_cast<const float4 *>(src);
            }
        }
#ifdef HX_TOP_HOOK
        HX_TOP_HOOK                                                 // (requests of the including kernel that a later part consumes: behind this phase's own)
#endif
        // (everything is in flight) the uniform number of this thread's instance's draw (pick_action): ten Philox rounds that need no memory
        const float u_pre = A.sample_mode == 1 ? pick_uniform(g0 + (tid >> 4), A.seed, A.counter) : 0.f;
        // now the stores that only needed the first few words (HX_NLDS: behind the first barrier — they lie where GAT tile 6 left node rows
        // that other waves may not have read yet)
        if (!HX_NLDS) {
            if (tid < 2 * HD) s_wc2[tid] = r_wc2;
            if (tid < HD) { s_vec[tid] = r_v0; s_vec[HD + tid] = r_v1; s_vec[2 * HD + tid] = r_v2; s_vec[3 * HD + tid] = r_v3; s_vec[4 * HD + tid] = r_v4; }
        }
        if (A.zero_stats && blockIdx.x == 0) for (int i = tid; i < A.zero_count; i += 512) A.zero_stats[i] = 0.0;
        if (A.zero_stats2 && blockIdx.x == 0) for (int i = tid; i < A.zero_count2; i += 512) A.zero_stats2[i] = 0.0;
        if (A.xbn_stats) {
#if HX_XCHG
            // The statistics of X are this launch's own: every workgroup has added its (sum | sum of squares) per column to the
            // count-carrying words of its dispatch group (the end of mtfjsp_gat3x_body.h) and now collects the eight groups' words —
            // a word that carries its group's size in the count field is complete (mtfjsp_gin_resident.h).  Everything this kernel
            // requests that does not depend on other workgroups is already in flight.  Thread t < 256 takes value t (sum of column t,
            // or sum of squares of column t - 128): 8 loads per poll.  The words of the wide-range set are empty unless a
            // contribution left the fine set's range: they are looked at from the fourth poll on.
            double *s_xch = reinterpret_cast<double *>(s_c2);       // (free until phase B)
            if (tid < 2 * HD) {
                (void)bsu; (void)bsq;
                const unsigned long long *w0 = XA.words + tid;      // [set fine | wide][group 8][sum | sumsq][column 128]
                const unsigned nblk = XA.nblk;
                unsigned long long sf = 0, sc = 0;
                unsigned nf = 0, nc = 0;
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                H3T_RT(1);
                for (int it = 0;; it++) {
                    unsigned long long vf[8], vc[8];
                    const bool wide = it >= 3;
#pragma unroll
                    for (int j = 0; j < 8; j++) vf[j] = __hip_atomic_load(w0 + j * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int j = 0; j < 8; j++) vc[j] = wide ? __hip_atomic_load(w0 + (8 + j) * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                    bool done = true;
                    sf = sc = 0; nf = nc = 0;
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const unsigned want = (nblk >> 3) + ((unsigned)j < (nblk & 7u) ? 1u : 0u);
                        const unsigned c0 = (unsigned)(vf[j] >> 58), d0 = (unsigned)(vc[j] >> 58);
                        nf += c0; sf += vf[j] & GR_FIX_PAYLOAD; nc += d0; sc += vc[j] & GR_FIX_PAYLOAD;
                        done = done && c0 + d0 == want;
                    }
                    if (it == 0) H3T_RT(2); else if (it == 1) H3T_RT(3); else if (it == 2) H3T_RT(4);
                    if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;                   // (waves 0..3 poll, each until all of its values are in)
                    if (__builtin_amdgcn_s_memrealtime() - t0 > 400000ull) {                 // 4 ms at 100 MHz: not all workgroups are resident
                        if (XA.fail) __hip_atomic_store(XA.fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (host-mapped: the next forward entry / mtfjsp_encoder_check reports it)
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
                X3_RT(4);
                s_xch[tid] = __builtin_ldexp((double)((long long)sf - (long long)nf * (long long)GR_FIX_BIAS), -20) +
                             __builtin_ldexp((double)((long long)sc - (long long)nc * (long long)GR_FIX_BIAS), -6);
            }
            LDS_BARRIER();
            if (HX_NLDS) {                                          // (every wave has its node rows in registers: LDS_BARRIER waits for the LDS reads first)
                if (tid < 2 * HD) s_wc2[tid] = r_wc2;
                if (tid < HD) { s_vec[tid] = r_v0; s_vec[HD + tid] = r_v1; s_vec[2 * HD + tid] = r_v2; s_vec[3 * HD + tid] = r_v3; s_vec[4 * HD + tid] = r_v4; }
            }
#endif
            if (tid < HD) {                                         // stage_bn() from the registers requested above; s_u is free until phase A
                double su = 0, sq = 0;
#if HX_XCHG
                su = s_xch[tid]; sq = s_xch[HD + tid];
#else
#pragma unroll
                for (int r = 0; r < STAT_REP; r++) { su += bsu[r]; sq += bsq[r]; }
                if (A.range_flag && (su != su || sq != sq)) __hip_atomic_store(A.range_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
                const double mean = su * A.xbn_inv_rows;
                double var = sq * A.xbn_inv_rows - mean * mean;
                if (var < 0) var = 0;
                const float rstd = 1.0f / sqrtf((float)(var + BN_EPS));
                const float sc = rstd * bga;
                s_u[tid] = sc;
                s_u[HD + tid] = bbe - (float)mean * sc;
            }
            LDS_BARRIER();
            xs0 = s_u[sc4]; xs1 = s_u[sc4 + 1]; xs2 = s_u[sc4 + 2]; xs3 = s_u[sc4 + 3];
            xh0 = s_u[HD + sc4]; xh1 = s_u[HD + sc4 + 1]; xh2 = s_u[HD + sc4 + 2]; xh3 = s_u[HD + sc4 + 3];
            LDS_BARRIER();
#if HX_XCHG
            X3_RT(5);
#endif
        }
        STAMP(6); H3_RT(6);
        {
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            float4 xp = z;
#if HX_XCHG
            if (by_inst) {                                          // the raw mean of the instance's R rows = of the rows this thread stages
                const float ir = 1.0f / (float)R;
#pragma unroll
                for (int t = 0; t < HCH; t++) {
                    const float wgt = t < R ? ir : 0.f;
                    praw.x = fmaf(xr[t].x, wgt, praw.x); praw.y = fmaf(xr[t].y, wgt, praw.y); praw.z = fmaf(xr[t].z, wgt, praw.z); praw.w = fmaf(xr[t].w, wgt, praw.w);
                }
            }
            if (sr < ng) {
                xp = make_float4(fmaf(praw.x, xs0, xh0), fmaf(praw.y, xs1, xh1), fmaf(praw.z, xs2, xh2), fmaf(praw.w, xs3, xh3));
                *reinterpret_cast<float4 *>(A.pooled_out + (size_t)(g0 + sr) * HD + sc4) = xp;
            }
#else
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
            } else if (sr < ng) xp = xp_top;
#endif
            STAMP(7); H3_RT(7);
#if HX_XCHG
            const float4 xo = sr < ng ? xo_pre : z;
#else
            const float4 xo = sr < ng ? xo_top : z;
#endif
            {
                const float vp[4] = {xp.x, xp.y, xp.z, xp.w}, vo[4] = {xo.x, xo.y, xo.z, xo.w};
                uint2 a0, a1, b0, b1;
                split2x4(vp, a0, a1); split2x4(vo, b0, b1);
                unsigned char *dp = s_pp + sr * X6_ROWB + (tid & 31) * 8, *dq = s_op + sr * X6_ROWB + (tid & 31) * 8;
                *reinterpret_cast<uint2 *>(dp) = a0; *reinterpret_cast<uint2 *>(dp + X6_PLANE) = a1;
                *reinterpret_cast<uint2 *>(dq) = b0; *reinterpret_cast<uint2 *>(dq + X6_PLANE) = b1;
            }
        }
        if (tid < nrows) s_mask[tid] = r_mask;                      // (requested with the first phase's other loads)
        short r_lk;
        {   // second link of the selection's chain: the machine of the job predecessor of row tid's task
            const int T_ = A.mf_on ? A.mf.T : 1, il = (int)__umulhi((unsigned)(tid < nrows ? tid : 0), invR);
            int a = A.gather_from ? r_gf : 0;
            if (a < 0 || a >= T_) a = 0;
            const size_t row = (size_t)(g0 + il) * T_ + a;
            r_lk = (A.mf_on ? reinterpret_cast<const short *>(A.mf.link) : reinterpret_cast<const short *>(A.X))[(A.mf_on && row > 0 ? row - 1 : 0) * 8 + 4];
            s_gf[tid] = r_gf;
        }
        for (int i = tid + 512; i < nrows; i += 512) s_mask[i] = A.mask[(size_t)g0 * R + i];
        LDS_BARRIER();
        STAMP(0); H3_RT(0);
        auto xnorm = [&](float4 v, bool valid) __attribute__((always_inline)) {
            if (!valid) return make_float4(0.f, 0.f, 0.f, 0.f);
            float y0 = fmaf(v.x, xs0, xh0), y1 = fmaf(v.y, xs1, xh1), y2 = fmaf(v.z, xs2, xh2), y3 = fmaf(v.w, xs3, xh3);
            if (A.xrelu) { y0 = fmaxf(y0, 0.f); y1 = fmaxf(y1, 0.f); y2 = fmaxf(y2, 0.f); y3 = fmaxf(y3, 0.f); }
            return make_float4(y0, y1, y2, y3);
        };
        // one 16-row tile (planes at `tp`) x this wave's column block: 12 products on two chains, fragments of the next k-step in flight
        auto tile_x6 = [&](const unsigned char *tp, const h16x8 (&wv)[2][4]) __attribute__((always_inline)) -> f32x4 {
            f32x4 aA = zero4, aB = zero4;
            h16x8 xv[4][2];                                         // all 8 fragments of the tile requested at once: 3 matrix instructions do not cover an LDS round trip
#pragma unroll
            for (int ks = 0; ks < 4; ks++)
#pragma unroll
                for (int p = 0; p < 2; p++) xv[ks][p] = *reinterpret_cast<const h16x8 *>(tp + xoff + p * X6_PLANE + 64 * ks);
#pragma unroll
            for (int ks = 0; ks < 4; ks++) X6_STEP(aA, aB, wv, xv[ks], ks);
            MFMA_SETTLE2(aA, aB);                                   // (margin behind the matrix pipe's write-back: see the macro)
            return aA + aB;
        };
        // a lane's 4 values of row m -> the two planes of a tile
        auto put_planes = [&](unsigned char *tp, const float (&v)[4]) __attribute__((always_inline)) {
            uint2 p0, p1;
            split2x4(v, p0, p1);
            unsigned char *d = tp + m * X6_ROWB + col4 * 2;
            *reinterpret_cast<uint2 *>(d) = p0; *reinterpret_cast<uint2 *>(d + X6_PLANE) = p1;
        };
        // ---- phase A: u = Wb pooled + Wc other + b0 ; c1 = tanh(Wc0 pooled + bc0)   (rows = the group's 16 instances)
        {
            const f32x4 au = tile_x6(s_pp, wA) + tile_x6(s_op, wB);
            const f32x4 ac = tile_x6(s_pp, wC);
            WCOLX(wA, A.Wc1x, 0);                                   // requested now, used in phase B
            WCOLX(wB, A.W0x, 0);                                    // Wa
            WCOLX(wC, A.W1x, 0);                                    // phase C
            const float4 b0v = *reinterpret_cast<const float4 *>(s_vec + col4), bc0v = *reinterpret_cast<const float4 *>(s_vec + HD + col4);
            *reinterpret_cast<float4 *>(s_u + m * HD + col4) = make_float4(fmaf(au[0], sW0, b0v.x), fmaf(au[1], sW0, b0v.y), fmaf(au[2], sW0, b0v.z), fmaf(au[3], sW0, b0v.w));
            const float c1v[4] = {fast_tanh(fmaf(ac[0], sWc0, bc0v.x)), fast_tanh(fmaf(ac[1], sWc0, bc0v.y)), fast_tanh(fmaf(ac[2], sWc0, bc0v.z)), fast_tanh(fmaf(ac[3], sWc0, bc0v.w))};
            put_planes(s_c1p, c1v);
        }
        s_pm[tid] = (int)r_lk;
        STAMP(1); H3_RT(1);
#if HX_VALUES_ONLY
        {   // c2 = tanh(Wc1 c1 + bc1), then the value head: the statements of the full kernel's first chunk (phase B's second half, the
            // head behind phase C), nothing of the scorer
            LDS_BARRIER();                                          // c1 planes are complete
            const f32x4 a0 = tile_x6(s_c1p, wA);
            const float4 bc1v = *reinterpret_cast<const float4 *>(s_vec + 2 * HD + col4);
            *reinterpret_cast<float4 *>(s_c2 + m * HX_CLDA + col4) =
                make_float4(fast_tanh(fmaf(a0[0], sWc1, bc1v.x)), fast_tanh(fmaf(a0[1], sWc1, bc1v.y)), fast_tanh(fmaf(a0[2], sWc1, bc1v.z)), fast_tanh(fmaf(a0[3], sWc1, bc1v.w)));
            LDS_BARRIER();                                          // c2 is complete
            if (tid < 256) {
                const int r = tid >> 4, part = tid & 15;
                float p0 = 0.f, p1 = 0.f;
                const float4 xa_ = *reinterpret_cast<const float4 *>(s_c2 + r * HX_CLDA + part * 8), xb_ = *reinterpret_cast<const float4 *>(s_c2 + r * HX_CLDA + part * 8 + 4);
                const float4 wa0 = *reinterpret_cast<const float4 *>(s_wc2 + part * 8), wb0 = *reinterpret_cast<const float4 *>(s_wc2 + part * 8 + 4);
                const float4 wa1 = *reinterpret_cast<const float4 *>(s_wc2 + HD + part * 8), wb1 = *reinterpret_cast<const float4 *>(s_wc2 + HD + part * 8 + 4);
                const float xs[8] = {xa_.x, xa_.y, xa_.z, xa_.w, xb_.x, xb_.y, xb_.z, xb_.w};
                const float w0s[8] = {wa0.x, wa0.y, wa0.z, wa0.w, wb0.x, wb0.y, wb0.z, wb0.w}, w1s[8] = {wa1.x, wa1.y, wa1.z, wa1.w, wb1.x, wb1.y, wb1.z, wb1.w};
#pragma unroll
                for (int k = 0; k < 8; k++) { p0 = fmaf(xs[k], w0s[k], p0); p1 = fmaf(xs[k], w1s[k], p1); }
                p0 = row_sum16(p0); p1 = row_sum16(p1);
                if (part == 0 && r < ng) {
                    const float v0 = p0 + A.bc2[0], v1 = p1 + A.bc2[1];
                    A.value[(size_t)(g0 + r) * 2] = v0; A.value[(size_t)(g0 + r) * 2 + 1] = v1;
                    if (A.range_flag && (v0 != v0 || v1 != v1)) __hip_atomic_store(A.range_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
            (void)xnorm; (void)b2; (void)invR; (void)xoff;
        }
#else
        const int ntl = (nrows + 15) >> 4;                          // 16-row tiles of this group (R for a full group of 16 instances)
        for (int tb = 0; tb < ntl; tb += HCH) {
            const int nt = (ntl - tb) < HCH ? (ntl - tb) : HCH;
            // ---- X rows of this chunk -> planes (rows beyond the group's are zero)
#pragma unroll
            for (int t = 0; t < HCH; t++) {
                if (tb > 0) {
                    const int grow = (tb + t) * 16 + sr;
                    xr[t] = (tb + t < ntl && grow < nrows) ? *reinterpret_cast<const float4 *>(xrow(grow)) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                const int srow = by_inst ? sr * R + t : (tb + t) * 16 + sr;   // (by_inst: one chunk, tb = 0)
                const float4 xv4 = xnorm(xr[t], srow < nrows);
                const float v[4] = {xv4.x, xv4.y, xv4.z, xv4.w};
                uint2 p0, p1;
                split2x4(v, p0, p1);
                unsigned char *d = by_inst ? s_xs + (srow >> 4) * X2_TILE + (srow & 15) * X6_ROWB + (tid & 31) * 8 : s_xs + t * X2_TILE + sr * X6_ROWB + (tid & 31) * 8;
                if (!by_inst || t < R) { *reinterpret_cast<uint2 *>(d) = p0; *reinterpret_cast<uint2 *>(d + X6_PLANE) = p1; }
            }
            LDS_BARRIER();                                          // X planes, u and c1 are complete
            STAMP(2); H3_RT(2);
            // ---- phase B: Wa x for every tile of the chunk (accumulators held); first chunk: c2 = tanh(Wc1 c1 + bc1)
            f32x4 accb[HCH];
#pragma unroll
            for (int t = 0; t < HCH; t++) accb[t] = t < nt ? tile_x6(s_xs + t * X2_TILE, wB) : zero4;
            H3S_RT(0);
            if (tb == 0) {
                const f32x4 a0 = tile_x6(s_c1p, wA);
                const float4 bc1v = *reinterpret_cast<const float4 *>(s_vec + 2 * HD + col4);
                *reinterpret_cast<float4 *>(s_c2 + m * HX_CLDA + col4) =
                    make_float4(fast_tanh(fmaf(a0[0], sWc1, bc1v.x)), fast_tanh(fmaf(a0[1], sWc1, bc1v.y)), fast_tanh(fmaf(a0[2], sWc1, bc1v.z)), fast_tanh(fmaf(a0[3], sWc1, bc1v.w)));
            }
            H3S_RT(1);
            LDS_BARRIER();                                          // every wave is done with the X planes: s1 overwrites them
            H3S_RT(2);
            // s1 = tanh(Wa x + u[instance]) -> planes
#pragma unroll
            for (int t = 0; t < HCH; t++) {
                if (t < nt) {
                    const int grow = (tb + t) * 16 + m;
                    const int i0 = grow < nrows ? (int)__umulhi((unsigned)grow, invR) : 0;
                    const float4 uv = *reinterpret_cast<const float4 *>(s_u + i0 * HD + col4);
                    const float sv[4] = {fast_tanh(fmaf(accb[t][0], sW0, uv.x)), fast_tanh(fmaf(accb[t][1], sW0, uv.y)), fast_tanh(fmaf(accb[t][2], sW0, uv.z)), fast_tanh(fmaf(accb[t][3], sW0, uv.w))};
                    put_planes(s_xs + t * X2_TILE, sv);
                }
            }
            H3S_RT(3);
            LDS_BARRIER();                                          // s1 planes and c2 are complete
            STAMP(3); H3_RT(3);
            // ---- phase C: s2 = tanh(W1 s1 + b1) ; partial scores of this wave's 16 columns
            {
                const float4 b1v = *reinterpret_cast<const float4 *>(s_vec + 3 * HD + col4), w2v = *reinterpret_cast<const float4 *>(s_vec + 4 * HD + col4);
#pragma unroll 1
                for (int t = 0; t < HCH; t++) {                     // (rolled; round 5 measured unroll 2 and 6, and two / three tile products interleaved per wave: no change)
                    if (t < nt) {
                        const f32x4 a0 = tile_x6(s_xs + t * X2_TILE, wC);
                        float v = fast_tanh(fmaf(a0[0], sW1, b1v.x)) * w2v.x;
                        v = fmaf(fast_tanh(fmaf(a0[1], sW1, b1v.y)), w2v.y, v);
                        v = fmaf(fast_tanh(fmaf(a0[2], sW1, b1v.z)), w2v.z, v);
                        v = fmaf(fast_tanh(fmaf(a0[3], sW1, b1v.w)), w2v.w, v);
#if HX_NPART == 32
                        s_part[(wave * 4 + q) * (HCH * 16) + t * 16 + m] = v;    // (the four row quarters are added with the waves below: it was two ds_bpermute round trips per tile)
#else
                        v += __shfl_xor(v, 16);
                        v += __shfl_xor(v, 32);
                        if (q == 0) s_part[wave * (HCH * 16) + t * 16 + m] = v;
#endif
                    }
                }
            }
            H3S_RT(4);
            if (tb == 0 && tid < 256) {   // value head: the 16 lanes of a DPP row per instance row, 8 columns each, both outputs (round 5: it was 32 lanes per row and
                                          // five ds_bpermute steps per output, one after the other)
                const int r = tid >> 4, part = tid & 15;
                float p0 = 0.f, p1 = 0.f;
                const float4 xa_ = *reinterpret_cast<const float4 *>(s_c2 + r * HX_CLDA + part * 8), xb_ = *reinterpret_cast<const float4 *>(s_c2 + r * HX_CLDA + part * 8 + 4);
                const float4 wa0 = *reinterpret_cast<const float4 *>(s_wc2 + part * 8), wb0 = *reinterpret_cast<const float4 *>(s_wc2 + part * 8 + 4);
                const float4 wa1 = *reinterpret_cast<const float4 *>(s_wc2 + HD + part * 8), wb1 = *reinterpret_cast<const float4 *>(s_wc2 + HD + part * 8 + 4);
                const float xs[8] = {xa_.x, xa_.y, xa_.z, xa_.w, xb_.x, xb_.y, xb_.z, xb_.w};
                const float w0s[8] = {wa0.x, wa0.y, wa0.z, wa0.w, wb0.x, wb0.y, wb0.z, wb0.w}, w1s[8] = {wa1.x, wa1.y, wa1.z, wa1.w, wb1.x, wb1.y, wb1.z, wb1.w};
#pragma unroll
                for (int k = 0; k < 8; k++) { p0 = fmaf(xs[k], w0s[k], p0); p1 = fmaf(xs[k], w1s[k], p1); }
                p0 = row_sum16(p0); p1 = row_sum16(p1);
                if (part == 0 && r < ng) {
                    const float v0 = p0 + A.bc2[0], v1 = p1 + A.bc2[1];
                    A.value[(size_t)(g0 + r) * 2] = v0; A.value[(size_t)(g0 + r) * 2 + 1] = v1;
                    if (A.range_flag && (v0 != v0 || v1 != v1)) __hip_atomic_store(A.range_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
            H3S_RT(5);
            LDS_BARRIER();
            if (tid < nt * 16) {
                const int grow = tb * 16 + tid;
                float v = b2;
#pragma unroll
                for (int w = 0; w < HX_NPART; w++) v += s_part[w * (HCH * 16) + tid];
                if (grow < nrows) s_score[grow] = v * A.scale;
            }
            H3S_RT(6);
            LDS_BARRIER();                                          // planes / s_part are reused by the next chunk
            STAMP(4); H3_RT(4);
        }
#if HX_XCHG
        X3_RT(6);
#endif
#ifdef HX_IDLE_HOOK
        // the selection below keeps waves 0-3 (16 lanes per instance) busy for ~4 us while waves 4-7 have nothing to do, and the scorer's
        // planes (the first 78 KB of LDS) are dead: the including kernel may give the idle waves work that touches neither s_score,
        // s_mask nor s_part (k_headsx_gat3x_headsx: the GAT part's weight staging)
        if (tid >= 256) { HX_IDLE_HOOK }
#endif
        // ---- masked softmax per instance (ac:266-278 / ac:487-491): 16 lanes per instance; optional action selection
        // (Round 4 requested gather_from[row] and the job predecessor's machine of EVERY scorer row at the top of the kernel, so that the
        // selection below would not fetch them behind one another — index, link, then the task's rows: the job heads' launch did not
        // change (46.2 against 46.0 us) and the machine heads, which only paid for the extra requests, lost 0.4 us: not kept.)
        {
            const int r0 = tid >> 4, l = tid & 15;
            // R <= 16 (round 5): lane l of the instance's 16 holds row l, and everything that went through LDS or a ds_bpermute per
            // step — the two butterflies of the softmax, the sequential scans of the draw by ONE lane, the hand-over of the picked
            // index — stays in registers: DPP partners within the row, the same additions in the same order (the same bits).
            const bool rowwise = R <= 16;
            float pr_l = 0.f;                                       // this lane's probability (rowwise)
            int pick_l = 0;
            if (r0 < ng && rowwise) {
                auto dpp = [&](float x, auto Ctrl) __attribute__((always_inline)) {
                    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(Ctrl)::value, 0xF, 0xF, true));
                };
                // partner lane ^ 8, ^ 4, ^ 2, ^ 1 (the order of __shfl_xor's butterfly): row rotations by 8 and by 4 / 12, quad permutes
                auto x8 = [&](float x) __attribute__((always_inline)) { return dpp(x, std::integral_constant<int, 0x128>{}); };
                auto x4 = [&](float x) __attribute__((always_inline)) {
                    const float up = dpp(x, std::integral_constant<int, 0x12C>{}), dn = dpp(x, std::integral_constant<int, 0x124>{});   // row_ror:12 = lane + 4, row_ror:4 = lane - 4
                    return (l & 4) ? dn : up;
                };
                auto x2 = [&](float x) __attribute__((always_inline)) { return dpp(x, std::integral_constant<int, 0x4E>{}); };
                auto x1 = [&](float x) __attribute__((always_inline)) { return dpp(x, std::integral_constant<int, 0xB1>{}); };
                const bool on = l < R && !s_mask[r0 * R + (l < R ? l : 0)];
                const float sc = s_score[r0 * R + (l < R ? l : 0)];
                float mx = on ? sc : -INFINITY;
                mx = fmaxf(mx, x8(mx)); mx = fmaxf(mx, x4(mx)); mx = fmaxf(mx, x2(mx)); mx = fmaxf(mx, x1(mx));
                const float ex = on ? __expf(sc - mx) : 0.f;
                float sum = ex;
                sum += x8(sum); sum += x4(sum); sum += x2(sum); sum += x1(sum);
                pr_l = on ? ex / sum : 0.f;
                if (l < R) {
                    A.prob[(size_t)(g0 + r0) * R + l] = pr_l;
                    if (A.range_flag && pr_l != pr_l) __hip_atomic_store(A.range_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                H3S_RT(7);
                if (A.sample_mode) {
                    // pick_action_u's scans as a chain along the row: lane k takes lane k-1's running sum (row_shr:1) + its own p
                    float run = pr_l;
#pragma unroll 1
                    for (int k = 1; k < R; k++) {
                        const float prev = dpp(run, std::integral_constant<int, 0x111>{});
                        if (l == k) run = prev + pr_l;
                    }
                    const float tot = __shfl(run, (lane & 48) + R - 1);            // (adding the masked rows' zeros is exact: the last row's sum is the total)
                    const float thr = u_pre * tot;
                    const bool pos = l < R && pr_l > 0.f;
                    const unsigned hit = (unsigned)(__builtin_amdgcn_ballot_w64(pos && thr < run) >> (lane & 48)) & 0xffffu;
                    const unsigned any = (unsigned)(__builtin_amdgcn_ballot_w64(pos) >> (lane & 48)) & 0xffffu;
                    pick_l = hit ? __builtin_ctz(hit) : any ? 31 - __builtin_clz(any) : 0;
                    if (A.sample_mode == 2) {                                       // greedy: the first row holding the maximum
                        float best = l < R ? pr_l : -INFINITY;
                        best = fmaxf(best, x8(best)); best = fmaxf(best, x4(best)); best = fmaxf(best, x2(best)); best = fmaxf(best, x1(best));
                        const unsigned eq = (unsigned)(__builtin_amdgcn_ballot_w64(l < R && pr_l == best) >> (lane & 48)) & 0xffffu;
                        pick_l = eq ? __builtin_ctz(eq) : 0;
                    }
                }
            }
            if (r0 < ng) {
              if (!rowwise) {
                float mx = -INFINITY;
                for (int r = l; r < R; r += 16) if (!s_mask[r0 * R + r]) mx = fmaxf(mx, s_score[r0 * R + r]);
                for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
                float sum = 0.f;
                for (int r = l; r < R; r += 16) if (!s_mask[r0 * R + r]) sum += __expf(s_score[r0 * R + r] - mx);
                for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
                for (int r = l; r < R; r += 16) {
                    const float pr = s_mask[r0 * R + r] ? 0.f : __expf(s_score[r0 * R + r] - mx) / sum;
                    A.prob[(size_t)(g0 + r0) * R + r] = pr;
                    if (A.range_flag && pr != pr) __hip_atomic_store(A.range_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    s_score[r0 * R + r] = pr;                           // lanes of one wave: visible to lane l == 0 below
                }
              }
                if (A.sample_mode) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    const float p_pick = rowwise ? __shfl(pr_l, (lane & 48) + pick_l) : 0.f;
                    if (l == 0) {
                        const int b = g0 + r0;
                        const int pick = rowwise ? pick_l : pick_action_u(s_score + r0 * R, R, A.sample_mode == 2, u_pre);
                        A.idx_out[b] = pick;
                        if (A.logp_out) A.logp_out[b] = logf(rowwise ? p_pick : s_score[r0 * R + pick]);
                        const int prow = r0 * R + pick;
                        const int gsel = !A.gather_from ? pick : prow < 512 ? s_gf[prow] : A.gather_from[(size_t)b * R + pick];
                        if (A.gather_from && A.gathered_out) A.gathered_out[b] = gsel;
                        s_part[r0] = __int_as_float(gsel);                     // hand the selected task to the instance's 16 lanes (s_part is free now)
                        s_part[HG + r0] = __int_as_float(prow);
                    }
                    H3T_RT(0);
                    if (A.mf_on) {
                        // = k_mfea1 (pe:152-214) for the task just selected: the 16 lanes of the instance take the machines
                        const int b = g0 + r0, T_ = A.mf.T, M_ = A.mf.M;
                        int a, prow_m;
                        if (rowwise) {                                          // every lane of the row knows the pick: no hand-over through LDS
                            prow_m = r0 * R + pick_l;
                            a = !A.gather_from ? pick_l : prow_m < 512 ? s_gf[prow_m] : A.gather_from[(size_t)b * R + pick_l];
                        } else {
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            a = __float_as_int(s_part[r0]); prow_m = __float_as_int(s_part[HG + r0]);
                        }
                        if (a < 0 || a >= T_) a = 0;
                        const size_t row = (size_t)b * T_ + a;
                        int pm = 0;
                        if (a % M_ != 0) {
                            pm = prow_m < 512 ? s_pm[prow_m] : (int)reinterpret_cast<const short *>(A.mf.link)[(row - 1) * 8 + 4];   // machine of the job predecessor
                            if (pm < 0) pm += M_;                                             // python negative index (pe:206)
                        }
                        H3T_RT(1);
                        for (int mm = l; mm < M_; mm += 16) {
                            const double tv = A.mf.t[row * M_ + mm], pv = A.mf.p[row * M_ + mm];
                            // (requested with them, not where a comparison on tv / pv selects them: that was a second round trip behind the first)
                            const double mean0 = A.mf.mean3[row * 3 + 0], mean1 = A.mf.mean3[row * 3 + 1], mean2 = A.mf.mean3[row * 3 + 2];
                            const int shop_m = A.mf.shop[(size_t)b * M_ + mm];
                            const double tt_pm = A.mf.tt[((size_t)b * M_ + pm) * M_ + mm];   // (pm = 0 where the task has no job predecessor: a valid row, value unused)
                            H3T_RT(2);
                            const double ptv = tv * fabs(pv);
                            const unsigned char mk = (unsigned char)!(tv >= 0);              // run:258-259 ~(t >= 0)
                            const double x = (a % M_ != 0) ? tt_pm : 0.0;
                            H3T_RT(3);
                            const double f0 = tv > 0 ? tv : mean0, f1 = ptv > 0 ? ptv : mean1;
                            const double f4 = pv > 0 ? pv : mean2;
                            H3T_RT(4);
                            const size_t o = ((size_t)b * M_ + mm) * 6;
                            const double f3 = (double)(1 - (int)mk), f5 = (double)(shop_m + 1);
                            if (A.mf.obs_f32) {
                                float *of = reinterpret_cast<float *>(A.mf.m_fea1_out) + o;
                                of[0] = (float)f0; of[1] = (float)f1; of[2] = (float)x; of[3] = (float)f3; of[4] = (float)f4; of[5] = (float)f5;
                            } else {
                                double *od = reinterpret_cast<double *>(A.mf.m_fea1_out) + o;
                                od[0] = f0; od[1] = f1; od[2] = x; od[3] = f3; od[4] = f4; od[5] = f5;
                            }
                            A.mf.mmask_out[(size_t)b * M_ + mm] = mk;
#ifdef HX_MF1_LDS
                            {   // the GAT statements that follow in this launch take the row from LDS (as f32: what they would make of it)
                                float *lf = (HX_MF1_LDS) + ((size_t)r0 * M_ + mm) * 6;
                                lf[0] = (float)f0; lf[1] = (float)f1; lf[2] = (float)x; lf[3] = (float)f3; lf[4] = (float)f4; lf[5] = (float)f5;
                            }
#endif
                        }
                    }
                }
            }
        }
#endif                                                              // !HX_VALUES_ONLY
        STAMP(5); H3_RT(5);
    }
#ifdef MTFJSP_STAMP
    if (A.stamps && lane == 0) for (int i = 0; i < 8; i++) A.stamps[((size_t)blockIdx.x * 8 + wave) * 8 + i] = ph[i];
#endif
