// mtfjsp_env_grp.h — k_env_grp: the register step kernel (k_env_reg, T <= 64, M*M <= 64) for GROUPS of 16 instances.
// Included by mtfjsp_env.hip and (without the kernels: MTFJSP_ENV_GRP_NO_KERNELS) by mtfjsp_encoder.hip; uses EnvParams, Link, MRec,
// rl_i, rl_d, uni, trunc_l and the S_* slots of mtfjsp_env_dev.h.
//
// Why: at the headline batch every SIMD holds four one-instance waves and the launch is bound by vector-instruction issue
// (≈920 VALU instructions per wave, SQ counters), and a third of those compute per-INSTANCE scalars — makespan, energy and
// idle sums, the four rewards, RewardScaling with its three f64 divisions and a square root, the machine feature row, the
// job mask — on all 64 lanes for one useful lane.  Here a workgroup is 16 waves = 16 instances:
//   * every wave runs the per-task part of the step for its instance exactly as k_env_reg does (lane = task: scheduling
//     decision, route / estimate updates, observation rows, ELL rows, state write-back) and leaves the per-instance inputs
//     of the scalar part in LDS;
//   * after one barrier, wave 0 runs the scalar part ONCE for all 16 instances with lane = (instance, reward channel):
//     the same f64 operations in the same order (the idle sum left to right over the terms in rank order, numpy's pairwise
//     energy sum, pt:54-124 per channel), so the results are bit-identical to k_env_reg.
// Nothing in the per-task part waits for the scalar part.
#pragma once
// Instances (= waves) per workgroup; the scalar part uses 4 lanes per instance.  Two builds (measured, tools/env_variants.py):
// 16 instances and 4 waves per SIMD (no spills) while all waves of the batch are resident at once (<= 8192 instances: 14.4 us at
// 4096 against 16.1 us for k_env_reg); 4 instances and 8 waves per SIMD for chip-filling batches, where occupancy and short
// barrier waits matter more than the amortisation (262 144 instances: 335 us = 0.78 of the copy rate against 442 us = 0.59).
#ifndef EG_ABL
#define EG_ABL 0           // timing ablations of the diagnostic builds (tools/ablate_env.sh): wrong results, never in the product
#endif
#define EG_SMALL 16
#define EG_LARGE 4
#define EG_SMALL_MAX_B 8192
enum { U_R0 = 0, U_NEWTR, U_D, U_PK, U_FTTAIL, U_STK, U_TAIL = 8 };     // s_un slots (8..15: ragged tail of the pairwise energy sum)
enum { I_VALID = 0, I_STATUS, I_NSCHED, I_M, I_JA, I_A, I_NK, I_LASTM, I_MERGED, I_F4A };   // s_in slots (12 per instance)

// NS task slots per lane: NS = 1 for T <= 64 (lane = task), NS = 2 for T <= 128 (lane holds tasks lane and lane + 64, and two
// transport-time entries: M*M <= 128).  Gathers with a uniform index pick the slot with a scalar condition and read one lane;
// gathers with a per-lane index read both slots' lanes and select.  The per-task LDS arrays have 64*NS entries.
template <typename OBS, int NS>
__device__ __forceinline__ void env_grp_wave(const EnvParams &P, const int b, const int lane, double *s_sorted, double *s_jmx, double *s_jrw,
                                             int *s_cn, double *s_scl, double *s_mf, double *s_un, int *s_in, MJRec *s_mj2 /*[2] records m | ja*/, double *s_row /*[3][16] ste | fte | pte of the acting job's ops*/, int *s_mp /*[2][64 NS] mach | prev*/,
                                             double *s_sdf /*[3][64 NS] st | dur | ft*/, double *s_ttl /*[64 NS]*/, unsigned long long *rt)
{
#ifdef MTFJSP_STAMP
#define RT(i) do { __builtin_amdgcn_sched_barrier(0); rt[i] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define RT(i) do { } while (0)
#endif
    constexpr int NL = 64 * NS;
    const int J = P.J, M = P.M, T = P.T;
    const unsigned invM = P.inv_M;
#define DIVM(x) ((int)__umulhi((unsigned)(x), invM))
    auto RLI = [&](const int (&x)[NS], int idx) __attribute__((always_inline)) { return NS == 1 ? rl_i(x[0], idx) : rl_i(idx < 64 ? x[0] : x[NS - 1], idx & 63); };
    auto RLD = [&](const double (&x)[NS], int idx) __attribute__((always_inline)) { return NS == 1 ? rl_d(x[0], idx) : rl_d(idx < 64 ? x[0] : x[NS - 1], idx & 63); };
    auto SHI = [&](const int (&x)[NS], int idx) __attribute__((always_inline)) {
        if (NS == 1) return __shfl(x[0], idx);
        const int a = __shfl(x[0], idx & 63), c = __shfl(x[NS - 1], idx & 63);
        return idx < 64 ? a : c;
    };
    auto SHD = [&](const double (&x)[NS], int idx) __attribute__((always_inline)) {
        if (NS == 1) return __shfl(x[0], idx);
        const double a = __shfl(x[0], idx & 63), c = __shfl(x[NS - 1], idx & 63);
        return idx < 64 ? a : c;
    };
    const size_t bT = (size_t)b * T;
    int v[NS]; bool isT[NS];
    // ---- Loads.  Round 6: the loads used to sit behind `if (lane < ...)` branches; hipcc turned every branch into "request, wait
    // for EVERYTHING outstanding, continue" — three dependent round trips (bulk state | per-job rows, scalars and the action |
    // the rows the action selects) where the data dependencies ask for one and a short second.  Now: the action first (scalar
    // loads), every action-independent row in ONE batch with clamped indices and no branch (lanes past the end re-read the last
    // element: same line, no traffic), and the rows the action selects as soon as the action is there, behind the batch.
    // State records (mtfjsp_env_dev.h): a lane's share of the batch is FIVE loads, three of them 16 bytes wide (round 5: eleven
    // 8-byte loads from eleven arrays, 2.1 KB per instance; now 1.5 KB — no finish times, no transport matrix).
    int a = uni(P.task_idx[b]), m = uni(P.mach_idx[b]);
    __builtin_amdgcn_sched_barrier(0);          // (the action's requests stay ahead of the batch)
    int mach[NS], prev[NS], next[NS], pos[NS];
    double st[NS], ft[NS], dur[NS], pte[NS];
    TaskSD sd_[NS]; TaskPL pl_[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) {
        v[s] = lane + 64 * s; isT[s] = v[s] < T;
        const size_t o = bT + (isT[s] ? v[s] : T - 1);
        sd_[s] = P.sd[o]; pl_[s] = P.pl[o];
    }
    const MJRec r_ = P.mj[(size_t)b * P.MJ + (lane < P.MJ ? lane : P.MJ - 1)];
    const JobR jr_ = P.jr[(size_t)b * J + (lane < J ? lane : J - 1)];
    const double sc = P.scal[(size_t)b * SCAL_N + (lane < SCAL_N ? lane : SCAL_N - 1)];
    // (the second hop's per-lane source array is known before the action: only a 32-bit offset waits for it)
    const bool h_mf = lane < 8, h_md = lane >= 16 && lane < 16 + M, h_tt = lane >= 40 && lane < 40 + M;
    const char *src0 = reinterpret_cast<const char *>(lane == 33 ? P.p + bT * M : P.t + bT * M);
    if (h_mf) src0 = reinterpret_cast<const char *>(P.mfea + (size_t)b * M * 8 + lane);
    if (h_md) src0 = reinterpret_cast<const char *>(&P.cst[bT + (lane - 16)].x);
    if (h_tt) src0 = reinterpret_cast<const char *>(P.ttT + (size_t)b * M * M + (lane - 40));
    const unsigned k_mf = h_mf ? ~0u : 0u, k_md = h_md ? ~0u : 0u, k_tt = h_tt ? ~0u : 0u, k_t = ~(k_mf | k_md | k_tt);   // (selects as masks: the ternary chain became four nested branches)
    __builtin_amdgcn_sched_barrier(0);          // every request of the batch goes out before the wait for the action
    bool valid = a >= 0 && a < T && m >= 0 && m < M;
    if (!valid) { a = 0; m = 0; }
    const int ja = DIVM(a), op = a - ja * M;
    // ---- second hop (depends on the action): ONE vector load with a per-lane source — lanes 0..7 the acting machine's feature row,
    // lanes 16..16+M-1 the acting job's minimal durations, lane 32 t[a, m], lane 33 p[a, m], lanes 40..40+M-1 column m of the
    // transport times (a row of the transposed copy: the step never needs another column — its own decision, the job edge of a, the
    // route edges of a and of its successor all end on machine m; the one entry of another column, for the merged edge that
    // reverts, is the previous step's job-edge transport time and waits in S_TRLAST); every other lane re-reads t[a, m].
    // As four loads, the two uniform ones became scalar loads that hipcc sank into the branch of their first use, BEHIND the
    // waits for the batch: a third dependent round trip.  The lane reads below are convergent and keep the request up here.
    double d, pk, md, mfr, ttc;
    {
        const unsigned off = (k_mf & ((unsigned)m * 64u)) | (k_md & ((unsigned)(ja * M) * 16u)) | (k_tt & ((unsigned)(m * M) * 8u)) | (k_t & ((unsigned)(a * M + m) * 8u));
        const double x2 = *reinterpret_cast<const double *>(src0 + off);
        d = rl_d(x2, 32); pk = rl_d(x2, 33);
        md = x2; mfr = x2; ttc = x2;
    }
#define MD_AT(c) rl_d(md, 16 + (c))
#define TTC_AT(x) rl_d(ttc, 40 + (x))           /* tt[x, m], uniform x */
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const Link l = pl_[s].link;
        mach[s] = isT[s] ? l.mach : -1; prev[s] = isT[s] ? l.prev : -1; pos[s] = isT[s] ? l.pos : 0; next[s] = isT[s] ? l.pad : -1;
        st[s] = isT[s] ? sd_[s].st : 0.0; dur[s] = isT[s] ? sd_[s].dur : 0.0; pte[s] = isT[s] ? pl_[s].pte : 0.0;
        ft[s] = st[s] + dur[s];                                                 // env:356: the addition that made the stored finish time
    }
    int head_ = lane < M ? r_.head : -1, tail_ = lane < M ? r_.tail : -1, len_ = lane < M ? r_.len : 0;
    int cnt_ = lane < J ? r_.cnt : 0;
    double jmax_ = lane < J ? jr_.jmax : -INFINITY, jrow_ = lane < J ? jr_.jrow : 0.0;
    const int lastm = (int)rl_d(sc, S_LASTM);
    int jv[NS], opv[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) { jv[s] = DIVM(v[s]); opv[s] = v[s] - jv[s] * M; }
    if (lane < SCAL_N) s_scl[lane] = sc;

    // =========================================================================================
    // A. scheduling (env:1476-1685)
    int status = 0, path = 0, Pk = -1, Nk = -1, ipos = 0;
    double st_k = 0.0;
    int mach_p = -1;
    if (valid) {
        if (RLI(mach, a) >= 0) valid = false;                                   // env:1504
        else if (op != 0) { mach_p = RLI(mach, a - 1); if (mach_p < 0) valid = false; }    // env:1520
    }
    const int len = rl_i(len_, m), head = rl_i(head_, m), tail = rl_i(tail_, m);
    const double ttmm = TTC_AT(m);
    RT(1);
    if (valid) {
        if (d < 0.0) status |= MTFJSP_ST_INFEASIBLE;                            // pe:246-248
        const double arr_k = op == 0 ? 0.0 : RLD(ft, a - 1) + TTC_AT(mach_p);     // dg:46-66
        bool do_append = false;
        if (len == 0) { path = MTFJSP_PATH_EMPTY; st_k = arr_k; ipos = 0; }                 // env:1684
        else if (!P.left_shift) do_append = true;                                               // env:1680
        else {
            const double lb_ft = arr_k + d;
            const int jh = DIVM(head);
            const double arr_f = (head == jh * M) ? 0.0 : RLD(ft, head - 1) + TTC_AT(RLI(mach, head - 1));
            if (lb_ft <= arr_f) { path = MTFJSP_PATH_FRONT; st_k = arr_k; ipos = 0; Nk = head; }   // env:1548
            else if (len == 1) do_append = true;                                                 // env:1577
            else {
                // gap test of env:1587-1604 on every task at once, then the first hit in route order by a scalar walk
                unsigned long long okm[NS];
#pragma unroll
                for (int s = 0; s < NS; s++) {
                    const int pi = prev[s] >= 0 ? prev[s] : 0, vi = v[s] > 0 ? v[s] - 1 : 0;
                    double ftP, ftj, ttj;
                    if constexpr (NS == 1) {                                    // the job predecessor is the lane below: two DPP moves; the two gathers that remain are ONE LDS round trip (they were two)
                        ftj = dpp_d<0x138>(ft[0]);
                        const int mj = dpp_i<0x138>(mach[0]);
                        ftP = SHD(ft, pi);
                        ttj = __shfl(ttc, 40 + (mj >= 0 ? mj : 0));
                    } else {
                        ftP = SHD(ft, pi); ftj = SHD(ft, vi);
                        const int mj = SHI(mach, vi);
                        ttj = __shfl(ttc, 40 + (mj >= 0 ? mj : 0));
                    }
                    const double jarr = (opv[s] == 0) ? 0.0 : ftj + ttj;
                    const double x = (DIVM(pi) == jv[s]) ? ttmm : 0.0;
                    const double nst = fmax(jarr, ftP + x);
                    const bool ok = isT[s] && mach[s] == m && prev[s] >= 0 && !(lb_ft > nst) && !((nst - ftP) < d);
                    okm[s] = __ballot(ok);
                }
                int cur = RLI(next, head);
                while (cur >= 0) {
                    const unsigned long long mk_ = NS == 1 ? okm[0] : (cur < 64 ? okm[0] : okm[NS - 1]);
                    if ((mk_ >> (cur & 63)) & 1ull) { Nk = cur; break; }
                    cur = RLI(next, cur);
                }
                if (Nk >= 0) {
                    path = MTFJSP_PATH_BETWEEN;
                    ipos = RLI(pos, Nk); Pk = RLI(prev, Nk);
                    const double xx = (DIVM(Pk) == ja) ? ttmm : 0.0;
                    st_k = fmax(arr_k, RLD(ft, Pk) + xx);                       // env:1619
                } else do_append = true;                                        // env:1676
            }
        }
        if (do_append) {                                                        // env:1689-1775
            path = MTFJSP_PATH_APPEND;
            const double xx = (DIVM(tail) == ja) ? ttmm : 0.0;
            st_k = fmax(arr_k, RLD(ft, tail) + xx);
            ipos = len; Pk = tail;
        }
        status |= path;
    } else status |= MTFJSP_ST_INVALID;
    if (!valid) {                                                               // nothing changes; the scalar part reports it
        if (lane == 0) { s_in[I_VALID] = 0; s_in[I_STATUS] = status; }
        return;
    }
    const double ft_k = st_k + d;
    RT(2);
    // ---- apply: register updates on the owning lanes
#pragma unroll
    for (int s = 0; s < NS; s++) {
        if (isT[s] && mach[s] == m && pos[s] >= ipos) pos[s] += 1;
        if (v[s] == a) { mach[s] = m; prev[s] = Pk; next[s] = Nk; pos[s] = ipos; st[s] = st_k; ft[s] = ft_k; dur[s] = d; pte[s] = d * pk; }    // env:356,2175
        if (v[s] == Nk) prev[s] = a;
        if (v[s] == Pk) next[s] = a;
    }
    if (lane == m) { if (ipos == 0) head_ = a; if (ipos == len) tail_ = a; len_ = len + 1; }
    if (lane == ja) cnt_ += 1;
    const int nsched = (int)rl_d(sc, S_NSCHED) + 1;

    // =========================================================================================
    // B. per-task side of the costs
    // estimated start/finish of the acting job's ops: the reference's left-to-right loop (env:1965-1995), run with
    // scalar indices; the lane of task ja*M+c keeps its own (ste, fte).  Ops of a job are scheduled in order: ops < op are
    // scheduled (their estimate IS their finish time, already folded into the job's running row maximum jrow_), op is being
    // scheduled now, ops > op are unscheduled -> only the tail is walked.
    double my_ste[NS], my_fte[NS], accp = ft_k;
#pragma unroll
    for (int s = 0; s < NS; s++) { my_ste[s] = 0.0; my_fte[s] = 0.0; }
    const double row_prev = rl_d(jrow_, ja);                                    // max real finish time of ops < op (0 if none)
    double jrow_new = op == 0 ? ft_k : fmax(row_prev, ft_k);                    // ppo:265-275 row maximum of real finish times
    double jmax_new = jrow_new;                                                 // estimated finish times of ops <= op are the real ones
#pragma unroll
    for (int s = 0; s < NS; s++) if (v[s] == a) { my_ste[s] = st_k; my_fte[s] = ft_k; }
    if (ft_k == 0.0) {                                                          // env:1977: a zero finish time is treated as "not set"
        accp = (op ? RLD(ft, a - 1) : 0.0) + MD_AT(op);
#pragma unroll
        for (int s = 0; s < NS; s++) if (v[s] == a) my_fte[s] = accp;
        jmax_new = op == 0 ? accp : fmax(row_prev, accp);
    }
    if (!(EG_ABL & 8))
    for (int c = op + 1; c < M; c++) {
        const double fte_c = accp + MD_AT(c);
#pragma unroll
        for (int s = 0; s < NS; s++) if (v[s] == ja * M + c) { my_ste[s] = accp; my_fte[s] = fte_c; }
        accp = fte_c;
        jmax_new = fmax(jmax_new, fte_c);
    }
    if (lane == ja) { jmax_ = jmax_new; jrow_ = jrow_new; }
    if (lane < J) { s_jmx[lane] = jmax_; s_jrw[lane] = jrow_; s_cn[lane] = cnt_; }
    // env:896 np.sum(pt_est), numpy's pairwise order (one leaf block: T <= 128): lanes 0..7 are its 8 accumulators r[k] = a[k] +
    // a[k+8] + ... (in that order), then its fixed tree ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) as an xor butterfly (fp addition
    // commutes); the ragged tail is added by the scalar part, in order
    {
        const int nb = (EG_ABL & 8) ? 0 : T < 8 ? 0 : T - (T & 7);
        if (nb) {
            double r = pte[0];
            if constexpr (NS == 1) {                                            // the same additions in the same order without LDS round trips (there were six)
                const double x8 = dpp_d<0x108>(pte[0]), x16 = up16_d(pte[0]), x24 = dpp_d<0x108>(x16);
                if (nb > 8) r += x8;
                if (nb > 16) r += x16;
                if (nb > 24) r += x24;
                if (nb > 32) {
                    const double y = up32_d(pte[0]), y8 = dpp_d<0x108>(y), y16 = up16_d(y), y24 = dpp_d<0x108>(y16);
                    r += y;
                    if (nb > 40) r += y8;
                    if (nb > 48) r += y16;
                    if (nb > 56) r += y24;
                }
                r += dpp_d<0xB1>(r); r += dpp_d<0x4E>(r); r += dpp_d<0x141>(r);     // fp addition commutes: both partners hold the same sum
            } else {
                for (int i = 8; i < nb; i += 8) r += SHD(pte, (lane & 7) + i);
                r += __shfl_xor(r, 1); r += __shfl_xor(r, 2); r += __shfl_xor(r, 4);
            }
            if (lane == 0) s_un[U_R0] = r;
        } else if (lane == 0) s_un[U_R0] = 0.0;
#pragma unroll
        for (int s = 0; s < NS; s++) if (v[s] >= nb && v[s] < T) s_un[U_TAIL + v[s] - nb] = pte[s];
    }
    // idle time (dg:144-170): one term per scheduled task, summed strictly left to right in (machine, route position)
    // order: every scheduled task's lane computes its rank in that order = (tasks on lower machines) + (its route position) and
    // drops its term at that index; the scalar part adds terms in order.
    if (!(EG_ABL & 4)) {
        int incl = lane < M ? len_ : 0;                                         // M <= 16: four DPP row shifts with zero fill
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, true);   // row_shr:1
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, true);   // row_shr:2
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, true);   // row_shr:4
        if (NS > 1) incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, true);   // row_shr:8 (M > 8 only with two slots)
        const int before = incl - (lane < M ? len_ : 0);
        // the other lanes fill the remaining slots with +0.0 (x + 0.0 == x: the running sum is never -0.0), so that the scalar
        // part adds a fixed number of terms: unscheduled tasks take nsched.. in task order, indices >= T their own index
        int ubase = nsched;
#pragma unroll
        for (int s = 0; s < NS; s++) {
            const int below = __shfl(before, mach[s] >= 0 ? mach[s] : 0);       // executed by ALL lanes: the source lanes must be active
            const double ftPr = SHD(ft, prev[s] >= 0 ? prev[s] : 0);            // (the three gathers go out together: one LDS round trip)
            __builtin_amdgcn_sched_barrier(0);
            const double term = prev[s] < 0 ? st[s] : st[s] - ftPr;
            const bool sch = isT[s] && mach[s] >= 0;
            const unsigned long long um = __ballot(isT[s] && mach[s] < 0);
            const int uidx = ubase + __builtin_amdgcn_mbcnt_hi((unsigned)(um >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)um, 0u));
            s_sorted[sch ? below + pos[s] : isT[s] ? uidx : v[s]] = sch ? term : 0.0;
            ubase += __builtin_popcountll(um);
        }
    }
    const double new_tr = (op == 0) ? 0.0 : TTC_AT(mach_p);                       // env:872-876
    {
        const double ft_tail = RLD(ft, rl_i(tail_, m));                         // env:2315-2340, column 0 of the acting machine's row
        if (lane == 0) {
            s_un[U_NEWTR] = new_tr; s_un[U_D] = d; s_un[U_PK] = pk; s_un[U_FTTAIL] = ft_tail; s_un[U_STK] = st_k;
            s_in[I_VALID] = 1; s_in[I_STATUS] = status; s_in[I_NSCHED] = nsched; s_in[I_M] = m; s_in[I_JA] = ja;
        }
        if (lane < 8) s_mf[lane] = mfr;
    }

    RT(3);
    // =========================================================================================
    // C. the observation rows that changed
    const double w30 = rl_d(sc, S_W3), w31 = rl_d(sc, S_W3 + 1), w32 = rl_d(sc, S_W3 + 2);
    const bool merged_a = Pk >= 0 && op != 0 && Pk == a - 1;
    // (the rows themselves — tasks a .. end of job, env:2245-2277 — are formed and stored by scalar-part waves for the whole group:
    // env_grp_rows; a lane of the acting job leaves its three per-task values)
#pragma unroll
    for (int s = 0; s < NS; s++)
        if (isT[s] && jv[s] == ja && opv[s] >= op) { s_row[opv[s]] = my_ste[s]; s_row[16 + opv[s]] = my_fte[s]; s_row[32 + opv[s]] = pte[s]; }
    {   // the in-edge (ELL) rows that changed — of a, its job successor, its new route successor and the node whose merged edge
        // reverts — are computed by the group's second scalar wave (env_grp_ell: 4 lanes per instance); it gathers from these
        const int merged_now = merged_a ? a : -1;
#pragma unroll
        for (int s = 0; s < NS; s++) {
            if (!(EG_ABL & 16) && isT[s]) { s_mp[v[s]] = mach[s]; s_mp[NL + v[s]] = prev[s]; s_sdf[v[s]] = st[s]; s_sdf[NL + v[s]] = dur[s]; s_sdf[2 * NL + v[s]] = ft[s]; }
        }
        if (lane >= 40 && lane < 40 + M) s_ttl[lane - 40] = ttc;                // column m of the transport times
        if (lane == 0) { s_in[I_A] = a; s_in[I_NK] = Nk; s_in[I_LASTM] = lastm; s_in[I_MERGED] = merged_now; s_in[I_F4A] = 1 + ((Pk >= 0 && !merged_a) ? 1 : 0); }
    }
    // ---- the state that changed.  Round 6: the one-lane stores of an instance (start / duration of a, the job's record, the two
    // records of machine m and job ja, the merged-edge words, the candidate) each cost this wave a branch, a 64-bit address and a
    // store instruction for ONE lane — without them the group reached its barrier 1.2 us earlier (profiles/r06_ablate_env.txt).  They
    // are now stored for ALL instances of the group by three instructions of a scalar-part wave (env_grp_state); only the task
    // records, whose lanes are tasks, are stored here.
    {
        MJRec r; r.head = (short)head_; r.tail = (short)tail_; r.len = (short)len_; r.cnt = (short)cnt_;
        if (lane == m) s_mj2[0] = r;
        if (lane == ja) s_mj2[1] = r;
    }
    if (!(EG_ABL & 1))
#pragma unroll
    for (int s = 0; s < NS; s++) {
        // only the records this decision changed: the acting task, its route neighbours, the tasks of machine m whose rank moved up
        const bool chg = isT[s] && (v[s] == a || v[s] == Pk || v[s] == Nk || (mach[s] == m && pos[s] > ipos));
        if (chg) {
            TaskPL y; y.pte = pte[s]; y.link.mach = (short)mach[s]; y.link.prev = (short)prev[s]; y.link.pos = (short)pos[s]; y.link.pad = (short)next[s];
            P.pl[bT + v[s]] = y;
        }
    }
    RT(4);
#undef MD_AT
#undef TTC_AT
#undef DIVM
#undef RT
}

// Row pitches of the group's per-instance LDS arrays.  The scalar-part waves read them with lane = (instance, channel): the eight
// instances of a 32-lane group read the same slot of eight rows at once, and with power-of-two pitches (round 5: 64 doubles, 8, 16,
// 128 ints ...) all eight reads fell on one bank — every ds_read of the scalar part took eight LDS cycles instead of one.  Pitches
// of 2 (doubles: 4) banks modulo 64 put the eight rows on different banks.
template <int NL>
struct EgPitch { static constexpr int SORTED = NL + 2, JOB = 66, CN = 65, MF = 9, UN = 17, MP = 2 * NL + 1, SDF = 3 * NL + 1, TTL = NL + 1; };
// where the per-instance inputs of the scalar part live: k_env_grp* keeps them in fixed-size arrays, k_env_step_grp in the
// instance's own LDS region
template <int NL>
struct EnvGrpRegAcc {
    static constexpr bool kBigT = false;                         // T <= 128: the pairwise energy sum is one leaf block
    using PT = EgPitch<NL>;
    const double (*s_sorted)[PT::SORTED], (*s_jmx)[PT::JOB], (*s_jrw)[PT::JOB]; const int (*s_cn)[PT::CN];
    const double (*s_scl)[SCAL_N], (*s_mf)[PT::MF], (*s_un)[PT::UN]; const int (*s_in)[12];
    __device__ __forceinline__ const double *sorted(int g) const { return s_sorted[g]; }
    __device__ __forceinline__ const double *jmx(int g) const { return s_jmx[g]; }
    __device__ __forceinline__ const double *jrw(int g) const { return s_jrw[g]; }
    __device__ __forceinline__ const int *cn(int g) const { return s_cn[g]; }
    __device__ __forceinline__ const double *scl(int g) const { return s_scl[g]; }
    __device__ __forceinline__ const double *mf(int g) const { return s_mf[g]; }
    __device__ __forceinline__ const double *un(int g) const { return s_un[g]; }
    __device__ __forceinline__ const int *in(int g) const { return s_in[g]; }
};
// the per-instance scalar part for the instances of the group: lane = (instance g = lane >> 2, reward channel ch = lane & 3)
template <typename OBS, typename ACC>
__device__ __forceinline__ void env_grp_tail_batched(const EnvParams &P, const int b0, const int lane, const int EG, const ACC &A, unsigned long long *rt2 = nullptr)
{
#ifdef MTFJSP_STAMP
#define RT2(i) do { __builtin_amdgcn_sched_barrier(0); if (rt2) rt2[i] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define RT2(i) do { } while (0)
#endif
    const int g = lane >> 2, ch = lane & 3;
    const int b = b0 + g;
    if (g >= EG || b >= P.B) return;
    const int J = P.J, M = P.M, T = P.T;
    // Every fixed-slot input of the scalar part is requested here in ONE batch of LDS reads: this wave runs alone after the
    // group's barrier, so a read per dependent step is all latency (the per-use form had twelve serial LDS round trips, about
    // 0.7 us of the 1.8 us this part took).  The first chunk of the job maxima and of the idle terms rides in the same batch.
    const int *in = A.in(g);
    const double *sc = A.scl(g), *un = A.un(g), *so = A.sorted(g), *jm = A.jmx(g);
    const int valid = in[I_VALID], status = in[I_STATUS], nsched = in[I_NSCHED], m = in[I_M];
    const double mk_prev = sc[S_MK_PREV], e1_prev = sc[S_E1_PREV], tr_prev = sc[S_TR_PREV], id_prev = sc[S_ID_PREV];
    const double tr_this0 = sc[S_TR_THIS], n0 = sc[S_N], ns0 = sc[S_NSCHED];
    const double sR0 = sc[S_R + ch], sM0 = sc[S_MEAN + ch], S0 = sc[S_S + ch];
    const double u_r0 = un[U_R0], new_tr = un[U_NEWTR], d = un[U_D], pk = un[U_PK], fttail = un[U_FTTAIL];
    double xt[7];
#pragma unroll
    for (int i = 0; i < 7; i++) xt[i] = un[U_TAIL + i];
    const double mf_ch = A.mf(g)[ch], mf_4 = A.mf(g)[4];
    double xj[8], xs[8];
#pragma unroll
    for (int k = 0; k < 8; k++) xj[k] = jm[k < J ? k : J - 1];
#pragma unroll
    for (int k = 0; k < 8; k++) xs[k] = so[k];
    RT2(0);
    if (!valid) {                                                               // rejected action: nothing changed; observations persist
        const bool all_done = ns0 == (double)T;
        P.obs.info[(size_t)b * 6 + ch] = (ch == 1 && all_done) ? 1.0 : 0.0;
        if (ch < 2) P.obs.info[(size_t)b * 6 + 4 + ch] = 0.0;
        if (P.obs.raw) { P.obs.raw[(size_t)b * 5 + ch] = 0.0; if (ch == 0) P.obs.raw[(size_t)b * 5 + 4] = 0.0; }
        if (P.rec_r4) P.rec_r4[(size_t)ch * P.B + b] = 0.f;
        if (P.rec_done && ch == 0) P.rec_done[b] = all_done ? 1.f : 0.f;
        if (ch == 1) P.obs.status[b] = status;
        return;
    }
    // (indices past the end are clamped — max / min are idempotent — or hit zero-filled slots)
    double mk = xj[0];                                                          // env:894 np.amax
#pragma unroll
    for (int k = 1; k < 8; k++) mk = fmax(mk, xj[k]);
    for (int j0 = 8; j0 < J; j0 += 4) {
        double x[4];
#pragma unroll
        for (int k = 0; k < 4; k++) x[k] = jm[j0 + k < J ? j0 + k : J - 1];
#pragma unroll
        for (int k = 0; k < 4; k++) mk = fmax(mk, x[k]);
    }
    double e1 = u_r0;                                                           // env:896 np.sum: pairwise part, then the ragged tail in order
    bool big = false;
    if constexpr (ACC::kBigT) big = T > 128;                                    // (more than one leaf block: the instance's wave walked numpy's recursion, U_R0 is the total)
    if (!big) {
        const int nt = T < 8 ? T : (T & 7);
#pragma unroll
        for (int i = 0; i < 7; i++) if (i < nt) e1 += xt[i];
    }
    e1 = 0.0 + e1;
    // dg:144-170, strictly left to right.  Slots >= nsched hold +0.0 and the running sum starts at +0.0, so it is never -0.0 and
    // adding them changes nothing: the loop stops at nsched; the next chunk's reads are in flight under the 8 dependent adds
    // Round 6: two register sets in turn (the chunk loop copied the prefetched chunk into the working set — 16 moves and 8 selects
    // per 8 additions; the loop took 0.5-0.65 us of this wave's 1.5).  Every slot below NL is written by the instance's wave (+0.0
    // past nsched), so a chunk past the end may be read and added.
    double idle = 0.0;
    {
        constexpr int NLr = (int)(sizeof(*A.s_sorted) / sizeof(double)) - 2;    // slots per row (EgPitch::SORTED = NL + 2)
        double ys[8];
        for (int i0 = 0; i0 < nsched; i0 += 16) {
            const int i1 = i0 + 8 < NLr ? i0 + 8 : NLr - 8, i2 = i0 + 16 < NLr ? i0 + 16 : NLr - 8;
#pragma unroll
            for (int k = 0; k < 8; k++) ys[k] = so[i1 + k];
#pragma unroll
            for (int k = 0; k < 8; k++) idle = idle + xs[k];
            if (i0 + 8 >= nsched) break;
#pragma unroll
            for (int k = 0; k < 8; k++) xs[k] = so[i2 + k];
#pragma unroll
            for (int k = 0; k < 8; k++) idle = idle + ys[k];
        }
    }
    RT2(1);
    const double trans_this = tr_this0 + new_tr;
    const double r_t = 1.0 * mk_prev - mk;                                      // env:1066
    double r_pt = 1.0 * e1_prev - e1;
    r_pt = r_pt / (double)T;                                                    // env:1073-1076
    const double r_tt = 1.0 * tr_prev - trans_this;                             // env:1083
    const double r_idle = 1.0 * id_prev - idle;                                 // env:1088
    const double tot_n = P.w_mk * r_t + P.w_ec * (r_pt + 1 * r_idle) + P.w_tt * r_tt * 1;                  // env:1164
    const double tot = P.divisor == 1.0 ? tot_n : tot_n / P.divisor;            // x / 1.0 == x exactly
    const bool done = nsched == T;                                              // env:797-800
    double *s = P.scal + (size_t)b * SCAL_N;
    {   // reward scaling of channel ch (pt:54-124)
        const double n = n0 + 1.0;
        const double x = ch == 0 ? r_t : ch == 1 ? r_idle : ch == 2 ? r_pt : r_tt;
        const double R = P.gamma * sR0 + x;
        double mean, S = S0, sd;
        if (n == 1.0) { mean = R; sd = fabs(R); }
        else { mean = sM0 + (R - sM0) / n; S = S + (R - sM0) * (R - mean); sd = sqrt(S / n); }
        const double scaled = x / (sd + 1e-8);
        s[S_R + ch] = R; s[S_MEAN + ch] = mean; s[S_S + ch] = S; s[S_STD + ch] = sd;
        P.obs.info[(size_t)b * 6 + 2 + ch] = scaled;
        if (P.rec_r4) P.rec_r4[(size_t)ch * P.B + b] = (float)scaled;
        if (ch == 0) {
            s[S_N] = n; s[S_NSCHED] = (double)nsched;
            s[S_MK_PREV] = mk; s[S_E1_PREV] = e1; s[S_TR_PREV] = trans_this; s[S_ID_PREV] = idle;    // env:932-936
            s[S_TR_THIS] = done ? 0.0 : trans_this;                              // env:950-960
            P.obs.info[(size_t)b * 6 + 0] = tot;
            P.obs.info[(size_t)b * 6 + 1] = done ? 1.0 : 0.0;
            if (P.rec_done) P.rec_done[b] = done ? 1.f : 0.f;
        }
        if (P.obs.raw) {
            P.obs.raw[(size_t)b * 5 + 1 + ch] = x;                            // raw: total, makespan, idle, energy, transport
            if (ch == 0) P.obs.raw[(size_t)b * 5] = tot;
        }
        if (ch == 1) P.obs.status[b] = status;
    }
    RT2(2);
    {   // machine features of the acting machine (env:2315-2340): columns 0..3 on the four lanes, column 4 with lane 0
        double mfr = mf_ch;
        if (ch == 0) mfr = fttail;
        else if (ch == 1) mfr += (pk * d) / (double)T;
        else if (ch == 2) mfr += new_tr;
        else mfr += idle - id_prev;
        const size_t o = ((size_t)b * M + m) * 8;
        P.mfea[o + ch] = mfr;
        reinterpret_cast<OBS *>(P.obs.m_fea2)[o + ch] = (OBS)mfr;
        if (ch == 0) {
            const double c4 = mf_4 + 1;
            P.mfea[o + 4] = c4;
            reinterpret_cast<OBS *>(P.obs.m_fea2)[o + 4] = (OBS)c4;
        }
    }
    RT2(3);
#undef RT2
}

// the same scalar part with every LDS input read where it is used: 30-odd fewer live registers.  The kernels that run several
// workgroups per CU (k_env_grp4 at 64 registers, k_env_grp4x2, k_env_step_grp) use this form — the batched one spilled k_env_grp4
// to scratch and cost it 26 % at 262 144 instances (342 -> 430 us), for 0.2 us gained where one workgroup per CU runs alone
template <typename OBS, typename ACC>
__device__ __forceinline__ void env_grp_tail_seq(const EnvParams &P, const int b0, const int lane, const int EG, const ACC &A)
{
    const int g = lane >> 2, ch = lane & 3;
    const int b = b0 + g;
    if (g >= EG || b >= P.B) return;
    const int J = P.J, M = P.M, T = P.T;
    const double *sc = A.scl(g);
    const int status = A.in(g)[I_STATUS];
    if (!A.in(g)[I_VALID]) {                                                    // rejected action: nothing changed; observations persist
        const bool all_done = sc[S_NSCHED] == (double)T;
        P.obs.info[(size_t)b * 6 + ch] = (ch == 1 && all_done) ? 1.0 : 0.0;
        if (ch < 2) P.obs.info[(size_t)b * 6 + 4 + ch] = 0.0;
        if (P.obs.raw) { P.obs.raw[(size_t)b * 5 + ch] = 0.0; if (ch == 0) P.obs.raw[(size_t)b * 5 + 4] = 0.0; }
        if (P.rec_r4) P.rec_r4[(size_t)ch * P.B + b] = 0.f;
        if (P.rec_done && ch == 0) P.rec_done[b] = all_done ? 1.f : 0.f;
        if (ch == 1) P.obs.status[b] = status;
        return;
    }
    const int nsched = A.in(g)[I_NSCHED], m = A.in(g)[I_M];
    // (loops in chunks whose LDS reads go out together: this wave runs alone, a read per dependent step would be all latency;
    // indices past the end are clamped — max / min are idempotent — or hit zero-filled slots)
    double mk = A.jmx(g)[0];                                                    // env:894 np.amax
    for (int j0 = 0; j0 < J; j0 += 4) {
        double x[4];
#pragma unroll
        for (int k = 0; k < 4; k++) x[k] = A.jmx(g)[j0 + k < J ? j0 + k : J - 1];
#pragma unroll
        for (int k = 0; k < 4; k++) mk = fmax(mk, x[k]);
    }
    double e1 = A.un(g)[U_R0];                                                  // env:896 np.sum: pairwise part, then the ragged tail in order
    bool big = false;
    if constexpr (ACC::kBigT) big = T > 128;                                    // (more than one leaf block: the instance's wave walked numpy's recursion, U_R0 is the total)
    if (!big) {
        const int nt = T < 8 ? T : (T & 7);
        double x[7];
#pragma unroll
        for (int i = 0; i < 7; i++) x[i] = A.un(g)[U_TAIL + i];
#pragma unroll
        for (int i = 0; i < 7; i++) if (i < nt) e1 += x[i];
    }
    e1 = 0.0 + e1;
    // dg:144-170, strictly left to right.  Slots >= nsched hold +0.0 and the running sum starts at +0.0, so it is never -0.0 and
    // adding them changes nothing: the loop stops at nsched; the next chunk's reads are in flight under the 8 dependent adds
    double idle = 0.0;
    {
        const double *so = A.sorted(g);
        double x[8];
#pragma unroll
        for (int k = 0; k < 8; k++) x[k] = so[k];
        for (int i0 = 0; i0 < nsched; i0 += 8) {
            double y[8];
            const bool more = i0 + 8 < nsched;
#pragma unroll
            for (int k = 0; k < 8; k++) y[k] = more ? so[i0 + 8 + k] : 0.0;
#pragma unroll
            for (int k = 0; k < 8; k++) idle = idle + x[k];
#pragma unroll
            for (int k = 0; k < 8; k++) x[k] = y[k];
        }
    }
    const double new_tr = A.un(g)[U_NEWTR], d = A.un(g)[U_D], pk = A.un(g)[U_PK];
    const double trans_this = sc[S_TR_THIS] + new_tr;
    const double mk_prev = sc[S_MK_PREV], e1_prev = sc[S_E1_PREV], tr_prev = sc[S_TR_PREV], id_prev = sc[S_ID_PREV];
    const double r_t = 1.0 * mk_prev - mk;                                      // env:1066
    double r_pt = 1.0 * e1_prev - e1;
    r_pt = r_pt / (double)T;                                                    // env:1073-1076
    const double r_tt = 1.0 * tr_prev - trans_this;                             // env:1083
    const double r_idle = 1.0 * id_prev - idle;                                 // env:1088
    const double tot_n = P.w_mk * r_t + P.w_ec * (r_pt + 1 * r_idle) + P.w_tt * r_tt * 1;                  // env:1164
    const double tot = P.divisor == 1.0 ? tot_n : tot_n / P.divisor;            // x / 1.0 == x exactly
    const bool done = nsched == T;                                              // env:797-800
    double *s = P.scal + (size_t)b * SCAL_N;
    {   // reward scaling of channel ch (pt:54-124)
        const double n = sc[S_N] + 1.0;
        const double x = ch == 0 ? r_t : ch == 1 ? r_idle : ch == 2 ? r_pt : r_tt;
        const double sR0 = sc[S_R + ch], sM0 = sc[S_MEAN + ch];
        const double R = P.gamma * sR0 + x;
        double mean, S = sc[S_S + ch], sd;
        if (n == 1.0) { mean = R; sd = fabs(R); }
        else { mean = sM0 + (R - sM0) / n; S = S + (R - sM0) * (R - mean); sd = sqrt(S / n); }
        const double scaled = x / (sd + 1e-8);
        s[S_R + ch] = R; s[S_MEAN + ch] = mean; s[S_S + ch] = S; s[S_STD + ch] = sd;
        P.obs.info[(size_t)b * 6 + 2 + ch] = scaled;
        if (P.rec_r4) P.rec_r4[(size_t)ch * P.B + b] = (float)scaled;
        if (ch == 0) {
            s[S_N] = n; s[S_NSCHED] = (double)nsched;
            s[S_MK_PREV] = mk; s[S_E1_PREV] = e1; s[S_TR_PREV] = trans_this; s[S_ID_PREV] = idle;    // env:932-936
            s[S_TR_THIS] = done ? 0.0 : trans_this;                              // env:950-960
            P.obs.info[(size_t)b * 6 + 0] = tot;
            P.obs.info[(size_t)b * 6 + 1] = done ? 1.0 : 0.0;
            if (P.rec_done) P.rec_done[b] = done ? 1.f : 0.f;
        }
        if (P.obs.raw) {
            P.obs.raw[(size_t)b * 5 + 1 + ch] = x;                            // raw: total, makespan, idle, energy, transport
            if (ch == 0) P.obs.raw[(size_t)b * 5] = tot;
        }
        if (ch == 1) P.obs.status[b] = status;
    }
    {   // machine features of the acting machine (env:2315-2340): columns 0..3 on the four lanes, column 4 with lane 0
        double mfr = A.mf(g)[ch];
        if (ch == 0) mfr = A.un(g)[U_FTTAIL];
        else if (ch == 1) mfr += (pk * d) / (double)T;
        else if (ch == 2) mfr += new_tr;
        else mfr += idle - id_prev;
        const size_t o = ((size_t)b * M + m) * 8;
        P.mfea[o + ch] = mfr;
        reinterpret_cast<OBS *>(P.obs.m_fea2)[o + ch] = (OBS)mfr;
        if (ch == 0) {
            const double c4 = A.mf(g)[4] + 1;
            P.mfea[o + 4] = c4;
            reinterpret_cast<OBS *>(P.obs.m_fea2)[o + 4] = (OBS)c4;
        }
    }
}

template <typename OBS, bool BATCHED, typename ACC>
__device__ __forceinline__ void env_grp_tail(const EnvParams &P, const int b0, const int lane, const int EG, const ACC &A, unsigned long long *rt2 = nullptr)
{
    if constexpr (BATCHED) env_grp_tail_batched<OBS>(P, b0, lane, EG, A, rt2);
    else env_grp_tail_seq<OBS>(P, b0, lane, EG, A);
}

// job mask (ppo:202-316) of the group's instances: lane = (instance, jobs ch, ch + 4, ...)
template <typename ACC>
__device__ __forceinline__ void env_grp_mask(const EnvParams &P, const int b0, const int lane, const int EG, const ACC &A)
{
    const int g = lane >> 2, ch = lane & 3;
    const int b = b0 + g;
    if (g >= EG || b >= P.B || !A.in(g)[I_VALID]) return;
    const int J = P.J, M = P.M;
    {   // job mask (ppo:202-316)
        int cmin = M;
        double mn = INFINITY;                                                   // min row maximum over the unfinished jobs
        for (int j0 = 0; j0 < J; j0 += 4) {
            int c[4]; double r[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { const int j = j0 + k < J ? j0 + k : J - 1; c[k] = A.cn(g)[j]; r[k] = A.jrw(g)[j]; }
#pragma unroll
            for (int k = 0; k < 4; k++) { cmin = c[k] < cmin ? c[k] : cmin; mn = fmin(mn, c[k] != M ? r[k] : INFINITY); }
        }
        for (int j = ch; j < J; j += 4) {
            const int cnt_ = A.cn(g)[j];
            unsigned char mk_;
            if (cmin == 0) mk_ = cnt_ >= 1;
            else if (cmin == M) mk_ = 1;
            else mk_ = !((cnt_ == M ? INFINITY : A.jrw(g)[j]) == mn);
            P.obs.job_mask[(size_t)b * J + j] = mk_;
        }
    }
}

// The per-instance state words a decision changes, for the group's instances: lane = (instance g, item): 0 start / duration of the
// acting task, 1 the acting job's record, 2 the merged-edge words of the scalar row; then the records of machine m and of job ja, the
// acting job's candidate (ppo:202-316).  Three store instructions for the group.
template <typename ACC>
__device__ __forceinline__ void env_grp_state(const EnvParams &P, const int b0, const int lane, const int EG, const ACC &A, const MJRec (*s_mj2)[2])
{
    const int g = lane >> 2, it = lane & 3;
    const int b = b0 + g;
    if (g >= EG || b >= P.B || !A.in(g)[I_VALID]) return;
    const int J = P.J, M = P.M;
    const int a = A.in(g)[I_A], ja = A.in(g)[I_JA], m = A.in(g)[I_M], merged_now = A.in(g)[I_MERGED];
    const double stk = A.un(g)[U_STK], d = A.un(g)[U_D], new_tr = A.un(g)[U_NEWTR], jmx = A.jmx(g)[ja], jrw = A.jrw(g)[ja];
    const int cnt = A.cn(g)[ja];
    const MJRec r0 = s_mj2[g][0], r1 = s_mj2[g][1];
    if (it < 3) {
        double2 x; double2 *dst;
        if (it == 0) { x = make_double2(stk, d); dst = reinterpret_cast<double2 *>(&P.sd[(size_t)b * P.T + a]); }
        else if (it == 1) { x = make_double2(jmx, jrw); dst = reinterpret_cast<double2 *>(&P.jr[(size_t)b * J + ja]); }
        else { x = make_double2(new_tr, (double)merged_now); dst = reinterpret_cast<double2 *>(P.scal + (size_t)b * SCAL_N + S_TRLAST); }
        *dst = x;
    }
    if (it < 2) P.mj[(size_t)b * P.MJ + (it == 0 ? m : ja)] = it == 0 ? r0 : r1;
    if (it == 0) P.obs.candidate[(size_t)b * J + ja] = ja * M + (cnt < M ? cnt : M - 1);
}

// The feature rows a decision changes (tasks a .. end of its job, env:2245-2277), for the group's instances: lane = (instance, op of
// the acting job), 8 (M <= 8) or 4 instances per pass; passes are dealt to the waves w0, w0 + nw, ...  Round 5 stored them from the
// instances' own waves: three 16-byte stores of at most M lanes each per instance, 48 store instructions per group instead of 6.
template <typename OBS, typename ACC>
__device__ __forceinline__ void env_grp_rows(const EnvParams &P, const int b0, const int lane, const int EG, const ACC &A, const double (*s_row)[48], const int w0, const int nw)
{
    const int M = P.M, T = P.T;
    const int sh = M <= 8 ? 3 : 4, ipp = 64 >> sh;
    for (int g0 = w0 * ipp; g0 < EG; g0 += nw * ipp) {
        const int g = g0 + (lane >> sh), c = lane & ((1 << sh) - 1);
        const int b = b0 + g;
        if (g >= EG || b >= P.B || c >= M || !A.in(g)[I_VALID]) continue;
        const int a = A.in(g)[I_A], ja = A.in(g)[I_JA], m = A.in(g)[I_M], f4a = A.in(g)[I_F4A];
        const int op = a - ja * M;
        if (c < op) continue;
        const double ste = s_row[g][c], fte = s_row[g][16 + c], pte = s_row[g][32 + c];
        const double d = A.un(g)[U_D], pk = A.un(g)[U_PK], w30 = A.scl(g)[S_W3], w31 = A.scl(g)[S_W3 + 1], w32 = A.scl(g)[S_W3 + 2];
        const bool isa = c == op;
        OBS f[12];
        f[0] = (OBS)ste; f[1] = (OBS)fte; f[2] = (OBS)pte;
        f[3] = (OBS)(isa ? 1.0 : 0.0);
        f[4] = (OBS)(isa ? f4a : 1);
        f[5] = (OBS)(isa ? m + 1 : 0);
        f[6] = (OBS)(isa ? d : 0.0);
        f[7] = (OBS)(isa ? pk : 0.0);
        f[8] = (OBS)(ja + 1);
        f[9] = (OBS)w30; f[10] = (OBS)w31; f[11] = (OBS)w32;
        uint4 *dst = reinterpret_cast<uint4 *>(reinterpret_cast<OBS *>(P.obs.tasks_fea) + ((size_t)b * T + ja * M + c) * 12);
        const uint4 *src = reinterpret_cast<const uint4 *>(f);
        for (int i = 0; i < (int)(12 * sizeof(OBS) / 16); i++) dst[i] = src[i];
    }
}

// The <= 4 in-edge (ELL) rows a decision changes, for the group's instances: lane = (instance g, row r): r = 0 the acting
// task a, 1 its job successor, 2 its new route successor, 3 the node whose merged job+machine edge reverts (env:1384-1422,
// 1607-1675, 1703-1765, 2019, 2060-2062) — k_env_reg's arithmetic with the gathers going to the instance's LDS arrays.
template <typename OBS, int NL>
__device__ __forceinline__ void env_grp_ell(const EnvParams &P, const int b0, const int lane, const int EG, const int (*s_in)[12],
                                            const int (*s_mp)[EgPitch<NL>::MP], const double (*s_sdf)[EgPitch<NL>::SDF], const double (*s_ttl)[EgPitch<NL>::TTL],
                                            const double (*s_scl)[SCAL_N])
{
    const int g = lane >> 2, r = lane & 3;
    const int b = b0 + g;
    if (g >= EG || b >= P.B || !s_in[g][I_VALID]) return;
    const int M = P.M, T = P.T;
    const unsigned invM = P.inv_M;
#define DIVM(x) ((int)__umulhi((unsigned)(x), invM))
    const size_t bT = (size_t)b * T;
    const int a = s_in[g][I_A], Nk = s_in[g][I_NK], lastm = s_in[g][I_LASTM], merged_now = s_in[g][I_MERGED];
    const int ja = DIVM(a), op = a - ja * M;
    const int vv = r == 0 ? a : r == 1 ? ((op + 1 < M) ? a + 1 : -1) : r == 2 ? Nk : lastm;
    if (vv < 0) return;
    const int *mach = s_mp[g], *prev = s_mp[g] + NL;
    const double *st = s_sdf[g], *dur = s_sdf[g] + NL, *ft = s_sdf[g] + 2 * NL, *ttc = s_ttl[g];      // ttc[x] = tt[x, m]
    const int mv = mach[vv], pr = prev[vv];
    const double st_v = st[vv];
    const int jvv = DIVM(vv), opvv = vv - jvv * M;
    const int u = vv > 0 ? vv - 1 : 0;
    const int mu = mach[u];
    const double dur_u = dur[u], ft_u = ft[u];
    const int pri = pr >= 0 ? pr : 0;
    const int mpr = mach[pri];
    const double dur_p = dur[pri], ft_p = ft[pri];
    // rows 0 and 2 (a, its new route successor) sit on machine m: column m.  Row 1 (the job successor) is unscheduled: no transport
    // term.  Row 3 (the merged edge of the previous step that reverts): tt[mach(lastm - 1), mach(lastm)] was that step's job-edge
    // transport time (S_TRLAST); when a lands between the merged pair, row 3 is row 2's node on machine m and both forms agree.
    const double tt_uv = r == 3 ? s_scl[g][S_TRLAST] : ttc[mu >= 0 ? mu : 0];
    const double tt_pv = ttc[mpr >= 0 ? mpr : 0];
    const bool s = mv >= 0;
    const bool merged = pr >= 0 && opvv != 0 && pr == vv - 1;
    int c_job = -1, c_mch = -1;
    float a_job = 0.f, a_mch = 0.f;
    if (opvv != 0) {
        double w, nd;
        if (mu < 0) { w = 1.0; nd = 1.0; }
        else {
            nd = dur_u;
            if (merged && vv == merged_now) w = nd + tt_uv + (st_v - ft_u);                                             // env:1607-1675,1703-1765
            else w = nd + (s ? tt_uv : 0.0);                                                                           // env:1384-1422
        }
        long A = trunc_l(w);
        if (A != 0) { A = trunc_l((double)A - nd) + 1; c_job = u; a_job = (float)A; }                                   // env:2019, 2060-2062
    }
    if (pr >= 0 && !merged) {
        const double x = (DIVM(pri) == jvv) ? tt_pv : 0.0;
        const double w = dur_p + x + (st_v - ft_p);
        long A = trunc_l(w);
        if (A != 0) { A = trunc_l((double)A - dur_p) + 1; c_mch = pr; a_mch = (float)A; }
    }
    reinterpret_cast<int2 *>(P.obs.ell_col)[bT + vv] = make_int2(c_job, c_mch);
    reinterpret_cast<float2 *>(P.obs.ell_val)[bT + vv] = make_float2(a_job, a_mch);
    if (r == 2) reinterpret_cast<OBS *>(P.obs.tasks_fea)[(bT + vv) * 12 + 4] = (OBS)(1 + ((pr >= 0 && !merged) ? 1 : 0));
#undef DIVM
}

template <typename OBS, int EG, int NS>
__device__ __forceinline__ void env_grp_body(const EnvParams &P)
{
    constexpr int NL = 64 * NS;                // task slots per instance
    using PT = EgPitch<NL>;
    __shared__ double s_sorted[EG][PT::SORTED];    // idle terms in (machine, route position) rank order
    __shared__ double s_jmx[EG][PT::JOB], s_jrw[EG][PT::JOB];
    __shared__ int s_cn[EG][PT::CN];
    __shared__ double s_scl[EG][SCAL_N];
    __shared__ double s_mf[EG][PT::MF];
    __shared__ double s_un[EG][PT::UN];
    __shared__ int s_in[EG][12];
    __shared__ MJRec s_mj2[EG][2];             // the records of machine m and of job ja (after the step)
    __shared__ double s_row[EG][48];           // estimated start | estimated finish | energy of the acting job's ops
    __shared__ int s_mp[EG][PT::MP];           // machine | route predecessor per task (after the step)
    __shared__ double s_sdf[EG][PT::SDF];      // start | duration | finish per task
    __shared__ double s_ttl[EG][PT::TTL];      // transport times
    const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b0 = blockIdx.x * EG;
    const int lane = threadIdx.x & 63;
    unsigned long long rt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#ifdef MTFJSP_STAMP
    rt[0] = __builtin_amdgcn_s_memrealtime();
#endif
    // (issue priority by quartet — s_setprio 1..3 for the waves whose data arrives later — was measured in round 6: it reverses the order in
    // which the quartets' loads are served as well, the barrier moved from 4.65 to 4.97 us: not kept)
    if (b0 + grp < P.B) env_grp_wave<OBS, NS>(P, b0 + grp, lane, s_sorted[grp], s_jmx[grp], s_jrw[grp], s_cn[grp], s_scl[grp], s_mf[grp], s_un[grp], s_in[grp], s_mj2[grp], s_row[grp], s_mp[grp], s_sdf[grp], s_ttl[grp], rt);
#ifdef MTFJSP_STAMP_WAVES                      // diagnostic: every wave's entry / first-hop / end of its instance's work (slots 16.. of the group's 128)
    if (P.stamps && lane == 0 && EG == 16) {
        P.stamps[2048 + (size_t)blockIdx.x * 64 + grp * 4] = rt[0];
        P.stamps[2048 + (size_t)blockIdx.x * 64 + grp * 4 + 1] = rt[1];
        P.stamps[2048 + (size_t)blockIdx.x * 64 + grp * 4 + 2] = rt[3];
        P.stamps[2048 + (size_t)blockIdx.x * 64 + grp * 4 + 3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    __syncthreads();
#ifdef MTFJSP_STAMP
    rt[5] = __builtin_amdgcn_s_memrealtime();
#endif
    // three scalar-part waves (on different SIMDs): rewards / scaler / machine row | ELL rows | job mask + state words
    if (grp == 0) {
        const EnvGrpRegAcc<NL> acc{s_sorted, s_jmx, s_jrw, s_cn, s_scl, s_mf, s_un, s_in};
#ifdef MTFJSP_STAMP_WAVES
        unsigned long long rt2[4] = {0, 0, 0, 0};
        env_grp_tail<OBS, EG == EG_SMALL>(P, b0, lane, EG, acc, rt2);
        if (P.stamps && lane == 0 && EG == 16) for (int i = 0; i < 4; i++) P.stamps[2048 + 256 * 64 + (size_t)blockIdx.x * 4 + i] = rt2[i];
#else
        env_grp_tail<OBS, EG == EG_SMALL>(P, b0, lane, EG, acc);
#endif
    } else if (grp == 1) {
        const EnvGrpRegAcc<NL> acc{s_sorted, s_jmx, s_jrw, s_cn, s_scl, s_mf, s_un, s_in};
        env_grp_ell<OBS, NL>(P, b0, lane, EG, s_in, s_mp, s_sdf, s_ttl, s_scl);
    } else if (grp == 2) {
        const EnvGrpRegAcc<NL> acc{s_sorted, s_jmx, s_jrw, s_cn, s_scl, s_mf, s_un, s_in};
        env_grp_mask(P, b0, lane, EG, acc);
        env_grp_state(P, b0, lane, EG, acc, s_mj2);
    } else if (grp >= 3) {
        const EnvGrpRegAcc<NL> acc{s_sorted, s_jmx, s_jrw, s_cn, s_scl, s_mf, s_un, s_in};
        env_grp_rows<OBS>(P, b0, lane, EG, acc, s_row, grp - 3, EG - 3);
    }
#ifdef MTFJSP_STAMP
    if (grp == 0) {
        rt[6] = __builtin_amdgcn_s_memrealtime();
        if (P.stamps && lane == 0) for (int i = 0; i < 7; i++) P.stamps[(size_t)blockIdx.x * 8 + i] = rt[i];
    } else if (grp == 1) {                     // slot 7: the ELL / job-mask wave's end
        if (P.stamps && lane == 0) P.stamps[(size_t)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    (void)rt;
}
// The same step for a group of EG_SMALL instances as the TAIL of another kernel's workgroup (k_headsx_envstep, mtfjsp_encoder.hip:
// the machine actor's heads have just selected the machines of exactly these instances): the per-instance arrays are carved from
// the caller's dynamic LDS (free once the caller's own phases are done) and NW waves take the instances in EG_SMALL / NW rounds.
// Same functions, same operations: bit-identical to k_env_grp16.
template <int NS>
struct EnvGrpDynLds {
    static constexpr int NL = 64 * NS, EG = EG_SMALL;
    using PT = EgPitch<NL>;
    static constexpr size_t o_sorted = 0, o_jmx = o_sorted + sizeof(double) * EG * PT::SORTED, o_jrw = o_jmx + sizeof(double) * EG * PT::JOB,
                            o_scl = o_jrw + sizeof(double) * EG * PT::JOB, o_mf = o_scl + sizeof(double) * EG * SCAL_N,
                            o_un = o_mf + sizeof(double) * EG * PT::MF, o_sdf = o_un + sizeof(double) * EG * PT::UN, o_ttl = o_sdf + sizeof(double) * EG * PT::SDF,
                            o_cn = o_ttl + sizeof(double) * EG * PT::TTL, o_in = o_cn + sizeof(int) * EG * PT::CN, o_mp = o_in + sizeof(int) * EG * 12,
                            o_mj2 = (o_mp + sizeof(int) * EG * PT::MP + 7) / 8 * 8, o_row = o_mj2 + sizeof(MJRec) * EG * 2,
                            bytes = (o_row + sizeof(double) * EG * 48 + 15) / 16 * 16;
};
template <typename OBS, int NS, int NW>
__device__ __forceinline__ void env_grp_body_dyn(const EnvParams &P, unsigned char *smem)
{
    using L = EnvGrpDynLds<NS>;
    using PT = typename L::PT;
    constexpr int NL = L::NL, EG = L::EG;
    auto s_sorted = reinterpret_cast<double (*)[PT::SORTED]>(smem + L::o_sorted);
    auto s_jmx = reinterpret_cast<double (*)[PT::JOB]>(smem + L::o_jmx);
    auto s_jrw = reinterpret_cast<double (*)[PT::JOB]>(smem + L::o_jrw);
    auto s_cn = reinterpret_cast<int (*)[PT::CN]>(smem + L::o_cn);
    auto s_scl = reinterpret_cast<double (*)[SCAL_N]>(smem + L::o_scl);
    auto s_mf = reinterpret_cast<double (*)[PT::MF]>(smem + L::o_mf);
    auto s_un = reinterpret_cast<double (*)[PT::UN]>(smem + L::o_un);
    auto s_in = reinterpret_cast<int (*)[12]>(smem + L::o_in);
    auto s_mp = reinterpret_cast<int (*)[PT::MP]>(smem + L::o_mp);
    auto s_sdf = reinterpret_cast<double (*)[PT::SDF]>(smem + L::o_sdf);
    auto s_ttl = reinterpret_cast<double (*)[PT::TTL]>(smem + L::o_ttl);
    auto s_mj2 = reinterpret_cast<MJRec (*)[2]>(smem + L::o_mj2);
    auto s_row = reinterpret_cast<double (*)[48]>(smem + L::o_row);
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b0 = blockIdx.x * EG;
    const int lane = threadIdx.x & 63;
    unsigned long long rt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 1
    for (int g = w; g < EG; g += NW)
        if (b0 + g < P.B) env_grp_wave<OBS, NS>(P, b0 + g, lane, s_sorted[g], s_jmx[g], s_jrw[g], s_cn[g], s_scl[g], s_mf[g], s_un[g], s_in[g], s_mj2[g], s_row[g], s_mp[g], s_sdf[g], s_ttl[g], rt);
    __syncthreads();
    if (w == 0) {
        const EnvGrpRegAcc<NL> acc{s_sorted, s_jmx, s_jrw, s_cn, s_scl, s_mf, s_un, s_in};
        env_grp_tail<OBS, false>(P, b0, lane, EG, acc);
    } else if (w == 1) {
        const EnvGrpRegAcc<NL> acc{s_sorted, s_jmx, s_jrw, s_cn, s_scl, s_mf, s_un, s_in};
        env_grp_ell<OBS, NL>(P, b0, lane, EG, s_in, s_mp, s_sdf, s_ttl, s_scl);
    } else if (w == 2) {
        const EnvGrpRegAcc<NL> acc{s_sorted, s_jmx, s_jrw, s_cn, s_scl, s_mf, s_un, s_in};
        env_grp_mask(P, b0, lane, EG, acc);
        env_grp_state(P, b0, lane, EG, acc, s_mj2);
    } else {
        const EnvGrpRegAcc<NL> acc{s_sorted, s_jmx, s_jrw, s_cn, s_scl, s_mf, s_un, s_in};
        env_grp_rows<OBS>(P, b0, lane, EG, acc, s_row, w - 3, NW - 3);
    }
    static_assert(NW >= 4, "four scalar-part waves");
    (void)rt;
}
#ifndef MTFJSP_ENV_GRP_NO_KERNELS
template <typename OBS>
__global__ __launch_bounds__(EG_SMALL * WAVE) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_env_grp16(EnvParams P) { env_grp_body<OBS, EG_SMALL, 1>(P); }
template <typename OBS>
__global__ __launch_bounds__(EG_LARGE * WAVE) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_env_grp4(EnvParams P) { env_grp_body<OBS, EG_LARGE, 1>(P); }
// two task slots per lane: 64 < T <= 128, M*M <= 128, M <= 16 (J10M10 and the like)
template <typename OBS>
__global__ __launch_bounds__(EG_SMALL * WAVE) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_env_grp16x2(EnvParams P) { env_grp_body<OBS, EG_SMALL, 2>(P); }
template <typename OBS>
__global__ __launch_bounds__(EG_LARGE * WAVE) void k_env_grp4x2(EnvParams P) { env_grp_body<OBS, EG_LARGE, 2>(P); }
#endif
