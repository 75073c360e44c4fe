// mtfjsp_env.hip — batched MT-FJSP disjunctive-graph environment for MI355X (gfx950 / CDNA4).
//
// One 64-lane wavefront (= one 64-thread workgroup) owns one instance for the duration of a launch; lanes are
// tasks.  A batched step is
//   A. left-shift scheduling decision  — gap search evaluated for all route positions in parallel, no list walk
//   B. cost update                     — ordered idle sum, per-job estimated-finish scan, makespan max-reduce,
//                                        numpy-order energy sum, rewards, Welford reward scaler
//   C. observation                     — only the rows the decision changes (task features of the acting job's
//                                        unscheduled tail, <= 4 ELL in-edge rows, one m_fea2 row); k_env_reset
//                                        writes the full observation once per episode
//   D. candidate / job mask            — derived from the per-job counters
// Kernels: k_env_reset, k_env_reg (T <= 64: state in registers, readlane gathers, no LDS), k_env_step (LDS).
// in ONE kernel launch (mtfjsp_step).  The graph is never materialised: with a simple digraph the
// in-edges of node v are exactly {job predecessor, route predecessor}, and every edge weight the
// reference stores is a closed form of (dur, st, ft, machine) of its two endpoints.
//
// Semantics follow the reference bit for bit (binary64, same operation order, no FMA contraction:
// build with -ffp-contract=off).  "env:" = graph-jsp-env/src/graph_jsp_env/disjunctive_graph_jsp_env_singlestep.py,
// "dg:" = trainer/DGenv_func.py, "pe:" = trainer/parallel_env.py, "pt:" = algorithm/ppo_trick.py,
// "ppo:" = algorithm/ppo_algorithm.py.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/mtfjsp.h"

#include "mtfjsp_env_dev.h"

// ---------------------------------------------------------------------------------------------
// numpy float64 add.reduce order: 0 + pairwise_sum (8 accumulators, 128-element leaf blocks).
// env:896 np.sum(pt_est) and pe:176-183 np.mean use it; restated so e1/r_pt are bit-identical.
__device__ __forceinline__ double pw_leaf(const double *a, int n)
{
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; i++) r += a[i];
        return r;
    }
    double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
        r0 += a[i]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3];
        r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; i++) res += a[i];
    return res;
}
template <int DEPTH>
__device__ __noinline__ double pw_sum(const double *a, int n)
{
    if (n <= 128) return pw_leaf(a, n);
    int n2 = n / 2;
    n2 -= n2 % 8;
    return pw_sum<DEPTH - 1>(a, n2) + pw_sum<DEPTH - 1>(a + n2, n - n2);
}
template <>
__device__ __noinline__ double pw_sum<0>(const double *a, int n) { return pw_leaf(a, n); }
__device__ __forceinline__ double np_sum(const double *a, int n) { return 0.0 + pw_sum<6>(a, n); }   // n <= 8192


// ---------------------------------------------------------------------------------------------
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
// grid = B workgroups of 64 threads for all three environment kernels.
// wave-local LDS fence: a workgroup is ONE wavefront, the LDS pipeline executes a wave's DS instructions in issue
// order, so cross-lane hand-offs through LDS only need the compiler not to reorder (and no vmcnt drain, unlike
// __syncthreads()).
#define WSYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

__device__ __forceinline__ double wave_max(double x)
{
    for (int o = 32; o > 0; o >>= 1) x = fmax(x, __shfl_xor(x, o));
    return x;
}
// the same reductions on the cross-lane data path (DPP: no LDS round trip per step): after four row shifts lane 15 of every row
// of 16 holds its row's result, row_bcast:15 / :31 carry it on; the wave's result is in LANE 63 only.  Lanes without a source
// keep their own value (max / min are idempotent).
#define DPP_I(x, ctrl) __builtin_amdgcn_update_dpp((x), (x), (ctrl), 0xF, 0xF, false)
__device__ __forceinline__ double wave_max_lane63(double x)
{
#define STEP_(ctrl)                                                                                       \
    {                                                                                                    \
        const int lo = DPP_I(__double2loint(x), ctrl), hi = DPP_I(__double2hiint(x), ctrl);              \
        x = fmax(x, __hiloint2double(hi, lo));                                                           \
    }
    STEP_(0x111) STEP_(0x112) STEP_(0x114) STEP_(0x118) STEP_(0x142) STEP_(0x143)
#undef STEP_
    return x;
}
__device__ __forceinline__ int wave_min_lane63(int x)
{
#define STEP_(ctrl) { const int y = DPP_I(x, ctrl); x = y < x ? y : x; }
    STEP_(0x111) STEP_(0x112) STEP_(0x114) STEP_(0x118) STEP_(0x142) STEP_(0x143)
#undef STEP_
    return x;
}
// inclusive prefix sum over the wave's lanes (zero fill; row_bcast adds the previous rows' totals)
__device__ __forceinline__ int wave_scan_incl(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);      // rows 1, 3 += lane 15 of the row before
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);      // rows 2, 3 += lane 31
    return x;
}

// Philox4x32-10 (counter-based; Salmon et al. 2011)
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
// env.generate_random_weights("01") (env:1253-1259) of instance b: three uniforms in [0,1), normalised by their sum (numpy's sum of three =
// left-to-right adds).  53-bit uniforms like python's random.random(): (a >> 5, b >> 6) -> (a*2^26 + b) / 2^53.
__device__ __forceinline__ void draw_w3(int b, uint64_t seed, uint64_t episode, double (&w)[3])
{
    double u[3];
    for (int i = 0; i < 3; i += 2) {
        uint32_t c[4] = {(uint32_t)b, (uint32_t)episode, (uint32_t)(episode >> 32), 0x77337733u + (uint32_t)i};
        philox4x32(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        u[i] = ((double)(c[0] >> 5) * 67108864.0 + (double)(c[1] >> 6)) * (1.0 / 9007199254740992.0);
        if (i + 1 < 3) u[i + 1] = ((double)(c[2] >> 5) * 67108864.0 + (double)(c[3] >> 6)) * (1.0 / 9007199254740992.0);
    }
    const double sum = (u[0] + u[1]) + u[2];
    w[0] = u[0] / sum; w[1] = u[1] / sum; w[2] = u[2] / sum;
}

// =================================================================================================
// k_env_reset — env.reset() of every instance (env:1183-1245 + load_instance env:397-714): all tasks unscheduled,
// estimated times from the per-job prefix sums of min_dur, full observation written once (the step kernels then
// only touch the rows a decision changes).  One wavefront per instance.
template <typename OBS>
__global__ __launch_bounds__(WAVE) void k_env_reset(EnvParams P)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int J = P.J, M = P.M, T = P.T;
    const unsigned invM = P.inv_M;
    double *s_mind = reinterpret_cast<double *>(smem);          // T
    double *s_pte = s_mind + T;                                 // T
    double *s_fte = s_pte + T;                                  // T
    OBS *s_stage = reinterpret_cast<OBS *>(s_fte + T);          // min(T,64) rows x 12
    const size_t bT = (size_t)b * T;
    for (int v = lane; v < T; v += WAVE) { const double2 c = P.cst[bT + v]; s_mind[v] = c.x; s_pte[v] = c.y; }
    double w30, w31, w32;
    if (P.draw) {                                               // (every lane of the instance's wave: the same three numbers)
        double w[3];
        draw_w3(b, P.draw_seed, P.draw_episode, w);
        w30 = w[0]; w31 = w[1]; w32 = w[2];
        if (lane < 3 && P.w3_out) P.w3_out[(size_t)b * 3 + lane] = lane == 0 ? w30 : lane == 1 ? w31 : w32;
    } else { w30 = P.w3[b * 3]; w31 = P.w3[b * 3 + 1]; w32 = P.w3[b * 3 + 2]; }
    WSYNC();
    double mkmax = -INFINITY;
    for (int c0 = 0; c0 < T; c0 += WAVE) {
        const int v = c0 + lane;
        const int rows = (T - c0) < WAVE ? (T - c0) : WAVE;
        if (v < T) {
            const int jv = (int)__umulhi((unsigned)v, invM), c = v - jv * M;
            double acc = 0.0, prev = 0.0;                                   // env:1965-1993 on an empty schedule
            for (int k = 0; k <= c; k++) { prev = acc; acc = acc + s_mind[jv * M + k]; }
            s_fte[v] = acc;
            mkmax = fmax(mkmax, acc);
            OBS *f = s_stage + lane * 12;                                   // env:2245-2277
            f[0] = (OBS)(c == 0 ? 0.0 : prev); f[1] = (OBS)acc; f[2] = (OBS)s_pte[v];
            f[3] = (OBS)0; f[4] = (OBS)1; f[5] = (OBS)0; f[6] = (OBS)0; f[7] = (OBS)0;
            f[8] = (OBS)(jv + 1); f[9] = (OBS)w30; f[10] = (OBS)w31; f[11] = (OBS)w32;
            reinterpret_cast<int2 *>(P.obs.ell_col)[bT + v] = make_int2(c != 0 ? v - 1 : -1, -1);   // job edge weight 1 (env:617-644)
            reinterpret_cast<float2 *>(P.obs.ell_val)[bT + v] = make_float2(c != 0 ? 1.f : 0.f, 0.f);
            Link l; l.mach = -1; l.prev = -1; l.pos = 0; l.pad = -1;                               // pad = route successor
            TaskPL y; y.pte = s_pte[v]; y.link = l;
            P.pl[bT + v] = y;
            TaskSD x; x.st = 0.0; x.dur = 0.0;
            P.sd[bT + v] = x;
        }
        WSYNC();
        const int n16 = rows * 12 * (int)sizeof(OBS) / 16;
        const uint4 *src = reinterpret_cast<const uint4 *>(s_stage);
        uint4 *dst = reinterpret_cast<uint4 *>(reinterpret_cast<OBS *>(P.obs.tasks_fea) + (bT + c0) * 12);
        for (int i = lane; i < n16; i += WAVE) dst[i] = src[i];
        WSYNC();
    }
    const double mk = wave_max(mkmax);                                      // env:683-705 initial "previous" values
    const double e1 = np_sum(s_pte, T);
    for (int j = lane; j < J; j += WAVE) {
        double fm = s_fte[j * M];
        for (int c = 1; c < M; c++) fm = fmax(fm, s_fte[j * M + c]);
        JobR r; r.jmax = fm; r.jrow = 0.0;
        P.jr[(size_t)b * J + j] = r;
        P.obs.candidate[(size_t)b * J + j] = j * M;                         // ppo:90-98 pool = first op of every job
        P.obs.job_mask[(size_t)b * J + j] = 0;
    }
    for (int i = lane; i < P.MJ; i += WAVE) { MJRec r; r.head = -1; r.tail = -1; r.len = 0; r.cnt = 0; P.mj[(size_t)b * P.MJ + i] = r; }
    for (int i = lane; i < M * 8; i += WAVE) {
        const int f = i & 7;
        const double x = f == 5 ? w30 : f == 6 ? w31 : f == 7 ? w32 : 0.0;  // env:2343-2354
        P.mfea[(size_t)b * M * 8 + i] = x;
        reinterpret_cast<OBS *>(P.obs.m_fea2)[(size_t)b * M * 8 + i] = (OBS)x;
    }
    if (lane < SCAL_N) {
        double x = 0.0;
        if (lane >= S_R && lane <= S_N && !(P.reset_returns && lane < S_R + 4)) x = P.scal[(size_t)b * SCAL_N + lane];   // the scaler survives resets (pe:70-85); its returns R are zeroed per episode (pt:123) when asked
        else if (lane == S_MK_PREV) x = mk;
        else if (lane == S_E1_PREV) x = e1;
        else if (lane == S_W3) x = w30;
        else if (lane == S_W3 + 1) x = w31;
        else if (lane == S_W3 + 2) x = w32;
        else if (lane == S_LASTM) x = -1.0;
        P.scal[(size_t)b * SCAL_N + lane] = x;
    }
    if (lane < 6) P.obs.info[(size_t)b * 6 + lane] = 0.0;
    if (lane < 5 && P.obs.raw) P.obs.raw[(size_t)b * 5 + lane] = 0.0;
    if (lane == 6) P.obs.status[b] = 0;
}
static size_t env_reset_lds_bytes(int T, bool f32) { return (size_t)3 * T * 8 + (size_t)(T < WAVE ? T : WAVE) * 12 * (f32 ? 4 : 8) + 16; }

// =================================================================================================
// k_env_step — incremental step kernel (one wavefront per instance, one launch per batched step).
// Observations are persistent in the bound buffers, and one scheduling decision only changes
//   * task a and the unscheduled rest of its job (estimated start/finish restart from ft[a])          -> M-op(a) feature rows
//   * the in-edge slots of a, of its job successor, of its new route successor and of the node whose merged edge
//     reverts to the job-edge refresh value (§DESIGN 3.1)                                              -> <= 4 ELL rows
//   * machine row m of m_fea2, candidate[job], the job mask, rewards
// so only those are rewritten; the per-task state needed for decisions (link, st, ft, dur) and for the two
// order-sensitive sums (pt_est for numpy's pairwise sum, st/ft for the ordered idle sum) is staged in LDS.
template <typename OBS>
__global__ __launch_bounds__(WAVE, 4) void k_env_step(EnvParams P)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    const int J = P.J, M = P.M, T = P.T;
    const unsigned invM = P.inv_M;
#define DIVM(x) ((int)__umulhi((unsigned)(x), invM))
#ifdef MTFJSP_STAMP
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_last)::"memory");
#endif
    const int Tp = (T + 7) & ~7;
    double *s_st = reinterpret_cast<double *>(smem);
    double *s_ft = s_st + T;
    double *s_dur = s_ft + T;
    double *s_pte = s_dur + T;
    double *s_term = s_pte + T;                // Tp
    double *s_tt = s_term + Tp;                // M*M
    double *s_mind = s_tt + M * M;             // M: min_dur of the acting job's ops
    double *s_jste = s_mind + M;               // M: estimated start of the acting job's ops
    double *s_jfte = s_jste + M;               // M
    double *s_jmax = s_jfte + M;               // J: max estimated finish per job
    double *s_jrow = s_jmax + J;               // J: max real finish of scheduled ops per job
    double *s_sc = s_jrow + J;                 // SCAL_N
    double *s_mfr = s_sc + SCAL_N;             // 8: m_fea2 row of machine m
    size_t off = (size_t)((4 * T + Tp + M * M + 3 * M + 2 * J + SCAL_N + 8) * sizeof(double));
    off = (off + 15) & ~(size_t)15;
    OBS *s_stage = reinterpret_cast<OBS *>(smem + off);        // M rows x 12
    off += (size_t)M * 12 * sizeof(OBS);
    off = (off + 15) & ~(size_t)15;
    int *s_mach = reinterpret_cast<int *>(smem + off);
    int *s_prev = s_mach + T;
    int *s_pos = s_prev + T;
    int *s_cnt = s_pos + T;                    // J
    int *s_head = s_cnt + J;                   // M
    int *s_tail = s_head + M;
    int *s_len = s_tail + M;
    int *s_mstart = s_len + M;                 // M+1

    const size_t bT = (size_t)b * T;
    // ---- bulk state first (independent of the action), the action-dependent second hop right behind it
    for (int v = lane; v < T; v += WAVE) {
        const TaskSD x = P.sd[bT + v]; const TaskPL y = P.pl[bT + v];
        s_st[v] = x.st; s_dur[v] = x.dur; s_ft[v] = x.st + x.dur; s_pte[v] = y.pte;      // ft == st + dur (env:356)
        const Link l = y.link;
        s_mach[v] = l.mach; s_prev[v] = l.prev; s_pos[v] = l.pos;
    }
    for (int i = lane; i < Tp; i += WAVE) s_term[i] = 0.0;
    for (int i = lane; i < M * M; i += WAVE) s_tt[i] = P.tt[(size_t)b * M * M + i];
    for (int i = lane; i < P.MJ; i += WAVE) { const MJRec r = P.mj[(size_t)b * P.MJ + i]; if (i < M) { s_head[i] = r.head; s_tail[i] = r.tail; s_len[i] = r.len; } if (i < J) s_cnt[i] = r.cnt; }
    for (int i = lane; i < J; i += WAVE) { const JobR r = P.jr[(size_t)b * J + i]; s_jmax[i] = r.jmax; s_jrow[i] = r.jrow; }
    if (lane < SCAL_N) s_sc[lane] = P.scal[(size_t)b * SCAL_N + lane];
    const int lastm = (int)P.scal[(size_t)b * SCAL_N + S_LASTM];
    int a = P.task_idx[b], m = P.mach_idx[b];
    bool valid = a >= 0 && a < T && m >= 0 && m < M;
    if (!valid) { a = 0; m = 0; }
    const int ja = DIVM(a), op = a - ja * M;
    const double d = P.t[(bT + a) * M + m];
    const double pk = P.p[(bT + a) * M + m];
    for (int i = lane; i < M; i += WAVE) s_mind[i] = P.cst[bT + ja * M + i].x;
    if (lane < 8) s_mfr[lane] = P.mfea[((size_t)b * M + m) * 8 + lane];
    WSYNC();
    STAMP(0);

    // =========================================================================================
    // A. scheduling (env:1476-1685)
    int status = 0, path = 0;
    int Pk = -1, Nk = -1, ipos = 0;
    double st_k = 0.0, ft_k = 0.0;
    if (valid) {
        if (s_mach[a] >= 0) valid = false;                         // already scheduled (env:1504)
        else if (op != 0 && s_mach[a - 1] < 0) valid = false;      // job predecessor unscheduled (env:1520)
    }
    if (valid) {
        if (d < 0.0) status |= MTFJSP_ST_INFEASIBLE;               // pe:246-248
        const double ttmm = s_tt[m * M + m];
        const double arr_k = op == 0 ? 0.0 : s_ft[a - 1] + s_tt[s_mach[a - 1] * M + m];      // dg:46-66 over the single in-edge
        const int len = s_len[m], head = s_head[m], tail = s_tail[m];
        bool do_append = false;
        if (len == 0) { path = MTFJSP_PATH_EMPTY; st_k = arr_k; ipos = 0; }                      // env:1684
        else if (!P.left_shift) do_append = true;                                                     // env:1680
        else {
            const double lb_ft = arr_k + d;
            const int jh = DIVM(head);
            const double arr_f = (head == jh * M) ? 0.0 : s_ft[head - 1] + s_tt[s_mach[head - 1] * M + m];
            if (lb_ft <= arr_f) { path = MTFJSP_PATH_FRONT; st_k = arr_k; ipos = 0; Nk = head; }   // env:1548
            else if (len == 1) do_append = true;                                                     // env:1577
            else {
                int key = 0x7fffffff;                               // gap search over all consecutive (P,N) at once (env:1587-1604)
                for (int v = lane; v < T; v += WAVE) {
                    if (s_mach[v] == m && s_prev[v] >= 0) {
                        const int Pp = s_prev[v];
                        const int jv = DIVM(v);
                        const double jarr = (v == jv * M) ? 0.0 : s_ft[v - 1] + s_tt[s_mach[v - 1] * M + m];
                        const double x = (DIVM(Pp) == jv) ? ttmm : 0.0;
                        const double nst = fmax(jarr, s_ft[Pp] + x);
                        const bool ok = !(lb_ft > nst) && !((nst - s_ft[Pp]) < d);
                        if (ok) { const int kk = (s_pos[v] << 16) | v; key = kk < key ? kk : key; }
                    }
                }
                for (int o = 32; o > 0; o >>= 1) { const int other = __shfl_xor(key, o); key = other < key ? other : key; }
                if (key != 0x7fffffff) {
                    path = MTFJSP_PATH_BETWEEN;
                    Nk = key & 0xffff; ipos = key >> 16; Pk = s_prev[Nk];
                    const double x = (DIVM(Pk) == ja) ? ttmm : 0.0;
                    st_k = fmax(arr_k, s_ft[Pk] + x);                          // env:1619
                } else do_append = true;                                       // env:1676
            }
        }
        if (do_append) {                                                       // env:1689-1775
            path = MTFJSP_PATH_APPEND;
            const double x = (DIVM(tail) == ja) ? ttmm : 0.0;
            st_k = fmax(arr_k, s_ft[tail] + x);
            ipos = len; Pk = tail;
        }
        ft_k = st_k + d;
        WSYNC();
        if (ipos < len)
            for (int v = lane; v < T; v += WAVE)
                if (s_mach[v] == m && s_pos[v] >= ipos) s_pos[v] += 1;
        WSYNC();
        if (lane == 0) {
            s_mach[a] = m; s_prev[a] = Pk; s_pos[a] = ipos;
            s_st[a] = st_k; s_ft[a] = ft_k; s_dur[a] = d; s_pte[a] = d * pk;      // env:356,2175
            if (Nk >= 0) s_prev[Nk] = a;
            if (ipos == 0) s_head[m] = a;
            if (ipos == len) s_tail[m] = a;
            s_len[m] = len + 1;
            s_cnt[ja] += 1;
            s_sc[S_NSCHED] += 1.0;
        }
        status |= path;
        WSYNC();
    } else status |= MTFJSP_ST_INVALID;
    STAMP(1);
    if (!valid) {                                                               // nothing changes; observations persist
        const bool all_done = s_sc[S_NSCHED] == (double)T;
        if (lane < 6) P.obs.info[(size_t)b * 6 + lane] = (lane == 1 && all_done) ? 1.0 : 0.0;
        if (P.obs.raw && lane < 5) P.obs.raw[(size_t)b * 5 + lane] = 0.0;
        if (P.rec_r4 && lane < 4) P.rec_r4[(size_t)lane * P.B + b] = 0.f;
        if (P.rec_done && lane == 4) P.rec_done[b] = all_done ? 1.f : 0.f;
        if (lane == 5) P.obs.status[b] = status;
        return;
    }

    // =========================================================================================
    // B. costs
    {   // machine route offsets: exclusive scan of the route lengths
        int x = lane < M ? s_len[lane] : 0, incl = x;
        for (int o = 1; o < WAVE; o <<= 1) { const int y = __shfl_up(incl, o); if (lane >= o) incl += y; }
        if (lane < M) s_mstart[lane] = incl - x;
        if (lane == M - 1) s_mstart[M] = incl;
    }
    // estimated start/finish of the acting job's ops (env:1920-1999): lanes = ops of job ja; every lane replays the
    // reference's left-to-right add sequence from the last op with a non-zero real finish time
    double my_fte = -INFINITY, my_rft = 0.0;
    if (lane < M) {
        const int c = lane, v = ja * M + c;
        const bool s = s_mach[v] >= 0;
        double ste, fte;
        if (s && s_ft[v] != 0.0) { ste = s_st[v]; fte = s_ft[v]; }
        else {
            int k0 = c;
            while (k0 > 0 && !(s_mach[ja * M + k0 - 1] >= 0 && s_ft[ja * M + k0 - 1] != 0.0)) k0--;
            double acc = k0 > 0 ? s_ft[ja * M + k0 - 1] : 0.0, prev = acc;
            for (int k = k0; k <= c; k++) { prev = acc; acc = acc + s_mind[k]; }
            fte = acc;
            ste = s ? s_st[v] : (c == 0 ? 0.0 : prev);
        }
        s_jste[c] = ste; s_jfte[c] = fte;
        my_fte = fte;
        my_rft = s ? s_ft[v] : 0.0;
    }
    {   // per-job maxima of the acting job (row maximum of ft_est for the makespan; of real ft for the job mask, ppo:265-275)
        double fm = my_fte, rm = lane < M ? my_rft : -INFINITY;
        for (int o = 32; o > 0; o >>= 1) { fm = fmax(fm, __shfl_xor(fm, o)); rm = fmax(rm, __shfl_xor(rm, o)); }
        if (lane == 0) { s_jmax[ja] = fm; s_jrow[ja] = rm; }
    }
    WSYNC();
    for (int v = lane; v < T; v += WAVE)                                       // idle-time terms in (machine, position) order (dg:144-170)
        if (s_mach[v] >= 0) {
            const int pr = s_prev[v];
            s_term[s_mstart[s_mach[v]] + s_pos[v]] = pr < 0 ? s_st[v] : s_st[v] - s_ft[pr];
        }
    double mk = -INFINITY;                                                     // env:894 np.amax over all tasks
    for (int j0 = 0; j0 < J; j0 += WAVE) { const int j = j0 + lane; if (j < J) mk = fmax(mk, s_jmax[j]); }
    mk = wave_max(mk);
    WSYNC();
    STAMP(2);
    double e1;                                                                 // env:896 np.sum in numpy's order
    if (T >= 8 && T <= 128) {
        double r = 0.0;
        const int nb = T - (T & 7);
        if (lane < 8) { r = s_pte[lane]; for (int i = 8 + lane; i < nb; i += 8) r += s_pte[i]; }
        r += __shfl_xor(r, 1); r += __shfl_xor(r, 2); r += __shfl_xor(r, 4);
        for (int i = nb; i < T; i++) r += s_pte[i];
        e1 = 0.0 + __shfl(r, 0);
    } else e1 = np_sum(s_pte, T);
    const int nsched = s_mstart[M];
    double idle = 0.0;                                                         // ordered idle sum (dg:147-168)
    for (int i = 0; i < nsched; i += 8) {
        const double t0 = s_term[i], t1 = s_term[i + 1], t2 = s_term[i + 2], t3 = s_term[i + 3];
        const double t4 = s_term[i + 4], t5 = s_term[i + 5], t6 = s_term[i + 6], t7 = s_term[i + 7];
        idle = idle + t0; idle = idle + t1; idle = idle + t2; idle = idle + t3;
        idle = idle + t4; idle = idle + t5; idle = idle + t6; idle = idle + t7;
    }
    const double new_tr = (op == 0) ? 0.0 : s_tt[s_mach[a - 1] * M + m];      // env:872-876
    const double trans_this = s_sc[S_TR_THIS] + new_tr;
    const double mk_prev = s_sc[S_MK_PREV], e1_prev = s_sc[S_E1_PREV];
    const double tr_prev = s_sc[S_TR_PREV], id_prev = s_sc[S_ID_PREV];
    const double r_t = 1.0 * mk_prev - mk;                                     // env:1066
    double r_pt = 1.0 * e1_prev - e1;
    r_pt = r_pt / (double)T;                                                   // env:1073-1076
    const double r_tt = 1.0 * tr_prev - trans_this;                            // env:1083
    const double r_idle = 1.0 * id_prev - idle;                                // env:1088
    const double tot_n = P.w_mk * r_t + P.w_ec * (r_pt + 1 * r_idle) + P.w_tt * r_tt * 1;                  // env:1164
    const double tot = P.divisor == 1.0 ? tot_n : tot_n / P.divisor;            // x / 1.0 == x exactly: skip the f64 divide for the default scaling_divisor
    const bool done = nsched == T;                                             // env:797-800
    const double scN = s_sc[S_N];
    double sR = 0, sMean = 0, sS = 0, sSd = 0, scaled = 0;
    if (lane < 4) {                                                            // reward scaling, one channel per lane (pt:54-124)
        const double x = lane == 0 ? r_t : lane == 1 ? r_idle : lane == 2 ? r_pt : r_tt;
        const double n = scN + 1.0;
        sR = P.gamma * s_sc[S_R + lane] + x;
        sS = s_sc[S_S + lane];
        if (n == 1.0) { sMean = sR; sSd = fabs(sR); }
        else {
            const double old = s_sc[S_MEAN + lane];
            sMean = old + (sR - old) / n;
            sS = sS + (sR - old) * (sR - sMean);
            sSd = sqrt(sS / n);
        }
        scaled = x / (sSd + 1e-8);
    }
    WSYNC();
    if (lane < 4) {
        s_sc[S_R + lane] = sR; s_sc[S_MEAN + lane] = sMean; s_sc[S_S + lane] = sS; s_sc[S_STD + lane] = sSd;
        P.obs.info[(size_t)b * 6 + 2 + lane] = scaled;
        if (P.rec_r4) P.rec_r4[(size_t)lane * P.B + b] = (float)scaled;
    }
    if (lane == 4) {
        P.obs.info[(size_t)b * 6 + 0] = tot;
        P.obs.info[(size_t)b * 6 + 1] = done ? 1.0 : 0.0;
        if (P.rec_done) P.rec_done[b] = done ? 1.f : 0.f;
        s_sc[S_N] = scN + 1.0;
        s_sc[S_MK_PREV] = mk; s_sc[S_E1_PREV] = e1; s_sc[S_TR_PREV] = trans_this; s_sc[S_ID_PREV] = idle;   // env:932-936
        s_sc[S_TR_THIS] = done ? 0.0 : trans_this;                             // env:950-960
        s_mfr[0] = s_ft[s_tail[m]];                                            // env:2315-2340
        s_mfr[1] += (pk * d) / (double)T;
        s_mfr[2] += new_tr;
        s_mfr[3] += idle - id_prev;
        s_mfr[4] += 1;
    }
    if (P.obs.raw && lane >= 8 && lane < 13) {
        const int i = lane - 8;
        P.obs.raw[(size_t)b * 5 + i] = i == 0 ? tot : i == 1 ? r_t : i == 2 ? r_idle : i == 3 ? r_pt : r_tt;
    }
    if (lane == 5) P.obs.status[b] = status;
    WSYNC();
    STAMP(3);

    // =========================================================================================
    // C. the observation rows that changed
    {   // feature rows a .. end of job (env:2245-2277)
        const int nrow = M - op;
        if (lane < nrow) {
            const int c = op + lane, v = a + lane;
            const int pr = s_prev[a];
            const bool merged0 = pr >= 0 && op != 0 && pr == a - 1;
            OBS *f = s_stage + lane * 12;
            f[0] = (OBS)s_jste[c]; f[1] = (OBS)s_jfte[c]; f[2] = (OBS)s_pte[v];
            const bool isa = lane == 0;
            f[3] = (OBS)(isa ? 1.0 : 0.0);
            f[4] = (OBS)(isa ? (1 + ((pr >= 0 && !merged0) ? 1 : 0)) : 1);    // len(G.in_edges)
            f[5] = (OBS)(isa ? m + 1 : 0);
            f[6] = (OBS)(isa ? d : 0.0);
            f[7] = (OBS)(isa ? pk : 0.0);
            f[8] = (OBS)(ja + 1);
            f[9] = (OBS)s_sc[S_W3]; f[10] = (OBS)s_sc[S_W3 + 1]; f[11] = (OBS)s_sc[S_W3 + 2];
        }
        WSYNC();
        const int n16 = nrow * 12 * (int)sizeof(OBS) / 16;
        const uint4 *src = reinterpret_cast<const uint4 *>(s_stage);
        uint4 *dst = reinterpret_cast<uint4 *>(reinterpret_cast<OBS *>(P.obs.tasks_fea) + (bT + a) * 12);
        for (int i = lane; i < n16; i += WAVE) dst[i] = src[i];
    }
    {   // in-edge (ELL) rows: a, its job successor, its new route successor, the node whose merged edge reverts
        const int merged_now = (Pk >= 0 && op != 0 && Pk == a - 1) ? a : -1;
        int v = -1;
        if (lane == 0) v = a;
        else if (lane == 1) v = (op + 1 < M) ? a + 1 : -1;
        else if (lane == 2) v = Nk;
        else if (lane == 3) v = lastm;
        if (v >= 0) {
            const int mv = s_mach[v];
            const bool s = mv >= 0;
            const int jv = DIVM(v), opv = v - jv * M;
            const int pr = s_prev[v];
            const bool merged = pr >= 0 && opv != 0 && pr == v - 1;
            int c_job = -1, c_mch = -1;
            float a_job = 0.f, a_mch = 0.f;
            if (opv != 0) {
                const int u = v - 1, mu = s_mach[u];
                double w, nd;
                if (mu < 0) { w = 1.0; nd = 1.0; }
                else {
                    nd = s_dur[u];
                    if (merged && v == merged_now) w = s_dur[u] + s_tt[mu * M + mv] + (s_st[v] - s_ft[u]);     // env:1607-1675,1703-1765
                    else w = s_dur[u] + (s ? s_tt[mu * M + mv] : 0.0);                                          // env:1384-1422
                }
                long A = trunc_l(w);
                if (A != 0) { A = trunc_l((double)A - nd) + 1; c_job = u; a_job = (float)A; }                   // env:2019, 2060-2062
            }
            if (pr >= 0 && !merged) {
                const double x = (DIVM(pr) == jv) ? s_tt[s_mach[pr] * M + mv] : 0.0;
                const double w = s_dur[pr] + x + (s_st[v] - s_ft[pr]);
                long A = trunc_l(w);
                if (A != 0) { A = trunc_l((double)A - s_dur[pr]) + 1; c_mch = pr; a_mch = (float)A; }
            }
            reinterpret_cast<int2 *>(P.obs.ell_col)[bT + v] = make_int2(c_job, c_mch);
            reinterpret_cast<float2 *>(P.obs.ell_val)[bT + v] = make_float2(a_job, a_mch);
            if (lane == 2) reinterpret_cast<OBS *>(P.obs.tasks_fea)[(bT + v) * 12 + 4] = (OBS)(1 + ((pr >= 0 && !merged) ? 1 : 0));
        }
        if (lane == 6) { s_sc[S_LASTM] = (double)merged_now; s_sc[S_TRLAST] = new_tr; }    // (written back with the scalar row below)
    }
    if (lane < 8) reinterpret_cast<OBS *>(P.obs.m_fea2)[((size_t)b * M + m) * 8 + lane] = (OBS)s_mfr[lane];
    STAMP(5);

    // =========================================================================================
    // D. candidate + job mask (ppo:202-316)
    {
        int cmin = M;
        double mn = INFINITY;
        for (int j0 = 0; j0 < J; j0 += WAVE) {
            const int j = j0 + lane;
            int c = j < J ? s_cnt[j] : M;
            for (int o = 32; o > 0; o >>= 1) { const int y = __shfl_xor(c, o); c = y < c ? y : c; }
            cmin = c < cmin ? c : cmin;
        }
        if (cmin > 0 && cmin < M)
            for (int j0 = 0; j0 < J; j0 += WAVE) {
                const int j = j0 + lane;
                double r = (j < J && s_cnt[j] != M) ? s_jrow[j] : INFINITY;
                for (int o = 32; o > 0; o >>= 1) r = fmin(r, __shfl_xor(r, o));
                mn = fmin(mn, r);
            }
        for (int j = lane; j < J; j += WAVE) {
            const int cnt = s_cnt[j];
            unsigned char mk_;
            if (cmin == 0) mk_ = cnt >= 1;
            else if (cmin == M) mk_ = 1;
            else mk_ = !((cnt == M ? INFINITY : s_jrow[j]) == mn);
            P.obs.job_mask[(size_t)b * J + j] = mk_;
        }
        if (lane == 0) P.obs.candidate[(size_t)b * J + ja] = ja * M + (s_cnt[ja] < M ? s_cnt[ja] : M - 1);
    }
    // ---- write back the state that changed
    WSYNC();
    for (int v = lane; v < T; v += WAVE) {
        Link l; l.mach = (short)s_mach[v]; l.prev = (short)s_prev[v]; l.pos = (short)s_pos[v]; l.pad = 0;
        P.pl[bT + v].link = l;
    }
    if (lane == 0) { TaskSD x; x.st = st_k; x.dur = d; P.sd[bT + a] = x; P.pl[bT + a].pte = d * pk; }
    if (lane == 1) { JobR r; r.jmax = s_jmax[ja]; r.jrow = s_jrow[ja]; P.jr[(size_t)b * J + ja] = r; P.mj[(size_t)b * P.MJ + ja].cnt = (short)s_cnt[ja]; }
    if (lane == 2) { MJRec *r = &P.mj[(size_t)b * P.MJ + m]; r->head = (short)s_head[m]; r->tail = (short)s_tail[m]; r->len = (short)s_len[m]; }
    if (lane >= 8 && lane < 16) P.mfea[((size_t)b * M + m) * 8 + lane - 8] = s_mfr[lane - 8];
    if (lane < SCAL_N) P.scal[(size_t)b * SCAL_N + lane] = s_sc[lane];
#ifdef MTFJSP_STAMP
    STAMP(6);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(7);
    if (P.stamps && lane == 0) for (int i = 0; i < 8; i++) P.stamps[(size_t)b * 8 + i] = ph[i];
#endif
#undef DIVM
}
static size_t env_step_lds_bytes(int J, int M, int T, bool f32)
{
    const int Tp = (T + 7) & ~7;
    size_t off = (size_t)(4 * T + Tp + M * M + 3 * M + 2 * J + SCAL_N + 8) * sizeof(double);
    off = (off + 15) & ~(size_t)15;
    off += (size_t)M * 12 * (f32 ? 4 : 8);
    off = (off + 15) & ~(size_t)15;
    off += (size_t)(3 * T + J + 4 * M + 1 + 4) * sizeof(int);
    return off;
}

// =================================================================================================
// k_env_reg — the step kernel for instances with T <= 64 and M*M <= 64 (J6M6E2 and the like): NO LDS.
// lane = task: each lane keeps its task's (machine, route links, st, ft, dur, pt_est) in registers; lanes < M also keep
// a machine record, lanes < J a job record, lanes < M*M one transport-time entry.  The acting task/machine are
// wave-uniform, so every gather of the scheduling decision is a v_readlane with a scalar index (no memory hop), the
// left-shift gap test runs on all lanes at once (ballot) and routes are walked by scalar loops.  Same incremental
// output contract as k_env_step.  The whole step is ~2 global round trips + register/scalar work.
// (rl_i / rl_d / uni: mtfjsp_env_dev.h)

template <typename OBS>
__global__ __launch_bounds__(WAVE) void k_env_reg(EnvParams P)
{
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    const int J = P.J, M = P.M, T = P.T;
    const unsigned invM = P.inv_M;
#define DIVM(x) ((int)__umulhi((unsigned)(x), invM))
    const size_t bT = (size_t)b * T;
    const int v = lane;
    const bool isT = v < T;
    // ---- bulk state (independent of the action)
    int mach = -1, prev = -1, next = -1, pos = 0;
    double st = 0.0, ft = 0.0, dur = 0.0, pte = 0.0;
    if (isT) {
        const TaskSD x = P.sd[bT + v]; const TaskPL y = P.pl[bT + v];
        const Link l = y.link;
        mach = l.mach; prev = l.prev; pos = l.pos; next = l.pad;
        st = x.st; dur = x.dur; ft = x.st + x.dur; pte = y.pte;                  // ft == st + dur (env:356)
    }
    const double ttv = lane < M * M ? P.tt[(size_t)b * M * M + lane] : 0.0;
    int head_ = -1, tail_ = -1, len_ = 0;
    int cnt_ = 0; double jmax_ = -INFINITY, jrow_ = 0.0;
    if (lane < P.MJ) { const MJRec r = P.mj[(size_t)b * P.MJ + lane]; if (lane < M) { head_ = r.head; tail_ = r.tail; len_ = r.len; } if (lane < J) cnt_ = r.cnt; }
    if (lane < J) { const JobR r = P.jr[(size_t)b * J + lane]; jmax_ = r.jmax; jrow_ = r.jrow; }
    const double sc = lane < SCAL_N ? P.scal[(size_t)b * SCAL_N + lane] : 0.0;
    const int lastm = (int)rl_d(sc, S_LASTM);
    int a = uni(P.task_idx[b]), m = uni(P.mach_idx[b]);
    bool valid = a >= 0 && a < T && m >= 0 && m < M;
    if (!valid) { a = 0; m = 0; }
    const int ja = DIVM(a), op = a - ja * M;
    // ---- second hop (depends on the action)
    const double d = P.t[(bT + a) * M + m];
    const double pk = P.p[(bT + a) * M + m];
    const double md = lane < M ? P.cst[bT + ja * M + lane].x : 0.0;
    double mfr = lane < 8 ? P.mfea[((size_t)b * M + m) * 8 + lane] : 0.0;
    const int jv = DIVM(v), opv = v - jv * M;

    // =========================================================================================
    // A. scheduling (env:1476-1685)
    int status = 0, path = 0, Pk = -1, Nk = -1, ipos = 0;
    double st_k = 0.0;
    int mach_p = -1;
    if (valid) {
        if (rl_i(mach, a) >= 0) valid = false;                                  // env:1504
        else if (op != 0) { mach_p = rl_i(mach, a - 1); if (mach_p < 0) valid = false; }   // env:1520
    }
    const int len = rl_i(len_, m), head = rl_i(head_, m), tail = rl_i(tail_, m);
    const double ttmm = rl_d(ttv, m * M + m);
    if (valid) {
        if (d < 0.0) status |= MTFJSP_ST_INFEASIBLE;                            // pe:246-248
        const double arr_k = op == 0 ? 0.0 : rl_d(ft, a - 1) + rl_d(ttv, mach_p * M + m);   // dg:46-66
        bool do_append = false;
        if (len == 0) { path = MTFJSP_PATH_EMPTY; st_k = arr_k; ipos = 0; }                 // env:1684
        else if (!P.left_shift) do_append = true;                                               // env:1680
        else {
            const double lb_ft = arr_k + d;
            const int jh = DIVM(head);
            const double arr_f = (head == jh * M) ? 0.0 : rl_d(ft, head - 1) + rl_d(ttv, rl_i(mach, head - 1) * M + m);
            if (lb_ft <= arr_f) { path = MTFJSP_PATH_FRONT; st_k = arr_k; ipos = 0; Nk = head; }   // env:1548
            else if (len == 1) do_append = true;                                                 // env:1577
            else {
                // gap test of env:1587-1604 on every lane at once, then the first hit in route order by a scalar walk
                const int pi = prev >= 0 ? prev : 0, vi = v > 0 ? v - 1 : 0;
                const double ftP = __shfl(ft, pi), ftj = __shfl(ft, vi);
                const int mj = __shfl(mach, vi);
                const double ttj = __shfl(ttv, (mj >= 0 ? mj : 0) * M + m);
                const double jarr = (opv == 0) ? 0.0 : ftj + ttj;
                const double x = (DIVM(pi) == jv) ? ttmm : 0.0;
                const double nst = fmax(jarr, ftP + x);
                const bool ok = isT && mach == m && prev >= 0 && !(lb_ft > nst) && !((nst - ftP) < d);
                const unsigned long long okm = __ballot(ok);
                int cur = rl_i(next, head);
                while (cur >= 0) {
                    if ((okm >> cur) & 1ull) { Nk = cur; break; }
                    cur = rl_i(next, cur);
                }
                if (Nk >= 0) {
                    path = MTFJSP_PATH_BETWEEN;
                    ipos = rl_i(pos, Nk); Pk = rl_i(prev, Nk);
                    const double xx = (DIVM(Pk) == ja) ? ttmm : 0.0;
                    st_k = fmax(arr_k, rl_d(ft, Pk) + xx);                      // env:1619
                } else do_append = true;                                        // env:1676
            }
        }
        if (do_append) {                                                        // env:1689-1775
            path = MTFJSP_PATH_APPEND;
            const double xx = (DIVM(tail) == ja) ? ttmm : 0.0;
            st_k = fmax(arr_k, rl_d(ft, tail) + xx);
            ipos = len; Pk = tail;
        }
        status |= path;
    } else status |= MTFJSP_ST_INVALID;
    if (!valid) {                                                               // nothing changes; observations persist
        const bool all_done = rl_d(sc, S_NSCHED) == (double)T;
        if (lane < 6) P.obs.info[(size_t)b * 6 + lane] = (lane == 1 && all_done) ? 1.0 : 0.0;
        if (P.obs.raw && lane < 5) P.obs.raw[(size_t)b * 5 + lane] = 0.0;
        if (P.rec_r4 && lane < 4) P.rec_r4[(size_t)lane * P.B + b] = 0.f;
        if (P.rec_done && lane == 4) P.rec_done[b] = all_done ? 1.f : 0.f;
        if (lane == 5) P.obs.status[b] = status;
        return;
    }
    const double ft_k = st_k + d;
    // ---- apply: register updates on the owning lanes
    if (isT && mach == m && pos >= ipos) pos += 1;
    if (v == a) { mach = m; prev = Pk; next = Nk; pos = ipos; st = st_k; ft = ft_k; dur = d; pte = d * pk; }    // env:356,2175
    if (v == Nk) prev = a;
    if (v == Pk) next = a;
    if (lane == m) { if (ipos == 0) head_ = a; if (ipos == len) tail_ = a; len_ = len + 1; }
    if (lane == ja) cnt_ += 1;
    const int nsched = (int)rl_d(sc, S_NSCHED) + 1;

    // =========================================================================================
    // B. costs
    // estimated start/finish of the acting job's ops: the reference's left-to-right loop (env:1965-1995), run with
    // scalar indices; lane ja*M+c keeps its own (ste, fte)
    // Ops of a job are scheduled in order: ops < op are scheduled (their estimate IS their finish time, already folded into
    // the job's running row maximum jrow_), op is being scheduled now, ops > op are unscheduled -> only the tail is walked.
    double my_ste = 0.0, my_fte = 0.0, accp = ft_k;
    const double row_prev = rl_d(jrow_, ja);                                    // max real finish time of ops < op (0 if none)
    double jrow_new = op == 0 ? ft_k : fmax(row_prev, ft_k);                    // ppo:265-275 row maximum of real finish times
    double jmax_new = jrow_new;                                                 // estimated finish times of ops <= op are the real ones
    if (v == a) { my_ste = st_k; my_fte = ft_k; }
    if (ft_k == 0.0) {                                                          // env:1977: a zero finish time is treated as "not set"
        accp = (op ? rl_d(ft, a - 1) : 0.0) + rl_d(md, op);
        if (v == a) my_fte = accp;
        jmax_new = op == 0 ? accp : fmax(row_prev, accp);
    }
    for (int c = op + 1; c < M; c++) {
        const double fte_c = accp + rl_d(md, c);
        if (v == ja * M + c) { my_ste = accp; my_fte = fte_c; }
        accp = fte_c;
        jmax_new = fmax(jmax_new, fte_c);
    }
    if (lane == ja) { jmax_ = jmax_new; jrow_ = jrow_new; }
    double mk = rl_d(jmax_, 0);                                                 // env:894 np.amax
    for (int j = 1; j < J; j++) mk = fmax(mk, rl_d(jmax_, j));
    double e1;                                                                  // env:896 np.sum, numpy's pairwise order
    if (T < 8) { e1 = 0.0; for (int i = 0; i < T; i++) e1 += rl_d(pte, i); }
    else {
        // lanes 0..7 are numpy's 8 accumulators r[k] = a[k] + a[k+8] + ... (in that order), then its fixed tree
        // ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) as an xor butterfly (fp addition commutes), then the ragged tail in order
        const int nb = T - (T & 7);
        double r = pte;
        for (int i = 8; i < nb; i += 8) r += __shfl(pte, (lane & 7) + i);
        r += __shfl_xor(r, 1); r += __shfl_xor(r, 2); r += __shfl_xor(r, 4);
        e1 = rl_d(r, 0);
        for (int i = nb; i < T; i++) e1 += rl_d(pte, i);
    }
    e1 = 0.0 + e1;
    // idle time (dg:144-170): one term per scheduled task, summed strictly left to right in (machine, route position)
    // order.  Instead of chasing the route links (a dependent readlane per element) every scheduled lane computes its rank
    // in that order = (tasks on lower machines) + (its route position), a forward permute puts the terms into rank order
    // and a scalar-indexed loop adds lanes 0..nsched-1.
    const double ftPr = __shfl(ft, prev >= 0 ? prev : 0);
    const double term = prev < 0 ? st : st - ftPr;
    double idle = 0.0;
    {
        // lanes < M: tasks on machines below this one = exclusive prefix sum of the route lengths (M <= 8 here: three DPP
        // row shifts with zero fill)
        int incl = lane < M ? len_ : 0;
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, true);   // row_shr:1
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, true);   // row_shr:2
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, true);   // row_shr:4
        const int before = incl - (lane < M ? len_ : 0);
        const bool sch = isT && mach >= 0;
        const int below = __shfl(before, mach >= 0 ? mach : 0);                 // executed by ALL lanes: the source lanes must be active
        const int rank = sch ? below + pos : 63;                                // lane 63 is free whenever anything is unscheduled
        const double tv = sch ? term : 0.0;
        const int lo = __builtin_amdgcn_ds_permute(rank << 2, __double2loint(tv));
        const int hi = __builtin_amdgcn_ds_permute(rank << 2, __double2hiint(tv));
        const double sorted = __hiloint2double(hi, lo);
        int i = 0;
        for (; i + 3 < nsched; i += 4) {
            idle = idle + rl_d(sorted, i); idle = idle + rl_d(sorted, i + 1);
            idle = idle + rl_d(sorted, i + 2); idle = idle + rl_d(sorted, i + 3);
        }
        for (; i < nsched; i++) idle = idle + rl_d(sorted, i);
    }
    const double new_tr = (op == 0) ? 0.0 : rl_d(ttv, mach_p * M + m);          // env:872-876
    const double trans_this = rl_d(sc, S_TR_THIS) + new_tr;
    const double mk_prev = rl_d(sc, S_MK_PREV), e1_prev = rl_d(sc, S_E1_PREV), tr_prev = rl_d(sc, S_TR_PREV), id_prev = rl_d(sc, S_ID_PREV);
    const double r_t = 1.0 * mk_prev - mk;                                      // env:1066
    double r_pt = 1.0 * e1_prev - e1;
    r_pt = r_pt / (double)T;                                                    // env:1073-1076
    const double r_tt = 1.0 * tr_prev - trans_this;                             // env:1083
    const double r_idle = 1.0 * id_prev - idle;                                 // env:1088
    const double tot_n = P.w_mk * r_t + P.w_ec * (r_pt + 1 * r_idle) + P.w_tt * r_tt * 1;                  // env:1164
    const double tot = P.divisor == 1.0 ? tot_n : tot_n / P.divisor;            // x / 1.0 == x exactly: skip the f64 divide for the default scaling_divisor
    const bool done = nsched == T;                                              // env:797-800
    // reward scaling, one channel per lane (pt:54-124): lane c gathers its channel's state slots
    {
        const int c = lane & 3;
        const double sR0 = __shfl(sc, S_R + c), sM0 = __shfl(sc, S_MEAN + c), sS0 = __shfl(sc, S_S + c);
        const double n = rl_d(sc, S_N) + 1.0;
        if (lane < 4) {
            const double x = lane == 0 ? r_t : lane == 1 ? r_idle : lane == 2 ? r_pt : r_tt;
            const double R = P.gamma * sR0 + x;
            double mean, S = sS0, sd;
            if (n == 1.0) { mean = R; sd = fabs(R); }
            else { mean = sM0 + (R - sM0) / n; S = S + (R - sM0) * (R - mean); sd = sqrt(S / n); }
            const double scaled = x / (sd + 1e-8);
            double *s = P.scal + (size_t)b * SCAL_N;
            s[S_R + lane] = R; s[S_MEAN + lane] = mean; s[S_S + lane] = S; s[S_STD + lane] = sd;
            P.obs.info[(size_t)b * 6 + 2 + lane] = scaled;
            if (P.rec_r4) P.rec_r4[(size_t)lane * P.B + b] = (float)scaled;
        }
        if (lane == 4) {
            double *s = P.scal + (size_t)b * SCAL_N;
            s[S_N] = n; s[S_NSCHED] = (double)nsched;
            s[S_MK_PREV] = mk; s[S_E1_PREV] = e1; s[S_TR_PREV] = trans_this; s[S_ID_PREV] = idle;    // env:932-936
            s[S_TR_THIS] = done ? 0.0 : trans_this;                              // env:950-960
            P.obs.info[(size_t)b * 6 + 0] = tot;
            P.obs.info[(size_t)b * 6 + 1] = done ? 1.0 : 0.0;
            if (P.rec_done) P.rec_done[b] = done ? 1.f : 0.f;
        }
        if (P.obs.raw && lane >= 8 && lane < 13) {
            const int i = lane - 8;
            P.obs.raw[(size_t)b * 5 + i] = i == 0 ? tot : i == 1 ? r_t : i == 2 ? r_idle : i == 3 ? r_pt : r_tt;
        }
        if (lane == 5) P.obs.status[b] = status;
    }
    // machine features of the acting machine (env:2315-2340): lanes 0..4 own one column each
    {
        const double ft_tail = rl_d(ft, rl_i(tail_, m));
        if (lane == 0) mfr = ft_tail;
        else if (lane == 1) mfr += (pk * d) / (double)T;
        else if (lane == 2) mfr += new_tr;
        else if (lane == 3) mfr += idle - id_prev;
        else if (lane == 4) mfr += 1;
        if (lane < 8) {
            P.mfea[((size_t)b * M + m) * 8 + lane] = mfr;
            reinterpret_cast<OBS *>(P.obs.m_fea2)[((size_t)b * M + m) * 8 + lane] = (OBS)mfr;
        }
    }

    // =========================================================================================
    // C. the observation rows that changed
    const double w30 = rl_d(sc, S_W3), w31 = rl_d(sc, S_W3 + 1), w32 = rl_d(sc, S_W3 + 2);
    const bool merged_a = Pk >= 0 && op != 0 && Pk == a - 1;
    if (isT && jv == ja && opv >= op) {                                         // feature rows a .. end of job (env:2245-2277)
        const bool isa = v == a;
        OBS f[12];
        f[0] = (OBS)my_ste; f[1] = (OBS)my_fte; f[2] = (OBS)pte;
        f[3] = (OBS)(isa ? 1.0 : 0.0);
        f[4] = (OBS)(isa ? (1 + ((Pk >= 0 && !merged_a) ? 1 : 0)) : 1);
        f[5] = (OBS)(isa ? m + 1 : 0);
        f[6] = (OBS)(isa ? d : 0.0);
        f[7] = (OBS)(isa ? pk : 0.0);
        f[8] = (OBS)(ja + 1);
        f[9] = (OBS)w30; f[10] = (OBS)w31; f[11] = (OBS)w32;
        uint4 *dst = reinterpret_cast<uint4 *>(reinterpret_cast<OBS *>(P.obs.tasks_fea) + (bT + v) * 12);
        const uint4 *src = reinterpret_cast<const uint4 *>(f);
        for (int i = 0; i < (int)(12 * sizeof(OBS) / 16); i++) dst[i] = src[i];
    }
    {   // in-edge (ELL) rows of a, its job successor, its new route successor, and the node whose merged edge reverts:
        // lanes 0..3 take one row each and gather what they need with shuffles (every lane executes the shuffles: their
        // source lanes must be active), so the edge arithmetic runs once instead of four times with scalar reads
        const int merged_now = merged_a ? a : -1;
        const int vv0 = lane == 0 ? a : lane == 1 ? ((op + 1 < M) ? a + 1 : -1) : lane == 2 ? Nk : lane == 3 ? lastm : -1;
        const bool act = vv0 >= 0;
        const int vv = act ? vv0 : 0;
        const int mv = __shfl(mach, vv), pr = __shfl(prev, vv);
        const double st_v = __shfl(st, vv);
        const int jvv = DIVM(vv), opvv = vv - jvv * M;
        const int u = vv > 0 ? vv - 1 : 0;
        const int mu = __shfl(mach, u);
        const double dur_u = __shfl(dur, u), ft_u = __shfl(ft, u);
        const int pri = pr >= 0 ? pr : 0;
        const int mpr = __shfl(mach, pri);
        const double dur_p = __shfl(dur, pri), ft_p = __shfl(ft, pri);
        const double tt_uv = __shfl(ttv, (mu >= 0 ? mu : 0) * M + (mv >= 0 ? mv : 0));
        const double tt_pv = __shfl(ttv, (mpr >= 0 ? mpr : 0) * M + (mv >= 0 ? mv : 0));
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
        if (act) {
            reinterpret_cast<int2 *>(P.obs.ell_col)[bT + vv] = make_int2(c_job, c_mch);
            reinterpret_cast<float2 *>(P.obs.ell_val)[bT + vv] = make_float2(a_job, a_mch);
            if (lane == 2) reinterpret_cast<OBS *>(P.obs.tasks_fea)[(bT + vv) * 12 + 4] = (OBS)(1 + ((pr >= 0 && !merged) ? 1 : 0));
        }
        if (lane == 6) { P.scal[(size_t)b * SCAL_N + S_LASTM] = (double)merged_now; P.scal[(size_t)b * SCAL_N + S_TRLAST] = new_tr; }
    }

    // =========================================================================================
    // D. candidate + job mask (ppo:202-316)
    {
        int cmin = M;
        for (int j = 0; j < J; j++) { const int c = rl_i(cnt_, j); cmin = c < cmin ? c : cmin; }
        double mn = INFINITY;
        if (cmin > 0 && cmin < M)
            for (int j = 0; j < J; j++) { const double r = rl_i(cnt_, j) != M ? rl_d(jrow_, j) : INFINITY; mn = fmin(mn, r); }
        if (lane < J) {
            unsigned char mk_;
            if (cmin == 0) mk_ = cnt_ >= 1;
            else if (cmin == M) mk_ = 1;
            else mk_ = !((cnt_ == M ? INFINITY : jrow_) == mn);
            P.obs.job_mask[(size_t)b * J + lane] = mk_;
            if (lane == ja) {
                P.obs.candidate[(size_t)b * J + ja] = ja * M + (cnt_ < M ? cnt_ : M - 1);
                JobR r; r.jmax = jmax_; r.jrow = jrow_;
                P.jr[(size_t)b * J + ja] = r;
            }
        }
    }
    // ---- write back the state that changed
    if (isT) {
        TaskPL y; y.pte = pte; y.link.mach = (short)mach; y.link.prev = (short)prev; y.link.pos = (short)pos; y.link.pad = (short)next;
        P.pl[bT + v] = y;
        if (v == a) { TaskSD x; x.st = st; x.dur = dur; P.sd[bT + a] = x; }
    }
    if (lane == m || lane == ja) { MJRec r; r.head = (short)head_; r.tail = (short)tail_; r.len = (short)len_; r.cnt = (short)cnt_; P.mj[(size_t)b * P.MJ + lane] = r; }
#undef DIVM
}

#include "mtfjsp_env_grp.h"

// k_env_step_grp — k_env_step for GROUPS of instances per workgroup (one wave each, its own LDS region): the per-task part
// of the step per wave, then — after one barrier — the per-instance scalar part (energy / idle sums, rewards, RewardScaling,
// machine feature row, job mask) once for the whole group on wave 0 with lane = (instance, reward channel)
// (env_grp_tail, mtfjsp_env_grp.h).  Same operations in the same order: bit-identical to k_env_step.
#define ENV_LDS_GMAX 8                         // instances (= waves) per workgroup of k_env_step_grp at most
struct EnvStepLds {                            // layout of one instance's LDS region
    int T, Tp, M, J, nleaf;
    // f64 part, offsets in doubles: start | finish | processing energy per task, idle terms in rank order, column m of the transport
    // times, the acting job's min durations / estimated starts / finishes, per-job maxima, scalars, the acting machine's feature
    // row, leaf sums of the pairwise energy sum.  (Round 3: 18.0 KB per J20M20 instance instead of 26.7 — durations and the other
    // transport columns are read from memory by the few lanes that need one, route links are 16-bit — so that 8 instances fit a CU
    // and 2048 of them run in ONE round of workgroups: 2 x 29 us of dependent chain -> 1 x.)
    int d_ft, d_pte, d_term, d_ttc, d_mind, d_jste, d_jfte, d_jmax, d_jrow, d_sc, d_mfr, d_leaf, d_end;
    size_t o_stage, o_link, o_int, o_un, o_in, bytes;          // bytes from the region's start
    __host__ __device__ EnvStepLds(int J_, int M_, int T_, bool f32, int nleaf_) : T(T_), Tp((T_ + 7) & ~7), M(M_), J(J_), nleaf(nleaf_)
    {
        d_ft = T; d_pte = 2 * T; d_term = 3 * T; d_ttc = d_term + Tp; d_mind = d_ttc + M; d_jste = d_mind + M; d_jfte = d_jste + M;
        d_jmax = d_jfte + M; d_jrow = d_jmax + J; d_sc = d_jrow + J; d_mfr = d_sc + SCAL_N; d_leaf = d_mfr + 8; d_end = d_leaf + nleaf;
        size_t off = (size_t)d_end * sizeof(double);
        off = (off + 15) & ~(size_t)15;
        o_stage = off; off += (size_t)M * 12 * (f32 ? 4 : 8);
        off = (off + 15) & ~(size_t)15;
        o_link = off; off += (size_t)3 * T * sizeof(short);     // machine | route predecessor | rank per task
        off = (off + 15) & ~(size_t)15;
        o_int = off; off += (size_t)(J + 4 * M + 1 + 4) * sizeof(int);
        off = (off + 15) & ~(size_t)15;
        o_un = off; off += 16 * sizeof(double);
        o_in = off; off += 8 * sizeof(int);
        bytes = (off + 15) & ~(size_t)15;
    }
};
// numpy's pairwise recursion (n > 128: halves, the left one rounded down to a multiple of 8) as a table the step kernel walks:
// [2l], [2l+1] = offset, length of leaf l (in order); then nleaf - 1 merges (i, j): leaf-sum slot i += slot j, in post-order, so
// that the total ends in slot 0.  Depends on T only; built once per handle.
static int pw_table(int off, int n, int depth, std::vector<short> &leaves, std::vector<short> &merges)
{
    if (n <= 128 || depth == 0) { const int idx = (int)leaves.size() / 2; leaves.push_back((short)off); leaves.push_back((short)n); return idx; }
    int n2 = n / 2;
    n2 -= n2 % 8;
    const int l = pw_table(off, n2, depth - 1, leaves, merges), r = pw_table(off + n2, n - n2, depth - 1, leaves, merges);
    merges.push_back((short)l); merges.push_back((short)r);
    return l;
}
struct EnvGrpLdsAcc {
    static constexpr bool kBigT = true;
    const unsigned char *base; EnvStepLds L;
    __device__ __forceinline__ const double *d(int g, int off) const { return reinterpret_cast<const double *>(base + (size_t)g * L.bytes) + off; }
    __device__ __forceinline__ const double *sorted(int g) const { return d(g, L.d_term); }
    __device__ __forceinline__ const double *jmx(int g) const { return d(g, L.d_jmax); }
    __device__ __forceinline__ const double *jrw(int g) const { return d(g, L.d_jrow); }
    __device__ __forceinline__ const double *scl(int g) const { return d(g, L.d_sc); }
    __device__ __forceinline__ const double *mf(int g) const { return d(g, L.d_mfr); }
    __device__ __forceinline__ const int *cn(int g) const { return reinterpret_cast<const int *>(base + (size_t)g * L.bytes + L.o_int); }
    __device__ __forceinline__ const double *un(int g) const { return reinterpret_cast<const double *>(base + (size_t)g * L.bytes + L.o_un); }
    __device__ __forceinline__ const int *in(int g) const { return reinterpret_cast<const int *>(base + (size_t)g * L.bytes + L.o_in); }
};
template <typename OBS>
__device__ __forceinline__ void env_step_wave(const EnvParams &P, const int b, const int lane, unsigned char *smem, const EnvStepLds &LL)
{
    const int J = P.J, M = P.M, T = P.T;
    const unsigned invM = P.inv_M;
#define DIVM(x) ((int)__umulhi((unsigned)(x), invM))
    const int Tp = LL.Tp;
    double *s_st = reinterpret_cast<double *>(smem);
    double *s_ft = s_st + LL.d_ft;
    double *s_pte = s_st + LL.d_pte;
    double *s_term = s_st + LL.d_term;         // Tp
    double *s_ttc = s_st + LL.d_ttc;           // M: transport times INTO the acting machine (column m of tt)
    double *s_mind = s_st + LL.d_mind;         // M: min_dur of the acting job's ops
    double *s_jste = s_st + LL.d_jste;         // M: estimated start of the acting job's ops
    double *s_jfte = s_st + LL.d_jfte;         // M
    double *s_jmax = s_st + LL.d_jmax;         // J: max estimated finish per job
    double *s_jrow = s_st + LL.d_jrow;         // J: max real finish of scheduled ops per job
    double *s_sc = s_st + LL.d_sc;             // SCAL_N
    double *s_mfr = s_st + LL.d_mfr;           // 8: m_fea2 row of machine m
    double *s_leaf = s_st + LL.d_leaf;         // leaf sums of the pairwise energy sum (T > 128)
    OBS *s_stage = reinterpret_cast<OBS *>(smem + LL.o_stage);        // M rows x 12
    short *s_mach = reinterpret_cast<short *>(smem + LL.o_link);
    short *s_prev = s_mach + T;
    short *s_pos = s_prev + T;
    int *s_cnt = reinterpret_cast<int *>(smem + LL.o_int);            // J
    int *s_head = s_cnt + J;                   // M
    int *s_tail = s_head + M;
    int *s_len = s_tail + M;
    int *s_mstart = s_len + M;                 // M+1
    double *s_un = reinterpret_cast<double *>(smem + LL.o_un);
    int *s_in = reinterpret_cast<int *>(smem + LL.o_in);

    const size_t bT = (size_t)b * T;
#ifdef MTFJSP_STAMP
#define ESW_RT(i) do { if (P.stamps && lane == 0) { __builtin_amdgcn_sched_barrier(0); P.stamps[(size_t)b * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define ESW_RT(i) do { } while (0)
#endif
    ESW_RT(0);
    // ---- the action first; then the bulk state with every request of up to 8 task slots per lane in flight before the first
    // LDS write (a plain loop pays one memory round trip per 64 tasks); the action-dependent requests go out behind the bulk
    // requests and before any of them is waited for (loads return in order: the action's two words are there first)
    const int lastm = (int)P.scal[(size_t)b * SCAL_N + S_LASTM];
    int a = P.task_idx[b], m = P.mach_idx[b];
    const double *ttb = P.tt + (size_t)b * M * M;
    bool valid = false;
    int ja = 0, op = 0;
    double d = 0.0, pk = 0.0, x_mind = 0.0, x_ttc = 0.0, x_mfr = 0.0;
    for (int v0 = 0; v0 < T; v0 += 8 * WAVE) {
        TaskSD x_sd[8]; TaskPL x_pl[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int v = v0 + k * WAVE + lane;
            if (v < T) { x_sd[k] = P.sd[bT + v]; x_pl[k] = P.pl[bT + v]; }
        }
        if (v0 == 0) {
            valid = a >= 0 && a < T && m >= 0 && m < M;
            if (!valid) { a = 0; m = 0; }
            ja = DIVM(a); op = a - ja * M;
            d = P.t[(bT + a) * M + m];
            pk = P.p[(bT + a) * M + m];
            if (lane < M) { x_mind = P.cst[bT + ja * M + lane].x; x_ttc = P.ttT[((size_t)b * M + m) * M + lane]; }
            if (lane < 8) x_mfr = P.mfea[((size_t)b * M + m) * 8 + lane];
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int v = v0 + k * WAVE + lane;
            if (v < T) { s_st[v] = x_sd[k].st; s_ft[v] = x_sd[k].st + x_sd[k].dur; s_pte[v] = x_pl[k].pte; s_mach[v] = x_pl[k].link.mach; s_prev[v] = x_pl[k].link.prev; s_pos[v] = x_pl[k].link.pos; }   // ft == st + dur (env:356)
        }
    }
    for (int i = lane; i < Tp; i += WAVE) s_term[i] = 0.0;
    for (int i = lane; i < P.MJ; i += WAVE) { const MJRec r = P.mj[(size_t)b * P.MJ + i]; if (i < M) { s_head[i] = r.head; s_tail[i] = r.tail; s_len[i] = r.len; } if (i < J) s_cnt[i] = r.cnt; }
    for (int i = lane; i < J; i += WAVE) { const JobR r = P.jr[(size_t)b * J + i]; s_jmax[i] = r.jmax; s_jrow[i] = r.jrow; }
    if (lane < SCAL_N) s_sc[lane] = P.scal[(size_t)b * SCAL_N + lane];
    if (lane < M) { s_mind[lane] = x_mind; s_ttc[lane] = x_ttc; }
    for (int i = WAVE + lane; i < M; i += WAVE) { s_mind[i] = P.cst[bT + ja * M + i].x; s_ttc[i] = P.ttT[((size_t)b * M + m) * M + i]; }
    if (lane < 8) s_mfr[lane] = x_mfr;
    WSYNC();

    ESW_RT(1);
    // =========================================================================================
    // A. scheduling (env:1476-1685)
    int status = 0, path = 0;
    int Pk = -1, Nk = -1, ipos = 0;
    double st_k = 0.0, ft_k = 0.0;
    if (valid) {
        if (s_mach[a] >= 0) valid = false;                         // already scheduled (env:1504)
        else if (op != 0 && s_mach[a - 1] < 0) valid = false;      // job predecessor unscheduled (env:1520)
    }
    if (valid) {
        if (d < 0.0) status |= MTFJSP_ST_INFEASIBLE;               // pe:246-248
        const double ttmm = s_ttc[m];
        const double arr_k = op == 0 ? 0.0 : s_ft[a - 1] + s_ttc[s_mach[a - 1]];      // dg:46-66 over the single in-edge
        const int len = s_len[m], head = s_head[m], tail = s_tail[m];
        bool do_append = false;
        if (len == 0) { path = MTFJSP_PATH_EMPTY; st_k = arr_k; ipos = 0; }                      // env:1684
        else if (!P.left_shift) do_append = true;                                                     // env:1680
        else {
            const double lb_ft = arr_k + d;
            const int jh = DIVM(head);
            const double arr_f = (head == jh * M) ? 0.0 : s_ft[head - 1] + s_ttc[s_mach[head - 1]];
            if (lb_ft <= arr_f) { path = MTFJSP_PATH_FRONT; st_k = arr_k; ipos = 0; Nk = head; }   // env:1548
            else if (len == 1) do_append = true;                                                     // env:1577
            else {
                int key = 0x7fffffff;                               // gap search over all consecutive (P,N) at once (env:1587-1604)
                for (int v = lane; v < T; v += WAVE) {
                    if (s_mach[v] == m && s_prev[v] >= 0) {
                        const int Pp = s_prev[v];
                        const int jv = DIVM(v);
                        const double jarr = (v == jv * M) ? 0.0 : s_ft[v - 1] + s_ttc[s_mach[v - 1]];
                        const double x = (DIVM(Pp) == jv) ? ttmm : 0.0;
                        const double nst = fmax(jarr, s_ft[Pp] + x);
                        const bool ok = !(lb_ft > nst) && !((nst - s_ft[Pp]) < d);
                        if (ok) { const int kk = ((int)s_pos[v] << 16) | v; key = kk < key ? kk : key; }
                    }
                }
                key = rl_i(wave_min_lane63(key), 63);
                if (key != 0x7fffffff) {
                    path = MTFJSP_PATH_BETWEEN;
                    Nk = key & 0xffff; ipos = key >> 16; Pk = s_prev[Nk];
                    const double x = (DIVM(Pk) == ja) ? ttmm : 0.0;
                    st_k = fmax(arr_k, s_ft[Pk] + x);                          // env:1619
                } else do_append = true;                                       // env:1676
            }
        }
        if (do_append) {                                                       // env:1689-1775
            path = MTFJSP_PATH_APPEND;
            const double x = (DIVM(tail) == ja) ? ttmm : 0.0;
            st_k = fmax(arr_k, s_ft[tail] + x);
            ipos = len; Pk = tail;
        }
        ft_k = st_k + d;
        WSYNC();
        if (ipos < len)
            for (int v0 = 0; v0 < T; v0 += 8 * WAVE) {
                short mc[8], ps[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int v = v0 + k * WAVE + lane;
                    mc[k] = -1; ps[k] = -1;
                    if (v < T) { mc[k] = s_mach[v]; ps[k] = s_pos[v]; }
                }
#pragma unroll
                for (int k = 0; k < 8; k++)
                    if (mc[k] == m && ps[k] >= ipos) s_pos[v0 + k * WAVE + lane] = (short)(ps[k] + 1);
            }
        WSYNC();
        if (lane == 0) {
            s_mach[a] = (short)m; s_prev[a] = (short)Pk; s_pos[a] = (short)ipos;
            s_st[a] = st_k; s_ft[a] = ft_k; s_pte[a] = d * pk;                    // env:356,2175
            if (Nk >= 0) s_prev[Nk] = (short)a;
            if (ipos == 0) s_head[m] = a;
            if (ipos == len) s_tail[m] = a;
            s_len[m] = len + 1;
            s_cnt[ja] += 1;
            s_sc[S_NSCHED] += 1.0;
        }
        status |= path;
        WSYNC();
    } else status |= MTFJSP_ST_INVALID;
    if (!valid) {                                                               // nothing changes; the scalar part reports it
        if (lane == 0) { s_in[I_VALID] = 0; s_in[I_STATUS] = status; }
        return;
    }

    ESW_RT(2);
    // the <= 4 in-edge (ELL) rows that change (written in part C): a, its job successor, its new route successor, the node whose
    // merged edge reverts.  Durations and transport times of their predecessors are requested from memory HERE (the step's own
    // duration is not there yet; column m of the transport times is in LDS) and used after the cost part
    const int merged_now = (Pk >= 0 && op != 0 && Pk == a - 1) ? a : -1;
    int ev = -1;
    if (lane == 0) ev = a;
    else if (lane == 1) ev = (op + 1 < M) ? a + 1 : -1;
    else if (lane == 2) ev = Nk;
    else if (lane == 3) ev = lastm;
    int e_mv = -1, e_pr = -1, e_mu = -1, e_opv = 0, e_jv = 0;
    double e_nd = 1.0, e_dp = 0.0, e_tt1 = 0.0, e_tt2 = 0.0, e_gap1 = 0.0, e_gap2 = 0.0;
    if (ev >= 0) {
        auto durg = [&](int x) __attribute__((always_inline)) { return x == a ? d : P.sd[bT + x].dur; };
        auto ttg = [&](int r, int c) __attribute__((always_inline)) { return c == m ? s_ttc[r] : ttb[r * M + c]; };
        e_mv = s_mach[ev];
        e_jv = DIVM(ev); e_opv = ev - e_jv * M;
        e_pr = s_prev[ev];
        if (e_opv != 0) {
            e_mu = s_mach[ev - 1];
            if (e_mu >= 0) { e_nd = durg(ev - 1); if (e_mv >= 0) e_tt1 = ttg(e_mu, e_mv); e_gap1 = s_st[ev] - s_ft[ev - 1]; }
        }
        if (e_pr >= 0) { e_dp = durg(e_pr); if (DIVM(e_pr) == e_jv) e_tt2 = ttg(s_mach[e_pr], e_mv); e_gap2 = s_st[ev] - s_ft[e_pr]; }
    }
    // =========================================================================================
    // B. costs
    {   // machine route offsets: exclusive scan of the route lengths
        const int x = lane < M ? s_len[lane] : 0, incl = wave_scan_incl(x);
        if (lane < M) s_mstart[lane] = incl - x;
        if (lane == M - 1) s_mstart[M] = incl;
    }
    // estimated start/finish of the acting job's ops (env:1920-1999): lanes = ops of job ja.  The reference restarts a
    // left-to-right add sequence behind the last op with a non-zero real finish time; every lane's sequence is a prefix of one
    // chain: chain(c) = (op c-1 has a real finish ? that finish : chain(c-1)) + min_dur(c), chain(-1) = 0 — walked ONCE with
    // uniform lane reads instead of a dependent LDS loop per lane (4.0 -> 0.5 us of this kernel at M = 20)
    double my_fte = -INFINITY, my_rft = 0.0;
    {
        const int c = lane, v = ja * M + (lane < M ? lane : 0);
        const bool s = lane < M && s_mach[v] >= 0;
        const double ftv = s_ft[v], stv = s_st[v], mdv = lane < M ? s_mind[lane] : 0.0;
        const bool q = s && ftv != 0.0;
        const unsigned long long qm = __ballot(q);
        double acc = 0.0, fte_c = 0.0, base_c = 0.0;
        const int kfirst = __builtin_ctzll(~qm);                   // the ops before the first one without a real finish use their real times: the chain starts there
        for (int k = kfirst; k < M; k++) {
            const bool pq = k > 0 && ((qm >> (k - 1)) & 1ull);
            const double base = pq ? rl_d(ftv, k > 0 ? k - 1 : 0) : acc;
            acc = base + rl_d(mdv, k);
            if (lane == k) { fte_c = acc; base_c = base; }
        }
        if (lane < M) {
            const double ste = q ? stv : s ? stv : (c == 0 ? 0.0 : base_c);
            const double fte = q ? ftv : fte_c;
            s_jste[c] = ste; s_jfte[c] = fte;
            my_fte = fte;
            my_rft = s ? ftv : 0.0;
        }
    }
    {   // per-job maxima of the acting job (row maximum of ft_est for the makespan; of real ft for the job mask, ppo:265-275)
        const double fm = wave_max_lane63(my_fte), rm = wave_max_lane63(lane < M ? my_rft : -INFINITY);
        if (lane == 63) { s_jmax[ja] = fm; s_jrow[ja] = rm; }
    }
    WSYNC();
    ESW_RT(3);
    for (int v0 = 0; v0 < T; v0 += 8 * WAVE) {                                 // idle-time terms in (machine, position) order (dg:144-170); the reads of 8 slots per level together
        int mc[8], pr[8], ps[8]; double sv[8], fp[8]; int ms[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int v = v0 + k * WAVE + lane;
            mc[k] = -1; pr[k] = -1; ps[k] = 0; sv[k] = 0.0;
            if (v < T) { mc[k] = s_mach[v]; pr[k] = s_prev[v]; ps[k] = s_pos[v]; sv[k] = s_st[v]; }
        }
#pragma unroll
        for (int k = 0; k < 8; k++) { ms[k] = s_mstart[mc[k] >= 0 ? mc[k] : 0]; fp[k] = s_ft[pr[k] >= 0 ? pr[k] : 0]; }
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (mc[k] >= 0) s_term[ms[k] + ps[k]] = pr[k] < 0 ? sv[k] : sv[k] - fp[k];
    }
    {   // what the scalar part needs beyond the arrays: env:896 np.sum's pairwise part and the uniforms
        if (T <= 128) {                                                          // one leaf block: its 8 accumulators on 8 lanes; the ragged tail goes to the scalar part
            const int nb = T < 8 ? 0 : T - (T & 7);
            double r = 0.0;
            if (nb) {
                if (lane < 8) { r = s_pte[lane]; for (int i = 8 + lane; i < nb; i += 8) r += s_pte[i]; }
                r += __shfl_xor(r, 1); r += __shfl_xor(r, 2); r += __shfl_xor(r, 4);
            }
            if (lane == 0) s_un[U_R0] = r;
            if (lane < T - nb) s_un[U_TAIL + lane] = s_pte[nb + lane];
        } else {
            // numpy's recursion over the table of its leaves (pw_table): 8 lanes per leaf = its 8 accumulators, 8 leaves per
            // round; every lane's <= 16 reads go out together, the adds follow in numpy's order; ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7))
            // by three exchanges (the pairs commute), the leaf's ragged tail left to right on its first lane; then the merges
            const short *tab = P.pw_tab;
            const int nleaf = LL.nleaf;
            for (int l0 = 0; l0 < nleaf; l0 += 8) {
                const int lf = l0 + (lane >> 3), k = lane & 7;
                const bool on = lf < nleaf;
                const int off = on ? tab[2 * lf] : 0, n = on ? tab[2 * lf + 1] : 8;
                const int nb = n - (n & 7);
                const double *src = s_pte + off + k;
                double x[16];
#pragma unroll
                for (int i = 0; i < 16; i++) x[i] = 8 * i < nb ? src[8 * i] : 0.0;
                double r = x[0];
#pragma unroll
                for (int i = 1; i < 16; i++) if (8 * i < nb) r += x[i];
                r += __shfl_xor(r, 1); r += __shfl_xor(r, 2); r += __shfl_xor(r, 4);
                if (on && k == 0) {
                    for (int i = nb; i < n; i++) r += s_pte[off + i];
                    s_leaf[lf] = r;
                }
            }
            WSYNC();
            if (lane == 0) {
                const short *mg = tab + 2 * nleaf;
                for (int q = 0; q < nleaf - 1; q++) { const int i = mg[2 * q], j = mg[2 * q + 1]; s_leaf[i] = s_leaf[i] + s_leaf[j]; }
                s_un[U_R0] = s_leaf[0];
            }
        }
        if (lane == 0) {
            s_un[U_NEWTR] = (op == 0) ? 0.0 : s_ttc[s_mach[a - 1]];             // env:872-876
            s_un[U_D] = d; s_un[U_PK] = pk; s_un[U_FTTAIL] = s_ft[s_tail[m]];  // env:2315-2340
            s_in[I_VALID] = 1; s_in[I_STATUS] = status; s_in[I_NSCHED] = s_mstart[M]; s_in[I_M] = m; s_in[I_JA] = ja;
        }
    }

    ESW_RT(4);
    // =========================================================================================
    // C. the observation rows that changed
    {   // feature rows a .. end of job (env:2245-2277)
        const int nrow = M - op;
        if (lane < nrow) {
            const int c = op + lane, v = a + lane;
            const int pr = s_prev[a];
            const bool merged0 = pr >= 0 && op != 0 && pr == a - 1;
            OBS *f = s_stage + lane * 12;
            f[0] = (OBS)s_jste[c]; f[1] = (OBS)s_jfte[c]; f[2] = (OBS)s_pte[v];
            const bool isa = lane == 0;
            f[3] = (OBS)(isa ? 1.0 : 0.0);
            f[4] = (OBS)(isa ? (1 + ((pr >= 0 && !merged0) ? 1 : 0)) : 1);    // len(G.in_edges)
            f[5] = (OBS)(isa ? m + 1 : 0);
            f[6] = (OBS)(isa ? d : 0.0);
            f[7] = (OBS)(isa ? pk : 0.0);
            f[8] = (OBS)(ja + 1);
            f[9] = (OBS)s_sc[S_W3]; f[10] = (OBS)s_sc[S_W3 + 1]; f[11] = (OBS)s_sc[S_W3 + 2];
        }
        WSYNC();
        const int n16 = nrow * 12 * (int)sizeof(OBS) / 16;
        const uint4 *src = reinterpret_cast<const uint4 *>(s_stage);
        uint4 *dst = reinterpret_cast<uint4 *>(reinterpret_cast<OBS *>(P.obs.tasks_fea) + (bT + a) * 12);
        for (int i = lane; i < n16; i += WAVE) dst[i] = src[i];
    }
    if (ev >= 0) {   // the in-edge (ELL) rows from the values requested after the scheduling part
        const int v = ev, mv = e_mv, pr = e_pr, opv = e_opv;
        const bool s = mv >= 0;
        const bool merged = pr >= 0 && opv != 0 && pr == v - 1;
        int c_job = -1, c_mch = -1;
        float a_job = 0.f, a_mch = 0.f;
        if (opv != 0) {
            const int u = v - 1;
            double w_, nd;
            if (e_mu < 0) { w_ = 1.0; nd = 1.0; }
            else {
                nd = e_nd;
                if (merged && v == merged_now) w_ = nd + e_tt1 + e_gap1;                         // env:1607-1675,1703-1765
                else w_ = nd + (s ? e_tt1 : 0.0);                                                // env:1384-1422
            }
            long A = trunc_l(w_);
            if (A != 0) { A = trunc_l((double)A - nd) + 1; c_job = u; a_job = (float)A; }       // env:2019, 2060-2062
        }
        if (pr >= 0 && !merged) {
            const double w_ = e_dp + e_tt2 + e_gap2;
            long A = trunc_l(w_);
            if (A != 0) { A = trunc_l((double)A - e_dp) + 1; c_mch = pr; a_mch = (float)A; }
        }
        reinterpret_cast<int2 *>(P.obs.ell_col)[bT + v] = make_int2(c_job, c_mch);
        reinterpret_cast<float2 *>(P.obs.ell_val)[bT + v] = make_float2(a_job, a_mch);
        if (lane == 2) reinterpret_cast<OBS *>(P.obs.tasks_fea)[(bT + v) * 12 + 4] = (OBS)(1 + ((pr >= 0 && !merged) ? 1 : 0));
    }
    if (lane == 6) { P.scal[(size_t)b * SCAL_N + S_LASTM] = (double)merged_now; P.scal[(size_t)b * SCAL_N + S_TRLAST] = (op == 0) ? 0.0 : s_ttc[s_mach[a - 1]]; }

    ESW_RT(5);
    // ---- candidate of the acting job (ppo:202-316; the mask is the scalar part's)
    if (lane == 0) P.obs.candidate[(size_t)b * J + ja] = ja * M + (s_cnt[ja] < M ? s_cnt[ja] : M - 1);
    // ---- write back the state that changed: the links of machine m's route from the insertion point on (the acting task, its new
    // successor with a new predecessor, the ranks that moved up)
    for (int v0 = 0; v0 < T; v0 += 8 * WAVE) {
        short mc[8], pr[8], ps[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int v = v0 + k * WAVE + lane;
            mc[k] = -1; pr[k] = -1; ps[k] = -1;
            if (v < T) { mc[k] = s_mach[v]; pr[k] = s_prev[v]; ps[k] = s_pos[v]; }
        }
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (mc[k] == m && ps[k] >= ipos) {
                Link l; l.mach = mc[k]; l.prev = pr[k]; l.pos = ps[k]; l.pad = 0;
                P.pl[bT + v0 + k * WAVE + lane].link = l;
            }
    }
    if (lane == 0) { TaskSD x; x.st = st_k; x.dur = d; P.sd[bT + a] = x; P.pl[bT + a].pte = d * pk; }
    if (lane == 1) { JobR r; r.jmax = s_jmax[ja]; r.jrow = s_jrow[ja]; P.jr[(size_t)b * J + ja] = r; P.mj[(size_t)b * P.MJ + ja].cnt = (short)s_cnt[ja]; }
    if (lane == 2) { MJRec *r = &P.mj[(size_t)b * P.MJ + m]; r->head = (short)s_head[m]; r->tail = (short)s_tail[m]; r->len = (short)s_len[m]; }
    ESW_RT(6);
#undef ESW_RT
#undef DIVM
}
template <typename OBS>
__global__ __launch_bounds__(ENV_LDS_GMAX * WAVE) void k_env_step_grp(EnvParams P, int G)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const EnvStepLds LL(P.J, P.M, P.T, sizeof(OBS) == 4, P.pw_nleaf);
    const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b0 = blockIdx.x * G;
    const int lane = threadIdx.x & 63;
    if (b0 + grp < P.B) env_step_wave<OBS>(P, b0 + grp, lane, smem + (size_t)grp * LL.bytes, LL);
    __syncthreads();
    if (grp == 0) {
        const EnvGrpLdsAcc acc{smem, LL};
        env_grp_tail<OBS, false>(P, b0, lane, G, acc);
#ifdef MTFJSP_STAMP
        if (P.stamps && lane == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); P.stamps[(size_t)b0 * 8 + 7] = __builtin_amdgcn_s_memrealtime(); }
#endif
    } else if (grp == 1) {
        const EnvGrpLdsAcc acc{smem, LL};
        env_grp_mask(P, b0, lane, G, acc);
    }
}



// ---------------------------------------------------------------------------------------------
// instance preparation: min_dur/min_pt (env:1932-1950) and the means of pe:176-183. thread = (b,task)
__global__ void k_prepare(int B, int T, int M, const double *t, const double *p, double2 *cst, double *mean3)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * T) return;
    const double *tr = t + (size_t)i * M, *pr = p + (size_t)i * M;
    double md = INFINITY, mp = INFINITY;
    double bt[64], bpt[64], bp[64];
    int nt = 0, npt = 0, np_ = 0;
    for (int m = 0; m < M; m++) {
        const double tv = tr[m], pv = pr[m];
        const double q = tv * fabs(pv);
        const double dd = tv < 0 ? INFINITY : tv;
        const double qq = q < 0 ? INFINITY : q;
        md = dd < md ? dd : md;
        mp = qq < mp ? qq : mp;
        if (tv > 0) bt[nt++] = tv;
        if (q > 0) bpt[npt++] = q;
        if (pv > 0) bp[np_++] = pv;
    }
    cst[i] = make_double2(md, mp);
    mean3[(size_t)i * 3 + 0] = np_sum(bt, nt) / (double)nt;
    mean3[(size_t)i * 3 + 1] = np_sum(bpt, npt) / (double)npt;
    mean3[(size_t)i * 3 + 2] = np_sum(bp, np_) / (double)np_;
}

// ttT[b][m][x] = tt[b][x][m]: thread = (b, m, x)
__global__ void k_transpose_tt(int B, int M, const double *tt, double *ttT)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * M * M) return;
    const int b = i / (M * M), r = i - b * M * M, m = r / M, x = r - m * M;
    ttT[i] = tt[((size_t)b * M + x) * M + m];
}

// m_fea1 (pe:152-214): thread = (b, machine)
template <typename OBS>
__global__ void k_mfea1(int B, int T, int M, const double *t, const double *p, const double *tt, const double *mean3,
                        const int *shop, const TaskPL *pl, const int *task_idx, const uint8_t *mmask_in, OBS *out,
                        uint8_t *mmask_out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * M) return;
    const int b = i / M, m = i % M;
    int a = task_idx[b];
    if (a < 0 || a >= T) a = 0;
    const size_t row = ((size_t)b * T + a);
    const double tv = t[row * M + m], pv = p[row * M + m];
    const double ptv = tv * fabs(pv);
    const uint8_t mk = mmask_in ? mmask_in[i] : (uint8_t)!(tv >= 0);          // run:258-259 ~(t >= 0)
    double x = 0.0;
    if (a % M != 0) {
        int pm = pl[row - 1].link.mach;                                       // == tasks_fea[a-1][5] - 1 (pe:206)
        if (pm < 0) pm += M;                                                  // python negative index on an unscheduled predecessor
        x = tt[((size_t)b * M + pm) * M + m];
    }
    OBS *o = out + (size_t)i * 6;
    o[0] = (OBS)(tv > 0 ? tv : mean3[row * 3 + 0]);
    o[1] = (OBS)(ptv > 0 ? ptv : mean3[row * 3 + 1]);
    o[2] = (OBS)x;
    o[3] = (OBS)(1 - (int)mk);
    o[4] = (OBS)(pv > 0 ? pv : mean3[row * 3 + 2]);
    o[5] = (OBS)(shop[i] + 1);
    if (mmask_out) mmask_out[i] = mk;
}

// dense adj_wrk export (compat path, pe:136): thread = (b, dst task); out must be zero-filled first
__global__ void k_dense_adj(int B, int T, const int *ell_col, const float *ell_val, double *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * T) return;
    const int v = i % T;
    double *row = out + (size_t)i * T;
    row[v] = 1.0;
    for (int s = 0; s < 2; s++) {
        const int c = ell_col[(size_t)i * 2 + s];
        if (c >= 0) row[c] += (double)ell_val[(size_t)i * 2 + s];
    }
}

// env.valid_action_mask (env:2535-2575): thread = (b, task)
__global__ void k_valid_mask(int B, int T, int M, const TaskPL *pl, uint8_t *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * T) return;
    const int v = i % T;
    bool ok = pl[i].link.mach < 0 && (v % M == 0 || pl[i - 1].link.mach >= 0);
    out[i] = ok;
}

// uniform random valid action per instance: thread = instance
__global__ void k_random_actions(int B, int J, int M, int T, const double *t, const int *cand, const uint8_t *jmask,
                                 uint64_t seed, uint64_t counter, int *task_idx, int *mach_idx, int *job_idx)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    uint32_t c[4] = {(uint32_t)b, (uint32_t)counter, (uint32_t)(counter >> 32), 0x6d746a73u};
    philox4x32(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    int n = 0;
    for (int j = 0; j < J; j++) n += jmask[(size_t)b * J + j] ? 0 : 1;
    int a = 0, jj = 0;
    if (n > 0) {
        int pick = (int)(((uint64_t)c[0] * (uint64_t)n) >> 32);
        for (int j = 0; j < J; j++)
            if (!jmask[(size_t)b * J + j]) { if (pick == 0) { jj = j; break; } pick--; }
        a = cand[(size_t)b * J + jj];
    }
    const double *tr = t + ((size_t)b * T + a) * M;
    int nf = 0;
    for (int m = 0; m < M; m++) nf += tr[m] >= 0 ? 1 : 0;
    int mm = 0;
    if (nf > 0) {
        int pick = (int)(((uint64_t)c[1] * (uint64_t)nf) >> 32);
        for (int m = 0; m < M; m++)
            if (tr[m] >= 0) { if (pick == 0) { mm = m; break; } pick--; }
    }
    task_idx[b] = a; mach_idx[b] = mm;
    if (job_idx) job_idx[b] = jj;
}

// env.generate_random_weights("01") (env:1253-1259) on the device: three uniforms in [0,1) per instance, normalised by their sum
// (numpy's sum of three = left-to-right adds).  53-bit uniforms like python's random.random(): (a >> 5, b >> 6) -> (a*2^26 + b) / 2^53.
__global__ void k_draw_w3(int B, uint64_t seed, uint64_t episode, double *w3)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double w[3];
    draw_w3(b, seed, episode, w);
    w3[(size_t)b * 3 + 0] = w[0]; w3[(size_t)b * 3 + 1] = w[1]; w3[(size_t)b * 3 + 2] = w[2];
}

// GAE reverse scan (ppo:444-457 / 500-510): thread = instance, coalesced over b at every step.  The recurrence is serial in s but its
// inputs are not: the four loads of GAE_U steps are issued together and the chain then runs from registers (180 dependent
// memory round trips per thread before: 54.6 us per call at [180, 4096]; same operations in the same order, same bits).
#define GAE_U 12
__global__ void k_gae(int B, int S, const float *r, long r_ss, long r_sb, const float *v, long v_ss, long v_sb, const float *vn, long n_ss, long n_sb,
                      const float *done, float gamma, float lam, float *adv)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float g = 0.f;
    int s = S - 1;
    for (; s >= GAE_U - 1; s -= GAE_U) {
        float rr[GAE_U], vv[GAE_U], nn[GAE_U], dd[GAE_U];
#pragma unroll
        for (int u = 0; u < GAE_U; u++) {
            const long t = s - u;
            rr[u] = r[t * r_ss + b * r_sb]; nn[u] = vn[t * n_ss + b * n_sb]; vv[u] = v[t * v_ss + b * v_sb]; dd[u] = done[(size_t)t * B + b];
        }
#pragma unroll
        for (int u = 0; u < GAE_U; u++) {
            const float delta = rr[u] + gamma * nn[u] - vv[u];
            g = delta + gamma * lam * g * (1.0f - dd[u]);
            adv[(size_t)(s - u) * B + b] = g;
        }
    }
    for (; s >= 0; s--) {
        const float delta = r[s * r_ss + b * r_sb] + gamma * vn[s * n_ss + b * n_sb] - v[s * v_ss + b * v_sb];
        g = delta + gamma * lam * g * (1.0f - done[(size_t)s * B + b]);
        adv[(size_t)s * B + b] = g;
    }
}

// scaler init / per-episode reset (pe:70-85, pt:123)
__global__ void k_scaler(int B, int full, double *scal, const uint8_t *mask)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (mask && !mask[b]) return;
    double *s = scal + (size_t)b * SCAL_N;
    for (int i = 0; i < 4; i++) s[S_R + i] = 0.0;
    if (full) { for (int i = 0; i < 4; i++) { s[S_MEAN + i] = 0.0; s[S_S + i] = 0.0; s[S_STD + i] = 0.0; } s[S_N] = 0.0; }
}

// observation snapshot into a trajectory slot (device-resident replacement of ReplayBuffer.store_operation's
// per-step copies, replaybuffer.py:82-139): up to 8 (src, dst, bytes) segments in ONE launch; 16-byte lanes for the bulk,
// bytes for a ragged tail.  Pointers are 16-byte aligned when bytes % 16 == 0 slots are used (checked on the host).
struct SnapSeg { const unsigned char *src; unsigned char *dst, *dst2; unsigned long long bytes; };
struct SnapArgs { SnapSeg seg[8]; int n; const double *info; float *reward; int B; };
__global__ __launch_bounds__(256) void k_snapshot(SnapArgs A)
{
    const size_t gt = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (int i = 0; i < A.n; i++) {
        const SnapSeg g = A.seg[i];
        const bool al = (((size_t)g.src | (size_t)g.dst | (size_t)g.dst2) & 15) == 0;
        const size_t n16 = al ? g.bytes >> 4 : 0;
        const uint4 *s4 = reinterpret_cast<const uint4 *>(g.src);
        uint4 *d4 = reinterpret_cast<uint4 *>(g.dst), *e4 = reinterpret_cast<uint4 *>(g.dst2);
        if (g.dst2) {                                                       // one read, two destinations (s' of step k = s of step k+1)
            for (size_t k = gt; k < n16; k += stride) { const uint4 v = s4[k]; d4[k] = v; e4[k] = v; }
            for (size_t k = (n16 << 4) + gt; k < g.bytes; k += stride) { const unsigned char v = g.src[k]; g.dst[k] = v; g.dst2[k] = v; }
        } else {
            for (size_t k = gt; k < n16; k += stride) d4[k] = s4[k];
            for (size_t k = (n16 << 4) + gt; k < g.bytes; k += stride) g.dst[k] = g.src[k];
        }
    }
    if (A.reward) for (size_t b = gt; b < (size_t)A.B; b += stride) A.reward[b] = (float)A.info[b * 6];   // info[:,0] as f32 (pe:255-262)
}

// On-device instance generator (distribution of instance/generate_allsize_mofjsp_dataset.py:133-296, values of
// instance/config_ins.json): per task mean duration / power uniform, per (task, machine) a uniform weight, a uniform number
// k in [0,M) of machines made infeasible (a uniformly random k-subset: partial Fisher-Yates), t and p negative there;
// transport times uniform by shop distance, symmetric, zero diagonal.  Philox counters (seed; instance, task | pair), so a
// shard generates exactly its own instances; the legacy MT19937 stream of the host generator is not reproduced (parity is
// distributional, instances.py keeps the bit-exact host path).
struct GenScope { double t_low, t_high, p_low, p_high, w_low, w_high, in_low, in_high, out_high; };
__device__ __forceinline__ double u01(uint32_t a, uint32_t b) { return ((double)(((uint64_t)a << 21) ^ (uint64_t)(b >> 11)) + 0.5) * (1.0 / 9007199254740992.0); }
__global__ void k_generate(int B, int T, int M, int E, uint64_t seed, uint64_t first_instance, GenScope S, double *t, double *p, double *tt, int *shop)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B * T) {
        const int b = i / T, v = i % T;
        const uint64_t inst = first_instance + (uint64_t)b;
        uint32_t c[4] = {(uint32_t)inst, (uint32_t)(inst >> 32), (uint32_t)v, 0x67656e31u};
        philox4x32(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        const double avg_t = S.t_low + (S.t_high - S.t_low) * u01(c[0], c[1]);
        const double avg_p = S.p_low + (S.p_high - S.p_low) * u01(c[2], c[3]);
        uint64_t bad = 0;                                         // k-subset of infeasible machines (M <= 64)
        {
            uint32_t d[4] = {(uint32_t)inst, (uint32_t)(inst >> 32), (uint32_t)v, 0x67656e32u};
            philox4x32(d, (uint32_t)seed, (uint32_t)(seed >> 32));
            const int k = (int)(u01(d[0], d[1]) * M);             // [0, M)
            int perm[64];
            for (int m = 0; m < M; m++) perm[m] = m;
            uint32_t ctr = 0;
            for (int j = 0; j < k; j++) {
                uint32_t e[4] = {(uint32_t)inst, (uint32_t)v, ctr++, 0x67656e33u};
                philox4x32(e, (uint32_t)seed, (uint32_t)(seed >> 32));
                const int r = j + (int)(u01(e[0], e[1]) * (M - j));
                const int tmp = perm[j]; perm[j] = perm[r]; perm[r] = tmp;
                bad |= 1ull << perm[j];
            }
        }
        for (int m = 0; m < M; m++) {
            uint32_t e[4] = {(uint32_t)inst, (uint32_t)v, (uint32_t)m, 0x67656e34u};
            philox4x32(e, (uint32_t)seed, (uint32_t)(seed >> 32));
            double tv = avg_t * (S.w_low + (S.w_high - S.w_low) * u01(e[0], e[1]));
            double pv = avg_p * (S.w_low + (S.w_high - S.w_low) * u01(e[2], e[3]));
            if ((bad >> m) & 1ull) { tv = -tv; pv = -pv; }
            t[((size_t)b * T + v) * M + m] = tv;
            p[((size_t)b * T + v) * M + m] = pv;
        }
    }
    if (i < B * M * M) {
        const int b = i / (M * M), r = (i / M) % M, cc = i % M;
        const int per = M / E, sr = r / per < E ? r / per : E - 1, scc = cc / per < E ? cc / per : E - 1;
        double val = 0.0;
        if (r != cc) {
            const int lo = r < cc ? r : cc, hi = r < cc ? cc : r;   // one draw per unordered pair: symmetric
            const uint64_t inst = first_instance + (uint64_t)b;
            uint32_t e[4] = {(uint32_t)inst, (uint32_t)(inst >> 32), (uint32_t)(lo * M + hi), 0x67656e35u};
            philox4x32(e, (uint32_t)seed, (uint32_t)(seed >> 32));
            const int d = sr > scc ? sr - scc : scc - sr;
            const double u = u01(e[0], e[1]);
            val = d == 0 ? S.in_low + (S.in_high - S.in_low) * u : S.in_high * d + (S.out_high * d - S.in_high * d) * u;
        }
        tt[i] = val;
        if (r == 0) shop[b * M + cc] = scc;
    }
}

// =================================================================================================
// host side
struct mtfjsp_env {
    mtfjsp_config_t cfg;
    int T;
    hipStream_t stream = nullptr;
    std::string err;
    bool loaded = false, was_reset = false;
    // device memory
    double *t = nullptr, *p = nullptr, *tt = nullptr, *mean3 = nullptr;
    double2 *cst = nullptr;
    int *shop = nullptr;
    double *ttT = nullptr;             // tt transposed per instance (the step kernels read one row of it behind the action)
    TaskSD *sd = nullptr; TaskPL *pl = nullptr; JobR *jr = nullptr; MJRec *mj = nullptr;   // dynamic state records (mtfjsp_env_dev.h)
    int MJ = 0;
    double *mfea = nullptr, *scal = nullptr;
    size_t lds_max = 64 * 1024;        // hipDeviceAttributeMaxSharedMemoryPerBlock of the handle's device (160 KiB on gfx950)
    bool grp_lds_ok = true;            // the grouped LDS step kernel may be launched with lds_max bytes
    int *d_task = nullptr, *d_mach = nullptr;
    double *d_w3 = nullptr;
    short *pw_tab = nullptr;           // pw_table(T) on the device
    int pw_nleaf = 1;
    double *dense_scratch = nullptr;   // [B,T,T] f64, allocated on the first mtfjsp_export_dense_adj_host
    mtfjsp_obs_t obs{};
    bool obs_bound = false;
    std::vector<void *> owned;
    // timing
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    double *adv_partial = nullptr;     // mtfjsp_normalize_advantages: per-block partial sums [16][ADV_NB][2]
    size_t ev_used = 0;
};

static thread_local std::string g_create_err;

#define HIPCHK(h, call)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                       \
            return MTFJSP_ERR_HIP;                                                              \
        }                                                                                       \
    } while (0)

template <typename Tp>
static int dalloc(mtfjsp_env *h, Tp **ptr, size_t n)
{
    HIPCHK(h, hipMalloc((void **)ptr, n * sizeof(Tp)));
    h->owned.push_back(*ptr);
    return 0;
}

extern "C" const char *mtfjsp_last_error(mtfjsp_handle_t h) { return h ? h->err.c_str() : g_create_err.c_str(); }

extern "C" int mtfjsp_create(const mtfjsp_config_t *cfg, mtfjsp_handle_t *out)
{
    if (!cfg || !out) { g_create_err = "null argument"; return MTFJSP_ERR_ARG; }
    if (cfg->n_job < 1 || cfg->n_machine < 2 || cfg->n_machine > 64 || cfg->batch < 1 ||
        (long)cfg->n_job * cfg->n_machine > 32767 || (cfg->obs_dtype != MTFJSP_OBS_F64 && cfg->obs_dtype != MTFJSP_OBS_F32) ||
        cfg->scaling_divisor == 0.0) {
        g_create_err = "bad configuration (need n_job>=1, 2<=n_machine<=64, n_job*n_machine<=32767, batch>=1, divisor!=0)";
        return MTFJSP_ERR_ARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { g_create_err = "no HIP device available"; return MTFJSP_ERR_HIP; }
    if (cfg->device_id < 0 || cfg->device_id >= ndev) { g_create_err = "device_id out of range"; return MTFJSP_ERR_ARG; }
    if (hipSetDevice(cfg->device_id) != hipSuccess) { g_create_err = "hipSetDevice failed"; return MTFJSP_ERR_HIP; }
    mtfjsp_env *h = new mtfjsp_env();
    h->cfg = *cfg;
    h->T = cfg->n_job * cfg->n_machine;
    const size_t B = cfg->batch, T = h->T, M = cfg->n_machine;
    size_t lds = env_step_lds_bytes(cfg->n_job, cfg->n_machine, h->T, cfg->obs_dtype == MTFJSP_OBS_F32);
    if (env_reset_lds_bytes(h->T, false) > lds) lds = env_reset_lds_bytes(h->T, false);
    {   // LDS a workgroup may use on THIS device (gfx950: 160 KiB; asked, not assumed)
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, cfg->device_id) == hipSuccess && v > 0) h->lds_max = (size_t)v;
    }
    if (lds > h->lds_max) { g_create_err = "instance too large for one CU's LDS"; delete h; return MTFJSP_ERR_ARG; }
    int rc = 0;
    rc |= dalloc(h, &h->t, B * T * M); rc |= dalloc(h, &h->p, B * T * M); rc |= dalloc(h, &h->tt, B * M * M);
    rc |= dalloc(h, &h->mean3, B * T * 3); rc |= dalloc(h, &h->cst, B * T); rc |= dalloc(h, &h->shop, B * M);
    h->MJ = cfg->n_job > cfg->n_machine ? cfg->n_job : cfg->n_machine;
    rc |= dalloc(h, &h->ttT, B * M * M);
    rc |= dalloc(h, &h->sd, B * T); rc |= dalloc(h, &h->pl, B * T); rc |= dalloc(h, &h->jr, B * (size_t)cfg->n_job); rc |= dalloc(h, &h->mj, B * (size_t)h->MJ);
    rc |= dalloc(h, &h->mfea, B * M * 8); rc |= dalloc(h, &h->scal, B * SCAL_N);
    rc |= dalloc(h, &h->d_task, B); rc |= dalloc(h, &h->d_mach, B); rc |= dalloc(h, &h->d_w3, B * 3);
    if (rc) { g_create_err = h->err; mtfjsp_destroy(h); return MTFJSP_ERR_HIP; }
    if (hipMemset(h->scal, 0, B * SCAL_N * sizeof(double)) != hipSuccess) { g_create_err = "memset failed"; mtfjsp_destroy(h); return MTFJSP_ERR_HIP; }
    {   // the pairwise-sum table of T elements (walked by k_env_step_grp when T > 128)
        std::vector<short> leaves, merges;
        pw_table(0, (int)T, 6, leaves, merges);
        h->pw_nleaf = (int)leaves.size() / 2;
        leaves.insert(leaves.end(), merges.begin(), merges.end());
        if (dalloc(h, &h->pw_tab, leaves.size()) || hipMemcpy(h->pw_tab, leaves.data(), leaves.size() * sizeof(short), hipMemcpyHostToDevice) != hipSuccess) {
            g_create_err = "pairwise table upload failed"; mtfjsp_destroy(h); return MTFJSP_ERR_HIP;
        }
    }
    // opt in to large dynamic LDS (every kernel that is launched with a dynamic allocation sized from T)
    const void *dyn_kernels[] = {(const void *)k_env_step<double>, (const void *)k_env_step<float>,
                                 (const void *)k_env_reset<double>, (const void *)k_env_reset<float>};
    for (const void *k : dyn_kernels)
        if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            g_create_err = "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed";
            mtfjsp_destroy(h);
            return MTFJSP_ERR_HIP;
        }
    {   // the grouped LDS kernel takes up to a CU's worth of LDS (G instance regions)
        const void *grp_kernels[] = {(const void *)k_env_step_grp<double>, (const void *)k_env_step_grp<float>};
        for (const void *k : grp_kernels)                                  // a refusal only means "no grouped LDS kernel": G = 1 (k_env_step) serves
            if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_max) != hipSuccess) h->grp_lds_ok = false;
    }
    *out = h;
    return MTFJSP_OK;
}

extern "C" int mtfjsp_destroy(mtfjsp_handle_t h)
{
    if (!h) return MTFJSP_OK;
    (void)hipSetDevice(h->cfg.device_id);
    (void)hipDeviceSynchronize();
    for (void *p : h->owned) (void)hipFree(p);
    for (auto &e : h->ev_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    delete h;
    return MTFJSP_OK;
}

extern "C" int mtfjsp_set_stream(mtfjsp_handle_t h, void *s) { if (!h) return MTFJSP_ERR_ARG; h->stream = (hipStream_t)s; return MTFJSP_OK; }
extern "C" int mtfjsp_synchronize(mtfjsp_handle_t h) { if (!h) return MTFJSP_ERR_ARG; HIPCHK(h, hipStreamSynchronize(h->stream)); return MTFJSP_OK; }

extern "C" int mtfjsp_alloc_obs(mtfjsp_handle_t h, mtfjsp_obs_t *out)
{
    if (!h || !out) return MTFJSP_ERR_ARG;
    const size_t B = h->cfg.batch, T = h->T, M = h->cfg.n_machine, J = h->cfg.n_job;
    const size_t es = h->cfg.obs_dtype == MTFJSP_OBS_F32 ? 4 : 8;
    mtfjsp_obs_t o{};
    unsigned char *tf, *mf;
    if (dalloc(h, &tf, B * T * 12 * es) || dalloc(h, &mf, B * M * 8 * es) || dalloc(h, &o.ell_col, B * T * 2) ||
        dalloc(h, &o.ell_val, B * T * 2) || dalloc(h, &o.info, B * 6) || dalloc(h, &o.raw, B * 5) ||
        dalloc(h, &o.candidate, B * J) || dalloc(h, &o.job_mask, B * J) || dalloc(h, &o.status, B))
        return MTFJSP_ERR_HIP;
    o.tasks_fea = tf; o.m_fea2 = mf;
    *out = o;
    return mtfjsp_bind_obs(h, &o);
}

extern "C" int mtfjsp_bind_obs(mtfjsp_handle_t h, const mtfjsp_obs_t *o)
{
    if (!h || !o) return MTFJSP_ERR_ARG;
    if (!o->tasks_fea || !o->ell_col || !o->ell_val || !o->m_fea2 || !o->info || !o->candidate || !o->job_mask || !o->status) {
        h->err = "bind_obs: every field except raw must be non-NULL"; return MTFJSP_ERR_ARG;
    }
    h->obs = *o; h->obs_bound = true;
    return MTFJSP_OK;
}

// = the observation part of ReplayBuffer.store_operation (replaybuffer.py:82-139): copy the CURRENT bound observation
// (tasks_fea, ELL adjacency, m_fea2, candidate, job_mask; info/raw/status when dst has them) into caller buffers of the
// same layout, one launch on the handle's stream.  dst fields left NULL are skipped.  With dst2 the same launch writes a
// second copy (the reference's buffer keeps s' of step k and s of step k+1 separately: replaybuffer.py:97-118); with
// reward_out it also stores the scalar reward info[:,0] as f32 [B] (replaybuffer.py:106).
static int snapshot_impl(mtfjsp_env *h, const mtfjsp_obs_t *dst, const mtfjsp_obs_t *dst2, float *reward_out)
{
    if (!h->obs_bound) { h->err = "no observation buffers bound (mtfjsp_alloc_obs / mtfjsp_bind_obs)"; return MTFJSP_ERR_STATE; }
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    const size_t B = h->cfg.batch, T = h->T, M = h->cfg.n_machine, J = h->cfg.n_job;
    const size_t es = h->cfg.obs_dtype == MTFJSP_OBS_F32 ? 4 : 8;
    SnapArgs a{};
    size_t total = 0;
    auto add = [&](const void *src, void *d, void *d2, size_t bytes) {
        if (!d) { d = d2; d2 = nullptr; }
        if (!d || !src) return;
        a.seg[a.n].src = (const unsigned char *)src; a.seg[a.n].dst = (unsigned char *)d; a.seg[a.n].dst2 = (unsigned char *)d2; a.seg[a.n].bytes = bytes; a.n++;
        total += bytes;
    };
#define SNAP2(f) (dst2 ? (void *)dst2->f : nullptr)
    add(h->obs.tasks_fea, dst->tasks_fea, SNAP2(tasks_fea), B * T * 12 * es);
    add(h->obs.ell_col, dst->ell_col, SNAP2(ell_col), B * T * 2 * 4);
    add(h->obs.ell_val, dst->ell_val, SNAP2(ell_val), B * T * 2 * 4);
    add(h->obs.m_fea2, dst->m_fea2, SNAP2(m_fea2), B * M * 8 * es);
    add(h->obs.candidate, dst->candidate, SNAP2(candidate), B * J * 4);
    add(h->obs.job_mask, dst->job_mask, SNAP2(job_mask), B * J);
    add(h->obs.info, dst->info, SNAP2(info), B * 6 * 8);
    add(h->obs.raw, dst->raw, SNAP2(raw), B * 5 * 8);
#undef SNAP2
    if (reward_out) {
        if (!h->obs.info) { h->err = "mtfjsp_snapshot_obs2: reward_out needs the bound observation to hold the step information (obs.info)"; return MTFJSP_ERR_STATE; }
        a.info = h->obs.info; a.reward = reward_out; a.B = (int)B;
    }
    if (a.n == 0 && !reward_out) return MTFJSP_OK;
    size_t blocks = (total / 16 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_snapshot, dim3((unsigned)blocks), dim3(256), 0, h->stream, a);
    HIPCHK(h, hipGetLastError());
    return MTFJSP_OK;
}
extern "C" int mtfjsp_snapshot_obs(mtfjsp_handle_t h, const mtfjsp_obs_t *dst)
{
    if (!h || !dst) return MTFJSP_ERR_ARG;
    return snapshot_impl(h, dst, nullptr, nullptr);
}
extern "C" int mtfjsp_snapshot_obs2(mtfjsp_handle_t h, const mtfjsp_obs_t *dst, const mtfjsp_obs_t *dst2, float *reward_out)
{
    if (!h || !dst) return MTFJSP_ERR_ARG;
    return snapshot_impl(h, dst, dst2, reward_out);
}

static int load_common(mtfjsp_env *h, const double *t, const double *p, const double *tt, const int32_t *shop, hipMemcpyKind kind)
{
    const size_t B = h->cfg.batch, T = h->T, M = h->cfg.n_machine;
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    HIPCHK(h, hipMemcpyAsync(h->t, t, B * T * M * 8, kind, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->p, p, B * T * M * 8, kind, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->tt, tt, B * M * M * 8, kind, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->shop, shop, B * M * 4, kind, h->stream));
    const int n = (int)(B * T);
    hipLaunchKernelGGL(k_prepare, dim3((n + 127) / 128), dim3(128), 0, h->stream, (int)B, (int)T, (int)M, h->t, h->p, h->cst, h->mean3);
    hipLaunchKernelGGL(k_transpose_tt, dim3((unsigned)((B * M * M + 127) / 128)), dim3(128), 0, h->stream, (int)B, (int)M, h->tt, h->ttT);
    HIPCHK(h, hipGetLastError());
    if (kind == hipMemcpyHostToDevice) HIPCHK(h, hipStreamSynchronize(h->stream));
    h->loaded = true; h->was_reset = false;
    return MTFJSP_OK;
}
extern "C" int mtfjsp_load_instances(mtfjsp_handle_t h, const double *t, const double *p, const double *tt, const int32_t *shop)
{
    if (!h || !t || !p || !tt || !shop) return MTFJSP_ERR_ARG;
    return load_common(h, t, p, tt, shop, hipMemcpyDeviceToDevice);
}
extern "C" int mtfjsp_load_instances_host(mtfjsp_handle_t h, const double *t, const double *p, const double *tt, const int32_t *shop)
{
    if (!h || !t || !p || !tt || !shop) return MTFJSP_ERR_ARG;
    return load_common(h, t, p, tt, shop, hipMemcpyHostToDevice);
}

// = Instance_Dataset generation (generate…py:133-296) directly into the handle's instance arrays: no host->device upload
// (SURVEY §8f N4).  scope9 = {t_low, t_high, p_low, p_high, weight_low, weight_high, transT_in_low, transT_in_high,
// transT_out_high}; first_instance offsets the Philox counter so that shards / successive batches draw distinct instances.
extern "C" int mtfjsp_generate_instances(mtfjsp_handle_t h, uint64_t seed, uint64_t first_instance, const double *scope9)
{
    if (!h || !scope9) return MTFJSP_ERR_ARG;
    const int B = h->cfg.batch, T = h->T, M = h->cfg.n_machine, E = h->cfg.n_edge;
    if (E < 1 || M % E != 0) { h->err = "n_machine must be divisible by n_edge"; return MTFJSP_ERR_ARG; }
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    GenScope S{scope9[0], scope9[1], scope9[2], scope9[3], scope9[4], scope9[5], scope9[6], scope9[7], scope9[8]};
    const int n = B * (T > M * M ? T : M * M);
    hipLaunchKernelGGL(k_generate, dim3((n + 127) / 128), dim3(128), 0, h->stream, B, T, M, E, seed, first_instance, S, h->t, h->p, h->tt, h->shop);
    hipLaunchKernelGGL(k_prepare, dim3((B * T + 127) / 128), dim3(128), 0, h->stream, B, T, M, h->t, h->p, h->cst, h->mean3);
    hipLaunchKernelGGL(k_transpose_tt, dim3((B * M * M + 127) / 128), dim3(128), 0, h->stream, B, M, h->tt, h->ttT);
    HIPCHK(h, hipGetLastError());
    h->loaded = true; h->was_reset = false;
    return MTFJSP_OK;
}
// instance arrays back to the host (t, p [B,T,M] f64; tt [B,M,M] f64; shop [B,M] i32) — e.g. to export a generated set in
// the reference's pickle layout
extern "C" int mtfjsp_read_instances_host(mtfjsp_handle_t h, double *t, double *p, double *tt, int32_t *shop)
{
    if (!h || !t || !p || !tt || !shop) return MTFJSP_ERR_ARG;
    if (!h->loaded) { h->err = "no instances loaded"; return MTFJSP_ERR_STATE; }
    const size_t B = h->cfg.batch, T = h->T, M = h->cfg.n_machine;
    int rc = mtfjsp_copy_to_host(h, t, h->t, B * T * M * 8);
    if (!rc) rc = mtfjsp_copy_to_host(h, p, h->p, B * T * M * 8);
    if (!rc) rc = mtfjsp_copy_to_host(h, tt, h->tt, B * M * M * 8);
    if (!rc) rc = mtfjsp_copy_to_host(h, shop, h->shop, B * M * 4);
    return rc;
}

static int scaler_launch(mtfjsp_env *h, int full, const uint8_t *mask = nullptr)
{
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    hipLaunchKernelGGL(k_scaler, dim3((h->cfg.batch + 127) / 128), dim3(128), 0, h->stream, h->cfg.batch, full, h->scal, mask);
    HIPCHK(h, hipGetLastError());
    return MTFJSP_OK;
}
extern "C" int mtfjsp_scaler_init(mtfjsp_handle_t h) { return h ? scaler_launch(h, 1) : MTFJSP_ERR_ARG; }
extern "C" int mtfjsp_scaler_reset_returns(mtfjsp_handle_t h) { return h ? scaler_launch(h, 0) : MTFJSP_ERR_ARG; }
extern "C" int mtfjsp_scaler_reset_returns_masked_host(mtfjsp_handle_t h, const uint8_t *mask_host)
{
    if (!h || !mask_host) return MTFJSP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    uint8_t *dm = reinterpret_cast<uint8_t *>(h->d_task);            // scratch: B int32 >= B bytes
    HIPCHK(h, hipMemcpyAsync(dm, mask_host, (size_t)h->cfg.batch, hipMemcpyHostToDevice, h->stream));
    int rc = scaler_launch(h, 0, dm);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MTFJSP_OK;
}

static EnvParams make_params(mtfjsp_env *h)
{
    EnvParams P{};
    P.B = h->cfg.batch; P.J = h->cfg.n_job; P.M = h->cfg.n_machine; P.T = h->T;
    P.left_shift = h->cfg.left_shift; P.obs_f32 = h->cfg.obs_dtype == MTFJSP_OBS_F32;
    P.w_mk = h->cfg.w_mk; P.w_ec = h->cfg.w_ec; P.w_tt = h->cfg.w_tt; P.divisor = h->cfg.scaling_divisor; P.gamma = h->cfg.gamma;
    P.t = h->t; P.p = h->p; P.tt = h->tt; P.cst = h->cst;
    P.ttT = h->ttT; P.sd = h->sd; P.pl = h->pl; P.jr = h->jr; P.mj = h->mj; P.MJ = h->MJ; P.mfea = h->mfea; P.scal = h->scal;
    P.inv_M = (unsigned)((0x100000000ull + (unsigned long long)P.M - 1) / (unsigned long long)P.M);
    P.obs = h->obs;
    P.pw_tab = h->pw_tab; P.pw_nleaf = h->pw_nleaf;
    return P;
}

static int check_ready(mtfjsp_env *h, bool need_reset)
{
    if (!h->loaded) { h->err = "load_instances has not been called"; return MTFJSP_ERR_STATE; }
    if (!h->obs_bound) { h->err = "no observation buffers bound (mtfjsp_alloc_obs / mtfjsp_bind_obs)"; return MTFJSP_ERR_STATE; }
    if (need_reset && !h->was_reset) { h->err = "reset has not been called"; return MTFJSP_ERR_STATE; }
    return MTFJSP_OK;
}

extern "C" int mtfjsp_reset(mtfjsp_handle_t h, const double *w3)
{
    if (!h || !w3) return MTFJSP_ERR_ARG;
    int rc = check_ready(h, false);
    if (rc) return rc;
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    EnvParams P = make_params(h);
    P.w3 = w3;
    const size_t lds = env_reset_lds_bytes(P.T, P.obs_f32);
    if (P.obs_f32) hipLaunchKernelGGL((k_env_reset<float>), dim3(P.B), dim3(WAVE), lds, h->stream, P);
    else hipLaunchKernelGGL((k_env_reset<double>), dim3(P.B), dim3(WAVE), lds, h->stream, P);
    HIPCHK(h, hipGetLastError());
    h->was_reset = true;
    return MTFJSP_OK;
}
// reset of an EPISODE in one launch: scaler_reset_returns + draw_reward_weights + reset (run:283-284, env:1253-1259, pe:87) — what the
// accelerated rollout issued as three launches per episode
extern "C" int mtfjsp_reset_episode(mtfjsp_handle_t h, uint64_t seed, uint64_t episode, double *w3_out, int32_t reset_returns)
{
    if (!h || !w3_out) return MTFJSP_ERR_ARG;
    int rc = check_ready(h, false);
    if (rc) return rc;
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    EnvParams P = make_params(h);
    P.w3 = nullptr; P.draw = 1; P.draw_seed = seed; P.draw_episode = episode; P.w3_out = w3_out; P.reset_returns = reset_returns ? 1 : 0;
    const size_t lds = env_reset_lds_bytes(P.T, P.obs_f32);
    if (P.obs_f32) hipLaunchKernelGGL((k_env_reset<float>), dim3(P.B), dim3(WAVE), lds, h->stream, P);
    else hipLaunchKernelGGL((k_env_reset<double>), dim3(P.B), dim3(WAVE), lds, h->stream, P);
    HIPCHK(h, hipGetLastError());
    h->was_reset = true;
    return MTFJSP_OK;
}
extern "C" int mtfjsp_reset_host(mtfjsp_handle_t h, const double *w3_host)
{
    if (!h || !w3_host) return MTFJSP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    HIPCHK(h, hipMemcpyAsync(h->d_w3, w3_host, (size_t)h->cfg.batch * 3 * 8, hipMemcpyHostToDevice, h->stream));
    int rc = mtfjsp_reset(h, h->d_w3);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MTFJSP_OK;
}

static int step_impl(mtfjsp_handle_t h, const int32_t *task_idx, const int32_t *mach_idx, float *r4, float *dn);
extern "C" int mtfjsp_step(mtfjsp_handle_t h, const int32_t *task_idx, const int32_t *mach_idx) { return step_impl(h, task_idx, mach_idx, nullptr, nullptr); }
extern "C" int mtfjsp_step_record(mtfjsp_handle_t h, const int32_t *task_idx, const int32_t *mach_idx, float *r4_out, float *done_out)
{
    if (!r4_out || !done_out) return MTFJSP_ERR_ARG;
    return step_impl(h, task_idx, mach_idx, r4_out, done_out);
}
// The parameter block of the step kernel for a launch that runs the step as its own tail (mtfjsp_encoder_arm_env_step: the machine
// actor's heads kernel).  Returns 1 and fills `out` when this handle's step is the 16-instance register kernel (k_env_grp16) and
// nothing asks for the stand-alone launch (kernel-time recording, MTFJSP_ENV_KERNEL); 0 when the caller has to call mtfjsp_step /
// mtfjsp_step_record itself; < 0 on errors.  The step has no host-side state: a parameter block that is never used costs nothing.
extern "C" int32_t mtfjsp_step_params_bytes(void) { return (int32_t)sizeof(EnvParams); }
extern "C" int mtfjsp_step_params(mtfjsp_handle_t h, const int32_t *task_idx, const int32_t *mach_idx, float *r4_out, float *done_out, void *out, int32_t out_bytes)
{
    if (!h || !task_idx || !mach_idx || !out || out_bytes != (int32_t)sizeof(EnvParams)) return MTFJSP_ERR_ARG;
    if ((r4_out == nullptr) != (done_out == nullptr)) return MTFJSP_ERR_ARG;
    int rc = check_ready(h, true);
    if (rc) return rc;
    EnvParams P = make_params(h);
    P.task_idx = task_idx; P.mach_idx = mach_idx; P.rec_r4 = r4_out; P.rec_done = done_out;
    const bool eligible = !h->timing && !getenv("MTFJSP_ENV_KERNEL") && !getenv("MTFJSP_ENV_LDS") && P.T <= 64 && P.M * P.M <= 64 && P.J <= 64 && P.B <= EG_SMALL_MAX_B;
    if (!eligible) return 0;
    memcpy(out, &P, sizeof(EnvParams));
    return 1;
}
static int step_impl(mtfjsp_handle_t h, const int32_t *task_idx, const int32_t *mach_idx, float *r4, float *dn)
{
    if (!h || !task_idx || !mach_idx) return MTFJSP_ERR_ARG;
    int rc = check_ready(h, true);
    if (rc) return rc;
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    EnvParams P = make_params(h);
    P.task_idx = task_idx; P.mach_idx = mach_idx; P.rec_r4 = r4; P.rec_done = dn;
    std::pair<hipEvent_t, hipEvent_t> *ev = nullptr;
    if (h->timing) {
        if (h->ev_used == h->ev_pool.size()) {
            hipEvent_t a, b;
            HIPCHK(h, hipEventCreate(&a)); HIPCHK(h, hipEventCreate(&b));
            h->ev_pool.push_back({a, b});
        }
        ev = &h->ev_pool[h->ev_used++];
        HIPCHK(h, hipEventRecord(ev->first, h->stream));
    }
#ifdef MTFJSP_STAMP
    static unsigned long long *d_st = nullptr;
    if (!d_st) (void)hipMalloc((void **)&d_st, (size_t)P.B * 64);
    P.stamps = d_st;
#endif
    // diagnostic / test overrides of the kernel selection (read per call): MTFJSP_ENV_KERNEL = lds | reg1 | grp16 | grp4
    const char *force = getenv("MTFJSP_ENV_KERNEL");
    const bool force_lds = getenv("MTFJSP_ENV_LDS") || (force && (!strcmp(force, "lds") || !strcmp(force, "lds1")));
    const bool force_reg1 = force && !strcmp(force, "reg1");
    const bool reg_ok = P.T <= 64 && P.M * P.M <= 64 && P.J <= 64 && !force_lds;
    const bool reg2_ok = !reg_ok && P.T <= 128 && P.M * P.M <= 128 && P.M <= 16 && P.J <= 64 && !force_lds && !force_reg1;
    if (reg2_ok) {                                                        // register kernel with two task slots per lane
        const bool small = force && !strcmp(force, "grp16") ? true : force && !strcmp(force, "grp4") ? false : P.B <= EG_SMALL_MAX_B / 2;
        if (small) {
            const int grid = (P.B + EG_SMALL - 1) / EG_SMALL;
            if (P.obs_f32) hipLaunchKernelGGL((k_env_grp16x2<float>), dim3(grid), dim3(EG_SMALL * WAVE), 0, h->stream, P);
            else hipLaunchKernelGGL((k_env_grp16x2<double>), dim3(grid), dim3(EG_SMALL * WAVE), 0, h->stream, P);
        } else {
            const int grid = (P.B + EG_LARGE - 1) / EG_LARGE;
            if (P.obs_f32) hipLaunchKernelGGL((k_env_grp4x2<float>), dim3(grid), dim3(EG_LARGE * WAVE), 0, h->stream, P);
            else hipLaunchKernelGGL((k_env_grp4x2<double>), dim3(grid), dim3(EG_LARGE * WAVE), 0, h->stream, P);
        }
    } else if (reg_ok && !force_reg1) {                                   // register kernel, groups of instances per workgroup
        const bool small = force && !strcmp(force, "grp16") ? true : force && !strcmp(force, "grp4") ? false : P.B <= EG_SMALL_MAX_B;
        if (small) {
            const int grid = (P.B + EG_SMALL - 1) / EG_SMALL;
            if (P.obs_f32) hipLaunchKernelGGL((k_env_grp16<float>), dim3(grid), dim3(EG_SMALL * WAVE), 0, h->stream, P);
            else hipLaunchKernelGGL((k_env_grp16<double>), dim3(grid), dim3(EG_SMALL * WAVE), 0, h->stream, P);
        } else {
            const int grid = (P.B + EG_LARGE - 1) / EG_LARGE;
            if (P.obs_f32) hipLaunchKernelGGL((k_env_grp4<float>), dim3(grid), dim3(EG_LARGE * WAVE), 0, h->stream, P);
            else hipLaunchKernelGGL((k_env_grp4<double>), dim3(grid), dim3(EG_LARGE * WAVE), 0, h->stream, P);
        }
    } else if (reg_ok) {                                                  // one instance per workgroup (the A/B reference of the grouped form)
        if (P.obs_f32) hipLaunchKernelGGL((k_env_reg<float>), dim3(P.B), dim3(WAVE), 0, h->stream, P);
        else hipLaunchKernelGGL((k_env_reg<double>), dim3(P.B), dim3(WAVE), 0, h->stream, P);
    } else {
        // LDS kernel: groups of G instances per workgroup where at least two instances' regions fit (MTFJSP_ENV_STEP_G overrides;
        // 1 = the one-instance kernel k_env_step)
        const EnvStepLds LL(P.J, P.M, P.T, P.obs_f32 != 0, P.pw_nleaf);
        const int gmax = h->grp_lds_ok ? (int)((h->lds_max - 512) / LL.bytes) : 1;
        int G = gmax >= 8 ? 8 : gmax >= 4 ? 4 : gmax >= 2 ? 2 : 1;
        if (const char *gs = getenv("MTFJSP_ENV_STEP_G")) { G = atoi(gs); G = G < 1 ? 1 : G > ENV_LDS_GMAX ? ENV_LDS_GMAX : G; G = G > gmax ? (gmax < 1 ? 1 : gmax) : G; }
        if (force && !strcmp(force, "lds1")) G = 1;
        if (G > 1) {
            const size_t lds_g = (size_t)G * LL.bytes;
            const int grid = (P.B + G - 1) / G;
            if (P.obs_f32) hipLaunchKernelGGL((k_env_step_grp<float>), dim3(grid), dim3(G * WAVE), lds_g, h->stream, P, G);
            else hipLaunchKernelGGL((k_env_step_grp<double>), dim3(grid), dim3(G * WAVE), lds_g, h->stream, P, G);
        } else {
            const size_t lds_s = env_step_lds_bytes(P.J, P.M, P.T, P.obs_f32);
            if (P.obs_f32) hipLaunchKernelGGL((k_env_step<float>), dim3(P.B), dim3(WAVE), lds_s, h->stream, P);
            else hipLaunchKernelGGL((k_env_step<double>), dim3(P.B), dim3(WAVE), lds_s, h->stream, P);
        }
    }
    if (ev) HIPCHK(h, hipEventRecord(ev->second, h->stream));
#ifdef MTFJSP_STAMP
    static int printed = 0;
    if (getenv("MTFJSP_STAMP_PRINT") && (printed++ % 9) == 4 && printed < 60) {
        (void)hipStreamSynchronize(h->stream);
        std::vector<unsigned long long> hst((size_t)P.B * 8);
        (void)hipMemcpy(hst.data(), d_st, (size_t)P.B * 64, hipMemcpyDeviceToHost);
        double m[8] = {0};
        for (int w = 0; w < P.B; w++) for (int i = 0; i < 8; i++) m[i] += (double)hst[(size_t)w * 8 + i] / P.B;
        if (P.T > 128) {                                                    // grouped LDS kernel: s_memrealtime stamps (100 MHz) per instance; slot 7 by the group's first
            unsigned long long t0 = ~0ull; double r[8] = {0}; int n7 = 0;
            for (int w = 0; w < P.B; w++) t0 = hst[(size_t)w * 8] < t0 ? hst[(size_t)w * 8] : t0;
            for (int w = 0; w < P.B; w++) for (int i = 0; i < 8; i++) { if (i == 7 && hst[(size_t)w * 8 + 7] < t0) continue; r[i] += (double)(hst[(size_t)w * 8 + i] - t0) / 100.0; if (i == 7) n7++; }
            for (int i = 0; i < 7; i++) r[i] /= P.B;
            r[7] /= n7 ? n7 : 1;
            printf("STAMP k_env_step_grp B=%d (us since the first wave's start): entry %.2f  loads in LDS %.2f  scheduled %.2f  estimates %.2f  terms+energy sum %.2f  observation+ELL %.2f  wave done %.2f  tail drained %.2f\n",
                   P.B, r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
        } else if (hst[1] > 1000000000ull) {                                // grouped kernel: s_memrealtime stamps (100 MHz) of wave 0 per workgroup
            const int ng = (P.B + 15) / 16; double r[8] = {0};
            unsigned long long t0 = ~0ull;
            for (int w = 0; w < ng; w++) t0 = hst[(size_t)w * 8] < t0 ? hst[(size_t)w * 8] : t0;
            for (int w = 0; w < ng; w++) for (int i = 0; i < 8; i++) r[i] += (double)(hst[(size_t)w * 8 + i] - t0) / 100.0 / ng;
#ifdef MTFJSP_STAMP_WAVES
            if (P.B == 4096) {
                double e[16] = {0}, f[16] = {0}, c[16] = {0}, dn[16] = {0};
                for (int w = 0; w < ng; w++) for (int g = 0; g < 16; g++) {
                    e[g] += (double)(hst[2048 + (size_t)w * 64 + g * 4] - t0) / 100.0 / ng; f[g] += (double)(hst[2048 + (size_t)w * 64 + g * 4 + 1] - t0) / 100.0 / ng;
                    c[g] += (double)(hst[2048 + (size_t)w * 64 + g * 4 + 2] - t0) / 100.0 / ng; dn[g] += (double)(hst[2048 + (size_t)w * 64 + g * 4 + 3] - t0) / 100.0 / ng;
                }
                for (int g = 0; g < 16; g++) printf("STAMPW wave %2d: entry %.2f first-hop %.2f costs %.2f done %.2f\n", g, e[g], f[g], c[g], dn[g]);
                double t2[4] = {0};
                for (int w = 0; w < ng; w++) for (int i = 0; i < 4; i++) t2[i] += (double)(hst[2048 + 256 * 64 + (size_t)w * 4 + i] - t0) / 100.0 / ng;
                printf("STAMPT scalar-part wave: inputs read %.2f  idle sum done %.2f  rewards + scaling done %.2f  machine row done %.2f\n", t2[0], t2[1], t2[2], t2[3]);
            }
#endif
            printf("STAMP k_env_grp16 B=%d (us since the first workgroup's start, wave 0): entry %.2f  first-hop data %.2f  decision %.2f  per-task costs done %.2f  wave done %.2f  barrier %.2f  scalar-part wave done %.2f  ELL/mask wave done %.2f\n",
                   P.B, r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
        } else
        printf("STAMP k_env B=%d: load %.0f  schedule %.0f  estimate+terms %.0f  lane0-costs %.0f  scaler+info %.0f  observation %.0f  mask %.0f+writeback-issue  drain %.0f  (cycles/wave)\n",
               P.B, m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7]);
    }
#endif
    HIPCHK(h, hipGetLastError());
    return MTFJSP_OK;
}
extern "C" int mtfjsp_step_host(mtfjsp_handle_t h, const int32_t *task_host, const int32_t *mach_host)
{
    if (!h || !task_host || !mach_host) return MTFJSP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    const size_t B = h->cfg.batch;
    HIPCHK(h, hipMemcpyAsync(h->d_task, task_host, B * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_mach, mach_host, B * 4, hipMemcpyHostToDevice, h->stream));
    int rc = mtfjsp_step(h, h->d_task, h->d_mach);
    if (rc) return rc;
    std::vector<int32_t> st(B);
    HIPCHK(h, hipMemcpyAsync(st.data(), h->obs.status, B * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (size_t i = 0; i < B; i++)
        if (st[i] & MTFJSP_ST_INVALID) {
            char buf[160];
            snprintf(buf, sizeof buf, "invalid action for instance %zu: task %d machine %d (already scheduled, job predecessor unscheduled, or out of range)", i, task_host[i], mach_host[i]);
            h->err = buf;
            return MTFJSP_ERR_ACTION;
        }
    return MTFJSP_OK;
}

extern "C" int mtfjsp_observe_mfea1(mtfjsp_handle_t h, const int32_t *task_idx, const uint8_t *mmask_in, void *out, uint8_t *mmask_out)
{
    if (!h || !task_idx || !out) return MTFJSP_ERR_ARG;
    int rc = check_ready(h, true);
    if (rc) return rc;
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    const int B = h->cfg.batch, M = h->cfg.n_machine, n = B * M;
    if (h->cfg.obs_dtype == MTFJSP_OBS_F32)
        hipLaunchKernelGGL((k_mfea1<float>), dim3((n + 255) / 256), dim3(256), 0, h->stream, B, h->T, M, h->t, h->p, h->tt, h->mean3, h->shop, h->pl, task_idx, mmask_in, (float *)out, mmask_out);
    else
        hipLaunchKernelGGL((k_mfea1<double>), dim3((n + 255) / 256), dim3(256), 0, h->stream, B, h->T, M, h->t, h->p, h->tt, h->mean3, h->shop, h->pl, task_idx, mmask_in, (double *)out, mmask_out);
    HIPCHK(h, hipGetLastError());
    return MTFJSP_OK;
}

extern "C" int mtfjsp_get_mfea1_context(mtfjsp_handle_t h, void *m_fea1_out, uint8_t *mmask_out, mtfjsp_mfea1_ctx_t *ctx)
{
    if (!h || !m_fea1_out || !mmask_out || !ctx) return MTFJSP_ERR_ARG;
    int rc = check_ready(h, false);
    if (rc) return rc;
    ctx->t = h->t; ctx->p = h->p; ctx->tt = h->tt; ctx->mean3 = h->mean3; ctx->shop = h->shop; ctx->link = h->pl;
    ctx->m_fea1_out = m_fea1_out; ctx->mmask_out = mmask_out;
    ctx->T = h->T; ctx->M = h->cfg.n_machine; ctx->obs_f32 = h->cfg.obs_dtype == MTFJSP_OBS_F32;
    ctx->m_fea2 = h->obs_bound ? h->obs.m_fea2 : nullptr;
    return MTFJSP_OK;
}

extern "C" int mtfjsp_random_actions(mtfjsp_handle_t h, uint64_t seed, uint64_t counter, int32_t *task_idx, int32_t *mach_idx, int32_t *job_idx)
{
    if (!h || !task_idx || !mach_idx) return MTFJSP_ERR_ARG;
    int rc = check_ready(h, true);
    if (rc) return rc;
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    const int B = h->cfg.batch;
    hipLaunchKernelGGL(k_random_actions, dim3((B + 127) / 128), dim3(128), 0, h->stream, B, h->cfg.n_job, h->cfg.n_machine, h->T,
                       h->t, h->obs.candidate, h->obs.job_mask, seed, counter, task_idx, mach_idx, job_idx);
    HIPCHK(h, hipGetLastError());
    return MTFJSP_OK;
}

extern "C" int mtfjsp_draw_reward_weights(mtfjsp_handle_t h, uint64_t seed, uint64_t episode, double *w3_out)
{
    if (!h || !w3_out) return MTFJSP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    const int B = h->cfg.batch;
    hipLaunchKernelGGL(k_draw_w3, dim3((B + 255) / 256), dim3(256), 0, h->stream, B, seed, episode, w3_out);
    HIPCHK(h, hipGetLastError());
    return MTFJSP_OK;
}

extern "C" int mtfjsp_export_dense_adj(mtfjsp_handle_t h, double *out)
{
    if (!h || !out) return MTFJSP_ERR_ARG;
    int rc = check_ready(h, true);
    if (rc) return rc;
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    const size_t B = h->cfg.batch, T = h->T;
    HIPCHK(h, hipMemsetAsync(out, 0, B * T * T * 8, h->stream));
    const int n = (int)(B * T);
    hipLaunchKernelGGL(k_dense_adj, dim3((n + 255) / 256), dim3(256), 0, h->stream, (int)B, (int)T, h->obs.ell_col, h->obs.ell_val, out);
    HIPCHK(h, hipGetLastError());
    return MTFJSP_OK;
}

extern "C" int mtfjsp_export_dense_adj_host(mtfjsp_handle_t h, double *out_host)
{
    if (!h || !out_host) return MTFJSP_ERR_ARG;
    const size_t n = (size_t)h->cfg.batch * h->T * h->T;
    if (!h->dense_scratch) {
        int rc = dalloc(h, &h->dense_scratch, n);
        if (rc) return MTFJSP_ERR_HIP;
    }
    int rc = mtfjsp_export_dense_adj(h, h->dense_scratch);
    if (rc) return rc;
    return mtfjsp_copy_to_host(h, out_host, h->dense_scratch, n * sizeof(double));
}

extern "C" int mtfjsp_valid_action_mask(mtfjsp_handle_t h, uint8_t *out)
{
    if (!h || !out) return MTFJSP_ERR_ARG;
    int rc = check_ready(h, true);
    if (rc) return rc;
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    const int n = h->cfg.batch * h->T;
    hipLaunchKernelGGL(k_valid_mask, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->cfg.batch, h->T, h->cfg.n_machine, h->pl, out);
    HIPCHK(h, hipGetLastError());
    return MTFJSP_OK;
}

extern "C" int mtfjsp_copy_to_host(mtfjsp_handle_t h, void *dst, const void *src, size_t n)
{
    if (!h || !dst || !src) return MTFJSP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    HIPCHK(h, hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MTFJSP_OK;
}

extern "C" int mtfjsp_read_state_host(mtfjsp_handle_t h, int which, void *out)
{
    if (!h || !out) return MTFJSP_ERR_ARG;
    int rc = check_ready(h, true);
    if (rc) return rc;
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    const size_t B = h->cfg.batch, T = h->T, M = h->cfg.n_machine;
    std::vector<TaskPL> pl(B * T);
    HIPCHK(h, hipMemcpyAsync(pl.data(), h->pl, B * T * sizeof(TaskPL), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    std::vector<Link> link(B * T);
    for (size_t i = 0; i < B * T; i++) link[i] = pl[i].link;
    switch (which) {
    case MTFJSP_STATE_MACHINE: {
        int32_t *o = (int32_t *)out;
        for (size_t i = 0; i < B * T; i++) o[i] = link[i].mach;
        return MTFJSP_OK;
    }
    case MTFJSP_STATE_START:
    case MTFJSP_STATE_FINISH: {
        double *o = (double *)out;
        std::vector<TaskSD> sd(B * T);
        rc = mtfjsp_copy_to_host(h, sd.data(), h->sd, B * T * sizeof(TaskSD));
        if (rc) return rc;
        for (size_t i = 0; i < B * T; i++) o[i] = link[i].mach < 0 ? NAN : which == MTFJSP_STATE_START ? sd[i].st : sd[i].st + sd[i].dur;   // ft == st + dur (env:356)
        return MTFJSP_OK;
    }
    case MTFJSP_STATE_ROUTES: {
        int32_t *o = (int32_t *)out;
        for (size_t i = 0; i < B * M * T; i++) o[i] = -1;
        for (size_t b = 0; b < B; b++)
            for (size_t v = 0; v < T; v++) {
                const Link &l = link[b * T + v];
                if (l.mach >= 0) o[(b * M + l.mach) * T + l.pos] = (int32_t)v;
            }
        return MTFJSP_OK;
    }
    case MTFJSP_STATE_PREV_COSTS:
    case MTFJSP_STATE_SCALER:
    case MTFJSP_STATE_W3: {
        std::vector<double> sc(B * SCAL_N);
        rc = mtfjsp_copy_to_host(h, sc.data(), h->scal, B * SCAL_N * 8);
        if (rc) return rc;
        double *o = (double *)out;
        for (size_t b = 0; b < B; b++) {
            const double *s = &sc[b * SCAL_N];
            if (which == MTFJSP_STATE_PREV_COSTS) { for (int i = 0; i < 4; i++) o[b * 4 + i] = s[S_MK_PREV + i]; }
            else if (which == MTFJSP_STATE_W3) { for (int i = 0; i < 3; i++) o[b * 3 + i] = s[S_W3 + i]; }
            else {
                for (int i = 0; i < 4; i++) { o[b * 17 + i] = s[S_R + i]; o[b * 17 + 5 + i] = s[S_MEAN + i]; o[b * 17 + 9 + i] = s[S_S + i]; o[b * 17 + 13 + i] = s[S_STD + i]; }
                o[b * 17 + 4] = s[S_N];
            }
        }
        return MTFJSP_OK;
    }
    default:
        h->err = "read_state: unknown selector";
        return MTFJSP_ERR_ARG;
    }
}

// restores the RewardScaling state of instances [first, first + count) from the layout MTFJSP_STATE_SCALER reads (R[4], n, mean[4],
// S[4], std[4]): the per-env gym step of the Python mirror drives ONE instance through the fused step kernel, which also applies
// RewardScaling — the reference's env.step (env:716-974) never touches the scaler, only the batched step does (pe:255-260)
extern "C" int mtfjsp_set_scaler_state_host(mtfjsp_handle_t h, int32_t first, int32_t count, const double *state17)
{
    if (!h || !state17 || first < 0 || count < 1 || (size_t)first + (size_t)count > (size_t)h->cfg.batch) return MTFJSP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    std::vector<double> sc((size_t)count * SCAL_N);
    int rc = mtfjsp_copy_to_host(h, sc.data(), h->scal + (size_t)first * SCAL_N, (size_t)count * SCAL_N * 8);
    if (rc) return rc;
    for (int b = 0; b < count; b++) {
        double *s = &sc[(size_t)b * SCAL_N];
        const double *o = state17 + (size_t)b * 17;
        for (int i = 0; i < 4; i++) { s[S_R + i] = o[i]; s[S_MEAN + i] = o[5 + i]; s[S_S + i] = o[9 + i]; s[S_STD + i] = o[13 + i]; }
        s[S_N] = o[4];
    }
    HIPCHK(h, hipMemcpyAsync(h->scal + (size_t)first * SCAL_N, sc.data(), (size_t)count * SCAL_N * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MTFJSP_OK;
}

extern "C" int mtfjsp_gae(mtfjsp_handle_t h, int32_t S, const float *r, int64_t r_ss, int64_t r_sb, const float *v, int64_t v_ss, int64_t v_sb,
                          const float *v_next, int64_t n_ss, int64_t n_sb, const float *done, float gamma, float lambda, float *adv)
{
    if (!h || S < 1 || !r || !v || !v_next || !done || !adv) return MTFJSP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    const int B = h->cfg.batch;
    hipLaunchKernelGGL(k_gae, dim3((B + 63) / 64), dim3(64), 0, h->stream, B, (int)S, r, (long)r_ss, (long)r_sb, v, (long)v_ss, (long)v_sb,
                       v_next, (long)n_ss, (long)n_sb, done, gamma, lambda, adv);
    HIPCHK(h, hipGetLastError());
    return MTFJSP_OK;
}

// ---------------------------------------------------------------------------------------------
// Global advantage normalisation of the rollout -> update hand-off (ppo:485,532: `(adv - adv.mean()) / (adv.std() + 1e-5)` over the
// whole [S, B_total] tensor, torch's unbiased std) on the gathered buffer G [world][K][n] (n = S * B_local; what one
// all_gather_into_tensor of the packed [K,S,B_local] advantages leaves on every rank): statistics per tensor k over all shards, this
// rank's block normalised, value targets = normalised advantage + value at act time (ppo:668-671,689), and — optionally — the
// reference's single-process layout [K][S][B_total] (rank-major column blocks).  Deterministic: per-block f64 partial sums in a
// fixed order (no atomics), so two runs over the same numbers give the same bits.
#define ADV_NB 128
struct AdvNormArgs {
    int K, Kt, world, rank, B_local; long n;        // K tensors are normalised, Kt >= K travel in G; n = S * B_local
    const float *G; float eps;
    const float *val[16]; long val_ss[16], val_sb[16];   // value at act time of tensor k as an [S,B_local] view (NULL: no target)
    float *norm, *targets, *full;                   // [K][n], [K][n] or NULL, [Kt][S][world * B_local] or NULL
    double *partial;                                // [K][ADV_NB][2]
};
__global__ __launch_bounds__(256) void k_adv_stats(AdvNormArgs A)
{
    __shared__ double s_red[2][256];
    const int k = blockIdx.y, j = blockIdx.x, tid = threadIdx.x;
    const long n_all = (long)A.world * A.n;
    double su = 0.0, sq = 0.0;
    for (long i = (long)j * 256 + tid; i < n_all; i += (long)ADV_NB * 256) {
        const long w = i / A.n, r = i - w * A.n;
        const double x = (double)A.G[((size_t)w * A.Kt + k) * A.n + r];
        su += x; sq += x * x;
    }
    s_red[0][tid] = su; s_red[1][tid] = sq;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { s_red[0][tid] += s_red[0][tid + o]; s_red[1][tid] += s_red[1][tid + o]; }
        __syncthreads();
    }
    if (tid == 0) { A.partial[((size_t)k * ADV_NB + j) * 2] = s_red[0][0]; A.partial[((size_t)k * ADV_NB + j) * 2 + 1] = s_red[1][0]; }
}
__global__ __launch_bounds__(256) void k_adv_norm(AdvNormArgs A)
{
    __shared__ double s_red[2][ADV_NB];
    __shared__ float s_ms[2];
    const int k = blockIdx.y, tid = threadIdx.x;
    const bool normed = k < A.K;                    // (block-uniform)
    if (normed) {
        if (tid < ADV_NB) { s_red[0][tid] = A.partial[((size_t)k * ADV_NB + tid) * 2]; s_red[1][tid] = A.partial[((size_t)k * ADV_NB + tid) * 2 + 1]; }
        __syncthreads();
        for (int o = ADV_NB / 2; o > 0; o >>= 1) {
            if (tid < o) { s_red[0][tid] += s_red[0][tid + o]; s_red[1][tid] += s_red[1][tid + o]; }
            __syncthreads();
        }
        if (tid == 0) {
            const double cnt = (double)A.world * (double)A.n;
            const double mean = s_red[0][0] / cnt;
            double var = cnt > 1.0 ? (s_red[1][0] - cnt * mean * mean) / (cnt - 1.0) : 0.0;      // unbiased, as torch.std
            if (var < 0) var = 0;
            s_ms[0] = (float)mean; s_ms[1] = (float)sqrt(var);
        }
        __syncthreads();
    } else if (!A.full) return;
    const float mean = normed ? s_ms[0] : 0.f, inv = normed ? 1.0f / (s_ms[1] + A.eps) : 0.f;
    const long n_all = (long)A.world * A.n;
    const long Bt = (long)A.world * A.B_local;
    for (long i = (long)blockIdx.x * 256 + tid; i < n_all; i += (long)gridDim.x * 256) {
        const long w = i / A.n, r = i - w * A.n;
        const float x = A.G[((size_t)w * A.Kt + k) * A.n + r];
        const long sidx = r / A.B_local, b = r - sidx * A.B_local;
        if (A.full) A.full[(size_t)k * A.n * A.world + sidx * Bt + w * A.B_local + b] = x;
        if (normed && w == A.rank) {
            const float a = (x - mean) * inv;
            A.norm[(size_t)k * A.n + r] = a;
            if (A.targets && A.val[k]) A.targets[(size_t)k * A.n + r] = a + A.val[k][sidx * A.val_ss[k] + b * A.val_sb[k]];
        }
    }
}
extern "C" int mtfjsp_normalize_advantages(mtfjsp_handle_t h, int32_t K, int32_t K_total, int32_t world, int32_t rank, int32_t S, const float *gathered, float eps,
                                           const float *const *values, const int64_t *value_stride_s, const int64_t *value_stride_b,
                                           float *norm_out, float *targets_out, float *full_out)
{
    if (!h || K < 1 || K > 16 || K_total < K || K_total > 16 || world < 1 || rank < 0 || rank >= world || S < 1 || !gathered || !norm_out) return MTFJSP_ERR_ARG;
    if (targets_out && (!values || !value_stride_s || !value_stride_b)) return MTFJSP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    if (!h->adv_partial) {
        HIPCHK(h, hipMalloc((void **)&h->adv_partial, (size_t)16 * ADV_NB * 2 * sizeof(double)));
        h->owned.push_back(h->adv_partial);
    }
    AdvNormArgs a{};
    a.K = K; a.Kt = K_total; a.world = world; a.rank = rank; a.B_local = h->cfg.batch; a.n = (long)S * h->cfg.batch; a.G = gathered; a.eps = eps;
    for (int k = 0; k < K; k++) {
        a.val[k] = values ? values[k] : nullptr;
        a.val_ss[k] = value_stride_s ? (long)value_stride_s[k] : 0; a.val_sb[k] = value_stride_b ? (long)value_stride_b[k] : 0;
    }
    a.norm = norm_out; a.targets = targets_out; a.full = full_out; a.partial = h->adv_partial;
    hipLaunchKernelGGL(k_adv_stats, dim3(ADV_NB, K), dim3(256), 0, h->stream, a);
    const long n_all = (long)world * a.n;
    int gx = (int)((n_all + 256 * 8 - 1) / (256 * 8));
    gx = gx < 1 ? 1 : gx > 1024 ? 1024 : gx;
    hipLaunchKernelGGL(k_adv_norm, dim3(gx, K_total), dim3(256), 0, h->stream, a);
    HIPCHK(h, hipGetLastError());
    return MTFJSP_OK;
}
// out[k][s][b] = src_k[s * ss_k + b * sb_k]: K strided [S,B] f32 views into one packed buffer (the value tensors that ride in the
// hand-off's all-gather beside the advantages), one launch
struct PackArgs { int K, S, B; const float *src[16]; long ss[16], sb[16]; float *out; };
__global__ __launch_bounds__(256) void k_pack_views(PackArgs A)
{
    const int k = blockIdx.y;
    const long n = (long)A.S * A.B;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long s = i / A.B, b = i - s * A.B;
        A.out[(size_t)k * n + i] = A.src[k][s * A.ss[k] + b * A.sb[k]];
    }
}
extern "C" int mtfjsp_pack_views(mtfjsp_handle_t h, int32_t K, int32_t S, const float *const *src, const int64_t *stride_s, const int64_t *stride_b, float *out)
{
    if (!h || K < 1 || K > 16 || S < 1 || !src || !stride_s || !stride_b || !out) return MTFJSP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    PackArgs a{};
    a.K = K; a.S = S; a.B = h->cfg.batch; a.out = out;
    for (int k = 0; k < K; k++) { if (!src[k]) return MTFJSP_ERR_ARG; a.src[k] = src[k]; a.ss[k] = (long)stride_s[k]; a.sb[k] = (long)stride_b[k]; }
    const long n = (long)S * a.B;
    int gx = (int)((n + 256 * 4 - 1) / (256 * 4));
    gx = gx < 1 ? 1 : gx > 1024 ? 1024 : gx;
    hipLaunchKernelGGL(k_pack_views, dim3(gx, K), dim3(256), 0, h->stream, a);
    HIPCHK(h, hipGetLastError());
    return MTFJSP_OK;
}

// ---------------------------------------------------------------------------------------------
// Measurement only ("fraction = achieved / measured copy bandwidth of a same-footprint streaming kernel"): a launch
// that reads `read_bytes` and writes `write_bytes` with W-byte accesses, perfectly coalesced, nothing else — the denominator the
// step kernel's achieved bytes/s are compared with at the same batch.  Word i of the read stream is copied to word i of the write
// stream while both last; the rest of the longer stream is read into a checksum / filled with it.
typedef unsigned fp_u2 __attribute__((ext_vector_type(2)));
typedef unsigned fp_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned fp_first(unsigned v) { return v; }
__device__ __forceinline__ unsigned fp_first(fp_u2 v) { return v.x; }
__device__ __forceinline__ unsigned fp_first(fp_u4 v) { return v.x; }
template <typename WORD>
__global__ __launch_bounds__(256) void k_footprint_copy(const WORD *__restrict__ src, WORD *__restrict__ dst, size_t nr, size_t nw, unsigned *sink)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t nc = nr < nw ? nr : nw;
    unsigned acc = 0;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < nc; i += stride) dst[i] = src[i];                         // the common part: a copy
    for (; i < nr; i += stride) acc ^= fp_first(src[i]);                 // the longer stream's tail: read only ...
    for (; i < nw; i += stride) dst[i] = WORD(acc);                      // ... or write only
    if (acc == 0x9e3779b9u) *sink = acc;                                 // keeps the read-only tail alive
}
extern "C" int mtfjsp_footprint_copy(mtfjsp_handle_t h, size_t read_bytes, size_t write_bytes, int32_t access_bytes, int32_t grid, int32_t reps,
                                     double *avg_us_out, double *min_us_out)
{
    if (!h || (access_bytes != 4 && access_bytes != 8 && access_bytes != 16) || grid < 1 || reps < 1 || !avg_us_out) return MTFJSP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    const size_t nr = read_bytes / access_bytes, nw = write_bytes / access_bytes;
    // every exit path frees what was allocated (bench.py calls this a dozen times per batch size)
    struct Scratch {
        void *src = nullptr, *dst = nullptr; unsigned *sink = nullptr; std::vector<hipEvent_t> ev;
        ~Scratch() { for (auto &e : ev) if (e) (void)hipEventDestroy(e); (void)hipFree(src); (void)hipFree(dst); (void)hipFree(sink); }
    } S;
    HIPCHK(h, hipMalloc(&S.src, nr * access_bytes + 16)); HIPCHK(h, hipMalloc(&S.dst, nw * access_bytes + 16)); HIPCHK(h, hipMalloc((void **)&S.sink, 4));
    HIPCHK(h, hipMemsetAsync(S.src, 1, nr * access_bytes + 16, h->stream));
    S.ev.assign(2 * (size_t)reps, nullptr);
    for (auto &e : S.ev) HIPCHK(h, hipEventCreate(&e));
    void *src = S.src, *dst = S.dst; unsigned *sink = S.sink;
    std::vector<hipEvent_t> &ev = S.ev;
    auto launch = [&]() {
        if (access_bytes == 16) hipLaunchKernelGGL(k_footprint_copy<fp_u4>, dim3(grid), dim3(256), 0, h->stream, (const fp_u4 *)src, (fp_u4 *)dst, nr, nw, sink);
        else if (access_bytes == 8) hipLaunchKernelGGL(k_footprint_copy<fp_u2>, dim3(grid), dim3(256), 0, h->stream, (const fp_u2 *)src, (fp_u2 *)dst, nr, nw, sink);
        else hipLaunchKernelGGL(k_footprint_copy<unsigned>, dim3(grid), dim3(256), 0, h->stream, (const unsigned *)src, (unsigned *)dst, nr, nw, sink);
    };
    for (int i = 0; i < 10; i++) launch();                    // warm: code object, clocks, address translation, caches in the state repeated launches see
    for (int i = 0; i < reps; i++) {
        HIPCHK(h, hipEventRecord(ev[2 * i], h->stream));
        launch();
        HIPCHK(h, hipEventRecord(ev[2 * i + 1], h->stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    double tot = 0, mn = 1e30;
    for (int i = 0; i < reps; i++) {
        float ms = 0.f;
        HIPCHK(h, hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
        tot += ms; mn = ms < mn ? ms : mn;
    }
    HIPCHK(h, hipGetLastError());
    *avg_us_out = tot / reps * 1e3;
    if (min_us_out) *min_us_out = mn * 1e3;
    return MTFJSP_OK;
}

extern "C" int mtfjsp_timing_begin(mtfjsp_handle_t h)
{
    if (!h) return MTFJSP_ERR_ARG;
    h->timing = true; h->ev_used = 0;
    return MTFJSP_OK;
}
extern "C" int mtfjsp_timing_end(mtfjsp_handle_t h, double *ms_total, int64_t *launches)
{
    if (!h) return MTFJSP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    double tot = 0.0;
    for (size_t i = 0; i < h->ev_used; i++) {
        float ms = 0.f;
        HIPCHK(h, hipEventElapsedTime(&ms, h->ev_pool[i].first, h->ev_pool[i].second));
        tot += ms;
    }
    if (ms_total) *ms_total = tot;
    if (launches) *launches = (int64_t)h->ev_used;
    h->timing = false; h->ev_used = 0;
    return MTFJSP_OK;
}
