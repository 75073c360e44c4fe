// mtfjsp_env_dev.h — device-side definitions of the environment state that more than one translation unit needs: the step
// kernels of mtfjsp_env.hip, and mtfjsp_encoder.hip, whose machine-actor heads kernel can run the grouped step of its 16 instances
// as its tail (k_headsx_envstep).  Kernel parameter block, per-task / per-machine records, scalar slots, uniform lane reads.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/mtfjsp.h"

#define WAVE 64
#define SCAL_N 28          // doubles of per-instance scalar state
// scalar slots
#define S_MK_PREV 0
#define S_E1_PREV 1
#define S_TR_PREV 2
#define S_ID_PREV 3
#define S_TR_THIS 4
#define S_W3 5             // 5,6,7
#define S_R 8              // 8..11   RewardScaling.R
#define S_MEAN 12          // 12..15
#define S_S 16             // 16..19
#define S_STD 20           // 20..23
#define S_N 24             // RunningMeanStd.n
#define S_NSCHED 25        // number of scheduled tasks
#define S_TRLAST 26        // transport time of the previous step's job edge (the entry tt[mach(lastm-1), mach(lastm)] that the reverting merged edge needs)
#define S_LASTM 27         // node whose merged job+machine edge was created by the previous step (-1), as a double

struct __align__(8) Link { short mach, prev, pos, pad; };      // per task: machine (-1), route predecessor (-1), rank in route, route successor (pad)
// Dynamic state, round 6: 16-byte records so that a lane's share of the state is three wide loads (it was eleven 8-byte ones) and
// fewer bytes: the finish time is not stored (ft == st + dur, the very addition that made it: env:356), the transport matrix is
// read as ONE row of its transpose behind the action (the step only ever needs column m), per-job and per-machine words share records.
struct __align__(16) TaskSD { double st, dur; };               // per task: start, duration (0, 0 while unscheduled)
struct __align__(16) TaskPL { double pte; Link link; };        // per task: estimated / real processing energy (env:1995), route links
struct __align__(16) JobR { double jmax, jrow; };              // per job: max estimated finish / max real finish
struct __align__(8) MJRec { short head, tail, len, cnt; };     // element i: route head / tail / length of MACHINE i, scheduled ops of JOB i

struct EnvParams {
    int B, J, M, T, left_shift, obs_f32;
    unsigned inv_M;                    // ceil(2^32 / M)
    double w_mk, w_ec, w_tt, divisor, gamma;
    // instance constants
    const double *t, *p, *tt;          // [B,T,M] [B,T,M] [B,M,M]
    const double2 *cst;                // [B,T] {min_dur, min_pt}
    const double *ttT;                 // [B,M,M] transport times transposed: ttT[b][m][x] = tt[b][x][m]
    // dynamic state
    TaskSD *sd;                        // [B,T]
    TaskPL *pl;                        // [B,T]
    JobR *jr;                          // [B,J]
    MJRec *mj;                         // [B,MJ], MJ = max(J, M) (ops of a job are scheduled in order: cnt is the job's next op)
    int MJ;
    double *mfea;                      // [B,M,8] f64 master copy of machines_fea
    double *scal;                      // [B,SCAL_N]
    // inputs
    const int *task_idx, *mach_idx;    // [B]
    const double *w3;                  // [B,3] (reset)
    // reset of an EPISODE in one launch (mtfjsp_reset_episode): draw = 1: the reward weights are drawn here (k_draw_w3's Philox stream for
    // (draw_seed, draw_episode, instance): same bits) and written to w3_out; reset_returns = 1: the discounted returns R of the reward
    // scaler are zeroed as well (pt:123, run:283-284)
    int draw, reset_returns;
    unsigned long long draw_seed, draw_episode;
    double *w3_out;
    // outputs
    mtfjsp_obs_t obs;
    unsigned long long *stamps;        // diagnostic build only
    float *rec_r4, *rec_done;          // optional f32 trajectory record of this step ([4,B], [B])
    const short *pw_tab;               // leaves and merges of numpy's pairwise sum over T elements (pw_table; T > 128 only)
    int pw_nleaf;
};

__device__ __forceinline__ long trunc_l(double x) { return (long)x; }   // numpy astype(int): toward zero
__device__ __forceinline__ int rl_i(int x, int l) { return __builtin_amdgcn_readlane(x, l); }
__device__ __forceinline__ double rl_d(double x, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
// lane moves that stay in the vector ALU (no LDS round trip): DPP controls (0xB1 / 0x4E: lane ^ 1 / ^ 2 inside a quad, 0x141: lane -> 7 - lane
// inside 8 lanes, 0x108: lane <- lane + 8 inside a row of 16, 0x138: lane <- lane - 1 across the wave; a lane without a source reads 0)
template <int CTRL> __device__ __forceinline__ int dpp_i(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, true); }
template <int CTRL> __device__ __forceinline__ double dpp_d(double x) { return __hiloint2double(dpp_i<CTRL>(__double2hiint(x)), dpp_i<CTRL>(__double2loint(x))); }
// gfx950's row swaps: lanes 0..15 <- lanes 16..31 (and 32..47 <- 48..63) | lanes 0..31 <- lanes 32..63
__device__ __forceinline__ int up16_i(int x) { return (int)__builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false)[1]; }
__device__ __forceinline__ int up32_i(int x) { return (int)__builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false)[1]; }
__device__ __forceinline__ double up16_d(double x) { return __hiloint2double(up16_i(__double2hiint(x)), up16_i(__double2loint(x))); }
__device__ __forceinline__ double up32_d(double x) { return __hiloint2double(up32_i(__double2hiint(x)), up32_i(__double2loint(x))); }
