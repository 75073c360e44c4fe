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

struct __align__(8) Link { short mach, prev, pos, pad; };      // per task: machine (-1), route predecessor (-1), rank in route
struct __align__(8) MRec { short head, tail, len, pad; };      // per machine

struct EnvParams {
    int B, J, M, T, left_shift, obs_f32;
    unsigned inv_M;                    // ceil(2^32 / M)
    double w_mk, w_ec, w_tt, divisor, gamma;
    // instance constants
    const double *t, *p, *tt;          // [B,T,M] [B,T,M] [B,M,M]
    const double2 *cst;                // [B,T] {min_dur, min_pt}
    // dynamic state
    double *st, *ft, *dur, *psel;      // [B,T]
    Link *link;                        // [B,T]
    MRec *mrec;                        // [B,M]
    short *jcnt;                       // [B,J] scheduled ops per job (ops of a job are scheduled in order)
    double *pte;                       // [B,T] estimated / real processing energy per task (env:1995)
    double *jmax, *jrow;               // [B,J] max estimated finish / max real finish per job
    int *lastm;                        // [B]   node whose merged job+machine edge was created by the previous step (-1)
    double *mfea;                      // [B,M,8] f64 master copy of machines_fea
    double *scal;                      // [B,SCAL_N]
    // inputs
    const int *task_idx, *mach_idx;    // [B]
    const double *w3;                  // [B,3] (reset)
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
