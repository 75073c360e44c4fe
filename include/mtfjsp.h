/*
 * mtfjsp.h — C ABI of libmtfjsp.so: MI355X-native batched MT-FJSP disjunctive-graph
 * environment (reset / step / observe / masks) and the rollout forward passes of
 * the GIN + GAT actors.  This is the drop-in boundary for the hot path of
 * RKWin93/E2E-MAPPO-for-MT-FJSP; every entry point cites the reference interface
 * it replaces ("pe:" = trainer/parallel_env.py, "env:" = graph-jsp-env/src/
 * graph_jsp_env/disjunctive_graph_jsp_env_singlestep.py, "ppo:" =
 * algorithm/ppo_algorithm.py, "ac:" = model/actor_critic.py, "agent:" =
 * algorithm/agent_func.py, "run:" = Run.py).  The reference-side ctypes binding
 * is shown in INTEGRATION.md.
 *
 * Conventions
 *  - plain C types only; every function returns 0 (MTFJSP_OK) or a negative
 *    mtfjsp_status; mtfjsp_last_error(h) describes the last failure on h.
 *  - One handle per GPU.  A handle is not thread-safe; distinct handles are.
 *  - All array arguments are DEVICE pointers unless the name ends in _host.
 *    Work is enqueued on the handle's stream (mtfjsp_set_stream; default: the
 *    NULL stream) and is asynchronous; *_host variants synchronise that stream.
 *  - Task index a in [0,T), T = n_job*n_machine, job = a / n_machine,
 *    op = a % n_machine (square instances: ops per job == n_machine, as in the
 *    reference generator).  Machine index m in [0,M).
 *  - Scheduling state and all times/rewards are IEEE binary64, evaluated in the
 *    reference's operation order without FMA contraction.
 */
#ifndef MTFJSP_H
#define MTFJSP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mtfjsp_env *mtfjsp_handle_t;
typedef struct mtfjsp_encoder *mtfjsp_encoder_t;

typedef enum {
    MTFJSP_OK = 0,
    MTFJSP_ERR_ARG = -1,      /* bad argument / configuration */
    MTFJSP_ERR_STATE = -2,    /* call order (e.g. step before load_instances/reset) */
    MTFJSP_ERR_HIP = -3,      /* HIP runtime error (no device, launch failure, OOM) */
    MTFJSP_ERR_ACTION = -4,   /* *_host step: at least one action was invalid */
    MTFJSP_ERR_RETRY = -5     /* encoder: an earlier forward on the single-launch GIN kernel failed asynchronously (time-out of a grid-wide statistics exchange);
                               * everything enqueued on the handle since then is invalid, the handle has switched to the streaming
                               * launches — repeat the work (see mtfjsp_encoder_check) */
} mtfjsp_status;

enum { MTFJSP_OBS_F64 = 0, MTFJSP_OBS_F32 = 1 };

/* per-instance status word written by mtfjsp_step (bound obs.status) */
enum {
    MTFJSP_PATH_MASK = 0x7,          /* scheduling path taken (env:1531-1685): */
    MTFJSP_PATH_EMPTY = 0,           /*   machine route was empty */
    MTFJSP_PATH_FRONT = 1,           /*   left-shifted to the front of the route */
    MTFJSP_PATH_BETWEEN = 2,         /*   left-shifted into a gap */
    MTFJSP_PATH_APPEND = 3,          /*   appended */
    MTFJSP_ST_INVALID = 0x100,       /* already scheduled / job predecessor unscheduled / index out of range:
                                        the instance is left untouched (the reference silently corrupts node
                                        attributes here, env:1496-1528; never reached under the masks) */
    MTFJSP_ST_INFEASIBLE = 0x200     /* t[a,m] < 0 chosen (pe:246-248 prints a warning and carries on; so do we) */
};

typedef struct {
    int32_t n_job, n_machine, n_edge;   /* pe:22-25 */
    int32_t batch;                      /* env_batch, pe:26 */
    int32_t left_shift;                 /* perform_left_shift_if_possible (pe:116; tester/pdrs.py:669 uses 0) */
    int32_t obs_dtype;                  /* MTFJSP_OBS_F64: observations as the reference returns them (f64)
                                           MTFJSP_OBS_F32: the same values rounded once to f32 — what ac:143,377
                                           `.float()` would produce — for the device-resident rollout */
    int32_t device_id;
    int32_t reserved;
    double gamma;                       /* GAMMA for RewardScaling, pe:81 */
    double w_mk, w_ec, w_tt;            /* config weights in the scalar reward, env:1119-1132 */
    double scaling_divisor;             /* reward_function_parameters['scaling_divisor'], env:1164 */
} mtfjsp_config_t;

/* Observation / step outputs, all DEVICE pointers, caller- or library-allocated.
 * Shapes use B = batch, T, M, J.                                                                  replaces        */
typedef struct {
    void *tasks_fea;      /* [B*T,12] obs_dtype: st_est, ft_est, pt_est, scheduled, in_degree,      env:2245-2277   */
                          /*          machine+1|0, t|0, p|0, job+1, w3[0..2]                                         */
    int32_t *ell_col;     /* [B*T,2]  in-edge sources of node v (task index inside the instance,   env:2019-2073   */
                          /*          -1 = none): slot 0 job edge, slot 1 machine edge                               */
    float *ell_val;       /* [B*T,2]  adj_wrk values of those edges (small integers); the self      (dense: see     */
                          /*          loop (value 1) is implicit                                     export below)   */
    void *m_fea2;         /* [B,M,8]  obs_dtype                                                     env:2315-2354   */
    double *info;         /* [B,6]    reward, done, mk_s, idle_s, pt_s, tt_s (scaled components)    pe:255-262      */
    double *raw;          /* [B,5]    reward, r_mk, r_idle, r_pt, r_tt (unscaled); may be NULL      env:1051-1171   */
    int32_t *candidate;   /* [B,J]    next selectable task of each job                              ppo:306-309     */
    uint8_t *job_mask;    /* [B,J]    1 = job not selectable                                        ppo:238-297     */
    int32_t *status;      /* [B]      see MTFJSP_PATH_* / MTFJSP_ST_*                                               */
} mtfjsp_obs_t;

/* ------------------------------------------------------------------ lifecycle */
/* = Parallel_env.__init__ (pe:20-37). Allocates all device state for `batch` instances. */
int mtfjsp_create(const mtfjsp_config_t *cfg, mtfjsp_handle_t *out);
int mtfjsp_destroy(mtfjsp_handle_t h);
const char *mtfjsp_last_error(mtfjsp_handle_t h);      /* h may be NULL: last create() error */
int mtfjsp_set_stream(mtfjsp_handle_t h, void *hip_stream);
int mtfjsp_synchronize(mtfjsp_handle_t h);

/* Library-owned observation buffers (freed by destroy) / caller-owned ones. */
int mtfjsp_alloc_obs(mtfjsp_handle_t h, mtfjsp_obs_t *out);
int mtfjsp_bind_obs(mtfjsp_handle_t h, const mtfjsp_obs_t *obs);
/* = the observation copies of ReplayBuffer.store_operation (trainer/replaybuffer.py:82-139): snapshot of the
 * CURRENT bound observation into caller buffers of the same layout (a trajectory slot), one launch on the handle's stream.
 * Fields of dst left NULL are skipped (status is never copied).  The adjacency stays ELL: the reference's dense
 * [steps,B,T,T] f64 buffer is 7.6 GB per copy at B=4096 J6M6. */
int mtfjsp_snapshot_obs(mtfjsp_handle_t h, const mtfjsp_obs_t *dst);
/* The same launch with a second destination and the scalar reward: the reference's buffer stores s' of step k (adj_, fea_, ...:
 * replaybuffer.py:108-118) and s of step k+1 (replaybuffer.py:97-104) separately although they are the same observation inside an
 * episode, and the reward as f32 [B] (replaybuffer.py:106 <- info[:,0], pe:255-262).  dst2 (may be NULL) receives a second copy of
 * every field it has; reward_out (may be NULL) receives (float)info[b][0]. */
int mtfjsp_snapshot_obs2(mtfjsp_handle_t h, const mtfjsp_obs_t *dst, const mtfjsp_obs_t *dst2, float *reward_out);

/* ------------------------------------------------------------------ instances */
/* = Parallel_env.get_batch (pe:39-66): t,p [B,T,M] (negative = machine infeasible), tt [B,M,M],
 * shop_of_machine [B,M] (0-based shop id; the reference's edge[E,M/E] table inverted).
 * Precomputes min_dur/min_pt (env:1932-1950) and the per-task means used by m_fea1 (pe:176-183). */
int mtfjsp_load_instances(mtfjsp_handle_t h, const double *t, const double *p, const double *tt,
                          const int32_t *shop_of_machine);
int mtfjsp_load_instances_host(mtfjsp_handle_t h, const double *t_host, const double *p_host,
                               const double *tt_host, const int32_t *shop_host);
/* = Instance_Dataset generation (instance/generate_allsize_mofjsp_dataset.py:133-296) ON the device, straight into the
 * handle's instance arrays: same distributions (scope9 = t_low, t_high, p_low, p_high, weight_low,
 * weight_high, transT_in_low, transT_in_high, transT_out_high of instance/config_ins.json), Philox stream keyed by
 * (seed, first_instance + b) — distributional parity only, the legacy MT19937 stream stays in the host generator. */
int mtfjsp_generate_instances(mtfjsp_handle_t h, uint64_t seed, uint64_t first_instance, const double *scope9);
/* t, p [B,T,M] f64, tt [B,M,M] f64, shop [B,M] i32 of the loaded / generated instances, to host memory. */
int mtfjsp_read_instances_host(mtfjsp_handle_t h, double *t, double *p, double *tt, int32_t *shop);

/* = init_RewardScaling_sameBATCH (pe:70-85) and RewardScaling.reset() per episode (run:283-284). */
int mtfjsp_scaler_init(mtfjsp_handle_t h);
int mtfjsp_scaler_reset_returns(mtfjsp_handle_t h);
/* same, only for instances with mask_host[b] != 0 (run:283-284 resets them one by one) */
int mtfjsp_scaler_reset_returns_masked_host(mtfjsp_handle_t h, const uint8_t *mask_host);

/* ------------------------------------------------------------------ reset / step */
/* = init_DGFJSPEnv_state0 (pe:87-149) = env.reset() on every instance (env:1183-1245).
 * w3 [B,3]: normalised reward weights; the host draws them (env:1253-1259 uses python `random`). */
int mtfjsp_reset(mtfjsp_handle_t h, const double *w3);
/* = env.generate_random_weights("01") (env:1253-1259) for every instance ON the device: w3_out [B,3] f64 (device), three
 * uniforms normalised by their sum, Philox stream keyed by (seed, episode, instance) — distributional parity, for rollouts
 * that must not wait for the host; parity runs draw the weights with python `random` and pass them to mtfjsp_reset. */
int mtfjsp_draw_reward_weights(mtfjsp_handle_t h, uint64_t seed, uint64_t episode, double *w3_out);
/* One episode's start in ONE launch (the accelerated rollout issued three per episode): mtfjsp_draw_reward_weights(seed, episode) into
 * w3_out [B,3] (kept: the trajectory record and the next episode's observations carry the weights), mtfjsp_reset with them, and — with
 * reset_returns != 0 — mtfjsp_scaler_reset_returns (run:283-284: RewardScaling.reset() of every instance, pt:123).  Bit-identical to
 * the three calls. */
int mtfjsp_reset_episode(mtfjsp_handle_t h, uint64_t seed, uint64_t episode, double *w3_out, int32_t reset_returns);
int mtfjsp_reset_host(mtfjsp_handle_t h, const double *w3_host);

/* = DGFJSPEnv_paral_step (pe:217-268): env.step (env:716-974) + RewardScaling (pe:255-260), fused with
 * the candidate / job-mask update of ppo:202-316.  One launch; writes every bound obs field. */
int mtfjsp_step(mtfjsp_handle_t h, const int32_t *task_idx, const int32_t *mach_idx);
/* same launch, additionally recording this step's trajectory entries as f32 for the advantage computation
 * (SURVEY Appendix A rows 11,17-20): r4_out [4,B] = scaled mk, idle, pt, tt (pe:255-262 order) ; done_out [B]. */
int mtfjsp_step_record(mtfjsp_handle_t h, const int32_t *task_idx, const int32_t *mach_idx, float *r4_out, float *done_out);
/* The step as the TAIL of the launch that selects the machines (Run.py:363-427: machine actor forward, then env.step, nothing in
 * between): mtfjsp_step_params fills an opaque parameter block (mtfjsp_step_params_bytes() bytes) for exactly the step that
 * mtfjsp_step / mtfjsp_step_record (r4_out, done_out both given or both NULL) with these arguments would run, and returns 1; or
 * returns 0 when this handle's step cannot ride in another launch (shapes beyond the 16-instance register kernel, kernel-time
 * recording on, a diagnostic kernel override) and the caller must call mtfjsp_step itself; < 0 on errors.  The block is handed to
 * mtfjsp_encoder_arm_env_step before the mtfjsp_machine_actor_forward whose armed selection writes `mach_idx`. */
int32_t mtfjsp_step_params_bytes(void);
int mtfjsp_step_params(mtfjsp_handle_t h, const int32_t *task_idx, const int32_t *mach_idx, float *r4_out, float *done_out,
                       void *params_out, int32_t params_bytes);
/* host variant: returns MTFJSP_ERR_ACTION if any status word carries MTFJSP_ST_INVALID */
int mtfjsp_step_host(mtfjsp_handle_t h, const int32_t *task_idx_host, const int32_t *mach_idx_host);

/* = cal_cur_task_machine_feature (pe:152-214): out [B,M,6] obs_dtype, mmask_out [B,M] (1 = infeasible; may be
 * NULL).  mmask_in [B,M] may be NULL (then feasibility t>=0 is used, which is what run:335 passes). */
int mtfjsp_observe_mfea1(mtfjsp_handle_t h, const int32_t *task_idx, const uint8_t *mmask_in,
                         void *out, uint8_t *mmask_out);

/* Uniform random valid actions on device (benchmark / property-test policy, SURVEY §8d "on-device Philox allowed
 * for perf runs"): job uniform over unmasked jobs of the bound job_mask, machine uniform over feasible ones. */
int mtfjsp_random_actions(mtfjsp_handle_t h, uint64_t seed, uint64_t counter, int32_t *task_idx,
                          int32_t *mach_idx, int32_t *job_idx);

/* ------------------------------------------------------------------ exports (compat / tests) */
/* dense adj_wrk [B,T,T] f64, row = destination, diagonal 1 (env:2066-2073) — what pe:136 returns. */
int mtfjsp_export_dense_adj(mtfjsp_handle_t h, double *out);
/* same, into HOST memory (the reference returns adj as a host numpy array, pe:136,263); the [B,T,T] device scratch is
 * allocated inside the handle on first use — compatibility path for small batches */
int mtfjsp_export_dense_adj_host(mtfjsp_handle_t h, double *out_host);
/* gym-style valid_action_mask (env:2535-2575): [B,T] 1 = selectable */
int mtfjsp_valid_action_mask(mtfjsp_handle_t h, uint8_t *out);
enum {
    MTFJSP_STATE_MACHINE = 0,   /* int32 [B,T]   machine of task, -1 unscheduled       (G.nodes[k]['machine'])      */
    MTFJSP_STATE_START = 1,     /* f64   [B,T]   start time, NaN unscheduled           (…['start_time'])            */
    MTFJSP_STATE_FINISH = 2,    /* f64   [B,T]   finish time, NaN unscheduled          (…['finish_time'], ppo:271)  */
    MTFJSP_STATE_ROUTES = 3,    /* int32 [B,M,T] task indices in processing order, -1 padded (env.machine_routes)   */
    MTFJSP_STATE_PREV_COSTS = 4,/* f64   [B,4]   makespan/e1/trans/idle _previous_step  (run:632-633)               */
    MTFJSP_STATE_SCALER = 5,    /* f64   [B,17]  R[4], n, mean[4], S[4], std[4]         (pt:54-124)                 */
    MTFJSP_STATE_W3 = 6         /* f64   [B,3]   reward_random_weight                   (run:478)                   */
};
int mtfjsp_read_state_host(mtfjsp_handle_t h, int which, void *out_host);
/* writes back the RewardScaling state (MTFJSP_STATE_SCALER layout, [count,17]) of instances [first, first+count): lets a caller
 * drive ONE instance through mtfjsp_step the way the reference's gym-style env.step does (env:716-974 — no reward scaling;
 * only the batched step applies it, pe:255-260) by restoring that instance's scaler afterwards */
int mtfjsp_set_scaler_state_host(mtfjsp_handle_t h, int32_t first, int32_t count, const double *state17);
/* copy `nbytes` from a device buffer of this handle's context to host (synchronises the stream) */
int mtfjsp_copy_to_host(mtfjsp_handle_t h, void *dst_host, const void *src_dev, size_t nbytes);

/* GAE reverse scan of ppo:438-536 for one reward channel: for every instance b, s = S-1..0:
 *   delta = r[s,b] + gamma*v_next[s,b] - v[s,b] ; gae = delta + gamma*lambda*gae*(1-done[s,b]) ; adv[s,b] = gae
 * r, v, v_next are f32 with element strides (stride_s, stride_b) so packed trajectory buffers can be passed as views;
 * done and adv are [S,B] contiguous.  The result is NOT normalised (that needs the all-gather across GPUs). */
int mtfjsp_gae(mtfjsp_handle_t h, int32_t S, const float *r, int64_t r_ss, int64_t r_sb, const float *v, int64_t v_ss, int64_t v_sb,
               const float *v_next, int64_t n_ss, int64_t n_sb, const float *done, float gamma, float lambda, float *adv);
/* Global advantage normalisation of the hand-off (ppo:485,532): `(adv - adv.mean()) / (adv.std() + 1e-5)` with torch's unbiased
 * std, per tensor over ALL shards' columns.  gathered = [world][K_total][S][B] f32 — what ONE all-gather of the packed
 * [K_total,S,B] buffer leaves on every rank (world = 1: the packed buffer itself); the first K <= K_total <= 16 tensors are
 * advantages to normalise, the others (the value tensors of the whole hand-off, ppo:628-703) only ride along.  norm_out [K][S][B]
 * receives this rank's normalised advantages; targets_out (may be NULL) [K][S][B] = normalised advantage + value at act time
 * (ppo:668-671,689) with values[k] an f32 [S,B] view given by element strides (NULL entries are skipped); full_out (may be NULL)
 * [K_total][S][world*B] receives every gathered tensor in the reference's single-process layout (rank-major column blocks).
 * Two launches on the handle's stream; f64 statistics from per-block partial sums in a fixed order: bit-reproducible. */
int mtfjsp_normalize_advantages(mtfjsp_handle_t h, int32_t K, int32_t K_total, int32_t world, int32_t rank, int32_t S, const float *gathered, float eps,
                                const float *const *values, const int64_t *value_stride_s, const int64_t *value_stride_b,
                                float *norm_out, float *targets_out, float *full_out);
/* K strided f32 [S,B] views -> one packed [K][S][B] buffer in one launch (the value tensors that ride in the hand-off's all-gather
 * beside the advantages; the reference stacks them with torch.cat, ppo:657-703) */
int mtfjsp_pack_views(mtfjsp_handle_t h, int32_t K, int32_t S, const float *const *src, const int64_t *stride_s, const int64_t *stride_b, float *out);

/* kernel timing hook for bench.py: HIP events recorded on the handle's stream around every step launch
 * between begin/end; returns accumulated milliseconds and launch count. */
int mtfjsp_timing_begin(mtfjsp_handle_t h);
int mtfjsp_timing_end(mtfjsp_handle_t h, double *step_ms_total, int64_t *step_launches);

/* measurement only (the step kernel's achieved bytes/s are to be compared with "the measured copy bandwidth of a
 * same-footprint streaming kernel on the same GPU"): `reps` launches of a kernel that reads read_bytes and writes write_bytes
 * with access_bytes-wide (4, 8 or 16), fully coalesced accesses on `grid` workgroups of 256 threads, HIP events around every
 * launch on the handle's stream; returns the average and the minimum launch duration in microseconds.  Scratch buffers are
 * allocated and freed inside the call.  Replaces nothing in the reference (no reference counterpart). */
int mtfjsp_footprint_copy(mtfjsp_handle_t h, size_t read_bytes, size_t write_bytes, int32_t access_bytes, int32_t grid, int32_t reps,
                          double *avg_us_out, double *min_us_out);

/* ------------------------------------------------------------------ encoder (rollout forward passes) */
typedef struct {
    int32_t n_job, n_machine, batch;
    int32_t hidden;            /* gcn_hidden_dim == machine_hidden_dim == 128 (parameters.py:106,109) */
    int32_t obs_dtype;         /* dtype of tasks_fea / m_fea1 / m_fea2 handed to the forwards */
    int32_t device_id;
} mtfjsp_encoder_config_t;

int mtfjsp_encoder_create(const mtfjsp_encoder_config_t *cfg, mtfjsp_encoder_t *out);
int mtfjsp_encoder_destroy(mtfjsp_encoder_t e);
const char *mtfjsp_encoder_last_error(mtfjsp_encoder_t e);
int mtfjsp_encoder_set_stream(mtfjsp_encoder_t e, void *hip_stream);
/* Weights by the reference's state_dict key, prefixed "job_actor." or "machine_actor."
 * (e.g. "job_actor.encoder.feature_extract.mlps.0.linears.0.weight"); f32, row-major as in torch. */
int mtfjsp_encoder_load_weight_host(mtfjsp_encoder_t e, const char *name, const float *data_host, int64_t numel);
int mtfjsp_encoder_weights_ready(mtfjsp_encoder_t e);     /* 0 when every required tensor has been loaded */

/* = Operation_Actor_JointAction_selfCritic.forward (ac:104-296) without the sampling:
 * GIN encoder (gcn:109-197, training-mode BatchNorm over all B*T rows), candidate scorer, masked softmax,
 * local critic.  h_m_prev may be NULL (first step: learned `_input`, ac:229-233).
 * Outputs (f32): prob [B,J], h_pooled [B,H], job_v [B,2]; h_nodes [B*T,H] optional (NULL = keep internal).
 * candidate [B,J]: any task row in [0, T) per job (ac:197-207 gathers by index).  The environment's candidates lie in their job's
 * own block of rows, candidate[b][j] in [j M, (j+1) M) (ppo:306-309), which the streaming pool/gather kernel of the large shapes
 * exploits (it picks them out of the row stream); a candidate outside its block takes a dependent gather there — same result. */
int mtfjsp_job_actor_forward(mtfjsp_encoder_t e, const void *tasks_fea, const int32_t *ell_col,
                             const float *ell_val, const int32_t *candidate, const uint8_t *job_mask,
                             const float *h_m_prev, float *prob, float *h_pooled, float *job_v, float *h_nodes);
/* = Machine_Actor_JointAction_selfGAT_selfCritic.forward (ac:359-498; GATLayer gat:82-159).
 * Outputs (f32): prob [B,M], h_pooled [B,H], machine_v [B,2]. */
int mtfjsp_machine_actor_forward(mtfjsp_encoder_t e, const void *m_fea1, const void *m_fea2,
                                 const float *h_pooled_o, const uint8_t *mmask, float *prob, float *h_pooled,
                                 float *machine_v);
/* = Global_Critic_JointAction_GAT.forward (ac:587-750; SURVEY §8f N1, used under no_grad by ppo:628-703 to sample the
 * global values for GAE): weights under the prefix "global_critic." (loaded with mtfjsp_encoder_load_weight_host);
 * value4 [B,4] f32 = (mk, pt, tt, it). */
int mtfjsp_global_critic_forward(mtfjsp_encoder_t e, const void *tasks_fea, const int32_t *ell_col, const float *ell_val,
                                 const void *m_fea1, const void *m_fea2, float *value4);
/* = select_operation_action / greedy_select_action / select_machine_action (agent:22-72):
 * categorical sample (Philox, (seed,counter)) or argmax from prob [B,N]; idx_out [B], logp_out [B];
 * gather_from (optional, [B,N] int32, e.g. candidate) -> gathered_out [B] (task index). */
int mtfjsp_sample_categorical(mtfjsp_encoder_t e, const float *prob, int32_t n, int32_t greedy, uint64_t seed,
                              uint64_t counter, int32_t *idx_out, float *logp_out, const int32_t *gather_from,
                              int32_t *gathered_out);
/* Device-side recipe of Parallel_env.cal_cur_task_machine_feature (pe:152-214) for ONE decision: everything
 * mtfjsp_observe_mfea1 reads and writes, as plain device pointers, so that the job actor's heads kernel can produce
 * m_fea1 / the machine mask for the task it has just selected (one launch less between the two actor forwards).
 * Filled by mtfjsp_get_mfea1_context (environment side), consumed by mtfjsp_encoder_arm_mfea1 (encoder side). */
typedef struct {
    const double *t, *p, *tt, *mean3;   /* [B,T,M], [B,T,M], [B,M,M], [B,T,3] */
    const int32_t *shop;                /* [B,M] */
    const void *link;                   /* [B,T] 16-byte task records {f64 energy estimate; i16 machine, prev, pos, next}: the machine of task v is the i16 at byte 16 v + 8 */
    void *m_fea1_out;                   /* [B,M,6] obs dtype */
    uint8_t *mmask_out;                 /* [B,M] */
    int32_t T, M, obs_f32;
    const void *m_fea2;                 /* [B,M,8] obs dtype: the environment's bound machine features (may be NULL).  With it the
                                         * job actor's heads launch can also run the machine actor's GAT passes on (m_fea1_out, m_fea2)
                                         * — the mtfjsp_machine_actor_forward that follows with exactly these two pointers skips them */
} mtfjsp_mfea1_ctx_t;
int mtfjsp_get_mfea1_context(mtfjsp_handle_t h, void *m_fea1_out, uint8_t *mmask_out, mtfjsp_mfea1_ctx_t *ctx);
/* The WHOLE machine actor forward (ac:359-498) inside the job actor's heads launch: three launches per rollout step.  With these
 * outputs armed (prob [B,M], h_pooled [B,H], machine_v [B,2]; one-shot), a machine selection armed (mtfjsp_encoder_arm_selection,
 * which = 1) and an mfea1 context with m_fea2 (above), the NEXT mtfjsp_job_actor_forward — where the shape allows it: one workgroup
 * of 16 instances per CU, all co-resident (the single-launch GIN kernel's census), split products — also runs the machine path's
 * GAT passes, exchanges the BatchNorm sums of the B*M machine nodes (ac:434) between its workgroups inside the launch
 * (count-carrying integer atomics as in the single-launch GIN kernel; the wait is bounded, a time-out surfaces as
 * MTFJSP_ERR_RETRY) and the machine heads with their selection.  The mtfjsp_machine_actor_forward that follows with exactly the
 * pointers involved (m_fea1_out, m_fea2, that job forward's h_pooled, mmask_out, these outputs) finds itself done and returns at
 * once; with any other argument it recomputes (and selects again with the machine selection the launch consumed).  The match is by
 * POINTER IDENTITY, as for m_fea2: a caller that rewrites m_fea1 / the machine mask IN PLACE between the two forwards (a forced task
 * through mtfjsp_observe_mfea1 into the same buffers) must not arm the machine heads for that step.  The mode setters
 * (set_bn_mode, set_stats_reduce, set_product_mode) and mtfjsp_global_critic_forward discard a forward done ahead.
 * Where the shape does not allow it nothing changes.  MTFJSP_NO_FUSED_MHEADS=1: off.
 * mtfjsp_encoder_fused_launches: how many forwards took the three-in-one launch so far (tests, bench). */
int mtfjsp_encoder_arm_machine_heads(mtfjsp_encoder_t e, float *prob, float *h_pooled, float *machine_v);
int mtfjsp_encoder_fused_launches(mtfjsp_encoder_t e, int64_t *three_in_one_out);
/* The post-terminal forward pair of an episode (Run.py:455-475: job actor on the terminal observation, machine actor on the last
 * decision's m_fea1) keeps only the two local critics' values (replaybuffer.py:131-139).  Armed with this call, the NEXT
 * mtfjsp_job_actor_forward and the NEXT mtfjsp_machine_actor_forward write job_v / machine_v and the pooled embeddings as always —
 * bit-identical — and skip the scorer: prob (and any selection output) is left untouched.  One-shot; ignored (full forward) for a
 * forward with an armed selection / m_fea1 context, with per-instance BatchNorm or with the f32-instruction heads. */
int mtfjsp_encoder_arm_values_only(mtfjsp_encoder_t e);
/* BatchNorm statistics of the two actor forwards (every BatchNorm in the reference is in training mode):
 * per_instance = 0 (default): over all rows of the device batch = one reference run with env_batch = B (training rollout);
 * per_instance = 1: over the rows of ONE instance = B independent reference runs with env_batch = 1, i.e. the greedy
 * evaluation of validate.py:60-297 batched over the evaluation set.  Not applied to the global critic. */
int mtfjsp_encoder_set_bn_mode(mtfjsp_encoder_t e, int32_t per_instance);
/* Exact big-batch BatchNorm over several shards (optional): with a reduction callback set, every BatchNorm of the
 * forwards normalises over the rows of ALL shards.  After the launch that completes a BatchNorm's column sums the library
 * synchronises its stream and calls fn(user, sums, count): `sums` is DEVICE memory holding `count` doubles, which the callback
 * replaces by their element-wise sum over the shards (e.g. one all-reduce) and returns 0 once that is in place.
 * global_batch = instances of all shards together (>= this handle's batch).  7 calls per actor pair forward; the GIN encoder
 * then runs as its streaming launches (the single-launch kernel cannot exchange data mid-launch).  fn = NULL: off.  Per-shard
 * statistics (the default) are what DESIGN.md §7 describes: each GPU behaves like a reference run with env_batch = its shard. */
typedef int (*mtfjsp_stats_reduce_fn)(void *user, double *sums, int32_t count);
int mtfjsp_encoder_set_stats_reduce(mtfjsp_encoder_t e, mtfjsp_stats_reduce_fn fn, void *user, int64_t global_batch);
/* deferred = 1: the forward entries stop polling the asynchronous failure words (see mtfjsp_encoder_check); a failure then
 * surfaces only at mtfjsp_encoder_check.  For callers whose forwards contain collectives (the reduction callback above): every
 * shard must issue the same sequence of forwards, so a shard-local MTFJSP_ERR_RETRY may only be acted upon at a point all shards
 * agree on — rollout.py checks once per step and lets the shards agree through one MAX all-reduce (the reference has no
 * counterpart: its update samples the critics in one process, ppo:422-435).  Default 0.
 * Diagnostic: MTFJSP_RANGE_FAIL_AT=n raises the range word at the n-th job-actor forward of a handle (tests). */
int mtfjsp_encoder_set_deferred_poll(mtfjsp_encoder_t e, int32_t deferred);
/* How the [rows,128]x[128,128] products of the actor forwards (Linear layers of gcn:95-153 / ac:205-293, the GAT weight of
 * gat:82) are formed.  0 (default): on the 16-bit matrix cores with f32 accumulation, from f32 operands split into two f16
 * pieces (relative representation error <= 2^-22; weights pre-scaled by a power of two) and the three significant piece
 * products — measured as accurate as an f32 FMA chain of the same length (tests/test_encoder_hip.py states
 * the bounds); the 12 -> 128 first Linear, whose inputs are raw features, uses an exact three-piece bf16 split instead.
 * Bits select the f32 matrix instruction instead, as the A/B reference: 1 = GIN products, 2 = GAT passes, 4 = actor/critic
 * heads, 8 = first GIN Linear (12 -> 128) on the vector ALU.  Bit 16 is not a numerics choice: it runs the GIN encoder as six
 * streaming launches (k_gemm_x6) even where the single-launch register-resident kernel (k_gin_res; in-kernel grid-wide statistics exchanges for
 * the batch statistics) is eligible (16 <= T <= 65, <= 576 node rows per CU, census launch passed in mtfjsp_encoder_create);
 * MTFJSP_NO_RESIDENT_GIN=1 in the environment does the same for every handle. */
int mtfjsp_encoder_set_product_mode(mtfjsp_encoder_t e, int32_t f32_instruction_mask);
/* Fuse the action selection of the NEXT mtfjsp_job_actor_forward (which = 0) / mtfjsp_machine_actor_forward (which = 1) call into
 * its heads kernel: same arguments and the same Philox stream as mtfjsp_sample_categorical on that forward's `prob`
 * (agent:22-72), one launch less per decision.  One-shot: applies to one forward call. */
int mtfjsp_encoder_arm_selection(mtfjsp_encoder_t e, int32_t which, int32_t greedy, uint64_t seed, uint64_t counter,
                                 int32_t *idx_out, float *logp_out, const int32_t *gather_from, int32_t *gathered_out);
/* One-shot like mtfjsp_encoder_arm_selection (and only together with it, which = 0): the next job actor forward also
 * writes m_fea1 / the machine mask of every instance's selected task (== mtfjsp_observe_mfea1 on gathered_out). */
int mtfjsp_encoder_arm_mfea1(mtfjsp_encoder_t e, const mtfjsp_mfea1_ctx_t *ctx);
/* One-shot: the NEXT mtfjsp_machine_actor_forward also runs the environment step described by `params` (mtfjsp_step_params) in its
 * heads launch — one launch and one launch boundary less per rollout step — provided its armed selection (which = 1) writes the
 * `mach_idx` the block names and the shapes agree.  mtfjsp_encoder_env_step_fused() tells afterwards whether it did (1) or the
 * caller still has to call mtfjsp_step / mtfjsp_step_record (0: the default — the combined launch is enabled with the environment
 * variable MTFJSP_FUSED_ENV when the handle is created, because it measured slower than two launches at the headline shape —, also
 * per-instance BatchNorm mode, f32-instruction heads, kernel-time recording).  Results are those of mtfjsp_step, bit for bit (the
 * same device code). */
int mtfjsp_encoder_arm_env_step(mtfjsp_encoder_t e, const void *params, int32_t params_bytes);
int mtfjsp_encoder_env_step_fused(mtfjsp_encoder_t e);
/* ---- host-side helpers of the instance generator (instance/generate_allsize_mofjsp_dataset.py:204-216, 241-272).
 * The reference draws "k machines infeasible per task" and the transport times from numpy's legacy RandomState one python call at
 * a time; these take the same draws from the same MT19937 state (key[624] + *pos as RandomState.get_state() returns them; updated
 * in place for set_state()) at native speed, bit for bit, and keep only samples [first, first+count): t [count,T,M] gets the sign
 * flips, tt [count,M,M] is written.  No GPU involved. */
int mtfjsp_hostgen_infeasible(uint32_t *key, int32_t *pos, int64_t samples, int32_t T, int32_t M, int64_t first, int64_t count, double *t);
int mtfjsp_hostgen_transport(uint32_t *key, int32_t *pos, int64_t samples, int32_t M, const int64_t *shop_of_machine,
                             double in_lo, double in_hi, double out_hi, int64_t first, int64_t count, double *tt);
/* Synchronises the encoder's stream and reports asynchronous failures of the forwards enqueued so far.  The single-launch GIN
 * kernel exchanges its BatchNorm statistics between its workgroups inside the launch (count-carrying integer atomics; the waits
 * for the other workgroups' contributions are bounded at 4 ms); it needs all of its workgroups
 * resident at once (one per CU), which another process or another stream using the same GPU can prevent — not a hang but a
 * time-out.  The kernel then sets a host-mapped word that EVERY job-actor / global-critic forward polls on entry (a plain host
 * read, no synchronisation) and that this call reads after synchronising: the first call to see it returns MTFJSP_ERR_RETRY,
 * switches the handle to the streaming launches (slower, no co-residency requirement) and leaves the caller to recompute what
 * it enqueued since the failed launch (the Python rollout restarts the episode and discards the trajectory buffer).  A later
 * mtfjsp_encoder_check re-runs the residency census on the idle stream and re-enables the single launch when it passes.
 * *gin_resident_out (may be NULL) = 1 when the single-launch kernel is in use for this handle (shape eligible, census passed in
 * mtfjsp_encoder_create or here, product-mode bits 1, 8, 16 clear), 0 when the streaming launches are.  Only one process / stream
 * per GPU should use the single-launch path at a time (use one stream, or order handles with events).
 * Diagnostic: MTFJSP_GIN_RES_FAIL_AT=n makes the n-th single-launch forward of a handle time out (tests). */
int mtfjsp_encoder_check(mtfjsp_encoder_t e, int32_t *gin_resident_out);
/* Range of the split products.  The default kernels run every 128-deep product on the f16 matrix cores from operands split into
 * two f16 pieces: f32-accurate, but an activation beyond 65 504 (a BatchNorm gamma of several hundred, an edge weight of several
 * thousand — nothing a trained reference checkpoint produces, nothing the reference's f32/f64 arithmetic forbids) cannot be
 * represented.  It is never clamped and never silent: its pieces are (inf | -inf), their products NaN, the NaN reaches every output
 * of that forward, the heads kernel sets a host-mapped word, and the next forward entry (or mtfjsp_encoder_check) returns
 * MTFJSP_ERR_RETRY after switching the handle to the f32-instruction kernels (product mode 15, no range limit, slower): repeat
 * the forward.  mtfjsp_encoder_set_product_mode(0) returns to the split products.  *count_out = number of such switches,
 * *product_mode_out (may be NULL) = the product mode now in force. */
int mtfjsp_encoder_range_fallbacks(mtfjsp_encoder_t e, int64_t *count_out, int32_t *product_mode_out);
/* diagnostic (tools/first_launch): the first `count` floats of the machine path's node buffer [B*M,128] — the output of the three
 * GAT passes of ac:409-420 before the BatchNorm of ac:434 — copied to host memory after synchronising the stream.  The buffer is
 * written by the separate GAT launches (k_gat3x, k_headsx_gat3x) and by the global critic; the three-in-one launch keeps its node
 * rows in LDS (round 6) and leaves the buffer alone unless MTFJSP_FUSED3_NODES_HBM=1. */
int mtfjsp_encoder_peek_nodes_host(mtfjsp_encoder_t e, float *out_host, int64_t count);
/* number of statistics-exchange time-outs of the single-launch kernels reported on this handle so far */
int mtfjsp_encoder_resident_failures(mtfjsp_encoder_t e, int64_t *count_out);
int mtfjsp_encoder_timing_begin(mtfjsp_encoder_t e);
int mtfjsp_encoder_timing_end(mtfjsp_encoder_t e, double *ms_total, int64_t *launches);
/* per kernel family (between begin and the next begin): "gin0_agg_linear12" (or, round 6, "gin0_moments" / "gin0_stats_only" + "gin0_bn_gemm":
 * the first Linear's output formed by the second launch's producers instead of stored), "gin_gemm_bn_relu", "gin_gemm_agg",
 * "job_pool_gather", "heads", "head_gemm", "gat3", "mach_bn_pool", "sample", "small", "gin_inst", "gat_inst", "gin_resident" */
int mtfjsp_encoder_timing_query(mtfjsp_encoder_t e, const char *family, double *ms_total, int64_t *launches);

#ifdef __cplusplus
}
#endif
#endif /* MTFJSP_H */
