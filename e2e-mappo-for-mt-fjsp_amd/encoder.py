"""Host side of the encoder half of the C ABI: the job actor (GIN + candidate scorer + local critic) and the
machine actor (GAT + scorer + local critic) of model/actor_critic.py as HIP kernels (csrc/mtfjsp_encoder.hip).

Weights are addressed by the reference's state_dict key names, so the shipped checkpoints
(trained_model/can_use/*/PPO_{job,machine}_actor_*.pth -> torch.load -> dict) load unchanged.
"""
import ctypes as C
import math

import os

import numpy as np
import torch

from . import capi

H = 128
FAMILIES = ["gin0_agg_linear12", "gin0_stats_only", "gin0_moments", "gin0_bn_gemm", "gin_gemm_bn_relu", "gin_gemm_stats_only", "gin_gemm_pool", "cand_fixup", "gin_gemm_pair", "gin_gemm_agg", "job_pool_gather", "heads", "head_gemm", "gat3", "mach_bn_pool",
            "sample", "small", "gin_inst", "gat_inst", "gin_resident", "heads_gat3", "heads_gat3_heads"]


def available():
    try:
        capi.lib()
        return torch.cuda.is_available()
    except Exception:
        return False


def random_init_weights(seed=0, with_critic=False):
    """Random weights with the reference modules' default initialisers (nn.Linear: U(+-1/sqrt(in)); BatchNorm: 1/0;
    `_input`: U(-1,1) ac:76-77; GAT W, a: xavier normal gat:60-66) under the reference's state_dict key names.
    -> (job actor, machine actor[, global critic (ac:506-586: own GIN encoder, own GAT path, MLPCritic(256 -> 128 -> 128 -> 4))])"""
    g = torch.Generator().manual_seed(seed)

    def lin(out, inp, bias=True):
        b = 1.0 / math.sqrt(inp)
        w = (torch.rand(out, inp, generator=g) * 2 - 1) * b
        return (w, (torch.rand(out, generator=g) * 2 - 1) * b) if bias else (w, None)

    ja, ma = {}, {}
    pre = "encoder.feature_extract."
    for l, inp in ((0, 12), (1, H)):
        for i, (o, k) in enumerate(((H, inp), (H, H), (H, H))):
            w, b = lin(o, k)
            ja[f"{pre}mlps.{l}.linears.{i}.weight"], ja[f"{pre}mlps.{l}.linears.{i}.bias"] = w, b
        for i in range(2):
            ja[f"{pre}mlps.{l}.batch_norms.{i}.weight"] = torch.ones(H); ja[f"{pre}mlps.{l}.batch_norms.{i}.bias"] = torch.zeros(H)
        ja[f"{pre}batch_norms.{l}.weight"] = torch.ones(H); ja[f"{pre}batch_norms.{l}.bias"] = torch.zeros(H)
    ja["_input"] = torch.rand(H, generator=g) * 2 - 1
    for name, d, out_last in (("o_policy", ja, 1), ("job_critic", ja, 2), ("m_policy", ma, 1), ("machine_critic", ma, 2)):
        first_in = 3 * H if "policy" in name else H
        for i, (o, k) in enumerate(((H, first_in), (H, H), (out_last, H))):
            w, b = lin(o, k)
            d[f"{name}.linears.{i}.weight"], d[f"{name}.linears.{i}.bias"] = w, b
    ma["bn.weight"] = torch.ones(H); ma["bn.bias"] = torch.zeros(H)
    ma["m_fea_1_fcl.weight"] = lin(H, 6, False)[0]; ma["m_fea_2_fcl.weight"] = lin(H, 8, False)[0]
    ma["gat_layer.W"] = torch.randn(H, H, generator=g) * math.sqrt(2.0 / (H + H))
    ma["gat_layer.a"] = torch.randn(1, 2 * H, 1, generator=g) * math.sqrt(2.0 / (2 * H + 1 * 1))
    if not with_critic:
        return {k: v.numpy() for k, v in ja.items()}, {k: v.numpy() for k, v in ma.items()}
    gc = {}
    for l, inp in ((0, 12), (1, H)):
        for i, (o, k) in enumerate(((H, inp), (H, H), (H, H))):
            w, b = lin(o, k)
            gc[f"{pre}mlps.{l}.linears.{i}.weight"], gc[f"{pre}mlps.{l}.linears.{i}.bias"] = w, b
        for i in range(2):
            gc[f"{pre}mlps.{l}.batch_norms.{i}.weight"] = torch.ones(H); gc[f"{pre}mlps.{l}.batch_norms.{i}.bias"] = torch.zeros(H)
        gc[f"{pre}batch_norms.{l}.weight"] = torch.ones(H); gc[f"{pre}batch_norms.{l}.bias"] = torch.zeros(H)
    for i, (o, k) in enumerate(((H, 2 * H), (H, H), (4, H))):
        w, b = lin(o, k)
        gc[f"critic.linears.{i}.weight"], gc[f"critic.linears.{i}.bias"] = w, b
    gc["bn.weight"] = torch.ones(H); gc["bn.bias"] = torch.zeros(H)
    gc["m_fea_1_fcl.weight"] = lin(H, 6, False)[0]; gc["m_fea_2_fcl.weight"] = lin(H, 8, False)[0]
    gc["gat_layer.W"] = torch.randn(H, H, generator=g) * math.sqrt(2.0 / (H + H))
    gc["gat_layer.a"] = torch.randn(1, 2 * H, 1, generator=g) * math.sqrt(2.0 / (2 * H + 1 * 1))
    return tuple({k: v.numpy() for k, v in d.items()} for d in (ja, ma, gc))


class Encoder:
    def __init__(self, n_job, n_machine, batch, device=0, obs_dtype="f32"):
        if not torch.cuda.is_available():
            raise RuntimeError("the encoder needs a GPU (MI355X); there is no CPU fallback")
        self.L = capi.lib()
        self.J, self.M, self.B = n_job, n_machine, batch
        self.T = n_job * n_machine
        self.device = torch.device("cuda", device)
        self.obs_f32 = obs_dtype in ("f32", torch.float32, np.float32)
        cfg = capi.EncoderConfig(n_job, n_machine, batch, H, capi.OBS_F32 if self.obs_f32 else capi.OBS_F64, device)
        h = C.c_void_p()
        capi.check(self.L.mtfjsp_encoder_create(C.byref(cfg), C.byref(h)), None, enc=True)
        self.h = h
        B, J, M = batch, n_job, n_machine
        f32 = dict(dtype=torch.float32, device=self.device)
        self.job_prob = torch.zeros(B, J, **f32); self.h_pooled_o = torch.zeros(B, H, **f32); self.job_v = torch.zeros(B, 2, **f32)
        self.mch_prob = torch.zeros(B, M, **f32); self.h_pooled_m = torch.zeros(B, H, **f32); self.mach_v = torch.zeros(B, 2, **f32)
        self.use_current_stream()

    def use_current_stream(self):
        s = torch.cuda.current_stream(self.device).cuda_stream
        capi.check(self.L.mtfjsp_encoder_set_stream(self.h, C.c_void_p(s)), self.h, enc=True)

    def close(self):
        if getattr(self, "h", None):
            self.L.mtfjsp_encoder_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_weights(self, job_actor, machine_actor, global_critic=None):
        """dicts keyed by the reference's state_dict names (numpy arrays or torch tensors)."""
        for prefix, d in (("job_actor.", job_actor), ("machine_actor.", machine_actor), ("global_critic.", global_critic or {})):
            for k, v in d.items():
                if "running_" in k or "num_batches_tracked" in k:
                    continue
                a = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
                a = np.ascontiguousarray(a, np.float32)
                capi.check(self.L.mtfjsp_encoder_load_weight_host(self.h, (prefix + k).encode(), a.ctypes.data, a.size), self.h, enc=True)
        capi.check(self.L.mtfjsp_encoder_weights_ready(self.h), self.h, enc=True)

    def job_actor_forward(self, tasks_fea, ell_col, ell_val, candidate, job_mask, h_m_prev=None, h_nodes=None, v_out=None):
        """-> (prob [B,J], h_pooled [B,H], job_v [B,2]) device f32 tensors (owned by this object)."""
        capi.check(self.L.mtfjsp_job_actor_forward(
            self.h, tasks_fea.data_ptr(), ell_col.data_ptr(), ell_val.data_ptr(), candidate.data_ptr(), job_mask.data_ptr(),
            h_m_prev.data_ptr() if h_m_prev is not None else 0, self.job_prob.data_ptr(), self.h_pooled_o.data_ptr(),
            (v_out if v_out is not None else self.job_v).data_ptr(), h_nodes.data_ptr() if h_nodes is not None else 0), self.h, enc=True)
        return self.job_prob, self.h_pooled_o, (v_out if v_out is not None else self.job_v)

    def machine_actor_forward(self, m_fea1, m_fea2, h_pooled_o, mmask, v_out=None):
        capi.check(self.L.mtfjsp_machine_actor_forward(
            self.h, m_fea1.data_ptr(), m_fea2.data_ptr(), h_pooled_o.data_ptr(), mmask.data_ptr(), self.mch_prob.data_ptr(),
            self.h_pooled_m.data_ptr(), (v_out if v_out is not None else self.mach_v).data_ptr()), self.h, enc=True)
        return self.mch_prob, self.h_pooled_m, (v_out if v_out is not None else self.mach_v)

    def global_critic_forward(self, tasks_fea, ell_col, ell_val, m_fea1, m_fea2, out=None):
        """-> value [B,4] f32 (mk, pt, tt, it) of the reference's Global_Critic_JointAction_GAT (needs its weights loaded)"""
        if out is None:
            out = torch.empty(self.B, 4, dtype=torch.float32, device=self.device)
        capi.check(self.L.mtfjsp_global_critic_forward(self.h, tasks_fea.data_ptr(), ell_col.data_ptr(), ell_val.data_ptr(),
                                                       m_fea1.data_ptr(), m_fea2.data_ptr(), out.data_ptr()), self.h, enc=True)
        return out

    def sample(self, prob, greedy, seed, counter, idx_out, logp_out=None, gather_from=None, gathered_out=None):
        capi.check(self.L.mtfjsp_sample_categorical(
            self.h, prob.data_ptr(), prob.shape[1], int(greedy), seed, counter, idx_out.data_ptr(),
            logp_out.data_ptr() if logp_out is not None else 0, gather_from.data_ptr() if gather_from is not None else 0,
            gathered_out.data_ptr() if gathered_out is not None else 0), self.h, enc=True)

    def set_bn_mode(self, per_instance):
        """per_instance=True: every BatchNorm of the actor forwards normalises over the rows of one instance (= the reference's
        greedy evaluation with env_batch 1, validate.py:60-297); False (default): over the whole device batch"""
        capi.check(self.L.mtfjsp_encoder_set_bn_mode(self.h, 1 if per_instance else 0), self.h, enc=True)

    def set_product_mode(self, f32_instruction_mask=0):
        """0 (default): 128x128 products on the 16-bit matrix cores from split f32 operands (f32-accurate: include/mtfjsp.h);
        bits select the f32 matrix instruction instead (1 GIN products, 2 GAT passes, 4 heads, 8 first GIN Linear on the VALU) —
        the A/B reference; 16 (not a numerics choice): the GIN encoder as its streaming launches even where the register-resident
        single-launch kernel is eligible (check() tells which one runs)"""
        capi.check(self.L.mtfjsp_encoder_set_product_mode(self.h, int(f32_instruction_mask)), self.h, enc=True)

    def set_stats_reduce(self, fn, global_batch=0):
        """exact big-batch BatchNorm over several shards (include/mtfjsp.h): fn(ptr, count) replaces the `count` f64 values at
        DEVICE address ptr by their sum over the shards (dist.bn_stats_allreduce() builds one on torch.distributed);
        global_batch = instances of all shards.  fn = None switches it off."""
        if fn is None:
            capi.check(self.L.mtfjsp_encoder_set_stats_reduce(self.h, None, None, 0), self.h, enc=True)
            self._reduce_cb = None
            return

        def tramp(user, ptr, n):
            try:
                fn(int(ptr), int(n))
                return 0
            except Exception:                                       # a Python exception must not unwind through the C frames
                import traceback
                traceback.print_exc()
                return 1

        self._reduce_cb = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int32)(tramp)
        capi.check(self.L.mtfjsp_encoder_set_stats_reduce(self.h, C.cast(self._reduce_cb, C.c_void_p), None, int(global_batch)), self.h, enc=True)

    def set_deferred_poll(self, deferred):
        """True: the forward entries stop polling the asynchronous failure words; only check() reports a failure (for callers whose
        forwards contain collectives and must stay aligned across ranks: Rollout(exact_bn=True))"""
        capi.check(self.L.mtfjsp_encoder_set_deferred_poll(self.h, 1 if deferred else 0), self.h, enc=True)

    def check(self):
        """synchronise and raise if a forward failed asynchronously (a bounded wait of the single-launch GIN kernel's grid-wide statistics exchange timed out:
        MtfjspError with code capi.ERR_RETRY — the handle has then switched to the streaming launches and whatever was enqueued
        since the failed launch has to be recomputed; the forwards poll the same condition on entry without synchronising);
        -> True when that kernel is in use for this shape, False when the streaming launches are"""
        r = C.c_int32(0)
        capi.check(self.L.mtfjsp_encoder_check(self.h, C.byref(r)), self.h, enc=True)
        return bool(r.value)

    def resident_failures(self):
        """statistics-exchange time-outs of the single-launch kernels reported on this handle so far"""
        n = C.c_int64(0)
        capi.check(self.L.mtfjsp_encoder_resident_failures(self.h, C.byref(n)), self.h, enc=True)
        return int(n.value)

    def peek_nodes(self):
        """diagnostic: the machine path's node rows [B*M,128] as the GAT passes left them (host array)"""
        import numpy as np
        out = np.empty((self.B * self.M, 128), dtype=np.float32)
        capi.check(self.L.mtfjsp_encoder_peek_nodes_host(self.h, out.ctypes.data, out.size), self.h, enc=True)
        return out

    def range_fallbacks(self):
        """-> (times the handle left the f16 split products because an activation exceeded their range, product mode in force)"""
        n, m = C.c_int64(0), C.c_int32(0)
        capi.check(self.L.mtfjsp_encoder_range_fallbacks(self.h, C.byref(n), C.byref(m)), self.h, enc=True)
        return int(n.value), int(m.value)

    def arm_selection(self, which, greedy, seed, counter, idx_out, logp_out=None, gather_from=None, gathered_out=None):
        """fuse the action selection of the next job (which=0) / machine (which=1) actor forward into its heads kernel; same
        stream and outputs as sample() on that forward's prob"""
        capi.check(self.L.mtfjsp_encoder_arm_selection(
            self.h, which, int(greedy), seed, counter, idx_out.data_ptr(), logp_out.data_ptr() if logp_out is not None else 0,
            gather_from.data_ptr() if gather_from is not None else 0, gathered_out.data_ptr() if gathered_out is not None else 0),
            self.h, enc=True)

    def arm_mfea1(self, ctx):
        """the next job actor forward (with an armed selection) also writes m_fea1 / the machine mask of the selected tasks"""
        capi.check(self.L.mtfjsp_encoder_arm_mfea1(self.h, C.byref(ctx)), self.h, enc=True)

    def arm_machine_heads(self, prob, h_pooled, machine_v):
        """the next job actor forward (selection + mfea1 with m_fea2 armed, machine selection armed too) also runs the WHOLE machine
        actor forward into these outputs where the shape allows it (include/mtfjsp.h); the machine_actor_forward that follows with
        the same tensors returns at once"""
        capi.check(self.L.mtfjsp_encoder_arm_machine_heads(self.h, prob.data_ptr(), h_pooled.data_ptr(), machine_v.data_ptr()), self.h, enc=True)

    def fused_launches(self):
        """forwards that took the three-in-one launch (job heads + GAT passes + machine heads) so far"""
        n = C.c_int64(0)
        capi.check(self.L.mtfjsp_encoder_fused_launches(self.h, C.byref(n)), self.h, enc=True)
        return int(n.value)

    def arm_env_step(self, params):
        """the next machine actor forward (with an armed selection into the `mach_idx` of DeviceBatchEnv.step_params) also runs that
        environment step in its heads launch; env_step_fused() tells afterwards whether it did"""
        capi.check(self.L.mtfjsp_encoder_arm_env_step(self.h, params, len(params)), self.h, enc=True)

    def arm_values_only(self):
        """the next job forward and the next machine forward produce the critic values (and pooled embeddings) only: no scorer, prob untouched"""
        capi.check(self.L.mtfjsp_encoder_arm_values_only(self.h), self.h, enc=True)

    def env_step_fused(self):
        return bool(self.L.mtfjsp_encoder_env_step_fused(self.h))

    def timing_begin(self):
        capi.check(self.L.mtfjsp_encoder_timing_begin(self.h), self.h, enc=True)

    def timing_end(self):
        out = {}
        for fam in FAMILIES:
            ms = C.c_double(); n = C.c_int64()
            capi.check(self.L.mtfjsp_encoder_timing_query(self.h, fam.encode(), C.byref(ms), C.byref(n)), self.h, enc=True)
            if n.value:
                out[fam] = {"ms_total": ms.value, "launches": n.value}
        ms = C.c_double(); n = C.c_int64()
        capi.check(self.L.mtfjsp_encoder_timing_end(self.h, C.byref(ms), C.byref(n)), self.h, enc=True)
        return out


class ActorPair:
    """The two actors wired as Run.py:290-412 wires them: the job actor sees the machine actor's graph embedding of the
    previous step (the learned `_input` on the first step of an episode), the machine actor sees the job actor's of this step."""

    def __init__(self, n_job, n_machine, batch, device=0, obs_dtype="f32", weights=None, greedy=False, seed=0):
        self.enc = Encoder(n_job, n_machine, batch, device=device, obs_dtype=obs_dtype)
        w = weights if weights is not None else random_init_weights(seed)
        self.enc.load_weights(w[0], w[1], w[2] if len(w) > 2 else None)   # (job actor, machine actor[, global critic])
        self.has_critic = len(w) > 2
        self.greedy, self.seed = greedy, seed
        self._mf_ctx, self._mf_env, self._mf_own = None, None, None
        self.last_mfea1 = self.last_mmask = None
        self.n_env_fused = 0                                     # decisions whose environment step rode in the machine heads' launch
        self.fuse_env = bool(os.environ.get("MTFJSP_FUSED_ENV")) # (the library reads the same switch when the handle is created; off by default: DESIGN.md §9)
        self.fused = not os.environ.get("MTFJSP_NO_FUSED_SELECT")   # action selection inside the heads kernels (same stream either way)
        self.fuse_mheads = not os.environ.get("MTFJSP_NO_FUSED_MHEADS")   # the machine forward inside the job heads' launch where the library allows it
        self.values_only_terminal = not os.environ.get("MTFJSP_NO_VALUES_ONLY")   # the post-terminal forward pair without its scorers
        self.fuse_env3 = bool(os.environ.get("MTFJSP_FUSED_ENV3"))        # ... and the environment step as that launch's tail (two launches per step; off by default: measured slower)
        dev = self.enc.device
        self.job_logp = torch.zeros(batch, dtype=torch.float32, device=dev)
        self.mch_logp = torch.zeros(batch, dtype=torch.float32, device=dev)
        self.have_hm = False

    def begin_episode(self):
        self.have_hm = False                                    # run:280 h_mch_pooled = None

    def act(self, env, counter, task_idx, mach_idx, job_idx, jv_out=None, mv_out=None, job_logp=None, mach_logp=None,
            after_mfea1=None, force=None, env_step=None, mfea1_out=None, mmask_out=None):
        """one joint decision for every instance; the optional outputs let a trajectory buffer receive action indices,
        log-probabilities and critic values in place (no copies).  force = (task [B], machine [B][, job [B]]) int32 tensors:
        replay these decisions instead of the selected ones (teacher forcing; the forwards and their outputs are unchanged,
        the recorded log-probabilities stay those of the actors' own selections).
        mfea1_out / mmask_out ([B,M,6] observation dtype / [B,M] bytes): when the job heads' launch produces m_fea1 and the machine
        mask of the selected tasks itself, it writes them THERE (a trajectory slot) instead of env.m_fea1 / env.mmask, and the
        machine actor reads them from there; decisions that go through env.observe_mfea1 (forced actions, unfused selection)
        keep env's buffers and `after_mfea1(env)` is the caller's hook to copy them."""
        e = self.enc
        jl = job_logp if job_logp is not None else self.job_logp
        ml = mach_logp if mach_logp is not None else self.mch_logp
        hm = e.h_pooled_m if self.have_hm else None
        fuse_mfea1 = self.fused and force is None
        armed_m = armed_e = False
        if self.fused:
            e.arm_selection(0, self.greedy, self.seed, 2 * counter, job_idx, jl, env.candidate, task_idx)
            if fuse_mfea1:
                if self._mf_env is not env:
                    self._mf_ctx, self._mf_env = env.mfea1_context(), env
                    self._mf_own = (self._mf_ctx.m_fea1_out, self._mf_ctx.mmask_out)
                redirect = mfea1_out is not None and mmask_out is not None
                self._mf_ctx.m_fea1_out = mfea1_out.data_ptr() if redirect else self._mf_own[0]
                self._mf_ctx.mmask_out = mmask_out.data_ptr() if redirect else self._mf_own[1]
                e.arm_mfea1(self._mf_ctx)
                if self.fuse_mheads and not (self.fuse_env and env_step is not None):
                    # the whole machine forward may ride in the job heads' launch (three launches per step): its selection and outputs
                    # are armed now; the machine_actor_forward call below then finds itself done
                    e.arm_selection(1, self.greedy, self.seed, 2 * counter + 1, mach_idx, ml)
                    e.arm_machine_heads(e.mch_prob, e.h_pooled_m, mv_out if mv_out is not None else e.mach_v)
                    armed_m = True
                    if self.fuse_env3 and env_step is not None:
                        # env_step = () or (r4_out, done_out): the step of this decision as the tail of that launch (the library takes it
                        # where the three-in-one launch runs and the step is the 16-instance register kernel; env_step_fused() tells)
                        params = env.step_params(task_idx, mach_idx, *env_step)
                        if params is not None:
                            e.arm_env_step(params)
                            armed_e = True
            prob, h_o, _ = e.job_actor_forward(env.tasks_fea, env.ell_col, env.ell_val, env.candidate, env.job_mask, hm, v_out=jv_out)
        else:
            prob, h_o, _ = e.job_actor_forward(env.tasks_fea, env.ell_col, env.ell_val, env.candidate, env.job_mask, hm, v_out=jv_out)
            e.sample(prob, self.greedy, self.seed, 2 * counter, job_idx, jl, env.candidate, task_idx)
        if force is not None:
            task_idx.copy_(force[0])
            if len(force) > 2:
                job_idx.copy_(force[2])
        mf1, mmk = env.m_fea1, env.mmask
        if not fuse_mfea1:
            env.observe_mfea1(task_idx)                         # -> env.m_fea1, env.mmask (else: written by the heads kernel)
            if after_mfea1 is not None:
                after_mfea1(env)
        elif mfea1_out is not None and mmask_out is not None:
            mf1, mmk = mfea1_out, mmask_out                     # the heads kernel wrote them into the caller's buffers
        elif after_mfea1 is not None:
            after_mfea1(env)
        self.last_mfea1, self.last_mmask = mf1, mmk             # (terminal_values: the machine actor's inputs of the last decision)
        stepped = False
        if self.fused:
            if not armed_m:
                e.arm_selection(1, self.greedy, self.seed, 2 * counter + 1, mach_idx, ml)
            if self.fuse_env and env_step is not None and force is None:
                # env_step = () or (r4_out, done_out): let the environment step of this decision ride in the machine heads' launch
                params = env.step_params(task_idx, mach_idx, *env_step)
                if params is not None:
                    e.arm_env_step(params)
            e.machine_actor_forward(mf1, env.m_fea2, h_o, mmk, v_out=mv_out)
            stepped = (armed_e or (self.fuse_env and env_step is not None and force is None)) and e.env_step_fused()
        else:
            mprob, _, _ = e.machine_actor_forward(mf1, env.m_fea2, h_o, mmk, v_out=mv_out)
            e.sample(mprob, self.greedy, self.seed, 2 * counter + 1, mach_idx, ml)
        if force is not None:
            mach_idx.copy_(force[1])
        self.have_hm = True
        self.n_env_fused += int(stepped)
        return stepped                                           # True: the environment has already taken this step

    def terminal_values(self, env, prev_job_mask, jv_out, mv_out):
        """The post-terminal forward pair of Run.py:455-475, to be called right after the env step that finished the
        episode: job actor on the post-step observation and candidates with the PREVIOUS job mask (the new one masks every
        job) and the last machine-graph embedding; machine actor on the last m_fea1 / machine mask, the post-step m_fea2
        and that job forward's graph embedding.  Only the local critic values are kept: they are v_ of the terminal step
        (replaybuffer.py:131-139), which every advantage of the episode depends on (ppo:473,523 have no (1-done) factor)."""
        e = self.enc
        hm = e.h_pooled_m if self.have_hm else None
        if self.values_only_terminal:
            e.arm_values_only()                                  # (round 6) the pair keeps nothing but the two values: no scorer in its heads launches
        _, h_o, _ = e.job_actor_forward(env.tasks_fea, env.ell_col, env.ell_val, env.candidate, prev_job_mask, hm, v_out=jv_out)
        mf1 = self.last_mfea1 if self.last_mfea1 is not None else env.m_fea1
        mmk = self.last_mmask if self.last_mmask is not None else env.mmask
        e.machine_actor_forward(mf1, env.m_fea2, h_o, mmk, v_out=mv_out)

    def timing_begin(self):
        self.enc.timing_begin()

    def timing_end(self):
        return self.enc.timing_end()

    def roofline(self, name, kd, B):
        """matrix-core roofline for the single-launch GIN kernel; HBM roofline for the streaming GIN products (k_gemm_x6 streams
        1 KiB per row: 512 B in, 512 B out; its matrix work is far from the peak); f32 MFMA roofline for the fused GAT kernel
        (DESIGN.md §4)."""
        import os
        J, M, T = self.enc.J, self.enc.M, self.enc.T
        avg_s = kd["ms_total"] / max(kd["launches"], 1) * 1e-3
        if name in ("gin_gemm_bn_relu", "gin_gemm_agg"):
            rows = B * T
            flops = 2.0 * rows * H * H
            if os.environ.get("MTFJSP_GEMM_F32MFMA"):    # (set_product_mode() is not reflected here: bench.py never calls it)
                ach = flops / avg_s / 1e12
                return {"kernel": f"k_gemm16p<{name}> ([{rows},128]x[128,128] f32 MFMA 16x16x4, software-pipelined, fused BN/aggregation prologue + stats epilogue)",
                        "bound": "mfma", "achieved": ach, "peak": 157.3, "unit": "TFLOP/s", "frac": ach / 157.3, "traffic": None,
                        "avg_launch_us": avg_s * 1e6, "launches": kd["launches"], "algorithmic_flops_per_launch": flops,
                        "hbm_GBps_same_launch": rows * H * 4 * 2 / avg_s / 1e9}
            nbytes = rows * H * 4 * 2 + 2 * H * H * 2 + 4 * H * 8 * 8     # rows in + rows out + weight planes + BatchNorm sums
            ach = nbytes / avg_s / 1e9
            return {"kernel": f"k_gemm_x6<{name}> ([{rows},128]x[128,128] at f32 accuracy on the f16 matrix cores: 2-way operand split, 3 piece "
                              "products, f32 accumulate; fused BN/aggregation prologue + BN-sums epilogue; 4 producer + 4 consumer waves)",
                    "bound": "hbm", "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0,
                    "frac_of_measured_copy_bw": ach / 6300.0, "traffic": None,
                    "copy_bw_note": "6.3 TB/s is the copy rate of buffers that fit the 256 MB memory-side cache; a chain of read-one-write-one passes "
                                    "over matrices of this size (419 MB at 819 200 rows) runs at 5.0-5.5 TB/s in one direction and at 6.5 TB/s with "
                                    "alternating directions + non-temporal reads, which these launches use (profiles/r03d_ubench_mall_order.txt)",
                    "avg_launch_us": avg_s * 1e6, "launches": kd["launches"], "algorithmic_bytes_per_launch": nbytes,
                    "algorithmic_flops_per_launch": flops, "f32_equivalent_TFLOPs": flops / avg_s / 1e12,
                    "f16_matrix_TFLOPs": 3 * flops / avg_s / 1e12}
        if name in ("heads", "heads_gat3", "heads_gat3_heads", "gat3"):
            # Latency chains: one workgroup per CU runs a dependent sequence of small products separated by workgroup barriers.  Two
            # bounds are priced: the matrix time of the executed f16 piece products at the dense 16-bit peak, and the LDS traffic of
            # the operand reads at the chip's ds_read_b128 rate (every wave re-reads every activation tile / the GAT weight image).
            R = J if name != "gat3" else 0
            heads_flops = lambda rows_per_inst: 2.0 * B * H * H * (2 * rows_per_inst + 4)          # Wa x, W1 s1 per scorer row; Wb, Wc, Wc0, Wc1 per instance
            gat_flops = 2.0 * (2 * B * M) * H * H * 2 + 2.0 * (2 * B * M) * 8 * H             # two 128-deep passes + the K = 8 first pass
            if name == "heads":
                rows = (J + M) / 2.0                                                           # job and machine heads launches are averaged in this family
                flops = heads_flops(rows)
                lds = B * (rows + 2) * H * 2 * 2 * 8 * 2                                       # planes (2 x f16) of every tile read by all 8 waves, twice (phases B, C)
            elif name == "heads_gat3":
                flops = heads_flops(J) + gat_flops
                lds = B * (J + 2) * H * 2 * 2 * 8 * 2 + (2 * B * M / 16.0) * 2 * 65536           # + the 64 KB weight image streamed per tile and pass
            elif name == "heads_gat3_heads":                                                   # job heads + GAT passes + machine heads in one launch
                flops = heads_flops(J) + gat_flops + heads_flops(M)
                lds = B * (J + M + 4) * H * 2 * 2 * 8 * 2 + (2 * B * M / 16.0) * 2 * 65536
            else:
                flops = gat_flops
                lds = (2 * B * M / 16.0) * 2 * 65536
            executed = 3.0 * flops
            t_mfma = executed / 2.5e15
            t_lds = lds / 150e12                                                               # ~150 TB/s aggregate ds_read_b128 (MI355X_MICROARCH.md, LDS)
            return {"kernel": {"heads": "k_headsx (both heads of an actor for 16 instances per workgroup)", "heads_gat3": "k_headsx_gat3x (job heads + the machine path's three GAT passes)",
                               "heads_gat3_heads": "k_headsx_gat3x_headsx (job heads + the machine path's three GAT passes + in-launch exchange of the node statistics + machine heads)",
                               "gat3": "k_gat3x (three GAT passes, stand-alone)"}[name],
                    "bound": "latency (dependent phases of one workgroup per CU); priced against the matrix peak and the LDS read rate",
                    "avg_launch_us": avg_s * 1e6, "launches": kd["launches"], "algorithmic_flops_per_launch": flops, "executed_matrix_flops_per_launch": executed,
                    "matrix_time_us_at_2.5_PFLOPs": t_mfma * 1e6, "frac_of_matrix_peak": t_mfma / avg_s,
                    "lds_operand_bytes_per_launch": lds, "lds_time_us_at_150_TBps": t_lds * 1e6, "frac_of_lds_read_rate": t_lds / avg_s,
                    "achieved": executed / avg_s / 1e12, "peak": 2500.0, "unit": "TFLOP/s", "frac": t_mfma / avg_s, "traffic": None}
        if name == "gin_resident":
            # k_gin_res: the six Linear products of one forward in one launch.  Algorithmic work = the f32 products of the
            # reference (2 * rows * (12*128 + 5*128*128)); executed on the f16 matrix cores as 3 piece products each (the first
            # Linear: 6 bf16 piece products of a 16-wide k-step).  Bound: the dense 16-bit matrix peak (2.5 PFLOP/s).
            rows = B * T
            flops = 2.0 * rows * (12 * H + 5 * H * H)
            executed = 2.0 * rows * (6 * 16 * H + 3 * 5 * H * H)
            ach = flops / avg_s / 1e12
            return {"kernel": f"k_gin_res (whole GIN encoder of one forward, [{rows},12]->128 + 5 x [{rows},128]x[128,128] + 6 batch-wide BatchNorms, "
                              "activations resident in registers, 2-way f16 operand split = 3 piece products per product, f32 accumulate, "
                              "6 in-kernel statistics exchanges through count-carrying integer atomics (no grid barrier); graph pool + candidate gather fused)",
                    # achieved = ALGORITHMIC flops (the reference's f32 products) / launch time; the piece products the
                    # matrix cores execute for them (3 per product) are the side keys
                    "bound": "mfma", "achieved": ach, "peak": 2500.0, "unit": "TFLOP/s", "frac": ach / 2500.0, "traffic": None,
                    "what_is_counted": "algorithmic f32 flops of the reference's six Linear products (SURVEY 8d) against the dense 16-bit matrix peak; "
                                       "executed piece-product flops (3 per product; 6 in the 12->128 Linear) in frac_executed_flops",
                    "frac_algorithmic_flops": ach / 2500.0, "frac_executed_flops": executed / avg_s / 1e12 / 2500.0,
                    "executed_TFLOPs": executed / avg_s / 1e12,
                    "avg_launch_us": avg_s * 1e6, "launches": kd["launches"], "executed_matrix_flops_per_launch": executed,
                    "algorithmic_flops_per_launch": flops, "f32_equivalent_TFLOPs": flops / avg_s / 1e12,
                    "f32_equivalent_frac_of_f32_matrix_peak_157.3": flops / avg_s / 1e12 / 157.3,
                    "hbm_bytes_algorithmic_per_launch": rows * (12 * 4 + 16) + B * (H + J * H) * 4,
                    # SURVEY §8(d) prices the job encoder at 64*T + 6 BatchNorm boundaries x 1024*T + 512*(J+1) bytes per env-step,
                    # i.e. WITH the [rows,128] f32 activations written and re-read at every boundary; this kernel keeps them in
                    # registers, so the same work priced on those bytes exceeds what a streaming design could reach
                    "survey_8d_job_encoder_bytes_per_launch": B * (64 * T + 6 * 1024 * T + 512 * (J + 1)),
                    "survey_8d_bytes_over_time_GBps": B * (64 * T + 6 * 1024 * T + 512 * (J + 1)) / avg_s / 1e9,
                    "survey_8d_bytes_over_time_frac_of_hbm_peak": B * (64 * T + 6 * 1024 * T + 512 * (J + 1)) / avg_s / 1e9 / 8000.0}
        if name == "gat3":
            rows = 2 * B * M
            flops = 2.0 * rows * H * H * 3
            ach = flops / avg_s / 1e12
            return {"kernel": "k_gat3 (3 fused GAT passes, f32 MFMA 16x16x4)",
                    "bound": "mfma", "achieved": ach, "peak": 157.3, "unit": "TFLOP/s", "frac": ach / 157.3, "traffic": None,
                    "avg_launch_us": avg_s * 1e6, "launches": kd["launches"], "algorithmic_flops_per_launch": flops,
                    "hbm_GBps_same_launch": rows * H * 4 * 2 / avg_s / 1e9}
        return {"kernel": name, "bound": "hbm", "achieved": None, "peak": 8000.0, "unit": "GB/s", "frac": None, "traffic": None,
                "avg_launch_us": avg_s * 1e6, "launches": kd["launches"]}


def smoke():
    """tiny forward of both actors on cuda:0 (shape/finite check; parity is in tests/test_encoder_hip.py)."""
    from .batch_env import DeviceBatchEnv
    from .instances import generate_instances, random_weights
    J, M, E, B = 6, 6, 2, 8
    t, p, tt, edge = generate_instances(B, J, M, E, seed=0)
    env = DeviceBatchEnv(J, M, E, B, obs_dtype="f32")
    env.load_instances(t, p, tt, edge=edge); env.scaler_init(); env.reset(random_weights(B))
    ap = ActorPair(J, M, B, obs_dtype="f32", seed=0)
    a = torch.zeros(B, dtype=torch.int32, device=env.device); m = torch.zeros_like(a); j = torch.zeros_like(a)
    for s in range(J * M):
        ap.act(env, s, a, m, j)
        env.step(a, m)
    torch.cuda.synchronize()
    assert bool(env.info[:, 1].all()) and int((env.status & capi.ST_INVALID).sum()) == 0
    assert torch.isfinite(ap.enc.job_prob).all() and torch.isfinite(ap.enc.mch_prob).all()
