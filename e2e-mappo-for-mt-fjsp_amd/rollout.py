"""Device-resident rollout driver: the loop of Run.py:229-665 (reset -> [job actor -> m_fea1 -> machine actor ->
sample -> env step + job mask] x T -> done) with every tensor staying in HBM.

policy="random": uniform random valid actions drawn on device (Philox) — env-only workload.
policy="actor" : the GIN job actor and the GAT machine actor run as HIP kernels (encoder.py) and sample the joint action.
"""
import os
import random as _random

import numpy as np
import torch

from . import capi
from .batch_env import DeviceBatchEnv
from .instances import generate_instances, random_weights


def actor_available():
    try:
        from . import encoder
        return encoder.available()
    except Exception:
        return False


class Rollout:
    def __init__(self, n_job, n_machine, n_edge, batch, device=0, policy="random", obs_dtype="f32",
                 instance_seed=0, rank=0, world=1, weights=None, w3_pool_episodes=32, greedy=False, seed=1234,
                 buffer_episodes=5, gamma=0.99, lam=0.98, collect=True, instances=None, w3_episodes=None, w3="device",
                 exact_bn=False, global_handoff=True, time_handoff=False):
        """instances: (t, p, tt, edge) host arrays for THIS shard; default = rows [rank*batch, (rank+1)*batch) of the
        reference generator's `Instance_Dataset(samples=world*batch, seed=instance_seed)` (every
        instance distinct).  w3_episodes: [n,B,3] reward weights to use episode by episode (tests); otherwise w3 = "device"
        (default) or "host", see below.  exact_bn (world > 1, torch.distributed initialised): every BatchNorm of the actor
        forwards normalises over the rows of ALL shards (one small all-reduce per BatchNorm, streaming GIN launches) instead of
        per shard — the reference's semantics for env_batch = world*batch; off by default.
        weights = (job actor, machine actor[, global critic]) state dicts.  With collect="full" and global-critic weights the
        hand-off is the reference's whole one (global_handoff; ppo:628-703).  time_handoff (bench.py): device events around the
        all-gather and a synchronisation to read them — off in production, where the hand-off stays asynchronous.
        Only rank's own instances are generated when none are passed (generate_instances(first=...))."""
        self.J, self.M, self.E, self.B = n_job, n_machine, n_edge, batch
        self.T = n_job * n_machine
        self.policy = policy
        self.rank, self.world = rank, world
        self.env = DeviceBatchEnv(n_job, n_machine, n_edge, batch, obs_dtype=obs_dtype, device=device)
        dev = self.env.device
        if instances is None:
            lo, hi = rank * batch, (rank + 1) * batch               # only this rank's rows are kept (the stream is the whole set's)
            instances = generate_instances(world * batch, n_job, n_machine, n_edge, seed=instance_seed, first=lo, count=batch)
            self.instances_desc = (f"Instance_Dataset(samples={world * batch}, n_job={n_job}, n_machine={n_machine}, n_edge={n_edge}, "
                                   f"seed={instance_seed}) rows [{lo},{hi}) — all distinct")
        else:
            self.instances_desc = "caller-provided instances"
        t, p, tt, edge = instances
        self.env.load_instances(t, p, tt, edge=edge)
        self.env.scaler_init()
        # reward weights, fresh for every episode and instance (env:1253-1259): w3="device" draws them on the device (Philox
        # keyed by (seed, episode, instance); nothing crosses PCIe and the host never stalls the launch queue); w3="host" keeps
        # the reference's python `random` stream (3 draws per instance in instance order), drawn in pools of
        # `w3_pool_episodes` episodes and uploaded — about 10 us per instance and episode of host time
        self.w3_mode = "fixed" if w3_episodes is not None else w3
        self._w3_rng = _random.Random(1000 + rank)
        self._w3_pool_n = w3_pool_episodes
        if self.w3_mode == "fixed":
            self.w3_pool = torch.as_tensor(np.asarray(w3_episodes), dtype=torch.float64, device=dev).contiguous()
        elif self.w3_mode == "host":
            self.w3_pool = torch.empty(w3_pool_episodes, batch, 3, dtype=torch.float64, device=dev)
            self._refill_w3()
        elif self.w3_mode == "device":
            self.w3_pool = torch.empty(2, batch, 3, dtype=torch.float64, device=dev)     # this episode's / (full trajectory) kept alive
        else:
            raise ValueError("w3 must be 'device' or 'host'")
        self.task = torch.zeros(batch, dtype=torch.int32, device=dev)
        self.mach = torch.zeros(batch, dtype=torch.int32, device=dev)
        self.job = torch.zeros(batch, dtype=torch.int32, device=dev)
        self.seed = seed + rank
        self.t_in_ep = 0
        self.episode = 0
        self.nsteps = 0
        # trajectory record of the quantities the advantage computation needs (Appendix A of SURVEY.md, rows 17-20 and
        # 23-26): 4 scaled reward components, done, local critic values — [S,B] device tensors, S = buffer_episodes*T
        # collect=True: what the advantage computation needs; collect="full": every field of the reference's ReplayBuffer
        # in a device-resident TrajectoryBuffer
        self.collect = bool(collect) and policy == "actor"
        self.full = self.collect and collect == "full"
        self.S = buffer_episodes * self.T
        self.buffer_episodes = buffer_episodes
        self.gamma, self.lam = gamma, lam
        self.buf_pos = 0
        self.last_adv = None
        self.last_gather = None
        self.n_handoffs = 0
        self.n_resident_failures = 0
        self.n_dropped_buffers = 0         # buffers every rank dropped together because one of them reported a failure (world > 1)
        self.tainted = False               # this rank's current buffer contains invalid steps and will be dropped at its boundary
        self._new_episode = False
        self._pre_slot = -1                # trajectory slot whose pre-decision observation the previous step's snapshot already wrote
        self.exact_bn = False
        self.global_handoff, self.time_handoff = global_handoff, time_handoff
        self.last_full = None
        self.traj = None
        # local critic values: T+1 slots per episode — slot t<T is the value at act time of step t, slot T the value of
        # the terminal state from the post-terminal forward pair (run:455-475); v_ of step t is slot t+1 (run:451-454)
        if self.full:
            from .trajectory import TrajectoryBuffer
            self.traj = TrajectoryBuffer({"n_job": n_job, "n_machine": n_machine, "buffer_size": buffer_episodes,
                                          "env_batch": batch, "gcn_input_dim": 12}, device=dev,
                                         obs_dtype=torch.float32 if obs_dtype == "f32" else torch.float64, alias_v_next=True)
            tb = self.traj
            self.buf_r, self.buf_done, self.buf_jv, self.buf_mv = tb.r4, tb.done_operation, tb._jv, tb._mv
        elif self.collect:
            f = dict(dtype=torch.float32, device=dev)
            self.buf_r = torch.zeros(self.S, 4, batch, **f)          # mk, idle, pt, tt (scaled, pe:255-262 order)
            self.buf_done = torch.zeros(self.S, batch, **f)
            self.buf_jv = torch.zeros(buffer_episodes, self.T + 1, batch, 2, **f)     # job critic: mk, idle   (ac:293)
            self.buf_mv = torch.zeros(buffer_episodes, self.T + 1, batch, 2, **f)     # machine critic: pt, tt (ac:495)
        if self.collect:
            self.prev_job_mask = torch.zeros(batch, n_job, dtype=torch.uint8, device=dev)
        self.actor = None
        if policy == "actor":
            from . import encoder
            self.actor = encoder.ActorPair(n_job, n_machine, batch, device=device, obs_dtype=obs_dtype, weights=weights,
                                           greedy=greedy, seed=self.seed)
            if exact_bn and world > 1:
                from . import dist as _dist
                self.actor.enc.set_stats_reduce(_dist.bn_stats_allreduce(), world * batch)
                self.actor.enc.set_deferred_poll(True)     # failures are acted upon once per step, by all ranks together (step())
                self.exact_bn = True

    def describe(self):
        if self.policy == "actor":
            return ("full rollout step: GIN job-actor forward + m_fea1 + GAT machine-actor forward + categorical sampling "
                    "+ fused env step (transition, rewards, reward scaling, observation, job mask); batched reset every T steps")
        return ("env-only step: on-device random valid action + fused env step (transition, rewards, reward scaling, "
                "observation, job mask); batched reset every T steps")

    def env_kernel_name(self):
        """which step kernel mtfjsp_step dispatches to for this shape (csrc/mtfjsp_env.hip launch selection)"""
        force = os.environ.get("MTFJSP_ENV_KERNEL", "")
        lds = bool(os.environ.get("MTFJSP_ENV_LDS")) or force in ("lds", "lds1")
        one = self.T <= 64 and self.M * self.M <= 64 and self.J <= 64 and not lds
        two = (not one) and self.T <= 128 and self.M * self.M <= 128 and self.M <= 16 and self.J <= 64 and not lds and force != "reg1"
        if two:
            return "k_env_grp16x2" if (force == "grp16" or (force != "grp4" and self.B <= 4096)) else "k_env_grp4x2"
        if not one:
            return "k_env_step" if force == "lds1" else "k_env_step_grp"
        if force == "reg1":
            return "k_env_reg"
        return "k_env_grp16" if (force == "grp16" or (force != "grp4" and self.B <= 8192)) else "k_env_grp4"

    def _refill_w3(self):
        self.w3_pool.copy_(torch.as_tensor(np.stack([random_weights(self.B, rng=self._w3_rng) for _ in range(self._w3_pool_n)])))

    def _episode_w3(self):
        n = self.w3_pool.shape[0]
        if self.w3_mode == "device":
            return self.env.draw_reward_weights(self.seed, self.episode, out=self.w3_pool[self.episode % n])
        if self.w3_mode == "host" and self.episode > 0 and self.episode % n == 0:
            self._refill_w3()                                              # fresh draws for every episode (env:1253-1259)
        return self.w3_pool[self.episode % n]

    def step(self, force=None):
        """one batched decision step; a reported time-out of the single-launch GIN kernel or an operand beyond the f16 range
        (capi.ERR_RETRY: every output since that launch is invalid, the encoder has switched kernels) discards the trajectory
        buffer collected so far and restarts the episode.
        Single process: the buffer restarts at once.  With a process group (`_lockstep`): the hand-off at the end of a buffer is
        a collective, so every rank must reach it after the same number of step() calls — the failing rank keeps its position in
        the buffer, marks the buffer as tainted, restarts its episode and runs the rest of the buffer without recording; at the
        buffer boundary the ranks agree (one MAX all-reduce of a flag, finish_buffer) and ALL of them drop the buffer.
        With exact_bn the forwards themselves contain collectives (the BatchNorm all-reduces), so the ranks must issue the SAME
        forwards at every step: the forward entries do not poll for failures at all (Encoder.set_deferred_poll); once per step,
        after draining the device, every rank checks its failure words and the ranks agree — if any failed, ALL restart their
        episode and taint the buffer together, and their forward sequences stay identical."""
        if self.exact_bn:
            # (exact_bn implies an active group with world > 1 and a deferred poll, with or without a trajectory record: a throughput
            # run over several ranks — collect=False — all-reduces BatchNorm sums too, and a range failure on one rank must still
            # switch every rank's kernels before NaN sums spread through the reduction: advisor r5)
            self._agree_on_failure_and_restart_together()
            self._step(force)
            return
        try:
            self._step(force)
        except capi.MtfjspError as ex:
            if ex.code != capi.ERR_RETRY:
                raise
            self._restart_after_failure()
            self._step(force)

    def _failed_since_last_check(self):
        """synchronise and read the encoder's asynchronous failure words -> True when a forward enqueued since the last check is
        invalid (capi.ERR_RETRY; the handle has switched kernels); any other error propagates"""
        try:
            self.actor.enc.check()
        except capi.MtfjspError as ex:
            if ex.code != capi.ERR_RETRY:
                raise
            return True
        return False

    def _agree_on_failure_and_restart_together(self):
        from . import dist as _dist
        if torch.cuda.is_available():                      # (always, in the product; the CPU control-flow test drives this with stubs)
            torch.cuda.synchronize()
        failed = self._failed_since_last_check()
        if _dist.agree_any(failed):
            self.n_resident_failures += int(failed)
            self.t_in_ep = 0                               # every rank: fresh episode, rest of the buffer unrecorded, dropped at its boundary
            self.tainted = self.collect                    # (without a trajectory record there is no buffer to drop)
            self.actor.begin_episode()

    @property
    def _lockstep(self):
        from . import dist as _dist
        return self.collect and _dist.active()

    def _restart_after_failure(self):
        self.n_resident_failures += 1
        self.t_in_ep = 0                                   # the next step resets every instance (fresh weights, scaler returns)
        if self._lockstep:
            self.tainted = True                            # buf_pos keeps counting: the collective stays aligned across ranks
        else:
            self.buf_pos = 0
            if self.full:
                self.traj.reset()
        self.actor.begin_episode()

    def _step(self, force=None):
        """one batched decision step.  force = (task [B], machine [B]) int32 device tensors: apply these decisions instead of
        the sampled ones (teacher forcing for the parity tests; the forwards and everything recorded are unchanged)."""
        env = self.env
        if self.t_in_ep == 0:
            if self.w3_mode == "device" and hasattr(env, "reset_episode") and not os.environ.get("MTFJSP_NO_RESET_EPISODE"):
                # one launch: RewardScaling.reset() of every instance (run:283-284), the episode's reward weights (env:1253-1259), reset (pe:87)
                n = self.w3_pool.shape[0]
                w3 = env.reset_episode(self.seed, self.episode, out=self.w3_pool[self.episode % n])
            else:
                w3 = self._episode_w3()
                env.scaler_reset_returns()                                 # run:283-284
                env.reset(w3)                                              # pe:87 / run:229
            if self.actor is not None:
                self.actor.begin_episode()
            if self.full and not self.tainted:             # (a tainted buffer records nothing: its slots are not episode-aligned any more)
                self.traj.begin_episode(w3)
        last = self.t_in_ep == self.T - 1
        if self.tainted:
            # rest of a buffer that will be dropped (see step()): decisions and env steps without recording
            self.actor.act(env, self.nsteps, self.task, self.mach, self.job, force=force, env_step=()) or env.step(self.task, self.mach)
            self.buf_pos += 1
            if self.buf_pos == self.S:
                self.finish_buffer()
        elif self.full:
            # Per step: the four launches of the decision + ONE snapshot launch.  The kernels write actions, log-probabilities,
            # critic values, rewards, m_fea1 and the machine mask straight into the slot; the snapshot after the step copies the
            # observation into this slot's s' fields, the NEXT slot's s fields (same observation inside an episode) and the scalar
            # reward.  Only an episode's first step takes its own pre-decision snapshot.
            tb = self.traj
            sl = tb.slot()
            if self._pre_slot != tb.count_operation or self.t_in_ep == 0:
                tb.snapshot(env, "pre")
            if not self.actor.act(env, self.nsteps, self.task, sl["mach_idx"], sl["job_idx"], sl["job_v"], sl["mach_v"],
                                  job_logp=sl["job_logp"], mach_logp=sl["mach_logp"], after_mfea1=tb.after_decision, force=force,
                                  env_step=(sl["r4"], sl["done"]), mfea1_out=sl["m_fea1"], mmask_out=sl["mmask"]):
                env.step_record(self.task, sl["mach_idx"], sl["r4"], sl["done"])
            if last:           # value of the terminal state (run:455-475) with the mask the last decision was taken under
                jv_t, mv_t = tb.terminal_slot()
                self.actor.terminal_values(env, tb.mask_operation[tb.count_operation], jv_t, mv_t)
            self._pre_slot = tb.count_operation + 1 if tb.after_step(env, next_pre=not last) else -1
            self.buf_pos += 1
            if self.buf_pos == self.S:
                self.finish_buffer()
                tb.reset()
        elif self.collect:
            e, t = divmod(self.buf_pos, self.T)   # critic values and rewards land directly in the trajectory slots (no copies)
            if last:
                self.prev_job_mask.copy_(env.job_mask)
            if not self.actor.act(env, self.nsteps, self.task, self.mach, self.job, self.buf_jv[e, t], self.buf_mv[e, t], force=force,
                                  env_step=(self.buf_r[self.buf_pos], self.buf_done[self.buf_pos])):
                env.step_record(self.task, self.mach, self.buf_r[self.buf_pos], self.buf_done[self.buf_pos])
            if last:
                self.actor.terminal_values(env, self.prev_job_mask, self.buf_jv[e, self.T], self.buf_mv[e, self.T])
            self.buf_pos += 1
            if self.buf_pos == self.S:
                self.finish_buffer()
        else:
            stepped = False
            if self.actor is not None:
                stepped = self.actor.act(env, self.nsteps, self.task, self.mach, self.job, force=force, env_step=())
            else:
                env.random_actions(self.seed, self.nsteps, self.task, self.mach, self.job)
            if not stepped:
                env.step(self.task, self.mach)
        self.nsteps += 1
        self.t_in_ep += 1
        if self.t_in_ep == self.T or self._new_episode:
            self._new_episode = False
            self.t_in_ep = 0
            self.episode += 1

    def finish_buffer(self):
        """Rollout -> update hand-off (ppo:438-489): local-critic GAE per reward channel on this shard, then the one
        collective of the data path — all-gather of the per-shard advantages (RCCL over xGMI when world > 1) for the
        GLOBAL normalisation (adv - mean) / (std + 1e-5) — leaving normalised advantages + value targets on device."""
        from . import advantages as A
        from . import dist as _dist
        S, T, B = self.S, self.T, self.B
        # one synchronisation per buffer: a forward of this buffer that failed asynchronously (single-launch GIN kernel, see
        # Encoder.check) must not reach the update.  Single process: capi.ERR_RETRY propagates and step() restarts the buffer.
        # With a process group the ranks first AGREE (MAX all-reduce of one flag) whether any of them has a tainted buffer; if so
        # every rank drops its buffer here and none enters the all-gather — the collective sequence stays identical on all ranks.
        mv4 = mv4_ = None
        whole = self.full and self.actor.has_critic and self.global_handoff
        if self._lockstep:
            # The ranks agree BEFORE any rank-conditional work that may contain collectives: the global critic's 2 S forwards of the
            # whole hand-off run either on every rank or on none (with exact_bn each of them all-reduces its BatchNorm sums), and
            # the ranks agree once more on what those forwards reported before anybody enters the all-gather.
            def drop(failed):
                self.n_dropped_buffers += 1
                self.buf_pos = 0
                if self.full:
                    self.traj.reset()
                if failed:                                  # this rank's episode is not aligned with the buffer any more (or invalid)
                    self._new_episode = True
                self.tainted = False
            failed = self._failed_since_last_check()
            self.n_resident_failures += int(failed and not self.tainted)
            failed = failed or self.tainted
            if _dist.agree_any(failed):
                return drop(failed)
            if whole:
                try:
                    mv4, mv4_ = A.sample_global_values(self.actor.enc, self.traj)
                    failed = self._failed_since_last_check()
                except capi.MtfjspError as ex:               # (a forward entry of the sampling polled the failure itself: no exact_bn, no collectives inside)
                    if ex.code != capi.ERR_RETRY:
                        raise
                    failed = True
                self.n_resident_failures += int(failed)
                if _dist.agree_any(failed):
                    return drop(failed)
        else:
            self.actor.enc.check()
        # v of step t = slot t, v_ of step t = slot t+1 of its episode; the terminal step's v_ is the post-terminal forward
        # (run:451-475).  The deltas carry NO (1-done) factor (ppo:473,523): the terminal v_ enters every advantage of the
        # episode; (1-done) only stops the carried gae at episode boundaries.
        jv, mv = self.buf_jv[:, :T].reshape(S, B, 2), self.buf_mv[:, :T].reshape(S, B, 2)
        jv_, mv_ = self.buf_jv[:, 1:].reshape(S, B, 2), self.buf_mv[:, 1:].reshape(S, B, 2)
        if whole:
            # the whole hand-off of ppo:628-703: the global critic sampled on every stored pre- and post-decision state (2 S
            # forwards, no gradient), 4 global + 4 local advantages and the 8 value tensors in ONE all-gather
            if mv4 is None:
                mv4, mv4_ = A.sample_global_values(self.actor.enc, self.traj)
            h = A.full_handoff(self.env, self.buf_r, jv, jv_, mv, mv_, mv4, mv4_, self.buf_done, self.gamma, self.lam,
                               timed=self.time_handoff)
            self.last_full = h
            norm, targets, raw, self.last_gather = h["local_adv"], h["local_targets"], h["raw_local"], h["gather"]
        else:
            norm, targets, raw, self.last_gather = A.local_advantages(self.env, self.buf_r, jv, jv_, mv, mv_, self.buf_done,
                                                                      self.gamma, self.lam, timed=self.time_handoff)
        self.last_adv = (norm, targets)
        self.last_raw_adv = raw
        self.n_handoffs += 1
        self.buf_pos = 0

    def timing_begin(self):
        self.env.timing_begin()
        if self.actor is not None:
            self.actor.timing_begin()

    def timing_end(self):
        ms, n = self.env.timing_end()
        out = {"env_step": {"ms_total": ms, "launches": n}}
        if self.actor is not None:
            out.update(self.actor.timing_end())
        return out

    def roofline(self, name, kd):
        return self.actor.roofline(name, kd, self.B)

    def check_finished_cleanly(self):
        st = self.env.status
        if self.actor is not None:
            self.actor.enc.check()
        assert int((st & capi.ST_INVALID).sum().item()) == 0, "rollout produced invalid actions"
        if self.t_in_ep == 0 and self.nsteps > 0:
            assert bool(self.env.info[:, 1].all().item()), "episode boundary without done"
