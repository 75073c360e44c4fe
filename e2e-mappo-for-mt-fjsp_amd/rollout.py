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
                 buffer_episodes=5, gamma=0.99, lam=0.98, collect=True):
        self.J, self.M, self.E, self.B = n_job, n_machine, n_edge, batch
        self.T = n_job * n_machine
        self.policy = policy
        self.rank, self.world = rank, world
        self.env = DeviceBatchEnv(n_job, n_machine, n_edge, batch, obs_dtype=obs_dtype, device=device)
        dev = self.env.device
        # synthetic instances (generator semantics of the reference); 256 distinct ones tiled over the shard
        base = min(batch, 256)
        t, p, tt, edge = generate_instances(base, n_job, n_machine, n_edge, seed=instance_seed)
        rep = (batch + base - 1) // base
        t, p, tt, edge = [np.concatenate([x] * rep)[:batch] for x in (t, p, tt, edge)]
        self.env.load_instances(t, p, tt, edge=edge)
        self.env.scaler_init()
        # reward weights: host `random` stream (env:1253-1259), pre-drawn for a pool of episodes and kept in HBM
        rng = _random.Random(1000 + rank)
        self.w3_pool = torch.as_tensor(np.stack([random_weights(batch, rng=rng) for _ in range(w3_pool_episodes)]),
                                       dtype=torch.float64, device=dev)
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
        self.gamma, self.lam = gamma, lam
        self.buf_pos = 0
        self.last_adv = None
        self.traj = None
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
            self.buf_jv = torch.zeros(self.S + 1, batch, 2, **f)     # job critic: mk, idle   (ac:293)
            self.buf_mv = torch.zeros(self.S + 1, batch, 2, **f)     # machine critic: pt, tt (ac:495)
        self.actor = None
        if policy == "actor":
            from . import encoder
            self.actor = encoder.ActorPair(n_job, n_machine, batch, device=device, obs_dtype=obs_dtype, weights=weights,
                                           greedy=greedy, seed=self.seed)

    def describe(self):
        if self.policy == "actor":
            return ("full rollout step: GIN job-actor forward + m_fea1 + GAT machine-actor forward + categorical sampling "
                    "+ fused env step (transition, rewards, reward scaling, observation, job mask); batched reset every T steps")
        return ("env-only step: on-device random valid action + fused env step (transition, rewards, reward scaling, "
                "observation, job mask); batched reset every T steps")

    def env_kernel_name(self):
        """which step kernel mtfjsp_step dispatches to for this shape (csrc/mtfjsp_env.hip launch selection)"""
        small = self.T <= 64 and self.M * self.M <= 64 and self.J <= 64 and not os.environ.get("MTFJSP_ENV_LDS")
        return "k_env_reg" if small else "k_env_step"

    def step(self):
        env = self.env
        if self.t_in_ep == 0:
            env.scaler_reset_returns()                                     # run:283-284
            env.reset(self.w3_pool[self.episode % self.w3_pool.shape[0]])  # pe:87 / run:229
            if self.actor is not None:
                self.actor.begin_episode()
            if self.full:
                self.traj.begin_episode(self.w3_pool[self.episode % self.w3_pool.shape[0]])
        if self.full:
            tb = self.traj
            sl = tb.slot()
            tb.snapshot(env, "pre")
            self.actor.act(env, self.nsteps, self.task, sl["mach_idx"], sl["job_idx"], sl["job_v"], sl["mach_v"],
                           job_logp=sl["job_logp"], mach_logp=sl["mach_logp"], after_mfea1=tb.after_decision)
            env.step_record(self.task, sl["mach_idx"], sl["r4"], sl["done"])
            tb.after_step(env)
            self.buf_pos += 1
            if self.buf_pos == self.S:
                self.finish_buffer()
                tb.reset()
        elif self.collect:
            k = self.buf_pos                  # critic values and rewards land directly in the trajectory slots (no copies)
            self.actor.act(env, self.nsteps, self.task, self.mach, self.job, self.buf_jv[k], self.buf_mv[k])
            env.step_record(self.task, self.mach, self.buf_r[k], self.buf_done[k])
            self.buf_pos += 1
            if self.buf_pos == self.S:
                self.finish_buffer()
        else:
            if self.actor is not None:
                self.actor.act(env, self.nsteps, self.task, self.mach, self.job)
            else:
                env.random_actions(self.seed, self.nsteps, self.task, self.mach, self.job)
            env.step(self.task, self.mach)
        self.nsteps += 1
        self.t_in_ep += 1
        if self.t_in_ep == self.T:
            self.t_in_ep = 0
            self.episode += 1

    def finish_buffer(self):
        """Rollout -> update hand-off (ppo:438-489): local-critic GAE per reward channel on this shard, then the one
        collective of the data path — all-gather of the per-shard advantages (RCCL over xGMI when world > 1) for the
        GLOBAL normalisation (adv - mean) / (std + 1e-5) — leaving normalised advantages + value targets on device."""
        from . import dist as D
        S = self.S
        # next-state values: v_ of step s is v of step s+1 (run:451-454); within an episode's last step the reference
        # runs one extra forward (run:455-475) — its (1-done) factor zeroes that term in the GAE recursion anyway
        jv, mv = self.buf_jv[:S], self.buf_mv[:S]
        jv_, mv_ = self.buf_jv[1:S + 1], self.buf_mv[1:S + 1]
        r = self.buf_r
        pairs = [(r[:, 0], jv[..., 0], jv_[..., 0]), (r[:, 2], mv[..., 0], mv_[..., 0]),
                 (r[:, 3], mv[..., 1], mv_[..., 1]), (r[:, 1], jv[..., 1], jv_[..., 1])]        # mk, pt, tt, it (ppo:441-443)
        advs = [self.env.gae(rr, v, v_, self.buf_done, self.gamma, self.lam) for rr, v, v_ in pairs]     # HIP reverse scan
        targets = [a + p[1] for a, p in zip(advs, pairs)]
        full = D.all_gather_advantages(advs)                      # [S,B_total] each
        norm = []
        lo = self.rank * self.B
        for a_full, a_loc in zip(full, advs):
            mean, std = a_full.mean(), a_full.std()
            norm.append((a_loc - mean) / (std + 1e-5))
        self.last_adv = (norm, targets)
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
        assert int((st & capi.ST_INVALID).sum().item()) == 0, "rollout produced invalid actions"
        if self.t_in_ep == 0 and self.nsteps > 0:
            assert bool(self.env.info[:, 1].all().item()), "episode boundary without done"
