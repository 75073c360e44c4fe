"""Device-resident rollout driver: the loop of Run.py:229-665 (reset -> [job actor -> m_fea1 -> machine actor ->
sample -> env step + job mask] x T -> done) with every tensor staying in HBM.

policy="random": uniform random valid actions drawn on device (Philox) — env-only workload.
policy="actor" : the GIN job actor and the GAT machine actor run as HIP kernels (encoder.py) and sample the joint action.
"""
import ctypes as C
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
                 instance_seed=0, rank=0, world=1, weights=None, w3_pool_episodes=32, greedy=False, seed=1234):
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

    def step(self):
        env = self.env
        if self.t_in_ep == 0:
            env.scaler_reset_returns()                                     # run:283-284
            env.reset(self.w3_pool[self.episode % self.w3_pool.shape[0]])  # pe:87 / run:229
            if self.actor is not None:
                self.actor.begin_episode()
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
