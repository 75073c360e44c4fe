#!/usr/bin/env python3
"""Diagnostic: does a kernel's second launch in a row run faster than its first (instruction cache kept across launches)?
The machine actor's forward issued three times in a row behind a job actor forward; torch events on the shared stream.
Measured (round 3): 47.1 / 44.9 / 44.6 us — and the PMC says the instruction cache hardly misses at all (profiles/r03h_pmc_sq_counters_heads.txt).
    gpurun -- 'python tools/warm_icache_probe.py'"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mtfjsp_amd  # noqa
from importlib import import_module
rollout = import_module("e2e-mappo-for-mt-fjsp_amd.rollout")

ro = rollout.Rollout(6, 6, 2, 4096, policy="actor", obs_dtype="f32")
for _ in range(72):
    ro.step()
env, act, e = ro.env, ro.actor, ro.actor.enc
# the forwards need the job actor's graph embedding: take it from a fresh job forward, then time 3 machine forwards in a row
hm = e.h_pooled_m
t1 = t2 = t3 = 0.0
n = 40
for it in range(n):
    prob, h_o, _ = e.job_actor_forward(env.tasks_fea, env.ell_col, env.ell_val, env.candidate, env.job_mask, hm)
    a, b, c, d = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    a.record(); e.machine_actor_forward(env.m_fea1, env.m_fea2, h_o, env.mmask)
    b.record(); e.machine_actor_forward(env.m_fea1, env.m_fea2, h_o, env.mmask)
    c.record(); e.machine_actor_forward(env.m_fea1, env.m_fea2, h_o, env.mmask)
    d.record(); torch.cuda.synchronize()
    t1 += a.elapsed_time(b); t2 += b.elapsed_time(c); t3 += c.elapsed_time(d)
print(f"machine actor forward (GAT + heads launches) right after the GIN/job launches: {t1 / n * 1e3:.1f} us; again: {t2 / n * 1e3:.1f} us; a third time: {t3 / n * 1e3:.1f} us")
