#!/usr/bin/env python3
"""Wall time of the batched greedy evaluation for B instances, per-instance BatchNorm."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mtfjsp_amd  # noqa
from importlib import import_module
ev = import_module("e2e-mappo-for-mt-fjsp_amd.evaluate"); inst = import_module("e2e-mappo-for-mt-fjsp_amd.instances")
enc = import_module("e2e-mappo-for-mt-fjsp_amd.encoder")
args = {"n_job": 6, "n_machine": 6, "n_edge": 2, "weight_mk": 0.4, "weight_ec": 0.4, "weight_tt": 0.2}
for B in (100, 1000, 4096):
    t, p, tt, edge = inst.generate_instances(B, 6, 6, 2, seed=1)
    actor = enc.ActorPair(6, 6, B, obs_dtype="f32", weights=None, greedy=True, seed=0)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        cost, final4, obj = ev.validate_cost_batched(None, t, p, tt, edge, args, actor=actor)
        dt = time.perf_counter() - t0
    print(f"B={B}: {dt * 1e3:.1f} ms per evaluation ({B * 36 / dt / 1e3:.0f} k env-steps/s), mean objective {obj.mean():.1f}")
