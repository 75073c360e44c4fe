#!/usr/bin/env python3
"""Host enqueue time per rollout step vs device time (is the launch path the bottleneck?)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mtfjsp_amd  # noqa
from importlib import import_module
rollout = import_module("e2e-mappo-for-mt-fjsp_amd.rollout")
ro = rollout.Rollout(6, 6, 2, 4096, policy="actor", obs_dtype="f32")
for _ in range(72):
    ro.step()
torch.cuda.synchronize()
n = 360
t0 = time.perf_counter()
for _ in range(n):
    ro.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e6 * (t1 - t0) / n:.1f} us/step, total {1e6 * (t2 - t0) / n:.1f} us/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(180):
    ro.step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
