"""Diagnostic: ms per rollout step in consecutive 200-step windows from process start (how long a fresh process takes to
reach its steady state).   gpurun -- 'python tools/warmup_curve.py'"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mtfjsp_amd
from importlib import import_module
ro_mod = import_module("e2e-mappo-for-mt-fjsp_amd.rollout")
ro = ro_mod.Rollout(6, 6, 2, 4096, device=0, policy="actor", obs_dtype="f32", instance_seed=0, collect=True)
t_start = time.perf_counter()
out = []
for w in range(30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): ro.step()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    out.append("%.3f@%.2fs" % ((t1 - t0) / 200 * 1e3, t1 - t_start))
print(" ".join(out))
