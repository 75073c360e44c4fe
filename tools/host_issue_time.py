#!/usr/bin/env python3
"""How much host time does one rollout step take to ISSUE?  Short bursts behind a drained queue (so the HIP queue never pushes
back), wall and thread-CPU time of the issuing loop, and the time after which the device has finished the burst.
    gpurun -- 'python tools/host_issue_time.py'
(round 4, MI355X box: 38 us/step in advantage mode, 70 us/step with the full trajectory buffer, against 200 / 217 us of device time)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mtfjsp_amd  # noqa
from importlib import import_module
R = import_module("e2e-mappo-for-mt-fjsp_amd.rollout")
for mode in (True, "full"):
    B = 4096
    ro = R.Rollout(6, 6, 2, B, policy="actor", obs_dtype="f32", collect=mode, buffer_episodes=5)
    for _ in range(400):
        ro.step()
    for n in (20, 50, 100, 170):
        torch.cuda.synchronize()
        c0 = time.thread_time(); t0 = time.perf_counter()
        for _ in range(n):
            ro.step()
        t1 = time.perf_counter(); c1 = time.thread_time()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("collect=%s B=%d burst of %d steps: issue %.1f us/step (thread CPU %.1f us/step), device done after %.1f us/step"
              % (mode, B, n, (t1 - t0) / n * 1e6, (c1 - c0) / n * 1e6, (t2 - t0) / n * 1e6))
