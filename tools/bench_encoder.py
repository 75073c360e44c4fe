#!/usr/bin/env python3
"""Per-kernel-family timing of the two actor forwards (HIP events on the launch stream)."""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mtfjsp_amd  # noqa
from importlib import import_module
rollout = import_module("e2e-mappo-for-mt-fjsp_amd.rollout")

ap = argparse.ArgumentParser()
ap.add_argument("--size", default="6x6x2"); ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--steps", type=int, default=72); ap.add_argument("--tag", default="")
a = ap.parse_args()
J, M, E = [int(x) for x in a.size.split("x")]
ro = rollout.Rollout(J, M, E, a.batch, policy="actor", obs_dtype="f32")
for _ in range(J * M):
    ro.step()
torch.cuda.synchronize()
ro.timing_begin()
for _ in range(a.steps):
    ro.step()
k = ro.timing_end()
out = {n: round(v["ms_total"] / v["launches"] * 1e3, 1) for n, v in k.items()}
out["sum_per_step_us"] = round(sum(v["ms_total"] for v in k.values()) / a.steps * 1e3, 1)
print(a.tag, json.dumps(out))
