#!/usr/bin/env python3
"""End-to-end tour on one MI355X: generate instances on the device, collect a PPO buffer with the two actors
(device-resident trajectory buffer with the reference's ReplayBuffer interface, GAE + normalised advantages), then
evaluate the same policy greedily on a separate evaluation set with per-instance BatchNorm statistics (the reference's
validate.py semantics).  Everything runs through include/mtfjsp.h; nothing here touches the oracle.

    python examples/rollout_and_eval.py [--batch 1024] [--episodes 5]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mtfjsp_amd  # noqa: F401  (alias of the package directory)
from importlib import import_module

rollout = import_module("e2e-mappo-for-mt-fjsp_amd.rollout")
evaluate = import_module("e2e-mappo-for-mt-fjsp_amd.evaluate")
instances = import_module("e2e-mappo-for-mt-fjsp_amd.instances")
encoder = import_module("e2e-mappo-for-mt-fjsp_amd.encoder")

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--episodes", type=int, default=5)
a = ap.parse_args()
J, M, E, B = 6, 6, 2, a.batch
T = J * M
weights = encoder.random_init_weights(seed=0)            # or (torch.load(job_actor.pth), torch.load(machine_actor.pth))

# ---- collect one buffer (episodes x T steps) on device-generated instances
ro = rollout.Rollout(J, M, E, B, policy="actor", obs_dtype="f32", weights=weights, collect="full", buffer_episodes=a.episodes)
ro.env.generate_instances(seed=2024)                      # replace the host-generated set: nothing is uploaded
ro.env.scaler_init()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.episodes * T):
    ro.step()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
norm_adv, targets = ro.last_adv                            # 4 reward channels: mk, pt, tt, it
print(f"collected {a.episodes} episodes x {B} instances in {dt * 1e3:.1f} ms ({a.episodes * T * B / dt / 1e6:.2f} M env-steps/s)")
print("trajectory buffer:", f"{ro.traj.nbytes() / 1e6:.0f} MB on device;",
      "the reference's dense adjacency alone would be", f"{2 * ro.traj.total_step * B * T * T * 8 / 1e9:.1f} GB")
print("normalised advantage (makespan channel): mean %.3e std %.3f" % (float(norm_adv[0].mean()), float(norm_adv[0].std())))
batch = ro.traj.numpy_to_tensor_operation()                # the reference's 27-tuple, adjacency as EllAdjacency
print("27-tuple entry shapes:", [tuple(x.shape) for x in batch[:6]], "...")

# ---- greedy evaluation of the same policy on 100 fresh instances (env_batch-1 semantics, batched)
t, p, tt, edge = instances.generate_instances(100, J, M, E, seed=1)
args = {"n_job": J, "n_machine": M, "n_edge": E, "weight_mk": 0.4, "weight_ec": 0.4, "weight_tt": 0.2}
t0 = time.perf_counter()
cost, final4, obj = evaluate.validate_cost_batched(weights, t, p, tt, edge, args)
print(f"evaluated 100 instances in {(time.perf_counter() - t0) * 1e3:.1f} ms: mean makespan {final4[:, 0].mean():.1f}, mean objective {obj.mean():.1f}")
