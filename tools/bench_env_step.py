#!/usr/bin/env python3
"""Micro-benchmark of the fused env step kernel (random valid actions on device) with a B sweep.
Prints env-steps/s, mean kernel time (HIP events on the launch stream) and algorithmic GB/s
(ENV_BYTES = 136*T + 156*M + 483 per env-step)."""
import argparse
import json
import sys
import os
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mtfjsp_amd  # noqa
from importlib import import_module

batch_env = import_module("e2e-mappo-for-mt-fjsp_amd.batch_env")
inst = import_module("e2e-mappo-for-mt-fjsp_amd.instances")


def run(J, M, E, B, episodes, obs, seed=0):
    T = J * M
    base = min(B, 256)
    t, p, tt, edge = inst.generate_instances(base, J, M, E, seed)
    rep = (B + base - 1) // base
    t, p, tt, edge = [np.concatenate([x] * rep)[:B] for x in (t, p, tt, edge)]
    env = batch_env.DeviceBatchEnv(J, M, E, B, obs_dtype=obs)
    env.load_instances(t, p, tt, edge=edge)
    env.scaler_init()
    w3 = torch.full((B, 3), 1.0 / 3, dtype=torch.float64, device=env.device)
    a = torch.zeros(B, dtype=torch.int32, device=env.device); m = torch.zeros_like(a)
    for ep in range(episodes + 1):
        if ep == 1:
            torch.cuda.synchronize(); env.timing_begin(); t0 = time.perf_counter()
        env.reset(w3)
        for s in range(T):
            env.random_actions(seed, ep * T + s, a, m)
            env.step(a, m)
    torch.cuda.synchronize(); wall = time.perf_counter() - t0
    ms, n = env.timing_end()
    assert bool(env.info[:, 1].all()), "episode did not finish"
    assert int((env.status & 0x100).sum()) == 0
    kt = ms / n * 1e-3
    bytes_alg = B * (136 * T + 156 * M + 483)
    return dict(J=J, M=M, B=B, obs=obs, steps_per_s_wall=episodes * T * B / wall, kernel_us=kt * 1e6,
                steps_per_s_kernel=B / kt, alg_GBps=bytes_alg / kt / 1e9)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="6x6x2")
    ap.add_argument("--batches", default="4096")
    ap.add_argument("--episodes", type=int, default=3)
    ap.add_argument("--obs", default="f32")
    a = ap.parse_args()
    for sz in a.sizes.split(","):
        J, M, E = [int(x) for x in sz.split("x")]
        for B in [int(x) for x in a.batches.split(",")]:
            print(json.dumps(run(J, M, E, B, a.episodes, a.obs)), flush=True)
