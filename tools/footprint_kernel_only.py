#!/usr/bin/env python3
"""The step kernel and SURVEY 8(d)'s same-footprint copy kernel in ONE process, for `rocprofv3 --kernel-trace`:
kernel-only durations of both (no HIP events, no host pairing) at one batch size.  tools/footprint_kernel_only.sh runs it
under the profiler at 4 096 ... 262 144 instances and reduces the per-dispatch trace to profiles/r05_footprint_kernel_only.json.

  python3 tools/footprint_kernel_only.py --batch 4096          (under rocprofv3: program directly after `--`)
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mtfjsp_amd  # noqa
from importlib import import_module

be = import_module("e2e-mappo-for-mt-fjsp_amd.batch_env")

GRIDS = (256, 512, 1024, 2048, 4096, 8192)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--size", default="6x6x2")
    ap.add_argument("--episodes", type=int, default=5)
    a = ap.parse_args()
    J, M, E = [int(x) for x in a.size.split("x")]
    T, B = J * M, a.batch
    env = be.DeviceBatchEnv(J, M, E, B, obs_dtype="f32")
    env.generate_instances(seed=123)
    env.scaler_init()
    w3 = torch.full((B, 3), 1.0 / 3, dtype=torch.float64, device=env.device)
    act, mch = torch.zeros(B, dtype=torch.int32, device=env.device), torch.zeros(B, dtype=torch.int32, device=env.device)
    r, w = 64 * T + 100 * M + 176, 72 * T + 56 * M + 307          # read / written bytes per env-step (bench.env_bytes_rw)
    # interleaved rounds (cdna_hip_programming.md rule 24): an episode of steps, then the copy at every grid, repeated.  Episodes
    # alternate between two launch patterns, because a dispatch's trace duration depends on how it is queued (round 6: in the
    # rollout's pattern — action kernel and step kernel queued back to back — the trace durations of BOTH kernels, the 16-workgroup
    # action kernel included, jump by about 2.5 us from one launch to the next; the copy launches, each between two HIP event
    # records, do not):
    #   episode 0            warm-up (not counted);
    #   odd episodes         the rollout's pattern;
    #   even episodes >= 2   every step launch between two HIP event records (mtfjsp_timing_begin): the copy launches' pattern.
    # tools/footprint_reduce.py compares like with like: steps and copies that were both launched between event records.
    for ep in range(a.episodes):
        env.reset(w3)
        if ep >= 2 and ep % 2 == 0:
            env.timing_begin()
        for s in range(T):
            env.random_actions(7, ep * T + s, act, mch)
            env.step(act, mch)
        if ep >= 2 and ep % 2 == 0:
            env.timing_end()
        for acc in (16, 8):
            for grid in GRIDS:
                env.footprint_copy(B * r, B * w, access_bytes=acc, grid=grid, reps=20)
        # (each footprint_copy call queues 10 warm-up launches back to back, then its timed launches with events between them: the
        # trace holds copies of both kinds)
    torch.cuda.synchronize()
    assert bool(env.info[:, 1].all()) and int((env.status & 0x100).sum()) == 0
    print(f"done B={B} read={B * r} written={B * w}")


if __name__ == "__main__":
    main()
