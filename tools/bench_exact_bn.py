#!/usr/bin/env python3
"""What reference-equivalent BatchNorm over shards costs (review r5 item 8a): the headline rollout on ONE GPU with every BatchNorm's
column sums handed to the all-reduce callback (`exact_bn`: Encoder.set_stats_reduce + deferred poll, exactly what Rollout arms for
world > 1) on a ONE-rank RCCL group — the streaming GIN launches, the seven stream synchronisations and the seven collective calls
per forward are all there; only the wire time of a 2 KB all-reduce between GPUs is missing (xGMI: a few us each).
    gpurun -- 'python tools/bench_exact_bn.py'  ->  one JSON line"""
import json
import os
import sys
import time

import torch
import torch.distributed as td

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mtfjsp_amd  # noqa
from importlib import import_module

rollout = import_module("e2e-mappo-for-mt-fjsp_amd.rollout")
D = import_module("e2e-mappo-for-mt-fjsp_amd.dist")


def run(exact, steps=720, B=4096):
    ro = rollout.Rollout(6, 6, 2, B, policy="actor", obs_dtype="f32", collect=True, seed=1)
    if exact:
        cb = D.bn_stats_allreduce()
        ro.actor.enc.set_stats_reduce(cb, B)
        ro.actor.enc.set_deferred_poll(True)
        ro.exact_bn = True
    for _ in range(180):
        ro.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ro.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ro.timing_begin()
    for _ in range(72):
        ro.step()
    kt = ro.timing_end()
    return {"ms_per_step": dt / steps * 1e3, "env_steps_per_s": B * steps / dt,
            "kernel_us_per_launch": {k: round(v["ms_total"] / max(v["launches"], 1) * 1e3, 1) for k, v in kt.items()},
            "launches_per_step": {k: round(v["launches"] / 72, 2) for k, v in kt.items()}}


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29731")
    td.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    D.COLLECT_ON_ONE_RANK = True
    out = {"what": "J6M6E2 x 4096 rollout on one MI355X: default (per-shard statistics, k_gin_res + three-in-one launch) against exact_bn "
                   "(statistics all-reduced between the streaming launches; one-rank RCCL group: every call and synchronisation, no wire time)",
           "default": run(False), "exact_bn": run(True)}
    out["exact_bn_over_default"] = out["exact_bn"]["ms_per_step"] / out["default"]["ms_per_step"]
    print(json.dumps(out))
    td.destroy_process_group()


if __name__ == "__main__":
    main()
