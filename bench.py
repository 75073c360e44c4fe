#!/usr/bin/env python3
"""bench.py — env-steps/sec of the batched MT-FJSP hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

One "step" = one pass of the hot path over one batch: every one of the B instances on this GPU takes one
(task, machine) decision — policy forward passes (when --policy actor), m_fea1, the fused env step kernel
(state transition + rewards + reward scaling + next observation + candidate/job mask) — with all inputs resident
in HBM.  Episodes end every T steps; the batched reset (new reward weights) is part of the timed region.
Instances shard across GPUs with no data-path collective (weak scaling: B per GPU fixed).

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP events on the launch stream) and
`cpu_baseline` (the C oracle port timed on the host cores; N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
HBM_MEASURED_GBPS = 6290.0


def env_bytes(J, M):
    """algorithmic bytes per env-step of the step kernel."""
    T = J * M
    return 136 * T + 156 * M + 483


def pmc_traffic(family):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/r01_pmc_traffic.json: FETCH_SIZE and
    WRITE_SIZE collected in separate passes of this same command, gfx950 FETCH_SIZE x2 correction applied).  PMC
    collection cannot run inside the timed process, so bench.py reports the last committed measurement (B=4096 J6M6E2)."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))["kernels"][family]
        return d["traffic_bytes"]
    except Exception:
        return None


def env_kernel_large_batch(J, M, E, device, B=262144, episodes=2):
    """The step kernel alone at a batch that fills the chip (at 4096 instances all waves are resident at once and a launch
    lasts as long as ONE wave's dependent chain): on-device random valid actions, HIP events around every step launch."""
    from importlib import import_module
    be = import_module("e2e-mappo-for-mt-fjsp_amd.batch_env")
    T = J * M
    env = be.DeviceBatchEnv(J, M, E, B, obs_dtype="f32", device=device)
    env.generate_instances(seed=123)                            # on-device generator: nothing to upload
    env.scaler_init()
    w3 = torch.full((B, 3), 1.0 / 3, dtype=torch.float64, device=env.device)
    a = torch.zeros(B, dtype=torch.int32, device=env.device); m = torch.zeros_like(a)
    for ep in range(episodes + 1):
        if ep == 1:
            torch.cuda.synchronize(); env.timing_begin()
        env.reset(w3)
        for s in range(T):
            env.random_actions(7, ep * T + s, a, m)
            env.step(a, m)
    torch.cuda.synchronize()
    ms, n = env.timing_end()
    assert bool(env.info[:, 1].all()) and int((env.status & 0x100).sum()) == 0
    sec = ms / n * 1e-3
    ach = B * env_bytes(J, M) / sec / 1e9
    del env
    torch.cuda.empty_cache()
    return {"kernel": "k_env_reg" if (T <= 64 and M * M <= 64) else "k_env_step", "instances": B, "bound": "hbm", "achieved": ach,
            "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "frac_of_measured_copy_bw": ach / HBM_MEASURED_GBPS,
            "avg_launch_us": sec * 1e6, "launches": n, "env_steps_per_s": B / sec}


def cpu_baseline(J, M, E, seconds_target=12.0):
    """Time the CPU oracle port (oracle/mtfjsp_oracle.c, scalar C, 1 core) on a bounded sample of the same workload."""
    from oracle.env_oracle import OracleBatch
    from importlib import import_module
    inst = import_module("e2e-mappo-for-mt-fjsp_amd.instances")
    B, T = 256, J * M
    t, p, tt, edge = inst.generate_instances(B, J, M, E, seed=0)
    orc = OracleBatch(t, p, tt, edge)
    orc.scaler_init()
    rs = np.random.RandomState(0)
    feas = t >= 0
    w3 = np.full((B, 3), 1.0 / 3)
    n = 0
    t_step = 0.0
    t_start = time.perf_counter()
    episodes = 0
    while time.perf_counter() - t_start < seconds_target:
        orc.reset(w3)
        cand, mask = orc.job_mask_state()
        for s in range(T):
            # uniform random valid action (host side, not timed)
            job = np.array([rs.choice(np.flatnonzero(mask[b] == 0)) for b in range(B)], np.int32)
            task = cand[np.arange(B), job].astype(np.int32)
            mach = np.array([rs.choice(np.flatnonzero(feas[b, task[b]])) for b in range(B)], np.int32)
            t0 = time.perf_counter()
            orc.step(task, mach)
            cand, mask = orc.job_mask_update(job)
            orc.observe(dense=False)
            t_step += time.perf_counter() - t0
            n += B
        episodes += 1
    return {"value": n / t_step, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"J{J}M{M}E{E}, B=256, {episodes} episodes ({n} env-steps), step+reward scaling+job mask+observe (ELL adj), "
                      f"random valid actions; oracle/mtfjsp_oracle.c -O2 scalar"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=720)
    ap.add_argument("--warmup", type=int, default=360)
    ap.add_argument("--batch", type=int, default=4096, help="instances per GPU")
    ap.add_argument("--size", default="6x6x2")
    ap.add_argument("--policy", default="auto", choices=["auto", "actor", "random"])
    ap.add_argument("--obs", default="f32", choices=["f32", "f64"])
    ap.add_argument("--trajectory", default="advantage", choices=["advantage", "full"],
                    help="what the rollout records per step: the advantage inputs (rewards, dones, critic values) or every "
                         "field of the reference's ReplayBuffer (device-resident TrajectoryBuffer, SURVEY 8f N2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-env-sweep", action="store_true", help="skip the chip-filling step-kernel measurement (N=1 only)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N>1 launch with python -m torch.distributed.run --nproc-per-node N (one process per GPU)")
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU fallback for the product path"
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from importlib import import_module
    import mtfjsp_amd  # noqa: F401
    rollout_mod = import_module("e2e-mappo-for-mt-fjsp_amd.rollout")
    J, M, E = [int(x) for x in args.size.split("x")]
    T = J * M
    B = args.batch
    policy = args.policy
    if policy == "auto":
        policy = "actor" if rollout_mod.actor_available() else "random"
    ro = rollout_mod.Rollout(J, M, E, B, device=local_rank, policy=policy, obs_dtype=args.obs,
                             instance_seed=rank, rank=rank, world=world, collect="full" if args.trajectory == "full" else True)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        ro.step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ro.step()
    sync()
    elapsed = time.perf_counter() - t0
    # second pass of the same steps with HIP events recorded on the launch stream around every kernel launch
    # (kept out of the timed region above: two event records per launch would perturb `value`)
    prof_steps = min(args.steps, 4 * T)
    ro.timing_begin()
    for _ in range(prof_steps):
        ro.step()
    ktimes = ro.timing_end()
    sync()
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ro.check_finished_cleanly()

    if rank == 0:
        value = world * B * args.steps / elapsed
        # dominant kernel by measured device time
        dom = max(ktimes, key=lambda k: ktimes[k]["ms_total"])
        kd = ktimes[dom]
        avg_s = kd["ms_total"] / max(kd["launches"], 1) * 1e-3
        if dom == "env_step":
            achieved = B * env_bytes(J, M) / avg_s / 1e9
            roof = {"kernel": ro.env_kernel_name() + " (fused state transition + rewards + scaler + incremental observation + job mask)",
                    "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBPS, "frac_of_measured_copy_bw": achieved / HBM_MEASURED_GBPS,
                    "traffic": pmc_traffic("env_step") if (B, J, M) == (4096, 6, 6) else None,
                    "avg_launch_us": avg_s * 1e6, "launches": kd["launches"],
                    "algorithmic_bytes_per_launch": B * env_bytes(J, M)}
        else:
            roof = ro.roofline(dom, kd)
            if (B, J, M) == (4096, 6, 6):
                roof["traffic"] = pmc_traffic(dom)
        # the north-star kernel is always reported as well (extra key)
        ke = ktimes["env_step"]
        es = ke["ms_total"] / max(ke["launches"], 1) * 1e-3
        roof_env = {"kernel": ro.env_kernel_name(), "bound": "hbm", "achieved": B * env_bytes(J, M) / es / 1e9, "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": B * env_bytes(J, M) / es / 1e9 / HBM_PEAK_GBPS,
                    "frac_of_measured_copy_bw": B * env_bytes(J, M) / es / 1e9 / HBM_MEASURED_GBPS,
                    "traffic": pmc_traffic("env_step") if (B, J, M) == (4096, 6, 6) else None, "avg_launch_us": es * 1e6}
        out = {
            "metric": "env-steps/sec (batched J%dM%dE%d)" % (J, M, E), "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64 (environment) / f32 (encoder: f32 storage and accumulation; GIN products as exact 3-way bf16 splits on the matrix cores, f32-accurate)", "data": "synthetic",
            "config": {"workload": f"J{J}M{M}E{E}, {B} parallel instances per GPU, {ro.describe()}",
                       "instances_per_gpu": B, "obs_dtype": args.obs, "policy": policy, "trajectory": args.trajectory,
                       "parallelism": f"instances sharded over {world} GPU(s); env/encoder path has no collective; one all-gather of advantages per {ro.S}-step buffer (RCCL when world>1)"},
            "roofline": roof, "roofline_env_step": roof_env,
            "kernel_times_ms": {k: v for k, v in ktimes.items()}, "kernel_times_steps": prof_steps,
        }
        if world == 1 and not args.no_env_sweep:
            del ro
            torch.cuda.empty_cache()
            out["roofline_env_step_large_batch"] = env_kernel_large_batch(J, M, E, local_rank)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(J, M, E)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
