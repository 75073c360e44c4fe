#!/usr/bin/env python3
"""bench.py — env-steps/sec of the batched MT-FJSP hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    N > 1, either form: `python bench.py --gpus N ...` starts its N ranks itself (a child `python -m torch.distributed.run`, one
    process per GPU, before this process has loaded torch or touched a GPU), or the launcher form
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...`.

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

torch = None                    # imported by main(): the cpu_baseline worker process must stay free of torch (and its OpenMP runtime)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
HBM_MEASURED_GBPS = 6290.0


def env_bytes(J, M):
    """algorithmic bytes per env-step of the step kernel."""
    T = J * M
    return 136 * T + 156 * M + 483


def env_bytes_rw(J, M):
    """the read and write halves of env_bytes() (same terms): what a same-footprint streaming kernel must move"""
    T = J * M
    return 64 * T + 100 * M + 176, 72 * T + 56 * M + 307


def same_footprint_copy(env, B, J, M):
    """SURVEY §8(d)'s denominator for the step kernel: a plain streaming kernel that reads and writes the step's algorithmic bytes
    at this batch (mtfjsp_footprint_copy: HIP events around every launch, 40 launches per setting), best grid per access width.
    16-byte accesses are the fastest a copy can do; 8-byte ones are the step kernel's own dominant width (f64 state)."""
    r, w = env_bytes_rw(J, M)
    out = {"read_bytes": B * r, "write_bytes": B * w}
    for acc in (16, 8):
        best = None
        for grid in (256, 512, 1024, 2048, 4096, 8192):
            avg, mn = env.footprint_copy(B * r, B * w, access_bytes=acc, grid=grid, reps=40)
            if best is None or avg < best["avg_launch_us"]:
                best = {"avg_launch_us": avg, "min_launch_us": mn, "grid": grid, "GBps": B * (r + w) / avg / 1e3}
        out[f"access_{acc}B"] = best
    return out


PMC_FILE = "r06_pmc_traffic.json" if os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r06_pmc_traffic.json")) else "r05_pmc_traffic.json"
FOOTPRINT_FILE = "r06_footprint_kernel_only.json"


def footprint_kernel_only(B, J, M, E, kernel):
    """The same-footprint fraction of the step kernel on KERNEL-ONLY durations: rocprofv3 --kernel-trace of the step kernel and of the
    SURVEY 8(d) copy kernel in one process (tools/footprint_kernel_only.sh; committed under profiles/), both launched the same way
    (between two HIP event records).  The fraction measured inside this run (frac_of_same_footprint_copy) pairs two HIP-event timings,
    each of which carries ~2.3 us of event cost: it flatters the step kernel at small batches.  The profiler cannot run inside the
    timed process, so the committed measurement is reported — for the size and the step kernel it was MEASURED on only (advisor r5:
    the file used to be looked up by batch alone, and a J10M10 run quoted the J6M6 kernel's fraction)."""
    try:
        f = json.load(open(os.path.join(ROOT, "profiles", FOOTPRINT_FILE)))
        d = f["batches"][str(B)]
        if f.get("size") != f"{J}x{M}x{E}" or kernel not in d["step_kernel"]:
            return None
        return {"frac": d["frac_of_same_footprint_copy_kernel_only"], "step_kernel": d["step_kernel"], "size": f["size"],
                "step_kernel_us": d["step_idle"]["ns_avg"] / 1e3, "copy_kernel_us": d["copy_16B_idle"]["ns_avg"] / 1e3, "copy_grid": d["copy_16B_idle"]["grid"],
                "step_kernel_us_in_the_rollouts_launch_pattern": (d.get("step_b2b") or {}).get("ns_avg", 0) / 1e3 or None,
                "frac_round5_method": d.get("frac_round5_method"),
                "source": f"profiles/{FOOTPRINT_FILE} (rocprofv3 --kernel-trace, End - Start per dispatch, step and copy launches both between HIP event "
                          "records; frac_round5_method = the copy between events over the step in the rollout's back-to-back pattern, round 5's quotient; "
                          "not re-measured inside this run)"}
    except Exception:
        return None


def pmc_traffic(family, shape=None):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/<PMC_FILE>: FETCH_SIZE and
    WRITE_SIZE collected in separate passes of this same command, gfx950 FETCH_SIZE x2 correction applied).  PMC
    collection cannot run inside the timed process, so bench.py reports the last committed measurement (B=4096 J6M6E2)."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE)))
        d = d["sizes"][shape][family] if shape else d["kernels"][family]      # shape: "J10M10E2_B8192" ... (tools/pmc_sizes.sh passes)
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
    copy = same_footprint_copy(env, B, J, M)
    del env
    torch.cuda.empty_cache()
    kname = (("k_env_grp16" if B <= 8192 else "k_env_grp4") if (T <= 64 and M * M <= 64) else
             ("k_env_grp16x2" if B <= 4096 else "k_env_grp4x2") if (T <= 128 and M * M <= 128 and M <= 16) else "k_env_step_grp")
    ko = footprint_kernel_only(B, J, M, E, kname)
    return {"kernel": kname, "instances": B, "bound": "hbm", "achieved": ach,
            "frac_of_same_footprint_copy_kernel_only": ko["frac"] if ko else None, "kernel_only": ko,
            "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "frac_of_measured_copy_bw": ach / HBM_MEASURED_GBPS,
            "avg_launch_us": sec * 1e6, "launches": n, "env_steps_per_s": B / sec,
            # SURVEY §8(d)'s own denominator: the same bytes through a plain streaming launch at the same batch
            "same_footprint_copy": copy,
            "frac_of_same_footprint_copy": copy["access_16B"]["avg_launch_us"] / (sec * 1e6),
            "frac_of_same_footprint_copy_8B_accesses": copy["access_8B"]["avg_launch_us"] / (sec * 1e6)}


def config_leg(rollout_mod, J, M, E, B, device, steps=240, warm=120):
    """A short leg of another BASELINE.json configuration on this GPU (extra keys of the N=1 line; `value` stays the headline
    configuration's): B distinct instances drawn by the on-device generator (the host stream is python-loop bound at these
    sizes), full rollout step, wall-clock rate over `steps` steps after `warm`, then HIP-event times per kernel family."""
    T = J * M
    be = import_module_("e2e-mappo-for-mt-fjsp_amd.batch_env")
    gen = be.DeviceBatchEnv(J, M, E, B, obs_dtype="f32", device=device)
    gen.generate_instances(seed=2024)
    ins = gen.read_instances()
    del gen
    ro = rollout_mod.Rollout(J, M, E, B, device=device, policy="actor", obs_dtype="f32", instances=ins, collect=True)
    for _ in range(warm):
        ro.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ro.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = min(steps, 2 * T)
    ro.timing_begin()
    for _ in range(prof):
        ro.step()
    kt = ro.timing_end()
    torch.cuda.synchronize()
    ro.check_finished_cleanly()
    dom = max(kt, key=lambda k: kt[k]["ms_total"])
    ke = kt["env_step"]
    sec_env = ke["ms_total"] / max(ke["launches"], 1) * 1e-3
    alg = B * env_bytes(J, M)
    shape = f"J{J}M{M}E{E}_B{B}"
    roof = ro.roofline(dom, kt[dom]) if dom != "env_step" else None
    if roof is not None:
        roof["traffic"] = pmc_traffic(dom, shape)
        roof["traffic_source"] = (f"profiles/{PMC_FILE} sizes.{shape} (rocprofv3 --pmc passes of `bench.py --size {J}x{M}x{E} --batch {B}`, tools/pmc_sizes.sh; "
                                  "not re-measured inside this run)") if roof["traffic"] is not None else None
    out = {"workload": f"J{J}M{M}E{E}, {B} parallel instances (on-device generator, all distinct), full rollout step",
           "value": B * steps / dt, "unit": "env-steps/s", "ms_per_step": dt / steps * 1e3, "steps_timed": steps,
           "roofline": roof,
           "roofline_env_step": {"kernel": ro.env_kernel_name(), "bound": "hbm", "achieved": alg / sec_env / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                 "traffic": pmc_traffic("env_step", shape),
                                 "frac": alg / sec_env / 1e9 / HBM_PEAK_GBPS, "frac_of_measured_copy_bw": alg / sec_env / 1e9 / HBM_MEASURED_GBPS,
                                 "avg_launch_us": sec_env * 1e6, "algorithmic_bytes_per_launch": alg, "env_steps_per_s": B / sec_env},
           "kernel_times_us_per_launch": {k: v["ms_total"] / max(v["launches"], 1) * 1e3 for k, v in kt.items()}}
    del ro
    torch.cuda.empty_cache()
    return out


def import_module_(name):
    from importlib import import_module
    import mtfjsp_amd  # noqa: F401
    return import_module(name)


def full_handoff_leg(rollout_mod, J, M, E, B, device, rank, world, dist=None):
    """The reference's whole rollout -> update hand-off once, outside the timed region (N > 1): a Rollout with the complete
    device trajectory buffer and a (random-init) global critic runs one buffer; its finish_buffer samples the global critic on
    every stored state, runs the 8 GAE scans and exchanges 16 tensors x [S, B] f32 in ONE all-gather.  The ranks first
    agree that every one of them could set the leg up (a rank that could not would leave the others waiting in the collective)."""
    enc_mod = import_module_("e2e-mappo-for-mt-fjsp_amd.encoder")
    ro, err = None, None
    try:
        ro = rollout_mod.Rollout(J, M, E, B, device=device, policy="actor", obs_dtype="f32", instance_seed=0, rank=rank, world=world,
                                 collect="full", weights=enc_mod.random_init_weights(1234, with_critic=True), time_handoff=True)
    except Exception as ex:
        err = repr(ex)
    if dist is not None:
        ok = torch.tensor([0.0 if err else 1.0], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) < 0.5:
            return {"error": err or "another rank could not set the leg up"}
    elif err:
        return {"error": err}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while ro.n_handoffs == 0:
        ro.step()
    torch.cuda.synchronize()
    g = ro.last_gather or {}
    out = {"tensors": 16, "buffer_steps": ro.S, "world": g.get("world"), "allgather_bytes_per_rank": g.get("bytes_per_rank"),
           "allgather_ms": g.get("ms"), "buffer_plus_handoff_seconds": time.perf_counter() - t0,
           "what": "global critic on 2 S stored states + 8 GAE scans + ONE packed all-gather of 8 advantage and 8 value tensors [16,S,B] f32 + normalisation (ppo:628-703)"}
    del ro
    torch.cuda.empty_cache()
    return out


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_topology():
    """(logical CPUs this process may run on, physical cores among them) from /proc/cpuinfo + the affinity mask"""
    try:
        allowed = set(os.sched_getaffinity(0))
    except AttributeError:
        allowed = set(range(os.cpu_count() or 1))
    cores, cur = set(), {}
    try:
        for line in open("/proc/cpuinfo"):
            if ":" in line:
                k, v = [x.strip() for x in line.split(":", 1)]
                cur[k] = v
            elif cur:
                if int(cur.get("processor", -1)) in allowed:
                    cores.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
                cur = {}
        if cur and int(cur.get("processor", -1)) in allowed:
            cores.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
    except (OSError, ValueError):
        pass
    n = len(allowed)
    return n, (len(cores) if cores else n)


def cpu_quota():
    """CPU time this container may use, in CPUs (cgroup v2 cpu.max / v1 cfs quota); None = unlimited.  A pod on a shared host can
    see every hardware thread (nproc) and still be throttled to a fraction of them."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def cpu_baseline(J, M, E, batch=4096, seconds_per_leg=3.0):
    """The CPU oracle port (oracle/mtfjsp_oracle.c) timed on the host cores, SURVEY §8d leg (ii): per env and step what the
    reference's batched step does per env (env.step + RewardScaling + candidate/job-mask update + observation; pe:229-265),
    random valid actions chosen in C, on the GPU workload's own instances (the first `batch` of the same generator stream).
    Envs are independent, so the multi-core legs give every thread a contiguous block of envs for whole episodes
    (or_batch_bench_blocks: thread-local env copies, no per-step barrier).  Legs: one thread at the GPU's batch; all physical
    cores; all hardware threads (when SMT is on); BASELINE config 0's batch (16) on one thread.  ENVIRONMENT ONLY: compare
    with `roofline_env_step.env_steps_per_s`, not with `value` (which also contains both actor forwards)."""
    from oracle.env_oracle import OracleBatch, max_threads
    from importlib import import_module
    inst = import_module("e2e-mappo-for-mt-fjsp_amd.instances")
    nlogical, ncores = cpu_topology()
    cap = max(1, max_threads())
    B = batch
    t, p, tt, edge = inst.generate_instances(B, J, M, E, seed=0)
    w3 = np.full((B, 3), 1.0 / 3)

    def leg(nb, threads):
        threads = max(1, min(threads, cap, nb))
        orc = OracleBatch(t[:nb], p[:nb], tt[:nb], edge[:nb])
        orc.bench_blocks(1, threads, w3[:nb])                                            # warm the caches, start the threads
        n, wall, _ = orc.bench_blocks(8, threads, w3[:nb])                               # calibrate
        episodes = max(1, min(20000, int(seconds_per_leg / max(wall / 8, 1e-6))))
        n, wall, steps = orc.bench_blocks(episodes, threads, w3[:nb])
        return {"env_steps_per_s": n / wall, "env_steps_per_s_step_loops_only": n / steps, "threads": threads, "batch": nb,
                "episodes": episodes, "env_steps": n, "seconds": wall}

    quota = cpu_quota()
    try:
        load1 = float(open("/proc/loadavg").read().split()[0])
    except (OSError, ValueError):
        load1 = None
    one = leg(B, 1)
    phys = leg(B, ncores) if ncores > 1 else dict(one)
    smt = leg(B, nlogical) if nlogical > ncores else None
    # thread counts in between: on a shared or quota-limited host the rate saturates well below the core count — the curve shows where
    curve = {}
    for th in (8, 16, 32, 64):
        if th < ncores:
            curve[str(th)] = leg(B, th)
    c0 = leg(16, 1)
    legs = [one, phys] + ([smt] if smt else []) + list(curve.values())
    best = max(legs, key=lambda d: d["env_steps_per_s"])
    out = {"value": best["env_steps_per_s"], "unit": "env-steps/s", "cores": best["threads"], "kind": "port",
           "cpu_model": cpu_model(), "logical_cpus": nlogical, "physical_cores": ncores, "cgroup_cpu_quota": quota,
           "host_loadavg_1min_before": load1, "thread_scaling": {k: {"env_steps_per_s": v["env_steps_per_s"], "per_thread": v["env_steps_per_s"] / v["threads"]} for k, v in curve.items()},
           "omp": {k: os.environ.get(k) for k in ("OMP_PLACES", "OMP_PROC_BIND")},
           "single_thread": one, "all_cores": phys, "all_smt_threads": smt, "config0_B16_single_thread": c0,
           "all_cores_over_single_thread": phys["env_steps_per_s"] / one["env_steps_per_s"],
           "best_over_single_thread": best["env_steps_per_s"] / one["env_steps_per_s"],
           "best_over_config0_single_thread": best["env_steps_per_s"] / c0["env_steps_per_s"],
           "sample": f"J{J}M{M}E{E}: the first {B} instances of Instance_Dataset(seed=0) (the GPU's batch), {best['episodes']} episodes "
                     f"({best['env_steps']} env-steps, {best['seconds']:.1f} s) in the reported leg; per-episode resets inside the timed region; "
                     "ENVIRONMENT ONLY (step + reward scaling + job mask + ELL observation per env and step, random valid actions; no actor "
                     "forwards) — compare with roofline_env_step.env_steps_per_s; oracle/mtfjsp_oracle.c -O2, every OpenMP thread owns a block "
                     "of envs for whole episodes; the reference's own Python path measured in the build container: "
                     "tests/golden/reference_cpu_speed.txt (about 330 env-steps/s, 1 core)"}
    return out


def cpu_baseline_subprocess(J, M, E, batch):
    """cpu_baseline() in a fresh interpreter WITHOUT torch, started before this process touches the GPU: torch carries its own
    OpenMP runtime, and with thread binding requested the first runtime to initialise pins the main thread to one core, after
    which a second runtime sees only that core (measured: 2 logical CPUs reported on a 128-thread host).  The child pins one
    OpenMP place per physical core (OMP_PLACES=cores, OMP_PROC_BIND=spread unless the caller set them)."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("OMP_PLACES", "cores")
    env.setdefault("OMP_PROC_BIND", "spread")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(J), str(M), str(E), str(batch)],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        return {"error": "cpu_baseline worker failed", "stderr": r.stderr[-2000:]}
    return json.loads(r.stdout.strip().splitlines()[-1])


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: one process per GPU through `python -m torch.distributed.run` on
    127.0.0.1 (free port), started as a child process of this GPU-free parent.  Rank 0's output is relayed line by line; the
    return code is the child's."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")           # dmabuf IPC: what RCCL needs on this pool
    env.setdefault("OMP_NUM_THREADS", "1")
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in p.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    return p.wait()


def main():
    global torch
    if len(sys.argv) >= 6 and sys.argv[1] == "--cpu-baseline-worker":
        J, M, E, batch = [int(x) for x in sys.argv[2:6]]
        print(json.dumps(cpu_baseline(J, M, E, batch=batch)), flush=True)
        return
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
    ap.add_argument("--min-seconds", type=float, default=0.5,
                    help="the timed region repeats blocks of exactly --steps steps until it has lasted at least this long")
    ap.add_argument("--min-warmup-seconds", type=float, default=1.0,
                    help="the untimed warm-up lasts at least this long (extra steps beyond --warmup; 0 = exactly --warmup + 20 probe steps)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-env-sweep", action="store_true", help="skip the step-kernel batch sweep (N=1 only)")
    ap.add_argument("--no-config-legs", action="store_true", help="skip the short legs of BASELINE configs 2 and 4's shard (N=1, headline size only)")
    ap.add_argument("--no-full-handoff", action="store_true", help="skip the untimed full hand-off leg (N>1)")
    ap.add_argument("--weights", default="random", choices=["random", "top1"],
                    help="actor weights: seeded random initialisation (default; throughput does not depend on the values unless a range "
                         "fallback trips, which encoder_status reports) or the reference's shipped J6M6E2 `top1` checkpoint (its tensors "
                         "as arrays in tests/golden/encoder_j6m6e2_top1.npz)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N` (N > 1): this process has not imported torch or touched a GPU; it starts the N ranks as a
        # CHILD torch.distributed.run (never an exec), relays rank 0's JSON line and exits with the child's return code
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:          # checked before the CPU legs and before torch is loaded: a mis-launched run exits at once
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    cpu_leg = None
    if world == 1 and not args.no_cpu_baseline:        # before anything initialises the GPU or loads torch in this process
        Jc, Mc, Ec = [int(x) for x in args.size.split("x")]
        cpu_leg = cpu_baseline_subprocess(Jc, Mc, Ec, args.batch)
    import torch as _torch
    torch = _torch
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU fallback for the product path"
    # MTFJSP_BENCH_ONE_DEVICE=1 (diagnostic, tests/test_bench_two_ranks_gpu.py): every rank on cuda:0 over gloo — exercises the N > 1
    # control flow of this file on a single-GPU box; its numbers mean nothing (the ranks share one GPU)
    one_device = bool(os.environ.get("MTFJSP_BENCH_ONE_DEVICE"))
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if one_device:
            dist.init_process_group("gloo")
        else:
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
    # Instance_Dataset(samples = world*B, seed = 0), shard `rank` owns rows [rank*B, (rank+1)*B) — all distinct
    weights = None
    if args.weights == "top1":
        if (J, M, E) != (6, 6, 2):
            raise SystemExit("--weights top1: the shipped checkpoint is a J6M6E2 model")
        gz = np.load(os.path.join(ROOT, "tests", "golden", "encoder_j6m6e2_top1.npz"))
        weights = ({k[len("w_ja."):]: gz[k] for k in gz.files if k.startswith("w_ja.")}, {k[len("w_ma."):]: gz[k] for k in gz.files if k.startswith("w_ma.")})
    ro = rollout_mod.Rollout(J, M, E, B, device=local_rank, policy=policy, obs_dtype=args.obs,
                             instance_seed=0, rank=rank, world=world, collect="full" if args.trajectory == "full" else True,
                             time_handoff=True, weights=weights)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def agree_max(x):
        if dist is None:
            return x
        tmax = torch.tensor([x], dtype=torch.float64, device="cpu" if one_device else "cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return float(tmax.item())

    t_w0 = time.perf_counter()
    for _ in range(args.warmup):
        ro.step()
    sync()
    # a fresh box / first process needs more than a handful of steps to reach its steady state (lazy module loads, clock
    # ramp): the untimed warm-up is extended to --min-warmup-seconds; the extra step count is agreed across ranks (the
    # hand-off collective fires on a step count)
    warm_steps = args.warmup
    t_probe0 = time.perf_counter()
    for _ in range(20):
        ro.step()
    sync()
    warm_steps += 20
    per_step = agree_max((time.perf_counter() - t_probe0) / 20)
    extra = int(agree_max(float(max(0, int(np.ceil((args.min_warmup_seconds - (time.perf_counter() - t_w0)) / max(per_step, 1e-6)))))))
    extra = min(extra, 20000)
    for _ in range(extra):
        ro.step()
    sync()
    warm_steps += extra
    # timed region: a whole number of blocks of --steps steps, bracketed by barrier + synchronize on both sides, long enough to
    # last --min-seconds (the driver's --steps 20 is 7 ms of work: too short to time).  The block count comes from one untimed
    # calibration block and is agreed across ranks, so every rank times the same number of steps.
    handoffs0 = ro.n_handoffs
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ro.step()
    sync()
    t_block = agree_max(time.perf_counter() - t0)
    blocks = max(1, min(100000, int(np.ceil(1.15 * args.min_seconds / max(t_block, 1e-6)))))   # (margin: the calibration block runs a little slow)
    sync()
    t0 = time.perf_counter()
    for _ in range(blocks * args.steps):
        ro.step()
    sync()
    elapsed_local = time.perf_counter() - t0
    elapsed = agree_max(elapsed_local)
    steps_timed = blocks * args.steps
    per_rank_ms = None
    if dist is not None:                                # every rank's own time for the same timed region (the line's value uses the MAX)
        tl = torch.tensor([elapsed_local / steps_timed * 1e3], dtype=torch.float64, device="cpu" if one_device else "cuda")
        allt = [torch.zeros_like(tl) for _ in range(world)]
        dist.all_gather(allt, tl)
        per_rank_ms = [float(x.item()) for x in allt]
    handoffs = ro.n_handoffs - handoffs0
    # the advantage all-gather of the rollout -> update hand-off (the only collective of the data path) fires once per
    # S = buffer_episodes*T steps; when the timed region was shorter than that (N > 1), run the steps up to the next hand-off
    # now so that the RCCL exchange is exercised and measured (reported separately, not part of `value`)
    gather = ro.last_gather
    if world > 1 and handoffs == 0 and ro.collect:
        while ro.n_handoffs == handoffs0:
            ro.step()
        sync()
        gather = ro.last_gather
    # second pass with HIP events recorded on the launch stream around every kernel launch
    # (kept out of the timed region above: two event records per launch would perturb `value`)
    prof_steps = min(max(args.steps, T), 4 * T)
    ro.timing_begin()
    for _ in range(prof_steps):
        ro.step()
    ktimes = ro.timing_end()
    sync()
    ro.check_finished_cleanly()
    enc_status = None
    if ro.actor is not None:        # a fallback (time-out of the single-launch GIN kernel / f16 range) would be a silent performance cliff: say so
        enc = ro.actor.enc
        n_range, mode = enc.range_fallbacks()
        enc_status = {"gin_single_launch_in_use": bool(enc.check()), "grid_barrier_timeouts": enc.resident_failures(),
                      "range_fallbacks": n_range, "product_mode": mode, "rollout_restarts": ro.n_resident_failures}

    full_handoff = None
    if world > 1 and policy == "actor" and not args.no_full_handoff:
        try:
            del ro.traj
        except AttributeError:
            pass
        # set-up failures are agreed across the ranks inside the leg and reported in the line; an exception while the ranks are
        # stepping towards the collective is fatal on purpose (the launcher then ends every rank — catching it on one rank would
        # leave the others waiting in the all-gather)
        full_handoff = full_handoff_leg(rollout_mod, J, M, E, B, local_rank, rank, world, dist)
    if rank == 0:
        value = world * B * steps_timed / elapsed
        # dominant kernel by measured device time
        dom = max(ktimes, key=lambda k: ktimes[k]["ms_total"])
        kd = ktimes[dom]
        avg_s = kd["ms_total"] / max(kd["launches"], 1) * 1e-3
        headline = (B, J, M) == (4096, 6, 6)
        traffic_note = (f"profiles/{PMC_FILE} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command at B=4096 J6M6E2, "
                        "FETCH_SIZE x2 gfx950 correction; not re-measured inside this run)")

        def env_roof(sec, name):
            alg = B * env_bytes(J, M)
            tr = pmc_traffic("env_step") if headline else None
            d = {"kernel": name, "bound": "hbm", "achieved": alg / sec / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                 "frac": alg / sec / 1e9 / HBM_PEAK_GBPS, "frac_of_measured_copy_bw": alg / sec / 1e9 / HBM_MEASURED_GBPS,
                 "traffic": tr, "traffic_source": traffic_note if tr is not None else None,
                 "avg_launch_us": sec * 1e6, "algorithmic_bytes_per_launch": alg,
                 "algorithmic_bytes_per_env_step": env_bytes(J, M), "env_steps_per_s": B / sec}
            if tr is not None:      # the same launch priced on the bytes the incremental kernel really moves (PMC)
                d["achieved_on_pmc_traffic_GBps"] = tr / sec / 1e9
                d["frac_of_measured_copy_bw_on_pmc_traffic"] = tr / sec / 1e9 / HBM_MEASURED_GBPS
            return d

        if dom == "env_step":
            roof = env_roof(avg_s, ro.env_kernel_name() + " (fused state transition + rewards + scaler + incremental observation + job mask)")
            roof["launches"] = kd["launches"]
        else:
            roof = ro.roofline(dom, kd)
            if headline:
                roof["traffic"] = pmc_traffic(dom)
                roof["traffic_source"] = traffic_note if roof["traffic"] is not None else None
                if roof["traffic"] is not None:            # the HBM side of the same launch on the bytes it really moved (its `frac` is the matrix-core one)
                    roof["achieved_GBps_on_pmc_traffic"] = roof["traffic"] / avg_s / 1e9
                    roof["frac_hbm_on_pmc_traffic"] = roof["traffic"] / avg_s / 1e9 / HBM_PEAK_GBPS
        # the north-star kernel is always reported as well (extra key)
        ke = ktimes["env_step"]
        roof_env = env_roof(ke["ms_total"] / max(ke["launches"], 1) * 1e-3, ro.env_kernel_name())
        out = {
            "metric": "env-steps/sec (batched J%dM%dE%d)" % (J, M, E), "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / steps_timed * 1e3,
            "steps_timed": steps_timed, "timed_blocks": blocks, "timed_seconds": elapsed, "warmup_steps_run": warm_steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64 (environment) / f32 (encoder: f32 storage and accumulation; 128x128 products on the 16-bit matrix cores from split f32 operands, f32-accurate: DESIGN.md §4)", "data": "synthetic",
            "config": {"workload": f"J{J}M{M}E{E}, {B} parallel instances per GPU ({ro.instances_desc}), {ro.describe()}",
                       "instances_per_gpu": B, "obs_dtype": args.obs, "policy": policy, "trajectory": args.trajectory,
                       "parallelism": f"instances sharded over {world} GPU(s); env/encoder path has no collective; one all-gather of advantages per {ro.S}-step buffer (RCCL when world>1)"},
            "roofline": roof, "roofline_env_step": roof_env,
            "handoff": {"buffer_steps": ro.S, "handoffs_in_timed_region": handoffs, "world": (gather or {}).get("world"),
                        "allgather_bytes_per_rank": (gather or {}).get("bytes_per_rank"), "allgather_ms": (gather or {}).get("ms"),
                        "what": "local-critic GAE (4 reverse scans) + ONE packed all-gather of the 4 advantage tensors [4,S,B] f32 + global normalisation"},
            "kernel_times_ms": {k: v for k, v in ktimes.items()}, "kernel_times_steps": prof_steps,
            "encoder_status": enc_status,
            "weights": "seeded random initialisation (encoder.random_init_weights)" if weights is None else "the reference's shipped J6M6E2 top1 checkpoint (tests/golden/encoder_j6m6e2_top1.npz)",
        }
        if per_rank_ms is not None:                        # N > 1: what each rank took for the timed region, so that a first real scaling run can be read
            out["per_rank_ms_per_step"] = per_rank_ms
            out["handoff"]["allgather_GBps_per_rank"] = ((gather or {}).get("bytes_per_rank") or 0) * (world - 1) / max((gather or {}).get("ms") or 1e30, 1e-9) / 1e6 if gather else None
        if full_handoff is not None:
            out["handoff_full"] = full_handoff
        # the two latency chains beside the dominant kernel (round-3 review: "no roofline at all"): matrix time and LDS traffic bound
        heads = {}
        for fam in ("heads", "heads_gat3", "heads_gat3_heads", "gat3"):
            if fam in ktimes and ktimes[fam]["launches"]:
                try:
                    heads[fam] = ro.roofline(fam, ktimes[fam])
                except Exception as ex:
                    heads[fam] = {"error": repr(ex)}
        if heads:
            out["roofline_heads"] = heads
        full_rate = None
        if world == 1 and policy == "actor" and args.trajectory == "advantage" and not args.no_config_legs:
            # the same rollout with the complete device trajectory buffer (every field of the reference's ReplayBuffer):
            # its cost under the driver's clock, as an extra key (`value` stays the advantage-inputs trajectory BASELINE's metric names)
            try:
                del ro.traj
            except AttributeError:
                pass
            ro2 = rollout_mod.Rollout(J, M, E, B, device=local_rank, policy=policy, obs_dtype=args.obs, instance_seed=0, rank=rank, world=world, collect="full")
            for _ in range(2 * T):
                ro2.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            nfull = 10 * T
            for _ in range(nfull):
                ro2.step()
            torch.cuda.synchronize()
            dtf = time.perf_counter() - t0
            full_rate = {"value": B * nfull / dtf, "unit": "env-steps/s", "ms_per_step": dtf / nfull * 1e3, "steps_timed": nfull,
                         "what": "--trajectory full: the device-resident TrajectoryBuffer records every field of the reference's ReplayBuffer per step"}
            del ro2
            torch.cuda.empty_cache()
            out["trajectory_full"] = full_rate
        if world == 1 and not args.no_env_sweep:
            del ro
            torch.cuda.empty_cache()
            sweep = [env_kernel_large_batch(J, M, E, local_rank, B=b, episodes=1 if b >= 65536 else 2) for b in (4096, 16384, 65536, 262144)]
            out["roofline_env_step_batch_sweep"] = sweep
            out["roofline_env_step_large_batch"] = sweep[-1]
            if B == sweep[0]["instances"]:
                # SURVEY 8(d)'s denominator for the step kernel as it runs INSIDE the rollout (the sweep's figure is the kernel alone)
                c16 = sweep[0]["same_footprint_copy"]["access_16B"]["avg_launch_us"]
                out["roofline_env_step"]["same_footprint_copy_us"] = c16
                out["roofline_env_step"]["frac_of_same_footprint_copy"] = c16 / out["roofline_env_step"]["avg_launch_us"]
                ko = footprint_kernel_only(B, J, M, E, out["roofline_env_step"]["kernel"])
                if ko:                                             # the figure to quote: both durations kernel-only (see footprint_kernel_only)
                    out["roofline_env_step"]["frac_of_same_footprint_copy_kernel_only"] = ko["frac"]
                    out["roofline_env_step"]["kernel_only"] = ko
        if world == 1 and headline and policy == "actor" and not args.no_config_legs:
            legs = {}
            for name, (cj, cm, ce, cb) in (("J10M10E2_x8192", (10, 10, 2, 8192)), ("J20M20E4_x2048", (20, 20, 4, 2048))):
                try:
                    legs[name] = config_leg(rollout_mod, cj, cm, ce, cb, local_rank)
                except Exception as ex:
                    legs[name] = {"error": repr(ex)}
            out["configs"] = legs
        if cpu_leg is not None:
            # like for like: the CPU figure is the ENVIRONMENT alone; so is gpu_env_only (the step kernel's own rate, HIP events around the
            # launch, inside the rollout).  `value` of the line is the FULL rollout step (both actor forwards included) — not the pair to compare.
            if "error" not in cpu_leg:
                cpu_leg["gpu_env_only_env_steps_per_s"] = roof_env["env_steps_per_s"]
                cpu_leg["gpu_env_only_over_cpu"] = roof_env["env_steps_per_s"] / cpu_leg["value"]
                cpu_leg["gpu_full_rollout_env_steps_per_s"] = value
                cpu_leg["compare"] = "value (CPU, environment only) with gpu_env_only_env_steps_per_s; gpu_full_rollout_env_steps_per_s is the line's `value` and includes both actor forwards"
            out["cpu_baseline"] = cpu_leg
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
