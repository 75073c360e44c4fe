#!/usr/bin/env python3
"""PCIe-inclusive rate of the drop-in `Parallel_env` (host numpy in / host numpy out every step, dense adjacency included,
exactly what an unmodified reference caller sees) — NOT what bench.py's `value` measures (device-resident rollout)."""
import os, sys, time, random
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mtfjsp_amd  # noqa
from importlib import import_module
pe = import_module("e2e-mappo-for-mt-fjsp_amd.parallel_env"); inst = import_module("e2e-mappo-for-mt-fjsp_amd.instances")
J, M, E = 6, 6, 2
T = J * M
for B in (16, 256, 1024):
    args = {"n_job": J, "n_machine": M, "n_edge": E, "env_batch": B, "gcn_input_dim": 12, "GAMMA": 0.99, "weight_mk": 0.4, "weight_ec": 0.4,
            "weight_tt": 0.2, "reward_scaling": {"scaling_divisor": 1}, "mask_value": 1, "m_scaling": 1}
    t, p, tt, edge = inst.generate_instances(B, J, M, E, seed=0)
    env = pe.Parallel_env(args)
    env.get_batch({"t": torch.tensor(t), "p": torch.tensor(p), "transT": torch.tensor(tt), "edge": torch.tensor(edge)})
    env.init_RewardScaling_sameBATCH(4)
    rs = np.random.RandomState(0); random.seed(0)
    feas = t >= 0
    n, t_step = 0, 0.0
    for ep in range(3):
        adj, mf2, tf = env.init_DGFJSPEnv_state0()
        cnt = np.zeros((B, J), int)
        for s in range(T):
            job = np.array([rs.choice(np.flatnonzero(cnt[b] < M)) for b in range(B)])
            task = job * M + cnt[np.arange(B), job]; cnt[np.arange(B), job] += 1
            mach = np.array([rs.choice(np.flatnonzero(feas[b, task[b]])) for b in range(B)])
            t0 = time.perf_counter()
            mm = torch.tensor(~feas[np.arange(B), task][:, None, :])
            mf1 = env.cal_cur_task_machine_feature(torch.tensor(task), mm, tf)
            adj, info, mf2, tf = env.DGFJSPEnv_paral_step(list(zip(task.tolist(), mach.tolist())))
            if ep > 0:
                t_step += time.perf_counter() - t0; n += B
        env.reset_data()
    print(f"B={B}: {n / t_step:.0f} env-steps/s through the host-numpy Parallel_env surface ({t_step / (n / B) * 1e3:.2f} ms per batched step; dense adj {B * T * T * 8 / 1e6:.1f} MB/step)")
