#!/usr/bin/env python3
"""Diagnostic: the three forms of the first GIN Linear's BatchNorm (MTFJSP_FUSE_GIN0 = 1 moments / 2 statistics-only launch / 0 two launches) against each
other on rollout states: max |difference| of the pooled embedding (relative to its scale), of the probabilities and of the values.
    gpurun -- 'python tools/check_gin0_modes.py'"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import mtfjsp_amd  # noqa
from importlib import import_module
enc_mod = import_module("e2e-mappo-for-mt-fjsp_amd.encoder"); rollout = import_module("e2e-mappo-for-mt-fjsp_amd.rollout")
for (J, M, E, B, steps) in [(10, 10, 2, 333, 37), (10, 10, 2, 2048, 95), (20, 20, 4, 64, 390), (6, 6, 2, 1000, 30)]:
    ja, ma = enc_mod.random_init_weights(seed=J + M)
    os.environ["MTFJSP_NO_RESIDENT_GIN"] = "1"
    ro = rollout.Rollout(J, M, E, B, policy="actor", obs_dtype="f32", weights=(ja, ma), collect=False)
    for _ in range(steps):
        ro.step()
    env = ro.env; hm = ro.actor.enc.h_pooled_m.clone()
    outs = {}
    for mode in ("1", "2", "0"):
        os.environ["MTFJSP_FUSE_GIN0"] = mode
        e = enc_mod.Encoder(J, M, B, obs_dtype="f32"); e.load_weights(ja, ma)
        p, h, v = e.job_actor_forward(env.tasks_fea, env.ell_col, env.ell_val, env.candidate, env.job_mask, hm)
        torch.cuda.synchronize()
        outs[mode] = (p.cpu().numpy().copy(), h.cpu().numpy().copy(), v.cpu().numpy().copy()); e.close()
    del os.environ["MTFJSP_FUSE_GIN0"]
    sc = max(1.0, float(np.abs(outs["0"][1]).max()))
    tf = env.tasks_fea.cpu().numpy()
    for a, b in (("1", "2"), ("2", "0"), ("1", "0")):
        print(f"J{J}M{M}E{E} x {B} after {steps} steps (|features| max {np.abs(tf).max():.0f}): mode {a} vs {b}: pooled {np.abs(outs[a][1]-outs[b][1]).max()/sc:.2e} of scale {sc:.2f}  prob {np.abs(outs[a][0]-outs[b][0]).max():.2e}  value {np.abs(outs[a][2]-outs[b][2]).max():.2e}", flush=True)
