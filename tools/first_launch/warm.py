"""Warm probe: N rollout decisions in ONE process with the library in MTFJSP_LIB, each compared with the oracle (job actor and
machine actor outputs).  Prints per decision the worst errors and how many instances are off — tells a build that is wrong on every
launch from one that is wrong on the first only.

    python tools/first_launch/warm.py <lib.so> <decisions> [fused|unfused]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
lib = os.path.abspath(sys.argv[1])
os.environ["MTFJSP_LIB"] = lib
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
if len(sys.argv) > 3 and sys.argv[3] != "fused":
    os.environ["MTFJSP_NO_FUSED_GAT"] = "1"

import numpy as np
import torch
import mtfjsp_amd  # noqa: F401
from importlib import import_module
rollout = import_module("e2e-mappo-for-mt-fjsp_amd.rollout")
enc_mod = import_module("e2e-mappo-for-mt-fjsp_amd.encoder")
from oracle import encoder_oracle as eo

J, M, E, B = 6, 6, 2, 4096
w = enc_mod.random_init_weights(7)
ro = rollout.Rollout(J, M, E, B, policy="actor", obs_dtype="f32", weights=w, collect=False, greedy=True)
env, e = ro.env, ro.actor.enc
w3 = ro._episode_w3(); env.scaler_reset_returns(); env.reset(w3); ro.actor.begin_episode()
bad = 0
for it in range(n):
    tf = env.tasks_fea.cpu().numpy(); col = env.ell_col.cpu().numpy().reshape(B, J * M, 2); val = env.ell_val.cpu().numpy().reshape(B, J * M, 2)
    cand, mask = env.candidate.cpu().numpy(), env.job_mask.cpu().numpy()
    hm = e.h_pooled_m.cpu().numpy().copy() if ro.actor.have_hm else None
    ro.actor.act(env, it, ro.task, ro.mach, ro.job)
    torch.cuda.synchronize()
    o = eo.job_actor_forward(w[0], tf, col, val, cand, mask, hm, B, J * M)
    mo = eo.machine_actor_forward(w[1], env.m_fea1.cpu().numpy(), env.m_fea2.cpu().numpy(), e.h_pooled_o.cpu().numpy(), env.mmask.cpu().numpy(), B, M)
    jerr = np.abs(e.job_prob.cpu().numpy() - o["prob"])
    jv = np.abs(e.job_v.cpu().numpy() - o["job_v"]).max()
    perr = np.abs(e.mch_prob.cpu().numpy() - mo["prob"])
    herr = np.abs(e.h_pooled_m.cpu().numpy() - mo["h_pooled"]).max(1) / max(1.0, float(np.abs(mo["h_pooled"]).max()))
    wrong = (jerr.max() > 1e-4) or (perr.max() > 1e-4) or (herr.max() > 1e-4)
    bad += bool(wrong)
    print("decision %2d: job prob %.2e (%4d inst off)  job_v %.1e | mach prob %.2e (%4d inst off)  h_pooled_m %.2e (%4d inst off)%s" % (
        it, jerr.max(), int((jerr.max(1) > 1e-4).sum()), jv, perr.max(), int((perr.max(1) > 1e-4).sum()), herr.max(), int((herr > 1e-4).sum()),
        "   <-- WRONG" if wrong else ""))
    env.step(ro.task, ro.mach)
print("=== %s %s: %d of %d decisions wrong" % (os.path.basename(lib), sys.argv[3] if len(sys.argv) > 3 else "fused", bad, n))
