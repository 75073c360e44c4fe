"""Dump the GAT node rows of decision 0 (unfused path: k_gat3x stand-alone, then peek before the machine heads run) for the library in
MTFJSP_LIB to <out.npy>; run once per build and compare with tools/first_launch/nodes_diff.py.
    python tools/first_launch/nodes.py <lib.so> <out.npy> [repeat]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["MTFJSP_LIB"] = os.path.abspath(sys.argv[1])
os.environ["MTFJSP_NO_FUSED_GAT"] = "1"
import numpy as np
import torch
import mtfjsp_amd  # noqa: F401
from importlib import import_module
rollout = import_module("e2e-mappo-for-mt-fjsp_amd.rollout")
enc_mod = import_module("e2e-mappo-for-mt-fjsp_amd.encoder")
capi = import_module("e2e-mappo-for-mt-fjsp_amd.capi")
J, M, E, B = 6, 6, 2, 4096
w = enc_mod.random_init_weights(7)
ro = rollout.Rollout(J, M, E, B, policy="actor", obs_dtype="f32", weights=w, collect=False, greedy=True)
env, e = ro.env, ro.actor.enc
w3 = ro._episode_w3(); env.scaler_reset_returns(); env.reset(w3); ro.actor.begin_episode()
ro.actor.act(env, 0, ro.task, ro.mach, ro.job)          # job heads, m_fea1, GAT, machine heads
torch.cuda.synchronize()
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
outs = []
for r in range(reps):                                    # the GAT launch alone, repeatedly, on the SAME m_fea1 / m_fea2
    L = capi.lib()
    # machine_actor_forward = GAT launch + heads launch; the node buffer still holds the GAT output afterwards (the heads normalise on the fly)
    e.machine_actor_forward(env.m_fea1, env.m_fea2, e.h_pooled_o, env.mmask)
    outs.append(e.peek_nodes())
np.save(sys.argv[2], np.stack(outs))
np.savez(sys.argv[2] + ".inputs.npz", mfea1=env.m_fea1.cpu().numpy(), mfea2=env.m_fea2.cpu().numpy())
print("saved", sys.argv[2], np.stack(outs).shape)
