"""Cold-start probe for the fused job-heads + GAT launch (tests/test_first_launch_gpu.py): runs ONE rollout decision
at B = 4096 as the first launches of a fresh interpreter, with the library given in MTFJSP_LIB, and reports where the machine
path's outputs differ from the oracle — as (workgroup, GAT row tile, wave, round) so that a pattern shows.

    python tools/first_launch/repro.py <lib.so> <cold starts> [fused|unfused]
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CHILD = r'''
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch
import mtfjsp_amd
from importlib import import_module
rollout = import_module("e2e-mappo-for-mt-fjsp_amd.rollout")
from oracle import encoder_oracle as eo
enc_mod = import_module("e2e-mappo-for-mt-fjsp_amd.encoder")
J, M, E, B = 6, 6, 2, 4096
w = enc_mod.random_init_weights(7)
ro = rollout.Rollout(J, M, E, B, policy="actor", obs_dtype="f32", weights=w, collect=False, greedy=True)
env, e = ro.env, ro.actor.enc
w3 = ro._episode_w3(); env.scaler_reset_returns(); env.reset(w3); ro.actor.begin_episode()
tf = env.tasks_fea.cpu().numpy(); col = env.ell_col.cpu().numpy().reshape(B, J * M, 2); val = env.ell_val.cpu().numpy().reshape(B, J * M, 2)
cand, mask = env.candidate.cpu().numpy(), env.job_mask.cpu().numpy()
ro.actor.act(env, 0, ro.task, ro.mach, ro.job)
torch.cuda.synchronize()
mo = eo.machine_actor_forward(w[1], env.m_fea1.cpu().numpy(), env.m_fea2.cpu().numpy(), e.h_pooled_o.cpu().numpy(), env.mmask.cpu().numpy(), B, M)
perr = np.abs(e.mch_prob.cpu().numpy() - mo["prob"])                       # [B, M]
herr = np.abs(e.h_pooled_m.cpu().numpy() - mo["h_pooled"]).max(1) / max(1.0, float(np.abs(mo["h_pooled"]).max()))
print("WORST prob %.3e h_pooled %.3e" % (perr.max(), herr.max()))
bad_inst = np.nonzero((perr.max(1) > 1e-4) | (herr > 1e-4))[0]
print("BAD instances %d of %d" % (len(bad_inst), B))
tiles = {{}}
for b in bad_inst:
    for m in range(M):
        if perr[b, m] > 1e-4:
            lm = (b % 16) * M + m
            key = (int(b // 16), int(lm // 8))
            tiles[key] = tiles.get(key, 0) + 1
for (wg, tile), n in sorted(tiles.items())[:60]:
    print("  workgroup %4d  tile %2d (wave %d, round %d): %d machine rows off" % (wg, tile, tile % 8, tile // 8, n))
# second decision of the same process (warm): must be right
ro.actor.act(env, 1, ro.task, ro.mach, ro.job)
torch.cuda.synchronize()
mo2 = eo.machine_actor_forward(w[1], env.m_fea1.cpu().numpy(), env.m_fea2.cpu().numpy(), e.h_pooled_o.cpu().numpy(), env.mmask.cpu().numpy(), B, M)
print("SECOND prob %.3e" % np.abs(e.mch_prob.cpu().numpy() - mo2["prob"]).max())
'''


def main():
    lib = os.path.abspath(sys.argv[1])
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    kind = sys.argv[3] if len(sys.argv) > 3 else "fused"
    env = dict(os.environ, MTFJSP_LIB=lib)
    if kind != "fused":
        env["MTFJSP_NO_FUSED_GAT"] = "1"
    bad = 0
    for i in range(n):
        r = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        out = r.stdout.strip().splitlines()
        print(f"--- {os.path.basename(lib)} {kind} cold start {i}: rc {r.returncode}")
        for line in out:
            print("   ", line)
        if r.returncode != 0:
            print(r.stderr[-1500:])
        w = [x for x in out if x.startswith("WORST")]
        if not w or float(w[0].split()[2]) > 1e-4 or float(w[0].split()[4]) > 1e-4:
            bad += 1
    print(f"=== {os.path.basename(lib)} {kind}: {bad} of {n} cold starts wrong")


if __name__ == "__main__":
    main()
