"""The FIRST launch of a kernel in a process (cold instruction cache, waves of a workgroup starting far apart) must be as right
as the thousandth.  Every other GPU test warms the kernels up by what it runs before; these run one forward each in a FRESH
interpreter and compare it with the reference's recorded outputs (tests/golden/encoder_j6m6e2_mid.npz, B = 320) — the machine
actor alone (its GAT kernel is the first launch), the actor pair of a rollout step (job heads + GAT in one launch), and the
job actor alone, on the split-product kernels and on the f32-instruction ones."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch
import mtfjsp_amd
from importlib import import_module
enc_mod = import_module("e2e-mappo-for-mt-fjsp_amd.encoder")
from oracle import encoder_oracle as eo
what, mode = sys.argv[1], int(sys.argv[2])
g = np.load(os.path.join({root!r}, "tests", "golden", "encoder_j6m6e2_mid.npz"))
batch = "b320"
J, M, E, B = [int(x) for x in g[batch + "_meta"]]; T = J * M
ja, ma = eo.split_weights(g)
p = batch + "_s%d_" % int(g[batch + "_steps"][0])
t = lambda x, dt=None: (torch.as_tensor(np.ascontiguousarray(x)).cuda().to(dt) if dt is not None else torch.as_tensor(np.ascontiguousarray(x)).cuda())
enc = enc_mod.Encoder(J, M, B, obs_dtype="f32"); enc.load_weights(ja, ma)
enc.set_product_mode(mode)
col, val = eo.ell_from_dense(g[p + "adj"])
jargs = (t(g[p + "tfea"], torch.float32), t(col.reshape(B * T, 2).astype(np.int32)), t(val.reshape(B * T, 2).astype(np.float32)),
         t(g[p + "cand"].astype(np.int32)), t(g[p + "mask"].astype(np.uint8)), t(g[p + "h_m_in"].astype(np.float32)))
margs = (t(g[p + "mfea1"], torch.float32), t(g[p + "mfea2"], torch.float32), t(g[p + "h_o"].astype(np.float32)), t(g[p + "mmask"].reshape(B, M).astype(np.uint8)))
torch.cuda.synchronize()
worst = 0.0
if what == "machine":
    mprob, h_m, mach_v = enc.machine_actor_forward(*margs)
    torch.cuda.synchronize()
    worst = max(float(np.abs(mprob.cpu().numpy() - g[p + "mch_prob"]).max()), float(np.abs(h_m.cpu().numpy() - g[p + "h_m"]).max()) / max(1.0, float(np.abs(g[p + "h_m"]).max())))
elif what == "job":
    prob, h_o, job_v = enc.job_actor_forward(*jargs)
    torch.cuda.synchronize()
    worst = max(float(np.abs(prob.cpu().numpy() - g[p + "job_prob"]).max()), float(np.abs(h_o.cpu().numpy() - g[p + "h_o"]).max()) / max(1.0, float(np.abs(g[p + "h_o"]).max())))
print("WORST %.3e" % worst)
'''

ROLLOUT_CHILD = r'''
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch
import mtfjsp_amd
from importlib import import_module
rollout = import_module("e2e-mappo-for-mt-fjsp_amd.rollout")
from oracle import encoder_oracle as eo
enc_mod = import_module("e2e-mappo-for-mt-fjsp_amd.encoder")
fused = sys.argv[1] == "fused"
if not fused:
    os.environ["MTFJSP_NO_FUSED_GAT"] = "1"
J, M, E, B = 6, 6, 2, 4096
w = enc_mod.random_init_weights(7)
ro = rollout.Rollout(J, M, E, B, policy="actor", obs_dtype="f32", weights=w, collect=False, greedy=True)
env, e = ro.env, ro.actor.enc
w3 = ro._episode_w3(); env.scaler_reset_returns(); env.reset(w3); ro.actor.begin_episode()
# ONE decision: job actor (its heads launch also runs the GAT passes when fused), machine actor; compare both with the oracle
tf = env.tasks_fea.cpu().numpy(); col = env.ell_col.cpu().numpy().reshape(B, J * M, 2); val = env.ell_val.cpu().numpy().reshape(B, J * M, 2)
cand, mask = env.candidate.cpu().numpy(), env.job_mask.cpu().numpy()
ro.actor.act(env, 0, ro.task, ro.mach, ro.job)
torch.cuda.synchronize()
o = eo.job_actor_forward(w[0], tf, col, val, cand, mask, None, B, J * M)
assert np.array_equal(ro.job.cpu().numpy(), o["prob"].argmax(1)) or (np.sort(o["prob"], 1)[:, -1] - np.sort(o["prob"], 1)[:, -2]).min() < 1e-4
mo = eo.machine_actor_forward(w[1], env.m_fea1.cpu().numpy(), env.m_fea2.cpu().numpy(), e.h_pooled_o.cpu().numpy(), env.mmask.cpu().numpy(), B, M)
worst = max(float(np.abs(e.job_prob.cpu().numpy() - o["prob"]).max()), float(np.abs(e.mch_prob.cpu().numpy() - mo["prob"]).max()),
            float(np.abs(e.h_pooled_m.cpu().numpy() - mo["h_pooled"]).max()) / max(1.0, float(np.abs(mo["h_pooled"]).max())))
print("WORST %.3e" % worst)
'''


def _run(code, *args):
    r = subprocess.run([sys.executable, "-c", code.format(root=ROOT), *[str(a) for a in args]], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [x for x in r.stdout.splitlines() if x.startswith("WORST")][-1]
    return float(line.split()[1])


@pytest.mark.parametrize("mode", [0, 2, 4, 16])
@pytest.mark.parametrize("what", ["machine", "job"])
def test_first_forward_of_a_fresh_process_matches_the_reference(what, mode):
    for _ in range(2):                                          # (the failure this guards against showed on every cold start)
        assert _run(CHILD, what, mode) < 1e-4


@pytest.mark.parametrize("kind", ["fused", "unfused"])
def test_first_rollout_decision_of_a_fresh_process_matches_the_oracle(kind):
    """B = 4096: the shape at which the job heads and the GAT passes share a launch (heads grid == the GAT's own grid)"""
    for _ in range(2):
        assert _run(ROLLOUT_CHILD, kind) < 1e-4
