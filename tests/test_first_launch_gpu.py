"""The FIRST launch of a kernel in a process (cold instruction cache, waves of a workgroup starting far apart) must be as right
as the thousandth.  Every other GPU test warms the kernels up by what it runs before; these run one forward each in a FRESH
interpreter and compare it with the reference's recorded outputs (tests/golden/encoder_j6m6e2_mid.npz, B = 320) — the machine
actor alone (its GAT kernel is the first launch), the actor pair of a rollout step (job heads + GAT in one launch), and the
job actor alone, on the split-product kernels and on the f32-instruction ones.

Round 4: the miscomputation these tests were written to watch (round 3: the heads / GAT statements as __forceinline__ functions) was
root-caused — a packed-f32 instruction form that gfx950 executes unreliably next to matrix instructions, produced by hipcc's SLP
vectoriser in that schedule (tools/isa_lint.py) — and it was never a first-launch effect: a few node rows per launch
were wrong on EVERY launch.  The function form is therefore built (-DMTFJSP_BODY_FUNCS=3) and tested too, cold and warm, and the
rollout decision is checked at the BASELINE sizes of configs 2 and 4 and with the environment step fused into the heads' launch."""
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
J, M, E, B = [int(x) for x in sys.argv[2:6]]
decisions = int(sys.argv[6]) if len(sys.argv) > 6 else 1
w = enc_mod.random_init_weights(7)
ro = rollout.Rollout(J, M, E, B, policy="actor", obs_dtype="f32", weights=w, collect=False, greedy=True)
env, e = ro.env, ro.actor.enc
w3 = ro._episode_w3(); env.scaler_reset_returns(); env.reset(w3); ro.actor.begin_episode()
worst = 0.0
for it in range(decisions):
    # ONE decision: job actor (its heads launch also runs the GAT passes when fused), machine actor; compare both with the oracle
    tf = env.tasks_fea.cpu().numpy(); col = env.ell_col.cpu().numpy().reshape(B, J * M, 2); val = env.ell_val.cpu().numpy().reshape(B, J * M, 2)
    cand, mask = env.candidate.cpu().numpy(), env.job_mask.cpu().numpy()
    hm = e.h_pooled_m.cpu().numpy().copy() if ro.actor.have_hm else None
    mf2 = env.m_fea2.cpu().numpy().copy()                    # (the machine actor's input: with the step fused into its launch, env.m_fea2 is the NEXT state's afterwards)
    stepped = ro.actor.act(env, it, ro.task, ro.mach, ro.job, env_step=() if os.environ.get("MTFJSP_FUSED_ENV") else None)
    torch.cuda.synchronize()
    o = eo.job_actor_forward(w[0], tf, col, val, cand, mask, hm, B, J * M)
    top2 = np.sort(o["prob"], 1)[:, -2:]
    clear = top2[:, 1] - top2[:, 0] > 2e-4
    assert np.array_equal(ro.job.cpu().numpy()[clear], o["prob"].argmax(1)[clear])
    mo = eo.machine_actor_forward(w[1], env.m_fea1.cpu().numpy(), mf2, e.h_pooled_o.cpu().numpy(), env.mmask.cpu().numpy(), B, M)
    worst = max(worst, float(np.abs(e.job_prob.cpu().numpy() - o["prob"]).max()), float(np.abs(e.mch_prob.cpu().numpy() - mo["prob"]).max()),
                float(np.abs(e.h_pooled_m.cpu().numpy() - mo["h_pooled"]).max()) / max(1.0, float(np.abs(mo["h_pooled"]).max())))
    if not stepped:
        env.step(ro.task, ro.mach)
torch.cuda.synchronize()
assert int((env.status & 0x100).sum()) == 0
print("WORST %.3e" % worst)
'''


def _funcs_lib():
    """the __forceinline__-function form of the heads / GAT kernel bodies (diagnostic build; built on demand)"""
    sys.path.insert(0, ROOT)
    import mtfjsp_amd  # noqa: F401
    from importlib import import_module
    b = import_module("e2e-mappo-for-mt-fjsp_amd._build")
    path = os.path.join(b.PKG, "libmtfjsp_funcs.so")
    if not os.path.exists(path) or any(os.path.getmtime(os.path.join(b.CSRC, f)) > os.path.getmtime(path) for f in os.listdir(b.CSRC)):
        path = b.build_variant("funcs", ["-DMTFJSP_BODY_FUNCS=3"])
    return path


def _run(code, *args, env=None):
    r = subprocess.run([sys.executable, "-c", code.format(root=ROOT), *[str(a) for a in args]], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900,
                       env=dict(os.environ, **(env or {})))
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
        assert _run(ROLLOUT_CHILD, kind, 6, 6, 2, 4096) < 1e-4


@pytest.mark.parametrize("size", [(10, 10, 2, 8192), (20, 20, 4, 2048)], ids=["J10M10E2x8192", "J20M20E4x2048"])
def test_first_rollout_decision_at_the_larger_baseline_sizes(size):
    """BASELINE configs 2 and 4's shard: streaming GIN launches, heads in one or two chunks, stand-alone GAT launch, the two-slot
    register step kernel / the grouped LDS step kernel — all as the first launches of a process"""
    assert _run(ROLLOUT_CHILD, "fused", *size) < 1e-4


def test_first_rollout_decisions_with_the_environment_step_in_the_heads_launch():
    """MTFJSP_FUSED_ENV=1 (three launches per step): first decisions of a fresh process, the step riding in the machine heads' launch"""
    assert _run(ROLLOUT_CHILD, "fused", 6, 6, 2, 4096, 3, env={"MTFJSP_FUSED_ENV": "1"}) < 1e-4


@pytest.mark.parametrize("kind", ["fused", "unfused"])
def test_function_form_of_the_kernel_bodies_is_right_cold_and_warm(kind):
    """the build round 3 had to avoid: -DMTFJSP_BODY_FUNCS=3.  Three cold starts of four decisions each (round 4's probes:
    tools/first_launch/warm.py showed 4 of 4 decisions wrong before the fix — 10-30 instances per launch; 20 cold starts after it:
    profiles/r04_first_launch_function_form.txt)"""
    lib = _funcs_lib()
    for _ in range(3):
        assert _run(ROLLOUT_CHILD, kind, 6, 6, 2, 4096, 4, env={"MTFJSP_LIB": lib}) < 1e-4
