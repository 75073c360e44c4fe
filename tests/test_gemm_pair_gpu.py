"""k_gemm_x6f (csrc/mtfjsp_gemm_pair.h): the two inner Linears of a GIN MLP (gcn:204-249) in one streaming launch behind a
statistics-only pass — same arithmetic per element as the two separate launches (MTFJSP_FUSE_PAIR=0), so the whole job-actor
forward agrees to the round-off of the BatchNorm sums' f64 atomics; and the fp32 oracle on top."""
import os
import sys
from importlib import import_module

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


def _forward(J, M, E, B, fuse, monkeypatch, steps=5):
    import mtfjsp_amd  # noqa: F401
    monkeypatch.setenv("MTFJSP_FUSE_PAIR", "1" if fuse else "0")
    monkeypatch.setenv("MTFJSP_NO_RESIDENT_GIN", "1")            # the streaming launches also where the single-launch kernel would run
    enc_mod = import_module("e2e-mappo-for-mt-fjsp_amd.encoder")
    rollout = import_module("e2e-mappo-for-mt-fjsp_amd.rollout")
    ja, ma = enc_mod.random_init_weights(seed=31)
    rs = np.random.RandomState(3)
    for k in ja:
        if "batch_norms" in k:
            ja[k] = (rs.uniform(0.5, 1.5, ja[k].shape) if k.endswith("weight") else rs.uniform(-0.5, 0.5, ja[k].shape)).astype(np.float32)
    ro = rollout.Rollout(J, M, E, B, policy="random", obs_dtype="f32", collect=False)
    for _ in range(steps):
        ro.step()
    env = ro.env
    e = enc_mod.Encoder(J, M, B)
    e.load_weights(ja, ma)
    T = J * M
    h_nodes = torch.zeros(B * T, 128, dtype=torch.float32, device="cuda")
    prob, h_o, job_v = e.job_actor_forward(env.tasks_fea, env.ell_col, env.ell_val, env.candidate, env.job_mask, None, h_nodes=h_nodes)
    torch.cuda.synchronize()
    out = dict(h_nodes=h_nodes.cpu().numpy(), h_pooled=h_o.cpu().numpy().copy(), prob=prob.cpu().numpy().copy(), job_v=job_v.cpu().numpy().copy())
    ins = dict(tf=env.tasks_fea.cpu().numpy(), col=env.ell_col.cpu().numpy().reshape(B, T, 2), val=env.ell_val.cpu().numpy().reshape(B, T, 2),
               cand=env.candidate.cpu().numpy(), mask=env.job_mask.cpu().numpy())
    return out, ins, ja


@pytest.mark.parametrize("size", [(10, 10, 2, 96), (20, 20, 4, 9), (6, 6, 2, 333), (5, 7, 1, 40)])
def test_pair_launch_equals_the_two_launches_and_the_oracle(size, monkeypatch):
    from oracle import encoder_oracle as eo
    J, M, E, B = size
    a, ins, ja = _forward(J, M, E, B, True, monkeypatch)
    b, _, _ = _forward(J, M, E, B, False, monkeypatch)
    scale = max(1.0, float(np.abs(b["h_nodes"]).max()))
    for k in ("h_nodes", "h_pooled", "prob", "job_v"):
        d = float(np.abs(a[k] - b[k]).max())
        assert d <= 2e-6 * (scale if k in ("h_nodes", "h_pooled") else max(1.0, float(np.abs(b[k]).max()))), (k, d)
    T = J * M
    o = eo.job_actor_forward(ja, ins["tf"], ins["col"], ins["val"], ins["cand"], ins["mask"], None, B, T)   # (None: the learned `_input`, ac:229-233)
    oscale = max(1.0, float(np.abs(o["h_nodes"]).max()))
    assert float(np.abs(a["h_nodes"] - o["h_nodes"]).max()) <= 1e-4 * oscale
    assert float(np.abs(a["prob"] - o["prob"]).max()) <= 1e-4
