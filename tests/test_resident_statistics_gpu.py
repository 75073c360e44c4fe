"""The single-launch GIN kernel's BatchNorm sums travel as 52-bit fixed-point words with an arrival count (csrc/mtfjsp_gin_resident.h,
gr_fix_encode: 20 fractional bits, a workgroup's contribution below 2^31).  Pinned where that can fail:
  (a) a workgroup's sum of squares in [2^31, f16 overflow): finite z, inside every operand range — the contribution raises the range
      word, check() returns MTFJSP_ERR_RETRY, the repeated forward (streaming f32-instruction kernels) equals the oracle;
  (b) the shipped `top1` checkpoint (the only trained weights) at B = 4096 — 576 rows per workgroup, 16x tighter than the B = 16 of the
      reference fixture — on mid-episode observations: no fallback, outputs at the whole-batch oracle;
  (c) bit-reproducible: two runs of the same forward give the same bits (integer sums are order-independent).
Reference BatchNorm: model/gcn_mlp.py:109-197, 204-249."""
import os
import sys
from importlib import import_module

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from error_budget import check as budget  # noqa: E402

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(ROOT, "tests", "golden")
J, M, E, B = 6, 6, 2, 4096
T = J * M


def _mid_episode(weights, steps=17, seed=3):
    import mtfjsp_amd  # noqa: F401
    rollout = import_module("e2e-mappo-for-mt-fjsp_amd.rollout")
    ro = rollout.Rollout(J, M, E, B, policy="actor", obs_dtype="f32", weights=weights, collect=False, seed=seed)
    for _ in range(steps):
        ro.step()
    torch.cuda.synchronize()
    return ro


def _job_forward(enc, env, hm):
    h_nodes = torch.zeros(B * T, 128, dtype=torch.float32, device="cuda")
    prob, h_o, job_v = enc.job_actor_forward(env.tasks_fea, env.ell_col, env.ell_val, env.candidate, env.job_mask, hm, h_nodes=h_nodes)
    torch.cuda.synchronize()
    return h_nodes, prob.clone(), h_o.clone(), job_v.clone()


def _oracle(ja, env, hm):
    from oracle import encoder_oracle as eo
    return eo.job_actor_forward(ja, env.tasks_fea.cpu().numpy(), env.ell_col.cpu().numpy().reshape(B, T, 2), env.ell_val.cpu().numpy().reshape(B, T, 2),
                                env.candidate.cpu().numpy(), env.job_mask.cpu().numpy(), hm.cpu().numpy(), B, T)


def test_sum_of_squares_beyond_the_fixed_point_range_falls_back_loudly():
    enc_mod = import_module("e2e-mappo-for-mt-fjsp_amd.encoder")
    capi = import_module("e2e-mappo-for-mt-fjsp_amd.capi")
    ja, ma = enc_mod.random_init_weights(seed=12)
    # the second Linear's weight x 4000: z of that layer has |z| ~ 4000 rms (a workgroup's sum of squares ~ 576 * 1.6e7 = 2^33 > 2^31)
    # while every OPERAND stays O(1) (the weight image is rescaled by a power of two, its input is a BatchNorm output, its output is
    # normalised by the next BatchNorm): nothing leaves the f16 range, only the statistics' fixed-point range
    k = "encoder.feature_extract.mlps.0.linears.1.weight"
    ja[k] = (ja[k] * 4000.0).astype(np.float32)
    ro = _mid_episode((ja, ma), steps=0)
    ro.env.scaler_reset_returns(); ro.env.reset(ro._episode_w3()); ro.actor.begin_episode()
    env, enc = ro.env, ro.actor.enc
    assert enc.check()                                            # single-launch kernel in use
    hm = torch.tensor(np.tile(ja["_input"][None, :], (B, 1)), device="cuda")
    _job_forward(enc, env, hm)
    with pytest.raises(capi.MtfjspError) as ei:
        enc.check()
    assert ei.value.code == capi.ERR_RETRY
    n, mode = enc.range_fallbacks()
    assert n == 1 and (mode & 15) == 15
    h_nodes, prob, h_o, job_v = _job_forward(enc, env, hm)        # repeated: streaming f32-instruction kernels, no range limit
    assert not enc.check()
    o = _oracle(ja, env, hm)
    z_scale = max(1.0, float(np.abs(o["h_nodes"]).max()))
    assert np.isfinite(o["h_nodes"]).all()
    assert float(np.abs(h_nodes.cpu().numpy() - o["h_nodes"]).max()) <= 1e-4 * z_scale
    assert float(np.abs(prob.cpu().numpy() - o["prob"]).max()) <= 1e-4


def test_shipped_checkpoint_at_the_headline_batch_stays_inside_the_range_and_on_the_oracle():
    from oracle import encoder_oracle as eo
    g = np.load(os.path.join(GOLDEN, "encoder_j6m6e2_top1.npz"))
    ja, ma = eo.split_weights(g)
    ro = _mid_episode((ja, ma), steps=17)
    env, enc = ro.env, ro.actor.enc
    assert enc.check() and enc.range_fallbacks()[0] == 0 and ro.n_resident_failures == 0
    hm = enc.h_pooled_m.clone()
    h_nodes, prob, h_o, job_v = _job_forward(enc, env, hm)
    assert enc.check() and enc.range_fallbacks()[0] == 0
    o = _oracle(ja, env, hm)
    scale = max(1.0, float(np.abs(o["h_nodes"]).max()))
    case = "whole_batch_oracle:top1_checkpoint:6x6x2x4096:resident"
    budget(case, "h_nodes", h_nodes.cpu().numpy(), o["h_nodes"], 1e-4, scale)
    budget(case, "h_pooled_o", h_o.cpu().numpy(), o["h_pooled"], 1e-4, scale)
    budget(case, "job_prob", prob.cpu().numpy(), o["prob"], 1e-4)
    budget(case, "job_v", job_v.cpu().numpy(), o["job_v"], 1e-3, relative=True)


def test_shipped_checkpoint_over_five_episodes_of_the_headline_rollout_never_leaves_the_fast_kernels():
    """review r5 item 6: the range / overflow fallbacks of the split products and of the fixed-point statistics, and the three-in-one
    launch's exchange, over a whole timed-region-like run with the only TRAINED weights — not one observation"""
    from oracle import encoder_oracle as eo
    g = np.load(os.path.join(GOLDEN, "encoder_j6m6e2_top1.npz"))
    ja, ma = eo.split_weights(g)
    steps = 5 * T
    ro = _mid_episode((ja, ma), steps=steps, seed=5)
    enc = ro.actor.enc
    assert enc.check()                                             # no asynchronous failure word raised
    assert enc.range_fallbacks()[0] == 0 and ro.n_resident_failures == 0
    assert enc.fused_launches() == steps, (enc.fused_launches(), steps)      # every decision took the three-in-one launch
    assert bool(ro.env.info[:, 1].all())                           # ... and the fifth episode ended on every instance
    assert bool(torch.isfinite(enc.mch_prob).all()) and bool(torch.isfinite(ro.env.info).all())


def test_two_runs_give_the_same_bits():
    enc_mod = import_module("e2e-mappo-for-mt-fjsp_amd.encoder")
    ja, ma = enc_mod.random_init_weights(seed=4)
    ro = _mid_episode((ja, ma), steps=9)
    env, enc = ro.env, ro.actor.enc
    assert enc.check()
    hm = enc.h_pooled_m.clone()
    a = _job_forward(enc, env, hm)
    for _ in range(3):
        b = _job_forward(enc, env, hm)
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    # ... and the whole decision (three-in-one launch: the machine nodes' sums are integer as well)
    ro2 = _mid_episode((ja, ma), steps=9)
    for r in (ro, ro2):
        r.actor.act(r.env, r.nsteps, r.task, r.mach, r.job)
    torch.cuda.synchronize()
    assert torch.equal(ro.actor.enc.mch_prob, ro2.actor.enc.mch_prob) and torch.equal(ro.actor.enc.h_pooled_m, ro2.actor.enc.h_pooled_m)
    assert torch.equal(ro.mach, ro2.mach) and torch.equal(ro.task, ro2.task)


def test_the_requests_on_behalf_of_the_heads_launch_change_no_bit(monkeypatch):
    """k_gin_res requests (and drops) one word of every line of the heads launch's weight images in its last phase
    (GinResArgs::warm, MTFJSP_NO_WARM_HEADS=1 switches that off when a handle is created): a rollout with and one without them
    take the same decisions and produce the same bits."""
    enc_mod = import_module("e2e-mappo-for-mt-fjsp_amd.encoder")
    ja, ma = enc_mod.random_init_weights(seed=5)
    ro_a = _mid_episode((ja, ma), steps=9, seed=11)
    monkeypatch.setenv("MTFJSP_NO_WARM_HEADS", "1")
    ro_b = _mid_episode((ja, ma), steps=9, seed=11)
    monkeypatch.delenv("MTFJSP_NO_WARM_HEADS")
    for r in (ro_a, ro_b):
        r.actor.act(r.env, r.nsteps, r.task, r.mach, r.job)
    torch.cuda.synchronize()
    assert torch.equal(ro_a.task, ro_b.task) and torch.equal(ro_a.mach, ro_b.mach)
    assert torch.equal(ro_a.actor.enc.mch_prob, ro_b.actor.enc.mch_prob) and torch.equal(ro_a.actor.enc.h_pooled_m, ro_b.actor.enc.h_pooled_m)
    assert torch.equal(ro_a.env.tasks_fea, ro_b.env.tasks_fea)
