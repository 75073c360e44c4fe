"""k_headsx_gat3x_headsx — job heads + the machine path's GAT passes + the machine heads in ONE launch, the BatchNorm sums of the
B*M machine nodes (ac:434) exchanged between the workgroups inside the launch — against the separate launches on identical inputs
and against the fp32 oracle of the machine actor (ac:359-498, gat:82-159).  The three-in-one launch needs one workgroup of 16
instances per CU: the headline shape (J6M6E2 x 4096)."""
import os
import sys
from importlib import import_module

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu
J, M, E, B = 6, 6, 2, 4096


def _rollout(seed=77, scale_fcl=1.0, shape=None, obs_dtype="f32", **kw):
    import mtfjsp_amd  # noqa: F401
    enc_mod = import_module("e2e-mappo-for-mt-fjsp_amd.encoder")
    rollout = import_module("e2e-mappo-for-mt-fjsp_amd.rollout")
    ja, ma = enc_mod.random_init_weights(seed=seed)
    rs = np.random.RandomState(seed)
    for d in (ja, ma):
        for k in d:
            if "batch_norms" in k or k.startswith("bn."):
                d[k] = (rs.uniform(0.5, 1.5, d[k].shape) if k.endswith("weight") else rs.uniform(-0.5, 0.5, d[k].shape)).astype(np.float32)
    if scale_fcl != 1.0:
        ma["m_fea_1_fcl.weight"] = (ma["m_fea_1_fcl.weight"] * scale_fcl).astype(np.float32)
        ma["m_fea_2_fcl.weight"] = (ma["m_fea_2_fcl.weight"] * scale_fcl).astype(np.float32)
    j, m, e = (tuple(shape) + (E,))[:3] if shape else (J, M, E)
    ro = rollout.Rollout(j, m, e, B, policy="actor", obs_dtype=obs_dtype, weights=(ja, ma), collect=False, **kw)
    return ro, ja, ma


def _one_decision(ro):
    """one joint decision WITHOUT the env step -> the machine forward's inputs stay in place"""
    e = ro.actor.enc
    n0 = e.fused_launches()
    ro.actor.act(ro.env, ro.nsteps, ro.task, ro.mach, ro.job)
    torch.cuda.synchronize()
    return e.fused_launches() - n0


@pytest.mark.parametrize("steps_before", [0, 1, 17])
def test_three_in_one_launch_equals_the_separate_launches_and_the_oracle(steps_before):
    from oracle import encoder_oracle as eo
    ro, ja, ma = _rollout()
    for _ in range(steps_before):
        ro.step()
    if steps_before == 0:                                         # an episode's first decision: the learned `_input` in place of the machine embedding
        ro.env.scaler_reset_returns(); ro.env.reset(ro._episode_w3()); ro.actor.begin_episode()
    env, e = ro.env, ro.actor.enc
    assert e.check()                                              # single-launch GIN kernel in use: the census passed
    assert _one_decision(ro) == 1                                 # the decision took the three-in-one launch
    fused = [x.clone() for x in (e.mch_prob, e.h_pooled_m, e.mach_v, ro.mach, ro.actor.mch_logp)]
    # the same machine forward through the separate launches (k_gat3x + k_headsx), same inputs, same Philox counter
    e.arm_selection(1, ro.actor.greedy, ro.actor.seed, 2 * ro.nsteps + 1, ro.mach, ro.actor.mch_logp)
    mprob, h_m, mach_v = e.machine_actor_forward(env.m_fea1, env.m_fea2, e.h_pooled_o, env.mmask)
    torch.cuda.synchronize()
    sep = [x.clone() for x in (mprob, h_m, mach_v, ro.mach, ro.actor.mch_logp)]
    scale = max(1.0, float(sep[1].abs().max()))
    assert float((fused[0] - sep[0]).abs().max()) <= 2e-6
    assert float((fused[1] - sep[1]).abs().max()) <= 2e-6 * scale
    assert float((fused[2] - sep[2]).abs().max()) <= 2e-5 * max(1.0, float(sep[2].abs().max()))
    same = float((fused[3] == sep[3]).float().mean())
    assert same >= 0.999, same                                    # (a sampled index can only differ where the cumulative probabilities are within round-off of the draw)
    # ... and the oracle of the machine actor on the inputs the launch used
    mo = eo.machine_actor_forward(ma, env.m_fea1.cpu().numpy(), env.m_fea2.cpu().numpy(), e.h_pooled_o.cpu().numpy(), env.mmask.cpu().numpy(), B, M)
    mscale = max(1.0, float(np.abs(mo["h_pooled"]).max()))
    assert float(np.abs(fused[0].cpu().numpy() - mo["prob"]).max()) <= 1e-4
    assert float(np.abs(fused[1].cpu().numpy() - mo["h_pooled"]).max()) <= 1e-4 * mscale
    assert float(np.abs(fused[2].cpu().numpy() - mo["mach_v"]).max()) <= 1e-3 * max(1.0, float(np.abs(mo["mach_v"]).max()))


@pytest.mark.parametrize("shape,obs_dtype", [((6, 6), "f64"), ((4, 8), "f32"), ((4, 8), "f64"), ((7, 5, 1), "f32"), ((8, 4), "f32")])
def test_three_in_one_launch_with_f64_observations_and_eight_machines(shape, obs_dtype):
    """The inputs the GAT part finds staged in LDS (m_fea2 rows copied and converted by the idle waves, m_fea1 rows left there by the job
    selection) for the other observation dtype and for the widest machine count the launch takes (M = 8: 128 machine rows per workgroup);
    M = 5: ten GAT tiles per workgroup (two waves take a pair, six a single tile); M = 4: eight (the one-tile-at-a-time loop)."""
    from oracle import encoder_oracle as eo
    ro, ja, ma = _rollout(shape=shape, obs_dtype=obs_dtype)
    j, m = shape[:2]
    for _ in range(3):
        ro.step()
    env, e = ro.env, ro.actor.enc
    assert e.check()
    assert _one_decision(ro) == 1
    fused = [x.clone() for x in (e.mch_prob, e.h_pooled_m, e.mach_v, ro.mach)]
    e.arm_selection(1, ro.actor.greedy, ro.actor.seed, 2 * ro.nsteps + 1, ro.mach, ro.actor.mch_logp)
    mprob, h_m, mach_v = e.machine_actor_forward(env.m_fea1, env.m_fea2, e.h_pooled_o, env.mmask)
    torch.cuda.synchronize()
    scale = max(1.0, float(h_m.abs().max()))
    assert float((fused[0] - mprob).abs().max()) <= 2e-6
    assert float((fused[1] - h_m).abs().max()) <= 2e-6 * scale
    assert float((fused[3] == ro.mach).float().mean()) >= 0.999
    mo = eo.machine_actor_forward(ma, env.m_fea1.cpu().numpy().astype(np.float32), env.m_fea2.cpu().numpy().astype(np.float32), e.h_pooled_o.cpu().numpy(),
                                  env.mmask.cpu().numpy(), B, m)
    assert float(np.abs(fused[0].cpu().numpy() - mo["prob"]).max()) <= 1e-4
    assert float(np.abs(fused[1].cpu().numpy() - mo["h_pooled"]).max()) <= 1e-4 * max(1.0, float(np.abs(mo["h_pooled"]).max()))


@pytest.mark.parametrize("limit", ["100", "5000"])
def test_contributions_beyond_the_fine_range_use_the_wide_words(monkeypatch, limit):
    """A contribution beyond the fine words' range (2^31: a workgroup's sum of squares of 96 node rows with |z| rms > 4 700 — the f16
    range of the split products leaves little room above that) travels in the wide-range words.  MTFJSP_XCHG_FINE_LIMIT lowers the
    limit so that ordinary contributions take that way — every sum of squares and most sums at 100, about half of the sums of squares
    at 5000 (both sets in use for one column): the counts of the two sets add up, the decoded totals agree with the separate
    launches' f64 sums, nothing falls back.  (With the limit lowered the wide words' 6 fractional bits are coarse for these small
    magnitudes — at the real limit they resolve 2^-37 of a contribution — hence the looser tolerance: a protocol error would be O(1).)"""
    monkeypatch.setenv("MTFJSP_XCHG_FINE_LIMIT", limit)
    ro, ja, ma = _rollout()
    for _ in range(9):
        ro.step()
    env, e = ro.env, ro.actor.enc
    assert _one_decision(ro) == 1
    fused = [x.clone() for x in (e.mch_prob, e.h_pooled_m, e.mach_v)]
    mprob, h_m, mach_v = e.machine_actor_forward(env.m_fea1, env.m_fea2, e.h_pooled_o, env.mmask)
    torch.cuda.synchronize()
    # (the node rows: of the SEPARATE launches just made on the same inputs — the three-in-one launch keeps its rows in LDS since round 6)
    node = torch.as_tensor(e.peek_nodes())
    per_wg = (node.double() ** 2).reshape(B // 16, 16 * M, 128).sum(1)              # a workgroup's sum of squares per column
    frac_wide = float((per_wg >= float(limit)).double().mean())
    assert (frac_wide > 0.99) if limit == "100" else (0.2 < frac_wide < 0.8), frac_wide
    assert e.range_fallbacks()[0] == 0 and e.check()
    scale = max(1.0, float(h_m.abs().max()))
    assert float((fused[0] - mprob).abs().max()) <= 2e-4
    assert float((fused[1] - h_m).abs().max()) <= 2e-4 * scale
    assert float((fused[2] - mach_v).abs().max()) <= 2e-3 * max(1.0, float(mach_v.abs().max()))


def test_exchange_timeout_is_reported_and_the_rollout_recovers(monkeypatch):
    """MTFJSP_FUSED3_FAIL_AT=n: the n-th three-in-one launch waits for contributions that never come (as if a CU were held by
    somebody else): bounded wait, host-mapped failure word, MTFJSP_ERR_RETRY at the next forward entry, the rollout restarts the
    episode on the separate launches, and check() brings the single-launch kernels back"""
    monkeypatch.setenv("MTFJSP_FUSED3_FAIL_AT", "5")
    capi = import_module("e2e-mappo-for-mt-fjsp_amd.capi")
    ro, _, _ = _rollout(seed=5)
    e = ro.actor.enc
    for _ in range(12):                                           # (the host runs ahead of the device: the 4 ms wait of launch 5 ends long after these are enqueued)
        ro.step()
    torch.cuda.synchronize()
    ro.step()                                                     # this forward entry sees the failure word: restart on the separate launches
    torch.cuda.synchronize()
    assert ro.n_resident_failures == 1 and e.resident_failures() == 1 and ro.t_in_ep == 1
    n3 = e.fused_launches()
    for _ in range(10):                                           # (no check() yet: the handle stays on the separate launches)
        ro.step()
    assert e.fused_launches() == n3
    assert e.check()                                              # idle stream: census again, single-launch kernels back
    for _ in range(40):
        ro.step()
    torch.cuda.synchronize()
    assert e.fused_launches() > n3 and ro.n_resident_failures == 1
    assert torch.isfinite(e.job_prob).all() and torch.isfinite(e.mch_prob).all()
    assert int((ro.env.status & capi.ST_INVALID).sum().item()) == 0


def test_three_launches_per_step_and_switch_off(monkeypatch):
    ro, _, _ = _rollout(seed=9)
    for _ in range(40):                                           # crosses an episode boundary (reset, post-terminal forward pair)
        ro.step()
    torch.cuda.synchronize()
    assert ro.actor.enc.fused_launches() == 40
    ro.check_finished_cleanly()
    monkeypatch.setenv("MTFJSP_NO_FUSED_MHEADS", "1")
    ro2, _, _ = _rollout(seed=9)
    for _ in range(40):
        ro2.step()
    torch.cuda.synchronize()
    assert ro2.actor.enc.fused_launches() == 0
    ro2.check_finished_cleanly()


@pytest.mark.parametrize("shape,nb", [((6, 6), 4096), ((6, 6), 1000), ((10, 10), 512)])
def test_values_only_forward_pair_gives_the_full_pairs_values_bit_for_bit(shape, nb, monkeypatch):
    """mtfjsp_encoder_arm_values_only (round 6: the post-terminal forward pair of Run.py:455-475 keeps only the two critics' values):
    the values and the pooled embeddings of the armed pair are the full pair's, bit for bit; prob is left untouched; one-shot."""
    import mtfjsp_amd  # noqa: F401
    rollout = import_module("e2e-mappo-for-mt-fjsp_amd.rollout")
    enc_mod = import_module("e2e-mappo-for-mt-fjsp_amd.encoder")
    j, m = shape
    ja, ma = enc_mod.random_init_weights(seed=31)
    ro = rollout.Rollout(j, m, 2, nb, policy="actor", obs_dtype="f32", weights=(ja, ma), collect=False, seed=3)
    for _ in range(j * m // 2 + 3):
        ro.step()
    env, act, e = ro.env, ro.actor, ro.actor.enc
    torch.cuda.synchronize()
    hm0 = e.h_pooled_m.clone()
    mask = env.job_mask.clone()

    def pair(armed):
        e.h_pooled_m.copy_(hm0)
        e.job_prob.fill_(-7.0); e.mch_prob.fill_(-7.0)
        jv = torch.zeros(nb, 2, device="cuda"); mv = torch.zeros(nb, 2, device="cuda")
        if armed:
            e.arm_values_only()
        _, h_o, _ = e.job_actor_forward(env.tasks_fea, env.ell_col, env.ell_val, env.candidate, mask, e.h_pooled_m, v_out=jv)
        e.machine_actor_forward(act.last_mfea1, env.m_fea2, h_o, act.last_mmask, v_out=mv)
        torch.cuda.synchronize()
        return jv.clone(), mv.clone(), h_o.clone(), e.h_pooled_m.clone(), e.job_prob.clone(), e.mch_prob.clone()

    full = pair(False)
    only = pair(True)
    again = pair(False)                                            # one-shot: the next pair is a full one again
    for k in range(4):
        assert torch.equal(full[k], only[k]), k
        assert torch.equal(full[k], again[k]), k
    assert bool((only[4] == -7.0).all()) and bool((only[5] == -7.0).all())          # the scorer did not run
    assert torch.equal(full[4], again[4]) and torch.equal(full[5], again[5]) and bool((full[4] >= 0).all())
    e.check()                                                      # (raises on an asynchronous failure; its value only says which GIN kernel is in use)
