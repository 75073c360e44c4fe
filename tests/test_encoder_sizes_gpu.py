"""HIP encoder vs the fp32 oracle restatement at sizes beyond the J6M6 reference fixtures: exercises the chunked heads
(R = J or M > 6 tiles per group), ragged tile counts (B*T not a multiple of 16), larger per-instance kernels, f64
observations, and both BatchNorm modes.  The oracle is generic tensor code pinned against the reference modules at J6M6
(tests/test_encoder_oracle_golden.py, tests/test_eval_per_instance_bn.py); inputs come from a real device rollout."""
import os
from importlib import import_module

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _perturb(ja, ma, seed):
    """non-trivial BatchNorm affine parameters (the default initialiser has gamma 1 / beta 0)"""
    rs = np.random.RandomState(seed)
    for d in (ja, ma):
        for k in d:
            if "batch_norms" in k or k.startswith("bn."):
                d[k] = (rs.uniform(0.5, 1.5, d[k].shape) if k.endswith("weight") else rs.uniform(-0.5, 0.5, d[k].shape)).astype(np.float32)


@pytest.mark.parametrize("J,M,E,B,steps,obs", [(10, 6, 2, 37, 23, "f32"), (10, 10, 2, 24, 41, "f64"), (20, 20, 4, 5, 150, "f32"),
                                             (3, 4, 2, 7, 5, "f32"), (5, 7, 1, 19, 17, "f32"), (7, 5, 5, 33, 9, "f64")])
def test_encoder_matches_oracle_at_other_sizes(J, M, E, B, steps, obs):
    import torch
    import mtfjsp_amd  # noqa: F401
    enc_mod = import_module("e2e-mappo-for-mt-fjsp_amd.encoder")
    rollout = import_module("e2e-mappo-for-mt-fjsp_amd.rollout")
    from oracle import encoder_oracle as eo
    T = J * M
    ja, ma = enc_mod.random_init_weights(seed=J * 100 + M)
    _perturb(ja, ma, 3)
    ro = rollout.Rollout(J, M, E, B, policy="actor", obs_dtype=obs, weights=(ja, ma), collect=False)
    for _ in range(steps):
        ro.step()
    env, e = ro.env, ro.actor.enc
    torch.cuda.synchronize()
    tf = env.tasks_fea.cpu().numpy().astype(np.float64)
    col = env.ell_col.cpu().numpy().reshape(B, T, 2); val = env.ell_val.cpu().numpy().reshape(B, T, 2).astype(np.float64)
    cand = env.candidate.cpu().numpy(); mask = env.job_mask.cpu().numpy()
    hm = e.h_pooled_m.cpu().numpy().copy()
    for per_instance in (False, True):
        e.set_bn_mode(per_instance)
        prob, h_o, job_v = e.job_actor_forward(env.tasks_fea, env.ell_col, env.ell_val, env.candidate, env.job_mask, torch.as_tensor(hm).cuda())
        torch.cuda.synchronize()
        prob, h_o, job_v = prob.cpu().numpy(), h_o.cpu().numpy().copy(), job_v.cpu().numpy().copy()
        if per_instance:
            outs = [eo.job_actor_forward(ja, tf[b * T:(b + 1) * T], col[b:b + 1], val[b:b + 1], cand[b:b + 1], mask[b:b + 1], hm[b:b + 1], 1, T)
                    for b in range(B)]
            o = {k: np.concatenate([x[k] for x in outs], 0) for k in ("prob", "h_pooled", "job_v")}
        else:
            o = eo.job_actor_forward(ja, tf, col, val, cand, mask, hm, B, T)
        scale = max(1.0, float(np.abs(o["h_pooled"]).max()))
        np.testing.assert_allclose(h_o, o["h_pooled"], rtol=0, atol=2e-4 * scale)
        np.testing.assert_allclose(prob, o["prob"], rtol=0, atol=2e-4)
        np.testing.assert_allclose(job_v, o["job_v"], rtol=2e-3, atol=2e-3)
        # machine actor on the task each instance would pick greedily
        task = torch.as_tensor(cand[np.arange(B), prob.argmax(1)].astype(np.int32)).cuda()
        env.observe_mfea1(task)
        mprob, h_m, mach_v = e.machine_actor_forward(env.m_fea1, env.m_fea2, torch.as_tensor(o["h_pooled"]).cuda(), env.mmask)
        torch.cuda.synchronize()
        f1 = env.m_fea1.cpu().numpy().astype(np.float64).reshape(B, M, 6); f2 = env.m_fea2.cpu().numpy().astype(np.float64).reshape(B, M, 8)
        mm = env.mmask.cpu().numpy().reshape(B, M)
        if per_instance:
            outs = [eo.machine_actor_forward(ma, f1[b:b + 1], f2[b:b + 1], o["h_pooled"][b:b + 1], mm[b:b + 1], 1, M) for b in range(B)]
            mo = {k: np.concatenate([x[k] for x in outs], 0) for k in ("prob", "h_pooled", "mach_v")}
            htol = 5e-3                                               # BatchNorm over the M rows of one instance: see tests/test_eval_per_instance_bn.py
        else:
            mo = eo.machine_actor_forward(ma, f1, f2, o["h_pooled"], mm, B, M)
            htol = 2e-4
        np.testing.assert_allclose(mprob.cpu().numpy(), mo["prob"], rtol=0, atol=5e-4)
        dh = np.abs(h_m.cpu().numpy() - mo["h_pooled"])
        assert dh.max() < htol * 4 and np.median(dh) < htol / 4, (dh.max(), np.median(dh))
        np.testing.assert_allclose(mach_v.cpu().numpy(), mo["mach_v"], rtol=5e-3, atol=5e-3)
    e.set_bn_mode(False)


def test_streaming_order_switches_do_not_change_results(monkeypatch):
    """The streaming GIN launches alternate the direction in which a workgroup walks its rows, read their input with non-temporal
    loads, and the pool / gather kernel follows the last product's row ranges from the back (the memory-side
    cache).  None of that may change a value: the same forward on a handle created with round 2's order
    (MTFJSP_NO_STREAM_ORDER) agrees to the accumulation-order noise of the BatchNorm sums.  J10M10 x 333: 33 300 rows, a ragged
    last tile, 261 row tiles over the workgroups, groups of 16 and of 8 instances in the heads."""
    import torch
    import mtfjsp_amd  # noqa: F401
    enc_mod = import_module("e2e-mappo-for-mt-fjsp_amd.encoder")
    rollout = import_module("e2e-mappo-for-mt-fjsp_amd.rollout")
    J, M, E, B = 10, 10, 2, 333
    ja, ma = enc_mod.random_init_weights(seed=77)
    _perturb(ja, ma, 5)
    ro = rollout.Rollout(J, M, E, B, policy="actor", obs_dtype="f32", weights=(ja, ma), collect=False)
    for _ in range(37):
        ro.step()
    env = ro.env
    hm = ro.actor.enc.h_pooled_m.clone()
    outs = []
    # MTFJSP_FUSE_GIN0 — default 1: the first Linear's output is never stored; its BatchNorm sums come from the second moments of the 12 aggregated
    # features (k_gin0_moments) and the second launch's producers form it again.  2: the sums by a statistics-only launch instead — the same matrix
    # instructions on the same operands as 0, the two-launch form, so only the order of the sums' atomics differs: those two are held to a tenth of the bound.
    # MTFJSP_FUSE_POOL — default 1: the last Linear's output is not stored either; a second pass of the product pools and gathers in its epilogue
    # (0: product with output + k_job_pool_gather).
    for switches in ({}, {"MTFJSP_FUSE_GIN0": "0"}, {"MTFJSP_FUSE_GIN0": "2"}, {"MTFJSP_NO_STREAM_ORDER": "1"}, {"MTFJSP_POOL_S": "0", "MTFJSP_STREAM_NT": "0"},
                     {"MTFJSP_NO_HEADS_HG8": "1"}, {"MTFJSP_FUSE_POOL": "0"}, {"MTFJSP_FUSE_POOL": "0", "MTFJSP_FUSE_GIN0": "0"}):
        for k in ("MTFJSP_NO_STREAM_ORDER", "MTFJSP_POOL_S", "MTFJSP_STREAM_NT", "MTFJSP_NO_HEADS_HG8", "MTFJSP_FUSE_GIN0", "MTFJSP_FUSE_POOL"):
            monkeypatch.delenv(k, raising=False)
        for k, v in switches.items():
            monkeypatch.setenv(k, v)
        e = enc_mod.Encoder(J, M, B, obs_dtype="f32")            # the switches are read when the handle is created
        e.load_weights(ja, ma)
        prob, h_o, job_v = e.job_actor_forward(env.tasks_fea, env.ell_col, env.ell_val, env.candidate, env.job_mask, hm)
        torch.cuda.synchronize()
        outs.append((prob.cpu().numpy().copy(), h_o.cpu().numpy().copy(), job_v.cpu().numpy().copy()))
        e.close()
    scale = max(1.0, float(np.abs(outs[0][1]).max()))
    np.testing.assert_allclose(outs[2][1], outs[1][1], rtol=0, atol=2e-6 * scale)     # (measured: identical, tools/check_gin0_modes.py)
    np.testing.assert_allclose(outs[2][0], outs[1][0], rtol=0, atol=2e-6)
    np.testing.assert_allclose(outs[0][1], outs[1][1], rtol=0, atol=5e-6 * scale)     # statistics from the moments (measured: 0.3-1.3e-6 of the scale)
    np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=0, atol=5e-6)
    for o in outs[1:]:
        np.testing.assert_allclose(o[1], outs[0][1], rtol=0, atol=2e-5 * scale)
        np.testing.assert_allclose(o[0], outs[0][0], rtol=0, atol=2e-5)
        np.testing.assert_allclose(o[2], outs[0][2], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("J,M,E,B", [(10, 10, 2, 333), (4, 6, 2, 50), (4, 4, 2, 97), (8, 4, 2, 53), (15, 10, 5, 11)])
def test_pooling_epilogue_serves_any_candidate(monkeypatch, J, M, E, B):
    """The pooling epilogue of the last streaming product (MTFJSP_FUSE_POOL, default) takes candidate j from job j's block of rows, where the
    environment's candidates lie; a caller-made candidate elsewhere — another job's row, a row two slots share, a finished job's -1 — goes through
    k_cand_fixup.  Same forward as the stored-output path (k_job_pool_gather) on a candidate array with all of those; T = 24 (J4M6) puts up to two
    instances into a 16-row tile and an instance across workgroup ranges; T = 16 (J4M4) is the smallest the epilogue takes (every tile one instance), T = 150
    with 11 instances a ragged last tile."""
    import torch
    import mtfjsp_amd  # noqa: F401
    enc_mod = import_module("e2e-mappo-for-mt-fjsp_amd.encoder")
    rollout = import_module("e2e-mappo-for-mt-fjsp_amd.rollout")
    T = J * M
    ja, ma = enc_mod.random_init_weights(seed=5 + J)
    _perturb(ja, ma, 9)
    monkeypatch.setenv("MTFJSP_NO_RESIDENT_GIN", "1")                # (J4M6 would otherwise take the single-launch kernel)
    ro = rollout.Rollout(J, M, E, B, policy="actor", obs_dtype="f32", weights=(ja, ma), collect=False)
    for _ in range(T // 3):
        ro.step()
    env = ro.env
    hm = ro.actor.enc.h_pooled_m.clone()
    cand = env.candidate.cpu().numpy().copy()
    rs = np.random.RandomState(3)
    for b in range(0, B, 2):
        j = rs.randint(J)
        cand[b, j] = ((j + 1) % J) * M + rs.randint(M)                # a row of another job's block
        cand[b, (j + 2) % J] = cand[b, j]                             # ... shared by two slots
    cand_t = torch.as_tensor(cand.astype(np.int32)).cuda()
    mask = torch.zeros_like(env.job_mask)
    outs = []
    for fuse in ("1", "0"):
        monkeypatch.setenv("MTFJSP_FUSE_POOL", fuse)
        monkeypatch.setenv("MTFJSP_FUSE_GIN0", fuse)                  # (the second handle: every Linear's output stored, as in round 5)
        e = enc_mod.Encoder(J, M, B, obs_dtype="f32")
        e.load_weights(ja, ma)
        prob, h_o, job_v = e.job_actor_forward(env.tasks_fea, env.ell_col, env.ell_val, cand_t, mask, hm)
        torch.cuda.synchronize()
        outs.append((prob.cpu().numpy().copy(), h_o.cpu().numpy().copy(), job_v.cpu().numpy().copy()))
        e.close()
    scale = max(1.0, float(np.abs(outs[0][1]).max()))
    np.testing.assert_allclose(outs[0][1], outs[1][1], rtol=0, atol=2e-5 * scale)
    np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=0, atol=2e-5)
    np.testing.assert_allclose(outs[0][2], outs[1][2], rtol=1e-4, atol=1e-4)


def test_adjacency_entries_in_either_order():
    """The two in-edge entries of a row may come in either order (the environment writes the job edge first): the same forward to rounding (the
    aggregation adds the two products in the other order)."""
    import torch
    import mtfjsp_amd  # noqa: F401
    enc_mod = import_module("e2e-mappo-for-mt-fjsp_amd.encoder")
    rollout = import_module("e2e-mappo-for-mt-fjsp_amd.rollout")
    J, M, E, B = 10, 10, 2, 200
    ja, ma = enc_mod.random_init_weights(seed=21)
    _perturb(ja, ma, 4)
    ro = rollout.Rollout(J, M, E, B, policy="actor", obs_dtype="f32", weights=(ja, ma), collect=False)
    for _ in range(45):
        ro.step()
    env = ro.env
    hm = ro.actor.enc.h_pooled_m.clone()
    col = env.ell_col.clone().view(-1, 2); val = env.ell_val.clone().view(-1, 2)
    assert int((col[:, 1] >= 0).sum()) > 0                           # (machine edges exist: the swap moves them into the first entry)
    col_sw = col.flip(1).contiguous().view_as(env.ell_col); val_sw = val.flip(1).contiguous().view_as(env.ell_val)
    e = ro.actor.enc
    a = [x.cpu().numpy().copy() for x in e.job_actor_forward(env.tasks_fea, env.ell_col, env.ell_val, env.candidate, env.job_mask, hm)]
    b = [x.cpu().numpy().copy() for x in e.job_actor_forward(env.tasks_fea, col_sw, val_sw, env.candidate, env.job_mask, hm)]
    scale = max(1.0, float(np.abs(a[1]).max()))
    np.testing.assert_allclose(b[1], a[1], rtol=0, atol=2e-5 * scale)
    np.testing.assert_allclose(b[0], a[0], rtol=0, atol=2e-5)
    np.testing.assert_allclose(b[2], a[2], rtol=1e-4, atol=1e-4)
