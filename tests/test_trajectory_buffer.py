"""SURVEY.md §8f N2 — trajectory buffer vs the reference's ReplayBuffer.

tests/golden/replaybuffer_j6m6e2_b2.npz holds what the REFERENCE class returned from numpy_to_tensor_operation() after one
recorded J6M6E2 episode was stored through it the way Run.py does (oracle/ref_harness/gen_golden_buffer.py).  The same
transitions go through TrajectoryBuffer.store_operation / store_v_next here (host tensors: the class is device-agnostic);
every one of the 27 outputs must be equal, same order, same dtype, the adjacency after `.dense()`.
"""
import os
from importlib import import_module

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
NAMES = ["adj", "tasks_fea", "candidate", "mask_operation", "a_operation", "a_logprob_operation",
         "adj_", "tasks_fea_", "candidate_", "mask_operation_", "r_operation", "done_operation",
         "machine_fea2", "a", "a_logprob", "machine_fea2_", "mask_machine_",
         "mk", "pt", "tt", "it", "machine_fea1", "rw", "job_v", "machine_v", "job_v_", "machine_v_"]


def _fill(buf, ell_inputs):
    g = np.load(os.path.join(GOLDEN, "trace_j6m6e2_train16_mask.npz"))
    f = np.load(os.path.join(GOLDEN, "replaybuffer_j6m6e2_b2.npz"))
    J, M, E, B, ep = [int(x) for x in f["meta"]]
    T = J * M
    traj = import_module("e2e-mappo-for-mt-fjsp_amd.trajectory")
    feas = g["t"][:B] >= 0
    conv = (lambda a: traj.dense_to_ell(a)) if ell_inputs else (lambda a: a)
    adj, fea = g["adj0"][ep][:B].astype(np.float64), g["tfea0"][ep][:B * T]
    cand, mask, mf2 = g["cand0"][ep][:B], g["mask0"][ep][:B].astype(bool), g["mfea2_0"][ep][:B]
    nv = 0
    for s in range(T):
        adj_, fea_ = g["adj"][ep, s][:B].astype(np.float64), g["tfea"][ep, s][:B * T]
        cand_, mask_, mf2_ = g["cand"][ep, s][:B], g["mask"][ep, s][:B].astype(bool), g["mfea2"][ep, s][:B]
        info = g["info"][ep, s][:B]
        task, mach = g["actions"][ep, s][:B, 0], g["actions"][ep, s][:B, 1]
        if s >= 1:
            buf.store_v_next(f["fed_j_v_"][nv], f["fed_m_v_"][nv]); nv += 1
        done = info[:, 1].astype(bool)
        if done.all():
            buf.store_v_next(f["fed_j_v_"][nv], f["fed_m_v_"][nv]); nv += 1
        buf.store_operation(conv(adj), fea, cand, torch.tensor(mask), torch.tensor(g["job_actions"][ep, s][:B]).long(),
                            torch.tensor(f["fed_a_o_logprob"][s]), info[:, 0], conv(adj_), fea_, cand_, torch.tensor(mask_),
                            g["mfea1"][ep, s][:B], mf2, mf2_, torch.tensor(mach).long(), torch.tensor(f["fed_a_m_logprob"][s]),
                            None, done, torch.tensor(~feas[np.arange(B), task][:, None, :]),
                            info[:, 2], info[:, 4], info[:, 5], info[:, 3], g["w3"][ep][:B], f["fed_j_v"][s], f["fed_m_v"][s])
        adj, fea, cand, mask, mf2 = adj_, fea_, cand_, mask_, mf2_
    return f, (J, M, B, T)


@pytest.mark.parametrize("ell_inputs", [False, True])
def test_buffer_returns_what_the_reference_buffer_returns(ell_inputs):
    traj = import_module("e2e-mappo-for-mt-fjsp_amd.trajectory")
    args = {"n_job": 6, "n_machine": 6, "buffer_size": 1, "env_batch": 2, "gcn_input_dim": 12}
    buf = traj.TrajectoryBuffer(args, device="cpu", obs_dtype=torch.float64)
    f, (J, M, B, T) = _fill(buf, ell_inputs)
    assert buf.full and buf.count_operation == T and buf.count_operation_ == T
    out = buf.numpy_to_tensor_operation()
    assert len(out) == len(NAMES) == 27
    for n, v in zip(NAMES, out):
        want = f["out_" + n]
        if n in ("adj", "adj_"):
            assert isinstance(v, traj.EllAdjacency) and v.shape == want.shape
            got = v.dense()
            assert got.dtype == torch.float32
            assert np.array_equal(got.numpy(), want.astype(np.float32)), n
            # a minibatch of steps, as the update slices it
            assert np.array_equal(v.dense(torch.tensor([3, 17])).numpy(), want[[3, 17]].astype(np.float32))
            continue
        assert str(v.dtype) == str(f["dtype_" + n]), (n, v.dtype, f["dtype_" + n])
        assert tuple(v.shape) == want.shape, (n, v.shape, want.shape)
        assert np.array_equal(v.numpy(), want), n
    buf.reset()
    assert buf.count_operation == 0 and buf.count_operation_ == 0


def test_store_operation_on_an_aliased_buffer_lands_in_the_shifted_value_slots():
    """alias_v_next (the device rollout's layout): the reference-interface store must write job_v / machine_v through the same
    (episode, step) mapping slot() uses — v_ of step t is then the stored v of step t+1"""
    traj = import_module("e2e-mappo-for-mt-fjsp_amd.trajectory")
    args = {"n_job": 6, "n_machine": 6, "buffer_size": 1, "env_batch": 2, "gcn_input_dim": 12}
    buf = traj.TrajectoryBuffer(args, device="cpu", obs_dtype=torch.float64, alias_v_next=True)
    buf.store_v_next = lambda *a: None                       # (the aliased layout has nothing to store there)
    f, (J, M, B, T) = _fill(buf, False)
    jv, mv, jv_, mv_ = buf.local_values()
    assert np.array_equal(jv.numpy(), f["out_job_v"]) and np.array_equal(mv.numpy(), f["out_machine_v"])
    assert np.array_equal(jv_.numpy()[:T - 1], f["out_job_v"][1:T]) and np.array_equal(mv_.numpy()[:T - 1], f["out_machine_v"][1:T])


def test_dense_to_ell_round_trip_and_rejects_non_graph_input():
    traj = import_module("e2e-mappo-for-mt-fjsp_amd.trajectory")
    g = np.load(os.path.join(GOLDEN, "trace_j10m10e2_b2_mask.npz"))
    adj = g["adj"][0, -1].astype(np.float64)                     # terminal state: every node scheduled
    col, val = traj.dense_to_ell(adj)
    B, T, _ = adj.shape
    back = traj.EllAdjacency(col.reshape(1, B * T, 2), val.reshape(1, B * T, 2), B, T).dense()[0]
    assert np.array_equal(back.numpy(), adj.astype(np.float32))
    bad = adj.copy()
    bad[0, 5, :4] = 3
    with pytest.raises(ValueError):
        traj.dense_to_ell(bad)


def test_memory_footprint_is_ell_not_dense():
    traj = import_module("e2e-mappo-for-mt-fjsp_amd.trajectory")
    args = {"n_job": 6, "n_machine": 6, "buffer_size": 1, "env_batch": 64, "gcn_input_dim": 12}
    buf = traj.TrajectoryBuffer(args, device="cpu")
    S, B, T = buf.total_step, 64, 36
    dense_ref = 2 * S * B * T * T * 8                              # the reference's adj + adj_ alone
    assert buf.nbytes() < dense_ref / 2


@pytest.mark.gpu
def test_device_rollout_fills_the_buffer_consistently():
    """Rollout(collect='full'): every slot holds what the environment / actors produced at that step — checked against
    a twin rollout (same seeds) whose observations are cloned step by step — and the 27-tuple has the reference's
    dtypes; the pre-state of step s+1 is the post-state of step s inside an episode."""
    import mtfjsp_amd  # noqa: F401
    rollout = import_module("e2e-mappo-for-mt-fjsp_amd.rollout")
    B, J, M = 96, 6, 6
    T = J * M
    a = rollout.Rollout(J, M, 2, B, policy="actor", obs_dtype="f32", collect="full", buffer_episodes=2)
    b = rollout.Rollout(J, M, 2, B, policy="actor", obs_dtype="f32", collect=True, buffer_episodes=2)
    S = 2 * T
    pre, post, acts, dec = [], [], [], []
    for s in range(S - 1):                                   # stop one step short of the hand-off (which resets the counters)
        b_env = b.env
        # the twin's pre-state is only defined after its reset, which happens inside step(); clone post-state instead
        a.step(); b.step()
        post.append((b_env.tasks_fea.clone(), b_env.ell_col.clone(), b_env.ell_val.clone(), b_env.m_fea2.clone(),
                     b_env.candidate.clone(), b_env.job_mask.clone(), b_env.info.clone()))
        acts.append((b.job.clone(), b.mach.clone(), b.task.clone()))
        dec.append((b_env.m_fea1.clone(), b_env.mmask.clone()))   # the twin's heads kernel writes them into the environment's buffers
    torch.cuda.synchronize()
    tb = a.traj
    assert tb.count_operation == S - 1
    for s in range(S - 1):
        tf, ec, ev, mf, cand, mask, info = post[s]
        assert torch.equal(tb.tasks_fea_[s], tf) and torch.equal(tb.ell_col_[s], ec) and torch.equal(tb.ell_val_[s], ev)
        assert torch.equal(tb.machine_fea2_[s], mf.reshape(B, M, 8)) and torch.equal(tb.candidate_[s], cand)
        assert torch.equal(tb.mask_operation_[s], mask.bool())
        assert torch.equal(tb.r_operation[s], info[:, 0].float()) and torch.equal(tb.done_operation[s], info[:, 1].float())
        assert torch.equal(tb.mk[s], info[:, 2].float()) and torch.equal(tb.it[s], info[:, 3].float())
        assert torch.equal(tb.pt[s], info[:, 4].float()) and torch.equal(tb.tt[s], info[:, 5].float())
        assert torch.equal(tb.a_operation[s], acts[s][0]) and torch.equal(tb.a[s], acts[s][1])
        # m_fea1 / machine mask: written by the job heads' launch straight into the slot (no copy) and read from there by the machine actor
        assert torch.equal(tb.machine_fea1[s], dec[s][0].reshape(B, M, 6)) and torch.equal(tb.mask_machine_[s], dec[s][1].bool().reshape(B, 1, M))
        if (s + 1) % T != 0 and s + 1 < S - 1:               # inside an episode: s' of step s is s of step s+1
            assert torch.equal(tb.tasks_fea[s + 1], tb.tasks_fea_[s]) and torch.equal(tb.ell_col[s + 1], tb.ell_col_[s])
            assert torch.equal(tb.candidate[s + 1], tb.candidate_[s]) and torch.equal(tb.machine_fea2[s + 1], tb.machine_fea2_[s])
            assert torch.equal(tb.ell_val[s + 1], tb.ell_val_[s]) and torch.equal(tb.mask_operation[s + 1], tb.mask_operation_[s])
    # an episode's first slot holds the observation after the reset (its own snapshot): not the previous episode's terminal state
    assert not torch.equal(tb.tasks_fea[T], tb.tasks_fea_[T - 1]) and torch.equal(tb.tasks_fea[T][:, 3], torch.zeros_like(tb.tasks_fea[T][:, 3]))
    # chosen machine is feasible under the stored machine mask; stored job is selectable under the stored job mask
    idx = torch.arange(B, device="cuda")
    for s in (0, 17, T, S - 2):
        assert not bool(tb.mask_machine_[s][idx, 0, tb.a[s].long()].any())
        assert not bool(tb.mask_operation[s][idx, tb.a_operation[s].long()].any())
    out = tb.numpy_to_tensor_operation()
    f = np.load(os.path.join(GOLDEN, "replaybuffer_j6m6e2_b2.npz"))
    for n, v in zip(NAMES, out):
        if n not in ("adj", "adj_"):
            assert str(v.dtype) == str(f["dtype_" + n]), n
    assert out[0].dense(torch.tensor([5])).shape == (1, B, T, T)
    # the hand-off still works on the buffer's own storage (GAE + normalisation), and resets the counters
    a.step()
    assert tb.count_operation == 0 and a.last_adv is not None


@pytest.mark.gpu
def test_two_destination_snapshot_copies_exactly_the_fields_each_destination_has():
    """mtfjsp_snapshot_obs2 (include/mtfjsp.h): one launch, every non-NULL field of dst and of dst2 receives the bound observation,
    reward_out receives (float)info[:,0]; NULL fields / a NULL dst2 / a NULL reward_out are skipped; an odd batch exercises the
    byte tails of the 16-byte copy loops."""
    import ctypes as C
    import mtfjsp_amd  # noqa: F401
    rollout = import_module("e2e-mappo-for-mt-fjsp_amd.rollout")
    capi = import_module("e2e-mappo-for-mt-fjsp_amd.capi")
    B, J, M = 37, 5, 3
    T = J * M
    ro = rollout.Rollout(J, M, 3, B, policy="random", obs_dtype="f32")
    for _ in range(7):
        ro.step()
    env = ro.env
    dev = env.tasks_fea.device
    like = lambda x: torch.full_like(x, 77) if x.dtype != torch.bool else torch.ones_like(x)
    a = dict(tf=like(env.tasks_fea), ec=like(env.ell_col), ev=like(env.ell_val), mf=like(env.m_fea2), cand=like(env.candidate), mask=like(env.job_mask),
             info=like(env.info))
    b = dict(tf=like(env.tasks_fea), cand=like(env.candidate), mask=like(env.job_mask), mf=like(env.m_fea2))
    rew = torch.full((B,), -5.0, dtype=torch.float32, device=dev)
    p = lambda t: t.data_ptr()
    oa = capi.Obs(p(a["tf"]), p(a["ec"]), p(a["ev"]), p(a["mf"]), p(a["info"]), 0, p(a["cand"]), p(a["mask"]), 0)
    ob = capi.Obs(p(b["tf"]), 0, 0, p(b["mf"]), 0, 0, p(b["cand"]), p(b["mask"]), 0)          # dst2 without the adjacency and info
    capi.check(env.L.mtfjsp_snapshot_obs2(env.h, C.byref(oa), C.byref(ob), rew.data_ptr()), env.h)
    torch.cuda.synchronize()
    for k, src in (("tf", env.tasks_fea), ("ec", env.ell_col), ("ev", env.ell_val), ("mf", env.m_fea2), ("cand", env.candidate), ("mask", env.job_mask),
                   ("info", env.info)):
        assert torch.equal(a[k], src), k
    for k, src in (("tf", env.tasks_fea), ("mf", env.m_fea2), ("cand", env.candidate), ("mask", env.job_mask)):
        assert torch.equal(b[k], src), k
    assert torch.equal(rew, env.info[:, 0].float())
    # a field only dst2 has is still copied; no dst2 and no reward: the plain snapshot
    c_tf, d_ec = like(env.tasks_fea), like(env.ell_col)
    oc = capi.Obs(p(c_tf), 0, 0, 0, 0, 0, 0, 0, 0)
    od = capi.Obs(0, p(d_ec), 0, 0, 0, 0, 0, 0, 0)
    capi.check(env.L.mtfjsp_snapshot_obs2(env.h, C.byref(oc), C.byref(od), None), env.h)
    torch.cuda.synchronize()
    assert torch.equal(c_tf, env.tasks_fea) and torch.equal(d_ec, env.ell_col)
    e_tf = like(env.tasks_fea)
    oe = capi.Obs(p(e_tf), 0, 0, 0, 0, 0, 0, 0, 0)
    capi.check(env.L.mtfjsp_snapshot_obs2(env.h, C.byref(oe), None, None), env.h)
    torch.cuda.synchronize()
    assert torch.equal(e_tf, env.tasks_fea)
    assert env.L.mtfjsp_snapshot_obs2(env.h, None, None, None) == capi.ERR_ARG
