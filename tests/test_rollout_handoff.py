"""SURVEY.md §8f N1/N2 — the rollout -> update hand-off against a REAL reference rollout.

tests/golden/rollout_gae_j6m6e2_b4.npz (oracle/ref_harness/gen_golden_gae.py) is a 2-episode rollout of the reference's own
loop (Run.py:229-661: Parallel_env + the three networks + ReplayBuffer incl. the post-terminal forward pair of
Run.py:455-475), the reference buffer's 27-tuple, the global critic's values over the buffer (ppo:628-655) and the outputs
of cal_local_job_machine_reward_GAE / separate_cal_4_reward_GAE (ppo:437-536).

CPU tests: the host restatement (dist.gae + global normalisation) reproduces the reference's advantages, also from two
gloo shards.  GPU tests: `mtfjsp_gae` does; the device rollout teacher-forced on the recorded decisions reproduces every
stored value — in particular v_ of the terminal steps — and from them the advantages and value targets; the device
TrajectoryBuffer (k_snapshot path) equals the reference buffer's tuple; the global critic over the device buffer
reproduces multi_v / multi_v_ and the global advantages.
"""
import os
import sys
from importlib import import_module

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
FIX = os.path.join(GOLDEN, "rollout_gae_j6m6e2_b4.npz")
NAMES = ["adj", "tasks_fea", "candidate", "mask_operation", "a_operation", "a_logprob_operation",
         "adj_", "tasks_fea_", "candidate_", "mask_operation_", "r_operation", "done_operation",
         "machine_fea2", "a", "a_logprob", "machine_fea2_", "mask_machine_",
         "mk", "pt", "tt", "it", "machine_fea1", "rw", "job_v", "machine_v", "job_v_", "machine_v_"]


def _mods():
    sys.path.insert(0, ROOT)
    import mtfjsp_amd  # noqa: F401
    return import_module("e2e-mappo-for-mt-fjsp_amd.dist")


def _channels(f, lo=None, hi=None):
    """(r, v, v_) per channel in the reference's order mk, pt, tt, it (ppo:441-443) for the local critics, and the global
    critic's; columns [lo,hi) of the batch"""
    sl = slice(lo, hi)
    t = lambda a: torch.tensor(np.ascontiguousarray(a))
    jv, jv_, mv, mv_ = [f["out_" + k][:, sl] for k in ("job_v", "job_v_", "machine_v", "machine_v_")]
    r = [f["out_" + k][:, sl] for k in ("mk", "pt", "tt", "it")]
    local = [(t(r[0]), t(jv[..., 0]), t(jv_[..., 0])), (t(r[1]), t(mv[..., 0]), t(mv_[..., 0])),
             (t(r[2]), t(mv[..., 1]), t(mv_[..., 1])), (t(r[3]), t(jv[..., 1]), t(jv_[..., 1]))]
    glob = [(t(r[i]), t(f["multi_v"][:, sl, i]), t(f["multi_v_"][:, sl, i])) for i in range(4)]
    return local, glob, t(f["out_done_operation"][:, sl])


# ----------------------------------------------------------------------------------------------------------- CPU
def test_fixture_holds_the_post_terminal_values():
    """what the reference stores: v_ of step s is v of step s+1 inside an episode, and the extra forward pair at the
    terminal step — NOT the value of the next episode's first state"""
    f = np.load(FIX)
    J, M, E, B, eps = [int(x) for x in f["meta"]]
    T = J * M
    for ep in range(eps):
        o = ep * T
        assert np.array_equal(f["out_job_v_"][o:o + T - 1], f["out_job_v"][o + 1:o + T])
        assert np.array_equal(f["out_machine_v_"][o:o + T - 1], f["out_machine_v"][o + 1:o + T])
        assert np.array_equal(f["out_job_v_"][o + T - 1], f["term_job_v_"][ep])
        assert np.array_equal(f["out_machine_v_"][o + T - 1], f["term_machine_v_"][ep])
        assert f["out_done_operation"][o + T - 1].all() and not f["out_done_operation"][o:o + T - 1].any()
    assert np.abs(f["out_job_v_"][T - 1] - f["out_job_v"][T]).max() > 1e-3


def test_host_gae_and_normalisation_reproduce_the_reference_advantages():
    D = _mods()
    f = np.load(FIX)
    g, lam = [float(x) for x in f["gamma_lambda"]]
    local, glob, done = _channels(f)
    for chans, want, want_t, vs in ((local, f["local_adv"], f["local_target"], None), (glob, f["global_adv"], f["global_target"], None)):
        for i, (r, v, v_) in enumerate(chans):
            adv = D.normalize_advantages_global(D.gae(r, v, v_, done, g, lam))
            np.testing.assert_allclose(adv.numpy(), want[i], rtol=1e-5, atol=1e-5)
            np.testing.assert_allclose((adv + v).numpy(), want_t[i], rtol=1e-5, atol=1e-5)


def _gloo_worker(rank, world, port, tmp):
    D = _mods()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    f = np.load(FIX)
    g, lam = [float(x) for x in f["gamma_lambda"]]
    B = int(f["meta"][3])
    lo, hi = D.shard_range(B, rank, world)
    local, glob, done = _channels(f, lo, hi)
    ok = True
    for chans, want in ((local, f["local_adv"]), (glob, f["global_adv"])):
        raw = [D.gae(r, v, v_, done, g, lam) for r, v, v_ in chans]
        full = D.all_gather_advantages(raw)                               # one packed collective, as the rollout does
        for i, a in enumerate(raw):
            n1 = (a - full[i].mean()) / (full[i].std() + 1e-5)
            n2 = D.normalize_advantages_global(a)
            ok &= bool(np.allclose(n1.numpy(), want[i][:, lo:hi], rtol=1e-5, atol=1e-5))
            ok &= bool(np.allclose(n2.numpy(), want[i][:, lo:hi], rtol=1e-5, atol=1e-5))
    open(os.path.join(tmp, f"ok{rank}"), "w").write("1" if ok else "0")
    dist.barrier()
    dist.destroy_process_group()


def test_two_gloo_shards_reproduce_the_reference_advantages(tmp_path):
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_gloo_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / "ok0").read() == "1" and open(tmp_path / "ok1").read() == "1"


# ----------------------------------------------------------------------------------------------------------- GPU
def _weights(f):
    out = {}
    for pre in ("ja", "ma", "gc"):
        out[pre] = {k[len(pre) + 3:]: f[k] for k in f.files if k.startswith(f"w_{pre}.")}
    return out


@pytest.mark.gpu
def test_gae_kernel_reproduces_the_reference_advantages():
    _mods()
    batch_env = import_module("e2e-mappo-for-mt-fjsp_amd.batch_env")
    A = import_module("e2e-mappo-for-mt-fjsp_amd.advantages")
    f = np.load(FIX)
    g, lam = [float(x) for x in f["gamma_lambda"]]
    J, M, E, B, eps = [int(x) for x in f["meta"]]
    env = batch_env.DeviceBatchEnv(J, M, E, B, obs_dtype="f32")
    c = lambda k: torch.tensor(f[k]).cuda()
    r4 = torch.stack([c("out_mk"), c("out_it"), c("out_pt"), c("out_tt")], 1).contiguous()     # [S,4,B], the step kernel's order
    done = c("out_done_operation")
    norm, targets, raw, _ = A.local_advantages(env, r4, c("out_job_v"), c("out_job_v_"), c("out_machine_v"), c("out_machine_v_"),
                                               done, g, lam)
    for i in range(4):
        np.testing.assert_allclose(norm[i].cpu().numpy(), f["local_adv"][i], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(targets[i].cpu().numpy(), f["local_target"][i], rtol=1e-5, atol=1e-5)
    norm, targets, raw = A.global_advantages(env, r4, c("multi_v"), c("multi_v_"), done, g, lam)
    for i in range(4):
        np.testing.assert_allclose(norm[i].cpu().numpy(), f["global_adv"][i], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(targets[i].cpu().numpy(), f["global_target"][i], rtol=1e-5, atol=1e-5)


def _forced_rollout(f, collect, gin="streaming", critic=False):
    _mods()
    rollout = import_module("e2e-mappo-for-mt-fjsp_amd.rollout")
    J, M, E, B, eps = [int(x) for x in f["meta"]]
    g, lam = [float(x) for x in f["gamma_lambda"]]
    w = _weights(f)
    ro = rollout.Rollout(J, M, E, B, policy="actor", obs_dtype="f32", weights=(w["ja"], w["ma"]) + ((w["gc"],) if critic else ()),
                         collect=collect, buffer_episodes=eps, gamma=g, lam=lam, instances=(f["t"], f["p"], f["tt"], f["edge"]),
                         w3_episodes=f["w3"])
    if gin == "resident":                               # single-launch GIN kernel: the default, J6M6 is eligible
        assert ro.actor.enc.check()
    else:                                               # six streaming launches
        ro.actor.enc.set_product_mode(16)
        assert not ro.actor.enc.check()
    return ro, (J, M, E, B, eps)


@pytest.mark.gpu
@pytest.mark.parametrize("collect,gin", [(True, "streaming"), ("full", "streaming"), (True, "resident")])
def test_device_rollout_reproduces_the_reference_values_and_advantages(collect, gin):
    """teacher-forced on the reference rollout's decisions: probabilities, critic values at act time, v_ of every step incl.
    the terminal ones (post-terminal forward pair), scaled rewards, and the normalised local advantages + value targets"""
    f = np.load(FIX)
    ro, (J, M, E, B, eps) = _forced_rollout(f, collect, gin)
    T, S = J * M, eps * J * M
    i32 = lambda a: torch.tensor(a.astype(np.int32)).cuda()
    for s in range(S):
        if s == S - 1:                                  # the hand-off below resets the slots' bookkeeping; keep the views
            jv_buf, mv_buf, r_buf, d_buf = ro.buf_jv, ro.buf_mv, ro.buf_r, ro.buf_done
        ro.step(force=(i32(f["task"][s]), i32(f["mach"][s]), i32(f["job"][s])))
        e = ro.actor.enc
        if s % T != T - 1:                              # (after a terminal step the outputs are the extra forward's)
            np.testing.assert_allclose(e.job_prob.cpu().numpy(), f["job_prob"][s], rtol=0, atol=1e-4)
            np.testing.assert_allclose(e.mch_prob.cpu().numpy(), f["mch_prob"][s], rtol=0, atol=1e-4)
    torch.cuda.synchronize()
    ro.check_finished_cleanly()
    assert ro.n_handoffs == 1
    jv = jv_buf[:, :T].reshape(S, B, 2).cpu().numpy(); jv_ = jv_buf[:, 1:].reshape(S, B, 2).cpu().numpy()
    mv = mv_buf[:, :T].reshape(S, B, 2).cpu().numpy(); mv_ = mv_buf[:, 1:].reshape(S, B, 2).cpu().numpy()
    tol = dict(rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(jv, f["out_job_v"], **tol); np.testing.assert_allclose(mv, f["out_machine_v"], **tol)
    np.testing.assert_allclose(jv_, f["out_job_v_"], **tol); np.testing.assert_allclose(mv_, f["out_machine_v_"], **tol)
    for ep in range(eps):                               # the terminal slots, explicitly
        np.testing.assert_allclose(jv_[ep * T + T - 1], f["term_job_v_"][ep], **tol)
        np.testing.assert_allclose(mv_[ep * T + T - 1], f["term_machine_v_"][ep], **tol)
    r = r_buf.cpu().numpy()                             # [S,4,B]: mk, idle, pt, tt — the f32 casts of the reference's f64 values
    for ch, k in enumerate(("out_mk", "out_it", "out_pt", "out_tt")):
        assert np.array_equal(r[:, ch], f[k]), k
    assert np.array_equal(d_buf.cpu().numpy(), f["out_done_operation"])
    norm, targets = ro.last_adv
    for i in range(4):
        np.testing.assert_allclose(norm[i].cpu().numpy(), f["local_adv"][i], rtol=2e-3, atol=2e-3)
        np.testing.assert_allclose(targets[i].cpu().numpy(), f["local_target"][i], rtol=2e-3, atol=2e-3)


@pytest.mark.gpu
def test_device_trajectory_buffer_equals_the_reference_buffer_and_global_critic_values():
    """Rollout(collect='full') on device tensors (k_snapshot path), teacher-forced: the 27-tuple of the device
    TrajectoryBuffer against the reference ReplayBuffer's (integers / masks / observations / rewards exact as f32, network
    outputs within tolerance), then the global critic over the device buffer against the reference's multi_v / multi_v_ and
    the global advantages computed from them (ppo:628-703)."""
    f = np.load(FIX)
    ro, (J, M, E, B, eps) = _forced_rollout(f, "full")
    A = import_module("e2e-mappo-for-mt-fjsp_amd.advantages")
    T, S = J * M, eps * J * M
    i32 = lambda a: torch.tensor(a.astype(np.int32)).cuda()
    tb = ro.traj
    for s in range(S - 1):
        ro.step(force=(i32(f["task"][s]), i32(f["mach"][s]), i32(f["job"][s])))
    # last step without the automatic hand-off, so that the buffer can be read while full
    ro.finish_buffer, finish = (lambda: None), ro.finish_buffer
    tb.reset, reset = (lambda: None), tb.reset
    ro.step(force=(i32(f["task"][S - 1]), i32(f["mach"][S - 1]), i32(f["job"][S - 1])))
    torch.cuda.synchronize()
    assert tb.full
    out = tb.numpy_to_tensor_operation()
    exact = {"tasks_fea", "candidate", "mask_operation", "a_operation", "tasks_fea_", "candidate_", "mask_operation_", "r_operation",
             "done_operation", "machine_fea2", "a", "machine_fea2_", "mask_machine_", "mk", "pt", "tt", "it", "machine_fea1", "rw"}
    for n, v in zip(NAMES, out):
        want = f["out_" + n]
        if n in ("adj", "adj_"):
            assert np.array_equal(v.dense().cpu().numpy(), want.astype(np.float32)), n
            continue
        got = v.cpu().numpy()
        assert got.shape == want.shape and got.dtype == want.dtype, (n, got.shape, got.dtype, want.shape, want.dtype)
        if n in exact:
            assert np.array_equal(got, want), n
        elif n in ("job_v", "machine_v", "job_v_", "machine_v_"):
            np.testing.assert_allclose(got, want, rtol=1e-3, atol=1e-3, err_msg=n)
        # a_logprob*: log-probabilities of the actors' own (not the forced) selections — covered by the probability checks
    # global critic over the buffer
    enc = ro.actor.enc
    enc.load_weights({}, {}, _weights(f)["gc"])
    mv, mv_ = A.sample_global_values(enc, tb)
    np.testing.assert_allclose(mv.cpu().numpy(), f["multi_v"], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(mv_.cpu().numpy(), f["multi_v_"], rtol=1e-3, atol=1e-3)
    g, lam = [float(x) for x in f["gamma_lambda"]]
    norm, targets, _ = A.global_advantages(ro.env, tb.r4, mv, mv_, tb.done_operation, g, lam)
    for i in range(4):
        np.testing.assert_allclose(norm[i].cpu().numpy(), f["global_adv"][i], rtol=2e-3, atol=2e-3)
        np.testing.assert_allclose(targets[i].cpu().numpy(), f["global_target"][i], rtol=2e-3, atol=2e-3)
    finish(); reset()
    assert tb.count_operation == 0 and ro.last_adv is not None


@pytest.mark.gpu
@pytest.mark.parametrize("gin", ["streaming", "resident"])
def test_rollout_full_handoff_reproduces_all_eight_reference_advantages(gin):
    """Rollout(collect="full") with the global critic's weights: finish_buffer runs the WHOLE hand-off of ppo:628-703 itself —
    global critic over the stored pre-/post-decision states, 4 global + 4 local GAE scans, one packed exchange of the 8 advantage
    and 8 value tensors, normalisation, value targets — teacher-forced on the reference's own rollout"""
    f = np.load(FIX)
    ro, (J, M, E, B, eps) = _forced_rollout(f, "full", gin, critic=True)
    S = eps * J * M
    i32 = lambda a: torch.tensor(a.astype(np.int32)).cuda()
    for s in range(S):
        ro.step(force=(i32(f["task"][s]), i32(f["mach"][s]), i32(f["job"][s])))
    torch.cuda.synchronize()
    assert ro.n_handoffs == 1 and ro.last_full is not None
    h = ro.last_full
    assert h["gather"] is None or h["gather"]["world"] == 1
    for i in range(4):
        np.testing.assert_allclose(h["global_adv"][i].cpu().numpy(), f["global_adv"][i], rtol=2e-3, atol=2e-3)
        np.testing.assert_allclose(h["global_targets"][i].cpu().numpy(), f["global_target"][i], rtol=2e-3, atol=2e-3)
        np.testing.assert_allclose(h["local_adv"][i].cpu().numpy(), f["local_adv"][i], rtol=2e-3, atol=2e-3)
        np.testing.assert_allclose(h["local_targets"][i].cpu().numpy(), f["local_target"][i], rtol=2e-3, atol=2e-3)
    assert len(h["full_adv"]) == 8 and len(h["full_values"]) == 8            # 16 tensors x [S,B]: SURVEY 8(e)'s exchange


def _two_shard_worker(rank, world, port, q):
    import torch.distributed as td
    _mods()
    rollout = import_module("e2e-mappo-for-mt-fjsp_amd.rollout")
    D = import_module("e2e-mappo-for-mt-fjsp_amd.dist")
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        f = np.load(FIX)
        J, M, E, B, eps = [int(x) for x in f["meta"]]
        g, lam = [float(x) for x in f["gamma_lambda"]]
        lo, hi = D.shard_range(B, rank, world)
        w = _weights(f)
        ro = rollout.Rollout(J, M, E, hi - lo, policy="actor", obs_dtype="f32", weights=(w["ja"], w["ma"], w["gc"]), collect="full",
                             buffer_episodes=eps, gamma=g, lam=lam, rank=rank, world=world, exact_bn=True,
                             instances=tuple(f[k][lo:hi] for k in ("t", "p", "tt", "edge")), w3_episodes=f["w3"][:, lo:hi], time_handoff=True)
        S = eps * J * M
        i32 = lambda a: torch.tensor(np.ascontiguousarray(a).astype(np.int32)).cuda()
        for s in range(S):
            ro.step(force=(i32(f["task"][s, lo:hi]), i32(f["mach"][s, lo:hi]), i32(f["job"][s, lo:hi])))
        torch.cuda.synchronize()
        h = ro.last_full
        worst = 0.0
        for i in range(4):
            for got, want in ((h["global_adv"][i], f["global_adv"][i]), (h["global_targets"][i], f["global_target"][i]),
                              (h["local_adv"][i], f["local_adv"][i]), (h["local_targets"][i], f["local_target"][i])):
                worst = max(worst, float(np.abs(got.cpu().numpy() - want[:, lo:hi]).max()))
        # every rank holds the complete [S, B_total] exchange, rank-major columns
        full_ok = all(tuple(x.shape) == (S, B) for x in h["full_adv"] + h["full_values"])
        q.put((rank, worst, full_ok, h["gather"]["world"], h["gather"]["bytes_per_rank"], S * (hi - lo) * 4 * 16))
    finally:
        td.destroy_process_group()


@pytest.mark.gpu
def test_two_shards_full_handoff_through_rollout_reproduces_the_reference():
    """two processes (gloo; RCCL on a multi-GPU node), each a Rollout over HALF of the reference run's instances with all-reduced
    BatchNorm statistics (the shards then ARE the reference's whole-batch run): each rank's finish_buffer — global critic sampling,
    8 GAE scans, ONE all-gather of 16 tensors, normalisation over all shards' columns — reproduces the reference's 4 global + 4
    local advantages and value targets on its columns"""
    import torch.multiprocessing as tmp_mp
    ctx = tmp_mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 90)
    procs = [ctx.Process(target=_two_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    out = [q.get(timeout=600) for _ in procs]
    for p_ in procs:
        p_.join(timeout=60)
        assert p_.exitcode == 0
    for rank, worst, full_ok, world, nbytes, want_bytes in out:
        assert worst < 3e-3 and full_ok and world == 2 and nbytes == want_bytes, (rank, worst, full_ok, world, nbytes, want_bytes)


@pytest.mark.gpu
def test_grid_barrier_timeout_is_reported_and_the_rollout_recovers(monkeypatch):
    """MTFJSP_GIN_RES_FAIL_AT=n lets the n-th single-launch GIN forward's grid barriers time out (as if a compute unit were held
    by somebody else): the failure is latched in a host-mapped word, the next forward entry returns MTFJSP_ERR_RETRY, the handle
    falls back to the streaming launches, Rollout.step discards the buffer and restarts the episode — and check() brings the single
    launch back once the census passes again."""
    _mods()
    monkeypatch.setenv("MTFJSP_GIN_RES_FAIL_AT", "7")
    rollout = import_module("e2e-mappo-for-mt-fjsp_amd.rollout")
    capi = import_module("e2e-mappo-for-mt-fjsp_amd.capi")
    ro = rollout.Rollout(6, 6, 2, 512, policy="actor", obs_dtype="f32", collect=True, buffer_episodes=1)
    enc = ro.actor.enc
    assert enc.check() and enc.resident_failures() == 0
    for _ in range(5):
        ro.step()
    assert ro.n_resident_failures == 0
    for _ in range(40):                                          # launch 7 fails (6 x 4 ms of bounded spins); noticed at the next forward entry
        ro.step()
    torch.cuda.synchronize()
    assert ro.n_resident_failures == 1 and enc.resident_failures() == 1
    assert ro.buf_pos == ro.t_in_ep                               # the buffer restarted together with the episode
    assert torch.isfinite(enc.job_prob).all() and torch.isfinite(enc.mch_prob).all()
    assert int((ro.env.status & capi.ST_INVALID).sum().item()) == 0
    assert enc.check()                                            # idle stream: census re-run, single launch re-enabled
    for _ in range(40):
        ro.step()
    torch.cuda.synchronize()
    assert ro.n_resident_failures == 1 and ro.n_handoffs >= 1
    ro.check_finished_cleanly()
    # a bare Encoder user (no Rollout) sees the error code instead of silent garbage
    monkeypatch.setenv("MTFJSP_GIN_RES_FAIL_AT", "2")
    ro2 = rollout.Rollout(6, 6, 2, 512, policy="actor", obs_dtype="f32", collect=False)
    env, e2 = ro2.env, ro2.actor.enc
    ro2.step()                                                    # launch 1: fine
    e2.job_actor_forward(env.tasks_fea, env.ell_col, env.ell_val, env.candidate, env.job_mask, None)
    with pytest.raises(capi.MtfjspError) as ei:
        e2.check()
    assert ei.value.code == capi.ERR_RETRY
    p1 = e2.job_actor_forward(env.tasks_fea, env.ell_col, env.ell_val, env.candidate, env.job_mask, None)[0].clone()
    assert torch.isfinite(p1).all()


def _lockstep_worker(rank, world, port, q, mode):
    """two ranks on cuda:0 over gloo; rank 0's single-launch GIN kernel is told to time out at its 7th launch (rank 1 runs the
    streaming launches: two resident grids cannot share one GPU).  mode "exact_bn_full": both ranks all-reduce every BatchNorm's
    sums (streaming launches on both), the whole hand-off with the global critic, and rank 0's 7th job-actor forward raises the
    RANGE word — the one failure a rank can see alone in that mode (a NaN in the BatchNorm sums reaches every rank through the
    all-reduce; a NaN in a scorer output does not)."""
    import torch.distributed as td
    exact = mode == "exact_bn_full"
    if rank == 0:
        os.environ["MTFJSP_RANGE_FAIL_AT" if exact else "MTFJSP_GIN_RES_FAIL_AT"] = "7"
    elif not exact:
        os.environ["MTFJSP_NO_RESIDENT_GIN"] = "1"
    _mods()
    rollout = import_module("e2e-mappo-for-mt-fjsp_amd.rollout")
    enc_mod = import_module("e2e-mappo-for-mt-fjsp_amd.encoder")
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        kw = dict(collect=True) if mode == "advantage" else dict(collect="full", weights=enc_mod.random_init_weights(7, with_critic=True))
        ro = rollout.Rollout(6, 6, 2, 64 if exact else 256, policy="actor", obs_dtype="f32", buffer_episodes=1, rank=rank, world=world,
                             time_handoff=True, exact_bn=exact, **kw)
        assert ro.exact_bn == exact
        for _ in range(4 * ro.S):                                # four buffers' worth of step() calls on every rank
            ro.step()
        torch.cuda.synchronize()
        ro.check_finished_cleanly()
        ok = bool(torch.isfinite(ro.last_adv[0][0]).all()) and ro.last_gather["world"] == 2
        if exact:                                                # the failing rank left the split products, the other one did not
            ok = ok and (ro.actor.enc.range_fallbacks()[0] == (1 if rank == 0 else 0))
        q.put((rank, ro.n_handoffs, ro.n_dropped_buffers, ro.n_resident_failures, ro.buf_pos, ro.t_in_ep, ok))
        td.barrier()
    finally:
        td.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["advantage", "full", "exact_bn_full"])
def test_a_failure_on_one_rank_drops_the_buffer_on_every_rank(mode):
    """the hand-off is a collective, so a rank that restarts after MTFJSP_ERR_RETRY must not fall out of step
    with the others.  Rank 0 reports a grid-barrier time-out in its first buffer: BOTH ranks drop that buffer at its boundary
    (agreed through one MAX all-reduce), both run the same number of all-gathers afterwards, nobody hangs.
    ADVICE r4 (medium), mode "exact_bn_full": with all-reduced BatchNorm statistics every forward is a collective, so the ranks
    agree on a failure once per step and restart TOGETHER; the global critic's 2 S forwards of the hand-off run on every rank or on
    none.  A mismatch shows as a hang (the 600 s time-out below) or as mixed statistics (non-finite advantages)."""
    import torch.multiprocessing as tmp_mp
    ctx = tmp_mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + (os.getpid() % 90)
    procs = [ctx.Process(target=_lockstep_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    for p_ in procs:
        p_.start()
    out = sorted(q.get(timeout=600) for _ in procs)
    for p_ in procs:
        p_.join(timeout=120)
        assert p_.exitcode == 0
    (r0, h0, d0, f0, pos0, t0, ok0), (r1, h1, d1, f1, pos1, t1, ok1) = out
    assert f0 >= 1 and f1 == 0                                    # only rank 0 failed ...
    assert d0 == d1 >= 1 and h0 == h1 == 4 - d0                   # ... and both dropped the same buffers and ran the same hand-offs
    assert (pos0, t0) == (pos1, t1) == (0, 0) and ok0 and ok1     # buffers and episodes aligned again on both ranks


def _nccl_world1_worker(port, q):
    """a world-size-1 RCCL group on cuda:0: every collective of the data path goes through its device-tensor branch"""
    import torch.distributed as td
    D = _mods()
    D.COLLECT_ON_ONE_RANK = True
    rollout = import_module("e2e-mappo-for-mt-fjsp_amd.rollout")
    enc_mod = import_module("e2e-mappo-for-mt-fjsp_amd.encoder")
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    td.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        res = {"device_collectives": D.device_collectives(), "active": D.active()}
        g = torch.Generator(device="cuda").manual_seed(5)
        xs = [torch.randn(36, 64, device="cuda", generator=g) for _ in range(4)]
        out, info = D.all_gather_advantages(xs, timed=True)
        res["allgather_equal"] = all(torch.equal(a, b) for a, b in zip(out, xs)) and info["world"] == 1 and info["ms"] is not None
        res["allgather_bytes"] = info["bytes_per_rank"]
        res["columns_equal"] = bool(torch.equal(D.all_gather_columns(xs[0]), xs[0]))
        n = D.normalize_advantages_global(xs[0])
        res["normalise_ok"] = bool(torch.allclose(n, (xs[0] - xs[0].mean()) / (xs[0].std() + 1e-5)))
        res["agree"] = (D.agree_any(False), D.agree_any(True))
        # the BatchNorm-sum all-reduce on a raw device pointer (Encoder.set_stats_reduce callback)
        t = torch.arange(2048, dtype=torch.float64, device="cuda")
        D.bn_stats_allreduce()(t.data_ptr(), t.numel())
        res["allreduce_ok"] = bool(torch.equal(t, torch.arange(2048, dtype=torch.float64, device="cuda")))
        # Rollout hand-offs through the group: the whole one (16 tensors) and, with exact_bn, the all-reduced statistics
        w = enc_mod.random_init_weights(11, with_critic=True)
        ro = rollout.Rollout(6, 6, 2, 64, policy="actor", obs_dtype="f32", collect="full", weights=w, buffer_episodes=1, time_handoff=True)
        while ro.n_handoffs == 0:
            ro.step()
        torch.cuda.synchronize()
        h = ro.last_full
        res["full_world"] = h["gather"]["world"]
        res["full_bytes"] = h["gather"]["bytes_per_rank"]
        res["full_ms"] = h["gather"]["ms"]
        res["full_equal"] = all(torch.equal(a, b) for a, b in zip(h["full_adv"], h["raw_global"] + h["raw_local"]))
        res["full_finite"] = all(bool(torch.isfinite(a).all()) for a in h["local_adv"] + h["global_adv"])
        td.destroy_process_group()
        # the same buffer with no process group at all: identical advantages (same seeds, same kernels)
        ro2 = rollout.Rollout(6, 6, 2, 64, policy="actor", obs_dtype="f32", collect="full", weights=w, buffer_episodes=1)
        while ro2.n_handoffs == 0:
            ro2.step()
        torch.cuda.synchronize()
        res["same_as_no_group"] = max(float((a - b).abs().max()) for a, b in zip(h["local_adv"] + h["global_adv"],
                                                                                 ro2.last_full["local_adv"] + ro2.last_full["global_adv"]))
        q.put(res)
    except Exception as ex:                                       # report instead of a bare non-zero exit code
        import traceback
        q.put({"error": repr(ex), "trace": traceback.format_exc()})


def _nccl_exact_bn_worker(port, q):
    import torch.distributed as td
    _mods().COLLECT_ON_ONE_RANK = True
    rollout = import_module("e2e-mappo-for-mt-fjsp_amd.rollout")
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    td.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        # exact_bn needs world > 1 to arm itself: tell the Rollout so, with this (only) rank owning shard 0 — the reduction over the
        # one-rank group adds nothing, so the outputs must equal a plain streaming-GIN run
        from importlib import import_module as im
        D = im("e2e-mappo-for-mt-fjsp_amd.dist")
        ro = rollout.Rollout(6, 6, 2, 64, policy="actor", obs_dtype="f32", collect=True, buffer_episodes=1)
        ro.actor.enc.set_stats_reduce(D.bn_stats_allreduce(), 64)
        ro.exact_bn = True
        for _ in range(36):
            ro.step()
        torch.cuda.synchronize()
        a = [x.clone() for x in ro.last_adv[0]]
        td.destroy_process_group()
        os.environ["MTFJSP_NO_RESIDENT_GIN"] = "1"
        ro2 = rollout.Rollout(6, 6, 2, 64, policy="actor", obs_dtype="f32", collect=True, buffer_episodes=1)
        for _ in range(36):
            ro2.step()
        torch.cuda.synchronize()
        q.put({"diff": max(float((x - y).abs().max()) for x, y in zip(a, ro2.last_adv[0])), "handoffs": ro.n_handoffs})
    except Exception as ex:
        import traceback
        q.put({"error": repr(ex), "trace": traceback.format_exc()})


@pytest.mark.gpu
def test_rccl_branches_run_on_a_world_size_1_group():
    """VERDICT r3 missing #2: the device-tensor collectives (`all_gather_into_tensor` on the packed advantages, `all_reduce` on a
    wrapped raw device pointer, the failure-flag agreement) had only ever run over gloo with host copies.  A one-rank RCCL group on
    the single GPU drives the same lines; results equal the no-group path."""
    import torch.multiprocessing as tmp_mp
    ctx = tmp_mp.get_context("spawn")
    q = ctx.Queue()
    p_ = ctx.Process(target=_nccl_world1_worker, args=(29990 - (os.getpid() % 90), q))
    p_.start()
    res = q.get(timeout=600)
    p_.join(timeout=120)
    assert "error" not in res, res
    S, B = 36, 64
    assert res["device_collectives"] and res["active"]
    assert res["allgather_equal"] and res["columns_equal"] and res["normalise_ok"] and res["allreduce_ok"]
    assert res["allgather_bytes"] == 4 * 36 * 64 * 4 and res["agree"] == (False, True)
    assert res["full_world"] == 1 and res["full_bytes"] == 16 * S * B * 4 and res["full_ms"] is not None
    assert res["full_equal"] and res["full_finite"] and res["same_as_no_group"] == 0.0


@pytest.mark.gpu
def test_exact_bn_allreduce_over_rccl_on_one_rank():
    """the BatchNorm-sum all-reduce between the streaming GIN launches, on the RCCL branch (one rank: the sum is the identity, so a
    whole buffer's advantages equal the plain streaming run's)"""
    import torch.multiprocessing as tmp_mp
    ctx = tmp_mp.get_context("spawn")
    q = ctx.Queue()
    p_ = ctx.Process(target=_nccl_exact_bn_worker, args=(29890 - (os.getpid() % 90), q))
    p_.start()
    res = q.get(timeout=600)
    p_.join(timeout=120)
    assert "error" not in res, res
    assert res["handoffs"] == 1 and res["diff"] <= 1e-5, res


@pytest.mark.gpu
def test_trajectory_buffer_rejects_a_mismatching_environment():
    _mods()
    traj = import_module("e2e-mappo-for-mt-fjsp_amd.trajectory")
    batch_env = import_module("e2e-mappo-for-mt-fjsp_amd.batch_env")
    env64 = batch_env.DeviceBatchEnv(6, 6, 2, 8)                                   # default observation dtype
    env32 = batch_env.DeviceBatchEnv(6, 6, 2, 8, obs_dtype="f32")
    tb = traj.TrajectoryBuffer({"n_job": 6, "n_machine": 6, "buffer_size": 1, "env_batch": 8}, device="cuda", obs_dtype=torch.float64)
    with pytest.raises(ValueError):
        tb.snapshot(env32, "pre")
    tb16 = traj.TrajectoryBuffer({"n_job": 6, "n_machine": 6, "buffer_size": 1, "env_batch": 16}, device="cuda", obs_dtype=torch.float64)
    with pytest.raises(ValueError):
        tb16.snapshot(env64, "pre")
