"""SURVEY.md §8f N4 — on-device instance generator: distributional parity with the reference's generator (the host
restatement in instances.py is bit-exact with the reference and serves as the sample to compare against), structural
invariants, reproducibility, and a rollout on generated instances."""
import os
import pickle
from importlib import import_module

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_device_generator_matches_reference_distribution(tmp_path):
    import mtfjsp_amd  # noqa: F401
    be = import_module("e2e-mappo-for-mt-fjsp_amd.batch_env")
    inst = import_module("e2e-mappo-for-mt-fjsp_amd.instances")
    J, M, E, B = 6, 6, 2, 4096
    T = J * M
    env = be.DeviceBatchEnv(J, M, E, B, obs_dtype="f32")
    env.generate_instances(seed=5)
    t, p, tt, edge = env.read_instances()
    ht, hp, htt, hedge = inst.generate_instances(B, J, M, E, seed=5)
    # structure
    assert np.array_equal(t < 0, p < 0) and (np.abs(t) >= 0.8).all() and (np.abs(t) <= 99 * 1.2).all()
    nbad = (t < 0).sum(-1)
    assert nbad.max() <= M - 1 and nbad.min() == 0                      # at least one feasible machine per task (k < M)
    assert np.array_equal(edge, hedge)
    assert np.allclose(tt, np.transpose(tt, (0, 2, 1))) and (np.diagonal(tt, axis1=1, axis2=2) == 0).all()
    shop = np.repeat(np.arange(E), M // E)
    same = shop[:, None] == shop[None, :]
    off = ~np.eye(M, dtype=bool)
    assert (tt[:, same & off] >= 1).all() and (tt[:, same & off] <= 10).all()
    assert (tt[:, ~same] >= 10).all() and (tt[:, ~same] <= 20).all()
    # distribution vs the reference generator's sample (two-sample comparisons, 147k task rows each)
    for a, b, tol in ((np.abs(t), np.abs(ht), 0.01), (np.abs(p), np.abs(hp), 0.01)):
        assert abs(a.mean() / b.mean() - 1) < tol and abs(a.std() / b.std() - 1) < tol
        qa, qb = np.quantile(a, [0.1, 0.5, 0.9]), np.quantile(b, [0.1, 0.5, 0.9])
        assert np.all(np.abs(qa / qb - 1) < 0.02)
    hist_d = np.bincount(nbad.ravel(), minlength=M) / nbad.size
    hist_h = np.bincount((ht < 0).sum(-1).ravel(), minlength=M) / nbad.size
    assert np.abs(hist_d - 1.0 / M).max() < 0.01 and np.abs(hist_d - hist_h).max() < 0.01      # k uniform on [0, M)
    which = (t < 0).mean((0, 1))
    assert np.abs(which - which.mean()).max() < 0.01                      # every machine equally likely to be infeasible
    for a, b in ((tt[:, same & off], htt[:, same & off]), (tt[:, ~same], htt[:, ~same])):
        assert abs(a.mean() / b.mean() - 1) < 0.01 and abs(a.std() / b.std() - 1) < 0.03
    # within a task the machine weights are independent of each other: correlation of |t| across machines comes from avg_t only
    w = np.abs(t) / np.abs(t).mean(-1, keepdims=True)
    assert abs(np.corrcoef(w[..., 0].ravel(), w[..., 1].ravel())[0, 1]) < 0.3
    # reproducible, seed- and offset-dependent
    env.generate_instances(seed=5)
    t2 = env.read_instances()[0]
    assert np.array_equal(t, t2)
    env.generate_instances(seed=6)
    assert not np.array_equal(t, env.read_instances()[0])
    env.generate_instances(seed=5, first_instance=B // 2)
    t3 = env.read_instances()[0]
    assert np.array_equal(t3[:B // 2], t[B // 2:])                        # shard b of a larger set = the same instances
    # exporter: the reference's pickle layout
    path = os.path.join(tmp_path, "ins.pkl")
    inst.export_pickle(path, t, p, tt, edge)
    with open(path, "rb") as f:
        back = pickle.load(f)
    assert isinstance(back, list) and len(back) == 4 and np.array_equal(back[0], t) and back[3].shape == (B, E, M // E)


def test_rollout_on_generated_instances_matches_the_oracle():
    import mtfjsp_amd  # noqa: F401
    import torch
    be = import_module("e2e-mappo-for-mt-fjsp_amd.batch_env")
    from oracle.env_oracle import OracleBatch
    J, M, E, B = 6, 6, 2, 64
    T = J * M
    env = be.DeviceBatchEnv(J, M, E, B, obs_dtype="f64")
    env.generate_instances(seed=11)
    t, p, tt, edge = env.read_instances()
    env.scaler_init()
    orc = OracleBatch(t, p, tt, edge); orc.scaler_init()
    w3 = np.full((B, 3), 1.0 / 3)
    env.reset(torch.as_tensor(w3, device="cuda")); orc.reset(w3)
    a = torch.zeros(B, dtype=torch.int32, device="cuda"); m = torch.zeros_like(a)
    for s in range(T):
        env.random_actions(3, s, a, m)
        env.step(a, m)
        info, raw, _ = orc.step(a.cpu().numpy(), m.cpu().numpy())
        assert np.array_equal(env.info.cpu().numpy(), info) and np.array_equal(env.raw.cpu().numpy(), raw)
    o = orc.observe(dense=False)
    assert np.array_equal(env.tasks_fea.cpu().numpy(), o["tfea"]) and np.array_equal(env.m_fea2.cpu().numpy().reshape(B, M, 8), o["mfea2"])


def test_device_reward_weights_follow_the_reference_distribution():
    """mtfjsp_draw_reward_weights = env.generate_random_weights("01") (env:1253-1259) on the device: three uniforms normalised
    by their sum, fresh per (episode, instance)."""
    import torch
    import mtfjsp_amd  # noqa: F401
    batch_env = import_module("e2e-mappo-for-mt-fjsp_amd.batch_env")
    inst = import_module("e2e-mappo-for-mt-fjsp_amd.instances")
    B = 20000
    env = batch_env.DeviceBatchEnv(6, 6, 2, B, obs_dtype="f32")
    w0 = env.draw_reward_weights(11, 0).cpu().numpy()
    w1 = env.draw_reward_weights(11, 1).cpu().numpy()
    w0b = env.draw_reward_weights(11, 0).cpu().numpy()
    assert np.array_equal(w0, w0b) and not np.array_equal(w0, w1)            # a pure function of (seed, episode, instance)
    assert len({r.tobytes() for r in w0}) == B
    np.testing.assert_allclose(w0.sum(1), 1.0, rtol=0, atol=1e-15)
    assert (w0 > 0).all() and (w0 < 1).all()
    import random
    ref = inst.random_weights(B, rng=random.Random(3))                       # the host stream of the reference
    for d in (w0, w1):
        np.testing.assert_allclose(d.mean(0), ref.mean(0), atol=0.01)
        np.testing.assert_allclose(d.std(0), ref.std(0), atol=0.01)
        np.testing.assert_allclose(np.quantile(d, [0.1, 0.5, 0.9], axis=0), np.quantile(ref, [0.1, 0.5, 0.9], axis=0), atol=0.015)


@pytest.mark.parametrize("obs", ["f32", "f64"])
def test_one_launch_episode_reset_equals_the_three_calls(obs):
    """mtfjsp_reset_episode (round 6: one launch per episode) against scaler_reset_returns + draw_reward_weights + reset: the drawn
    weights, every observation, the scaler state and the first steps of the next episode, bit for bit — in the middle of a run, so
    that the scaler carries statistics across the reset (pe:70-85) and only its returns R are zeroed (pt:123)."""
    import torch
    import mtfjsp_amd  # noqa: F401
    batch_env = import_module("e2e-mappo-for-mt-fjsp_amd.batch_env")
    capi = import_module("e2e-mappo-for-mt-fjsp_amd.capi")
    J, M, E, B = 6, 6, 2, 257
    T = J * M
    envs = []
    for k in range(2):
        env = batch_env.DeviceBatchEnv(J, M, E, B, obs_dtype=obs)
        env.generate_instances(seed=5)
        env.scaler_init()
        envs.append(env)
    a = torch.zeros(B, dtype=torch.int32, device=envs[0].device); m = torch.zeros_like(a)
    for ep in range(3):
        w_sep = envs[0].draw_reward_weights(21, ep)
        envs[0].scaler_reset_returns()
        envs[0].reset(w_sep)
        w_one = envs[1].reset_episode(21, ep)
        assert torch.equal(w_sep, w_one)
        for s in range(T if ep < 2 else 7):
            for k, env in enumerate(envs):
                env.random_actions(9, ep * T + s, a, m)
                env.step(a, m)
            for name in ("tasks_fea", "ell_col", "ell_val", "m_fea2", "info", "raw", "candidate", "job_mask", "status"):
                assert torch.equal(getattr(envs[0], name), getattr(envs[1], name)), (ep, s, name)
        assert np.array_equal(envs[0].read_state(capi.STATE_SCALER), envs[1].read_state(capi.STATE_SCALER), equal_nan=True)
        assert np.array_equal(envs[0].read_state(capi.STATE_W3), envs[1].read_state(capi.STATE_W3))
