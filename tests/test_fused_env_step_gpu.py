"""The environment step as the tail of another launch — of the machine actor's heads launch (k_headsx_envstep, MTFJSP_FUSED_ENV=1) and
of the three-in-one launch (k_headsx_gat3x_headsx<1|2>, MTFJSP_FUSED_ENV3=1: two launches per rollout step at the headline shape;
Run.py:363-427: machine forward, then env.step, nothing in between) — against the stand-alone launch: the same device code on the same
inputs, so every observation, reward and scaler word must agree bit for bit over whole episodes, with and without the trajectory
record, f32 and f64 observations."""
from importlib import import_module

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(mode, obs, collect, monkeypatch, B, steps):
    import torch
    import mtfjsp_amd  # noqa: F401
    rollout = import_module("e2e-mappo-for-mt-fjsp_amd.rollout")
    monkeypatch.delenv("MTFJSP_FUSED_ENV", raising=False)
    monkeypatch.delenv("MTFJSP_FUSED_ENV3", raising=False)
    if mode == "machine_heads":
        monkeypatch.setenv("MTFJSP_FUSED_ENV", "1")              # (both off by default: measured slower; read when the handle is created)
    elif mode == "three_in_one":
        monkeypatch.setenv("MTFJSP_FUSED_ENV3", "1")
    ro = rollout.Rollout(6, 6, 2, B, policy="actor", obs_dtype=obs, collect=collect, seed=11)
    snaps = []
    for s in range(steps):
        ro.step()
        if s % 7 == 0 or s >= steps - 3:
            env = ro.env
            torch.cuda.synchronize()
            snaps.append([x.cpu().numpy().copy() for x in (env.tasks_fea, env.ell_col, env.ell_val, env.m_fea2, env.info, env.raw, env.job_mask,
                                                          env.candidate, env.status, ro.task, ro.mach)])
    extra = [ro.buf_r.cpu().numpy().copy(), ro.buf_done.cpu().numpy().copy()] if collect else []
    sc = ro.env.scaler_state() if hasattr(ro.env, "scaler_state") else None
    return snaps, extra, ro.actor.n_env_fused, sc


@pytest.mark.parametrize("obs,collect,B", [("f32", True, 4096), ("f64", False, 1000), ("f32", False, 37)])
def test_fused_step_is_the_stand_alone_step(obs, collect, B, monkeypatch):
    steps = 36 * 2 + 5
    b, eb, nfb, sb = _run("separate", obs, collect, monkeypatch, B, steps)
    assert nfb == 0, nfb                                          # the stand-alone launch really ran
    for mode in ("machine_heads", "three_in_one"):
        a, ea, nfa, sa = _run(mode, obs, collect, monkeypatch, B, steps)
        # the fused path really ran (the three-in-one launch exists where groups of 16 instances cover the batch, one per CU)
        assert nfa == (steps if mode == "machine_heads" or B == 4096 else 0), (mode, nfa)
        for x, y in zip(a, b):
            for u, v in zip(x, y):
                np.testing.assert_array_equal(u, v)
        for u, v in zip(ea, eb):
            np.testing.assert_array_equal(u, v)
        if sa is not None:
            for u, v in zip(sa, sb):
                np.testing.assert_array_equal(np.asarray(u), np.asarray(v))
