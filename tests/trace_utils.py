"""Shared replay helper: drive an env implementation through a golden trace and
compare every recorded output.  `impl` is anything with the OracleBatch method set
(oracle/env_oracle.py) — the HIP-backed batch exposes the same methods."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TRACES = ["trace_j6m6e2_train16_mask", "trace_j6m6e2_eval16_free", "trace_j6m6e2_eval8_sticky",
          "trace_j6m6e2_eval4_noleftshift", "trace_j10m6e2_b3_free", "trace_j10m10e2_b2_free",
          "trace_j10m10e2_b2_mask", "trace_j20m20e4_b1_free"]


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def routes_equal(got, ref):
    return np.array_equal(np.asarray(got, np.int64), np.asarray(ref, np.int64))


def replay(g, make_impl, exact=True, rtol=0.0, atol=0.0, check=None):
    """make_impl(t,p,tt,edge,left_shift,w_cfg,divisor,gamma,J) -> impl.
    exact=True: floats compared with array_equal; else allclose(rtol, atol).
    Integer state is always compared exactly."""
    J, M, E, B, episodes, left_shift, keep_every = [int(x) for x in g["meta"]]
    T = J * M
    w = g["cfg_w"]
    impl = make_impl(g["t"], g["p"], g["tt"], g["edge"], bool(left_shift), tuple(w[:3]), float(w[3]), float(w[4]), J)
    feas = g["t"] >= 0

    def feq(a, b, what):
        a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
        if exact:
            ok = np.array_equal(a, b, equal_nan=True)
        else:
            ok = np.allclose(a, b, rtol=rtol, atol=atol, equal_nan=True)
        assert ok, f"{what}: max abs diff {np.nanmax(np.abs(a - b))}"

    def ieq(a, b, what):
        assert np.array_equal(np.asarray(a).astype(np.int64), np.asarray(b).astype(np.int64)), what

    impl.scaler_init()
    n_checked = 0
    for ep in range(episodes):
        obs = impl.reset(g["w3"][ep])
        ieq(obs["adj"], g["adj0"][ep], f"ep{ep} adj0")
        feq(obs["tfea"], g["tfea0"][ep], f"ep{ep} tfea0")
        feq(obs["mfea2"], g["mfea2_0"][ep], f"ep{ep} mfea2_0")
        cand, mask = impl.job_mask_state()
        ieq(cand, g["cand0"][ep], "cand0"); ieq(mask, g["mask0"][ep], "mask0")
        impl.scaler_reset_returns()
        kept = list(g["kept_steps"][ep])
        tfea = obs["tfea"]
        for s in range(T):
            act = g["actions"][ep, s]
            task_idx, m_idx = act[:, 0], act[:, 1]
            mm = ~feas[np.arange(B), task_idx]
            mf1 = impl.mfea1(task_idx, mm, tfea)
            info, raw, paths = impl.step(task_idx, m_idx)
            cand, mask = impl.job_mask_update(g["job_actions"][ep, s])
            obs = impl.observe()
            tfea = obs["tfea"]
            if s in kept:
                i = kept.index(s)
                tag = f"ep{ep} step{s}"
                feq(mf1, g["mfea1"][ep, i], tag + " mfea1")
                ieq(obs["adj"], g["adj"][ep, i], tag + " adj")
                feq(obs["tfea"], g["tfea"][ep, i], tag + " tfea")
                ieq(obs["tfea"][:, [3, 4, 5, 8]], g["tfea"][ep, i][:, [3, 4, 5, 8]], tag + " tfea int cols")
                feq(obs["mfea2"], g["mfea2"][ep, i], tag + " mfea2")
                feq(raw, g["raw_rewards"][ep, i], tag + " raw rewards")
                ieq(info[:, 1], g["info"][ep, i][:, 1], tag + " done")
                feq(info, g["info"][ep, i], tag + " info (scaled rewards)")
                ieq(cand, g["cand"][ep, i], tag + " candidate"); ieq(mask, g["mask"][ep, i], tag + " job mask")
                st = impl.state()
                ieq(st["mach"], g["mach"][ep, i], tag + " machine"); ieq(st["sched"], g["sched"][ep, i], tag + " sched")
                feq(st["st"], g["st"][ep, i], tag + " st"); feq(st["ft"], g["ft"][ep, i], tag + " ft")
                ieq(st["routes"], g["routes"][ep, i], tag + " routes")
                feq(st["prev"], g["prev"][ep, i], tag + " prev costs")
                feq(st["scaler"], g["scaler"][ep, i], tag + " scaler state")
                ieq(impl.valid_action_mask(), g["vmask"][ep, i], tag + " valid_action_mask")
                if check:
                    check(impl, g, ep, i, s)
                n_checked += 1
    return n_checked
