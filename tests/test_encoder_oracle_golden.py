"""Pins the fp32 encoder restatement (oracle/encoder_oracle.py) against outputs of the
reference modules (tests/golden/encoder_*.npz).  CPU only.
Tolerances: f32 round-off of a different op order (1e-6-level diffs observed)."""
import os

import numpy as np
import pytest

from oracle import encoder_oracle as eo

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["encoder_j6m6e2_top1", "encoder_j6m6e2_rand", "encoder_j10m10e2_rand", "encoder_j20m20e4_rand"])
def test_encoder_oracle_matches_reference_modules(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    J, M, E, B = [int(x) for x in g["meta"]]
    T = J * M
    ja, ma = eo.split_weights(g)
    for s in g["steps"]:
        p = f"s{int(s)}_"
        col, val = eo.ell_from_dense(g[p + "adj"])
        out = eo.job_actor_forward(ja, g[p + "tfea"], col, val, g[p + "cand"], g[p + "mask"], g[p + "h_m_in"], B, T)
        scale = max(1.0, float(np.abs(g[p + "h_nodes"]).max()))
        np.testing.assert_allclose(out["h_nodes"], g[p + "h_nodes"], rtol=0, atol=2e-5 * scale)
        np.testing.assert_allclose(out["h_pooled"], g[p + "h_o"], rtol=0, atol=2e-5 * scale)
        np.testing.assert_allclose(out["prob"], g[p + "job_prob"], rtol=0, atol=1e-5)
        np.testing.assert_allclose(out["job_v"], g[p + "job_v"], rtol=1e-4, atol=1e-4)
        assert np.array_equal(out["greedy_job"], g[p + "job_index"])
        assert np.array_equal(out["greedy_task"], g[p + "task_index"])
        np.testing.assert_allclose(out["greedy_logp"], g[p + "job_logp"], rtol=0, atol=1e-5)
        mo = eo.machine_actor_forward(ma, g[p + "mfea1"], g[p + "mfea2"], g[p + "h_o"], g[p + "mmask"], B, M)
        np.testing.assert_allclose(mo["prob"], g[p + "mch_prob"], rtol=0, atol=2e-5)
        np.testing.assert_allclose(mo["h_pooled"], g[p + "h_m"], rtol=0, atol=2e-5)
        np.testing.assert_allclose(mo["mach_v"], g[p + "mach_v"], rtol=1e-4, atol=1e-4)
        gv = eo.global_critic_forward(eo.critic_weights(g), g[p + "tfea"], col, val, g[p + "mfea1"], g[p + "mfea2"], B, T, M)
        np.testing.assert_allclose(gv, g[p + "global_v"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("batch", ["b320", "b512"])
def test_encoder_oracle_matches_reference_modules_mid_batches(batch):
    """B = 320 / 512 (whole-batch BatchNorm over 11 520 / 18 432 rows): the fixture the GPU tests use to pin k_gin_res in its
    several-instances-per-workgroup partition (node embeddings are not stored in this one)"""
    g = np.load(os.path.join(GOLDEN, "encoder_j6m6e2_mid.npz"))
    J, M, E, B = [int(x) for x in g[batch + "_meta"]]
    T = J * M
    ja, ma = eo.split_weights(g)
    for s in g[batch + "_steps"]:
        p = f"{batch}_s{int(s)}_"
        col, val = eo.ell_from_dense(g[p + "adj"])
        out = eo.job_actor_forward(ja, g[p + "tfea"], col, val, g[p + "cand"], g[p + "mask"], g[p + "h_m_in"], B, T)
        scale = max(1.0, float(np.abs(g[p + "h_o"]).max()))
        np.testing.assert_allclose(out["h_pooled"], g[p + "h_o"], rtol=0, atol=2e-5 * scale)
        np.testing.assert_allclose(out["prob"], g[p + "job_prob"], rtol=0, atol=1e-5)
        np.testing.assert_allclose(out["job_v"], g[p + "job_v"], rtol=1e-4, atol=1e-4)
        top2 = np.sort(g[p + "job_prob"], axis=1)[:, -2:]
        clear = top2[:, 1] - top2[:, 0] > 1e-5
        assert np.array_equal(out["greedy_job"][clear], g[p + "job_index"][clear])
        mo = eo.machine_actor_forward(ma, g[p + "mfea1"], g[p + "mfea2"], g[p + "h_o"], g[p + "mmask"], B, M)
        np.testing.assert_allclose(mo["prob"], g[p + "mch_prob"], rtol=0, atol=2e-5)
        np.testing.assert_allclose(mo["h_pooled"], g[p + "h_m"], rtol=0, atol=2e-5 * max(1.0, float(np.abs(g[p + "h_m"]).max())))
        np.testing.assert_allclose(mo["mach_v"], g[p + "mach_v"], rtol=1e-4, atol=1e-4)
        gv = eo.global_critic_forward(eo.critic_weights(g), g[p + "tfea"], col, val, g[p + "mfea1"], g[p + "mfea2"], B, T, M)
        np.testing.assert_allclose(gv, g[p + "global_v"], rtol=1e-4, atol=1e-4)
