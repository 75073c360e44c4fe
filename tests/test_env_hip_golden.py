"""PARITY (GPU): the HIP environment, called through the C ABI, replays the reference's golden traces.

Bar (BASELINE.json north_star): integer scheduling state bit-exact; floats within 1e-5.
The kernels evaluate binary64 in the reference's operation order (no FMA contraction), so this test
demands MORE than the bar: every float (st/ft, 5 rewards, 4 scaled rewards, cumulative costs, tasks_fea,
m_fea1, m_fea2, scaler state) must be bit-identical (np.array_equal) to the reference's output.
"""
import numpy as np
import pytest

from trace_utils import TRACES, load, replay

pytestmark = pytest.mark.gpu


def _make(obs_dtype):
    def mk(t, p, tt, edge, left_shift, w_cfg, divisor, gamma, J):
        from hip_impl import HipImpl
        return HipImpl(t, p, tt, edge, left_shift, w_cfg, divisor, gamma, J, obs_dtype=obs_dtype)
    return mk


@pytest.mark.parametrize("name", TRACES)
def test_hip_env_bit_exact_on_reference_trace(name):
    g = load(name)
    paths = []

    def chk(impl, g_, ep, i, s):
        pass
    n = replay(g, _make("f64"), exact=True, check=chk)
    assert n > 0


@pytest.mark.parametrize("name", ["trace_j6m6e2_eval16_free", "trace_j10m10e2_b2_free"])
def test_hip_env_f32_observations_are_the_rounded_f64_ones(name):
    """obs_dtype=f32 must be exactly float32(reference f64 observation) — what actor_critic.py:143 `.float()` does.
    State, rewards and masks stay f64/int and bit-exact."""
    g = load(name)
    J, M, E, B, episodes, left_shift, keep_every = [int(x) for x in g["meta"]]
    T = J * M
    from hip_impl import HipImpl
    w = g["cfg_w"]
    impl = HipImpl(g["t"], g["p"], g["tt"], g["edge"], bool(left_shift), tuple(w[:3]), float(w[3]), float(w[4]), J, obs_dtype="f32")
    impl.scaler_init()
    feas = g["t"] >= 0
    obs = impl.reset(g["w3"][0])
    assert impl.env.tasks_fea.dtype.is_floating_point and impl.env.tasks_fea.element_size() == 4
    assert np.array_equal(impl.env.tasks_fea.cpu().numpy(), g["tfea0"][0].astype(np.float32))
    kept = list(g["kept_steps"][0])
    for s in range(T):
        act = g["actions"][0, s]
        mm = ~feas[np.arange(B), act[:, 0]]
        mf1 = impl.env.observe_mfea1(act[:, 0].astype(np.int32), mm).cpu().numpy()
        info, raw, _ = impl.step(act[:, 0], act[:, 1])
        if s in kept:
            i = kept.index(s)
            assert mf1.dtype == np.float32 and np.array_equal(mf1, g["mfea1"][0, i].astype(np.float32))
            assert np.array_equal(impl.env.tasks_fea.cpu().numpy(), g["tfea"][0, i].astype(np.float32))
            assert np.array_equal(impl.env.m_fea2.cpu().numpy(), g["mfea2"][0, i].astype(np.float32))
            assert np.array_equal(impl.env.dense_adj().cpu().numpy(), g["adj"][0, i].astype(np.float64))
            assert np.array_equal(raw, g["raw_rewards"][0, i])
            assert np.array_equal(info, g["info"][0, i])


def test_invalid_actions_are_rejected_without_touching_state():
    g = load("trace_j6m6e2_eval8_sticky")
    J, M, E, B = [int(x) for x in g["meta"][:4]]
    from hip_impl import HipImpl, capi
    w = g["cfg_w"]
    impl = HipImpl(g["t"], g["p"], g["tt"], g["edge"], True, tuple(w[:3]), float(w[3]), float(w[4]), J)
    impl.scaler_init()
    impl.reset(g["w3"][0])
    act = g["actions"][0, 0]
    impl.step(act[:, 0], act[:, 1])
    before = impl.state(); obs_b = impl.observe()
    with pytest.raises(capi.MtfjspError) as ei:
        impl.step(act[:, 0], act[:, 1])            # same tasks again: already scheduled
    assert ei.value.code == capi.ERR_ACTION
    st = impl.env.status.cpu().numpy()
    assert (st & capi.ST_INVALID).all()
    after = impl.state(); obs_a = impl.observe()
    for k in before:
        assert np.array_equal(before[k], after[k], equal_nan=True), k
    for k in obs_b:
        assert np.array_equal(obs_b[k], obs_a[k]), k
    bad = act[:, 0].copy(); bad[:] = 1              # op 1 of job 0 while op 0 may be unscheduled
    unsched0 = before["mach"][:, 0] < 0
    if unsched0.any():
        with pytest.raises(capi.MtfjspError):
            impl.step(bad, act[:, 1])
        st = impl.env.status.cpu().numpy()
        assert ((st & capi.ST_INVALID) != 0)[unsched0].all()
