"""Pins the CPU oracle (oracle/mtfjsp_oracle.c) against golden vectors captured
from the reference itself (oracle/ref_harness/gen_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle.env_oracle import OracleBatch, lib
from trace_utils import TRACES, load, replay


def _make(t, p, tt, edge, left_shift, w_cfg, divisor, gamma, J):
    return OracleBatch(t, p, tt, edge, left_shift=left_shift, w_cfg=w_cfg, divisor=divisor, gamma=gamma, n_job=J)


@pytest.mark.parametrize("name", TRACES)
def test_oracle_bit_exact_on_reference_trace(name):
    g = load(name)
    n = replay(g, _make, exact=True)
    assert n > 0


def test_np_sum_restated():
    rs = np.random.RandomState(0)
    L = lib()
    for n in (1, 5, 7, 8, 9, 36, 100, 128, 129, 400, 1000):
        for _ in range(50):
            a = rs.uniform(0.5, 2500.0, n)
            assert L.or_np_sum(a, n) == np.sum(a)
