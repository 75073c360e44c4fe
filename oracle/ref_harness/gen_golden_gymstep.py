#!/usr/bin/env python3
"""Golden fixture for the gym-style surface of ONE environment (env:434-467 spaces, env:716-974 step's 14-tuple, env:2515 reset's
9-tuple): the REFERENCE's DisjunctiveGraphJspEnv_singleStep, constructed exactly as trainer/parallel_env.py:110-118 constructs it,
replays instance 0 of tests/golden/trace_j6m6e2_eval16_free.npz; per step the entries the batched trace does not hold are recorded —
ft_s, it_s, the 3-column tasks_fea — plus the two spaces' shape / n / bounds / dtype.

ORACLE HARNESS ONLY (build container; imports /root/reference, copies nothing of it).
Usage: python oracle/ref_harness/gen_golden_gymstep.py
"""
import contextlib
import io
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from bootstrap import bootstrap, default_config  # noqa: E402

REF = bootstrap(models=False)
with contextlib.redirect_stdout(io.StringIO()):
    from graph_jsp_env.disjunctive_graph_jsp_env_singlestep import DisjunctiveGraphJspEnv_singleStep  # noqa: E402

GOLDEN = os.path.join(HERE, "..", "..", "tests", "golden")


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def main():
    g = np.load(os.path.join(GOLDEN, "trace_j6m6e2_eval16_free.npz"))
    J, M, E = [int(x) for x in g["meta"][:3]]
    T = J * M
    cfg = default_config(J, M, E, 1)
    env = quiet(DisjunctiveGraphJspEnv_singleStep, jps_instance=np.array([g["t"][0], g["p"][0]]), reward_function_parameters=cfg["reward_scaling"],
                default_visualisations=["gantt_console", "graph_console"], reward_function='wrk', ability_tr_mm=g["tt"][0],
                perform_left_shift_if_possible=True, configs=cfg)
    random.seed(1)                                      # the trace's w_seed: instance 0 takes the first three draws
    r = quiet(env.reset)
    out = {"meta": np.array([J, M, E], np.int32),
           "obs_shape": np.array(env.observation_space.shape, np.int64), "obs_low": np.array(float(np.min(env.observation_space.low))),
           "obs_high": np.array(float(np.max(env.observation_space.high))), "obs_dtype": np.array(str(np.dtype(env.observation_space.dtype))),
           "act_n": np.array(int(env.action_space.n), np.int64),
           "reset_ft_s": np.asarray(r[1], np.float64), "reset_it_s": np.asarray(r[2]), "reset_tfea3": np.asarray(r[4], np.float64),
           "reset_tfea": np.asarray(r[6], np.float64)}
    assert np.array_equal(np.asarray(r[6]), g["tfea0"][0][:T]), "the replay does not reproduce the trace's reset observation"
    ft, it, t3 = [], [], []
    for step in range(T):
        a, m = [int(x) for x in g["actions"][0, step][0]]
        res = quiet(env.step, [a, m])
        assert np.array_equal(np.asarray(res[13]), g["tfea"][0, step][:T]), f"step {step}: the replay left the trace"
        ft.append(np.asarray(res[8], np.float64).copy()); it.append(np.asarray(res[9]).copy()); t3.append(np.asarray(res[11], np.float64).copy())
    out["ft_s"] = np.stack(ft); out["it_s"] = np.stack(it); out["tfea3"] = np.stack(t3)
    np.savez_compressed(os.path.join(GOLDEN, "gymstep_j6m6e2.npz"), **out)
    print("wrote gymstep_j6m6e2.npz", {k: getattr(v, "shape", None) for k, v in out.items()})


if __name__ == "__main__":
    main()
