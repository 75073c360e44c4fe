#!/usr/bin/env python3
"""Golden fixture for the trajectory buffer: feed one recorded J6M6E2 episode (first 2 instances of
tests/golden/trace_j6m6e2_train16_mask.npz, itself recorded from the reference env) through the REFERENCE's
`trainer.replaybuffer.ReplayBuffer` exactly as Run.py:440-514 does, and record what `numpy_to_tensor_operation()` returns.

ORACLE HARNESS ONLY (build container; imports /root/reference, copies nothing of it).
Usage: python oracle/ref_harness/gen_golden_buffer.py
"""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from bootstrap import bootstrap  # noqa: E402

bootstrap(models=True)   # installs the CPU `trainer.train_device` stand-in the buffer imports
import torch  # noqa: E402

with contextlib.redirect_stdout(io.StringIO()):
    from trainer.replaybuffer import ReplayBuffer  # noqa: E402

GOLDEN = os.path.join(HERE, "..", "..", "tests", "golden")
NAMES = ["adj", "tasks_fea", "candidate", "mask_operation", "a_operation", "a_logprob_operation",
         "adj_", "tasks_fea_", "candidate_", "mask_operation_", "r_operation", "done_operation",
         "machine_fea2", "a", "a_logprob", "machine_fea2_", "mask_machine_",
         "mk", "pt", "tt", "it", "machine_fea1", "rw", "job_v", "machine_v", "job_v_", "machine_v_"]


def main():
    g = np.load(os.path.join(GOLDEN, "trace_j6m6e2_train16_mask.npz"))
    J, M, E, B16 = [int(x) for x in g["meta"][:4]]
    T, B, ep = J * M, 2, 0
    assert list(g["kept_steps"][ep]) == list(range(T))
    args = {"n_job": J, "n_machine": M, "buffer_size": 1, "env_batch": B, "gcn_input_dim": 12}
    rb = ReplayBuffer(args)
    rs = np.random.RandomState(7)
    feas = g["t"][:B] >= 0
    fed = {k: [] for k in ("a_o_logprob", "a_m_logprob", "j_v", "m_v", "j_v_", "m_v_")}
    adj, fea = g["adj0"][ep][:B].astype(np.float64), g["tfea0"][ep][:B * T]
    cand, mask, mf2 = g["cand0"][ep][:B], g["mask0"][ep][:B].astype(bool), g["mfea2_0"][ep][:B]
    j_v_prev = m_v_prev = None
    for s in range(T):
        adj_, fea_ = g["adj"][ep, s][:B].astype(np.float64), g["tfea"][ep, s][:B * T]
        cand_, mask_, mf2_ = g["cand"][ep, s][:B], g["mask"][ep, s][:B].astype(bool), g["mfea2"][ep, s][:B]
        info = g["info"][ep, s][:B]
        task, mach = g["actions"][ep, s][:B, 0], g["actions"][ep, s][:B, 1]
        a_o = torch.tensor(g["job_actions"][ep, s][:B], dtype=torch.long)
        lp_o, lp_m = torch.tensor(rs.randn(B), dtype=torch.float), torch.tensor(rs.randn(B), dtype=torch.float)
        j_v, m_v = torch.tensor(rs.randn(B, 2), dtype=torch.float), torch.tensor(rs.randn(B, 2), dtype=torch.float)
        if s >= 1:                                            # Run.py:451-454
            rb.store_v_next(j_v_=j_v, m_v_=m_v)
            fed["j_v_"].append(j_v.numpy()); fed["m_v_"].append(m_v.numpy())
        done = info[:, 1].astype(bool)
        if done.all():                                        # Run.py:455-474: value of the terminal state
            jl, ml = torch.tensor(rs.randn(B, 2), dtype=torch.float), torch.tensor(rs.randn(B, 2), dtype=torch.float)
            rb.store_v_next(j_v_=jl, m_v_=ml)
            fed["j_v_"].append(jl.numpy()); fed["m_v_"].append(ml.numpy())
        mmask = torch.tensor(~feas[np.arange(B), task][:, None, :])
        rb.store_operation(adj=adj, fea=fea, candidate=cand, mask=torch.tensor(mask), a_o=a_o, a_o_logprob=lp_o,
                           r=info[:, 0], adj_=adj_, fea_=fea_, candidate_=cand_, mask_=torch.tensor(mask_),
                           mch_fea1=g["mfea1"][ep, s][:B], mch_fea2=mf2, mch_fea2_=mf2_,
                           a_m=torch.tensor(mach, dtype=torch.long), a_m_logprob=lp_m, dw=None, done=done,
                           mask_machine_=mmask, mk=info[:, 2], pt=info[:, 4], tt=info[:, 5], it=info[:, 3],
                           rw=g["w3"][ep][:B], j_v=j_v, m_v=m_v)
        for k, v in (("a_o_logprob", lp_o), ("a_m_logprob", lp_m), ("j_v", j_v), ("m_v", m_v)):
            fed[k].append(v.numpy())
        adj, fea, cand, mask, mf2 = adj_, fea_, cand_, mask_, mf2_
    assert rb.count_operation == T and rb.count_operation_ == T
    out = rb.numpy_to_tensor_operation()
    assert len(out) == len(NAMES)
    d = {"meta": np.array([J, M, E, B, ep], np.int32)}
    for n, v in zip(NAMES, out):
        a = v.cpu().numpy()
        if n in ("adj", "adj_"):
            assert np.array_equal(a, a.astype(np.int16))
            a = a.astype(np.int16)
        d["out_" + n] = a
        d["dtype_" + n] = np.array(str(v.dtype))
    for k, v in fed.items():
        d["fed_" + k] = np.stack(v)
    path = os.path.join(GOLDEN, "replaybuffer_j6m6e2_b2.npz")
    np.savez_compressed(path, **d)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
