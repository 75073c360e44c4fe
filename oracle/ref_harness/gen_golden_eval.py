#!/usr/bin/env python3
"""Golden fixture for the batched greedy evaluation: the reference evaluates with env_batch = 1
(trainer/validate.py:60-297), so every training-mode BatchNorm normalises over the rows of ONE instance.  This script
takes the recorded batched inputs of tests/golden/encoder_j6m6e2_{rand,top1}.npz (observations from the reference env),
feeds each instance ALONE (batch 1) through the REFERENCE job actor / machine actor with the recorded weights, and stores
the stacked per-instance outputs next to a copy of nothing else — inputs and weights stay in the encoder fixture.

ORACLE HARNESS ONLY (build container; imports /root/reference, copies nothing of it).
Usage: python oracle/ref_harness/gen_golden_eval.py
"""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from bootstrap import bootstrap, default_config  # noqa: E402

bootstrap(models=True)
import torch  # noqa: E402

torch.set_num_threads(1)
with contextlib.redirect_stdout(io.StringIO()):
    from algorithm.ppo_algorithm import PPOAlgorithm  # noqa: E402
    from model.gcn_mlp import g_pool_cal  # noqa: E402

GOLDEN = os.path.join(HERE, "..", "..", "tests", "golden")


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def main():
    NB = 8                                                        # instances evaluated one by one
    for name in ("encoder_j6m6e2_rand", "encoder_j6m6e2_top1"):
        g = np.load(os.path.join(GOLDEN, name + ".npz"))
        J, M, E, B = [int(x) for x in g["meta"]]
        T = J * M
        cfg = default_config(J, M, E, 1)
        ppo = quiet(PPOAlgorithm, cfg, False)
        for net, pre in ((ppo.job_actor, "w_ja."), (ppo.machine_actor_gcn, "w_ma.")):
            sd = net.state_dict()
            for k in sd:
                if pre + k in g.files:
                    sd[k] = torch.tensor(g[pre + k])
            net.load_state_dict(sd)
        gpool = g_pool_cal("average", 1, T, torch.device("cpu"))
        out = {"meta": np.array([J, M, E, NB], np.int32), "steps": g["steps"]}
        for s in g["steps"]:
            p = f"s{int(s)}_"
            rec = {k: [] for k in ("job_prob", "h_o", "job_v", "job_index", "task_index", "mch_prob", "h_m", "mach_v")}
            for b in range(NB):
                hm_in = g[p + "h_m_in"]
                with torch.no_grad():
                    task_index, action_index, log_a, prob, h_o, job_v = ppo.job_actor(
                        x_fea=g[p + "tfea"][b * T:(b + 1) * T], graph_pool_avg=gpool, padded_nei=None,
                        adj=g[p + "adj"][b:b + 1].astype(np.float64), candidate=g[p + "cand"][b:b + 1],
                        h_g_m_pooled=None if hm_in.size == 0 else torch.tensor(hm_in[b:b + 1]),
                        mask_operation=torch.tensor(g[p + "mask"][b:b + 1]).bool(), use_greedy=True)
                    # machine actor on the RECORDED m_fea1 / mask (they belong to the recorded batched decision) and on this
                    # instance's own job embedding
                    mch_prob, h_m, mach_v = ppo.machine_actor_gcn(
                        machine_fea_1=g[p + "mfea1"][b:b + 1], machine_fea_2=g[p + "mfea2"][b:b + 1], h_pooled_o=h_o,
                        machine_mask=torch.tensor(g[p + "mmask"][b:b + 1]).bool())
                for k, v in (("job_prob", prob), ("h_o", h_o), ("job_v", job_v), ("job_index", action_index),
                             ("task_index", task_index), ("mch_prob", mch_prob), ("h_m", h_m), ("mach_v", mach_v)):
                    rec[k].append(v.numpy())
            for k, v in rec.items():
                out[p + k] = np.concatenate(v, 0)
        path = os.path.join(GOLDEN, name.replace("encoder_", "eval_b1_") + ".npz")
        np.savez_compressed(path, **out)
        print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
