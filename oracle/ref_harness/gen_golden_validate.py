#!/usr/bin/env python3
"""Golden fixture for the batched greedy evaluation driver: run the REFERENCE's
`trainer.validate.validate_cost_gcn_jointActor_GAT` (env_batch 1, greedy, shipped top1 checkpoints and seeded random
weights) on the first instances of tests/golden/trace_j6m6e2_eval16_free.npz and record what it returns per instance,
plus the weights used (random case).

ORACLE HARNESS ONLY (build container; imports /root/reference, copies nothing of it).
Usage: python oracle/ref_harness/gen_golden_validate.py
"""
import contextlib
import io
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from bootstrap import bootstrap, default_config  # noqa: E402

REF = bootstrap(models=True)
import torch  # noqa: E402

torch.set_num_threads(1)
with contextlib.redirect_stdout(io.StringIO()):
    from algorithm.ppo_algorithm import PPOAlgorithm  # noqa: E402
    import trainer.validate as V  # noqa: E402

GOLDEN = os.path.join(HERE, "..", "..", "tests", "golden")


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def main():
    g = np.load(os.path.join(GOLDEN, "trace_j6m6e2_eval16_free.npz"))
    J, M, E = [int(x) for x in g["meta"][:3]]
    NB = 12
    data = types.SimpleNamespace(t=g["t"], p=g["p"], transT=g["tt"], edge=g["edge"])
    cfg = default_config(J, M, E, 1)
    out = {"meta": np.array([J, M, E, NB], np.int32)}
    enc = np.load(os.path.join(GOLDEN, "encoder_j6m6e2_rand.npz"))
    for tag in ("top1", "rand"):
        ppo = quiet(PPOAlgorithm, cfg, False)
        if tag == "top1":
            base = os.path.join(REF, "trained_model", "can_use", "No_lr_decay")
            ppo.job_actor.load_state_dict(torch.load(os.path.join(base, "PPO_job_actor_J6M6E2_top1.pth"), map_location="cpu"))
            ppo.machine_actor_gcn.load_state_dict(torch.load(os.path.join(base, "PPO_machine_actor_J6M6E2_top1.pth"), map_location="cpu"))
        else:
            for net, pre in ((ppo.job_actor, "w_ja."), (ppo.machine_actor_gcn, "w_ma.")):
                sd = net.state_dict()
                for k in sd:
                    if pre + k in enc.files:
                        sd[k] = torch.tensor(enc[pre + k])
                net.load_state_dict(sd)
        cum, final, obj, margin, actions, probs = [], [], [], [], [], []
        gaps, acts, prs = [], [], []

        def hook_job(mod, inp, outp):
            pr = outp[3].detach().flatten().sort(descending=True).values
            gaps.append(float(pr[0] - pr[1]) if pr.numel() > 1 else 1.0)
            acts.append(int(outp[0].item())); prs.append(outp[3].detach().flatten().numpy().copy())

        def hook_mch(mod, inp, outp):
            pr = outp[0].detach().flatten().sort(descending=True).values
            gaps.append(float(pr[0] - pr[1]) if pr.numel() > 1 else 1.0)
            acts.append(int(outp[0].argmax(1).item())); prs.append(outp[0].detach().flatten().numpy().copy())
        h1 = ppo.job_actor.register_forward_hook(hook_job)
        h2 = ppo.machine_actor_gcn.register_forward_hook(hook_mch)
        for i in range(NB):
            gaps.clear(); acts.clear(); prs.clear()
            c, f4, o = quiet(V.validate_cost_gcn_jointActor_GAT, ppo, False, data, i, "eval", True, cfg)
            actions.append(np.array(acts, np.int32).reshape(-1, 2)); probs.append(np.stack(prs).reshape(-1, 2, prs[0].size))
            margin.append(min(gaps))                              # smallest top-1 / top-2 probability gap of the episode's 72 decisions
            cum.append([float(c[k]) for k in ("opr_Gt", "opr_mk", "opr_idleT", "opr_pt", "opr_transT")])
            final.append([float(x) for x in f4]); obj.append(float(o))
        h1.remove(); h2.remove()
        out[tag + "_cumsum"] = np.array(cum); out[tag + "_final4"] = np.array(final); out[tag + "_objective"] = np.array(obj)
        out[tag + "_min_margin"] = np.array(margin)
        out[tag + "_actions"] = np.stack(actions)                  # [NB, T, 2] (task, machine) of every greedy decision
        out[tag + "_probs"] = np.stack(probs).astype(np.float32)   # [NB, T, 2, J] job / machine probabilities behind them
        print(tag, "min decision margins:", np.round(margin, 5))
        print(tag, "final4[0] =", final[0], "objective[0] =", obj[0])
    out["cfg_w"] = np.array([cfg["weight_mk"], cfg["weight_ec"], cfg["weight_tt"]], np.float64)
    path = os.path.join(GOLDEN, "validate_j6m6e2_eval12.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
