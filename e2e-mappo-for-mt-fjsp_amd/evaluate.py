"""Batched greedy evaluation — the reference's `validate_cost_gcn_jointActor_GAT` (trainer/validate.py:60-297) for a
whole evaluation set in ONE device rollout.

The reference evaluates instance by instance with env_batch = 1 (Run.py:672-745: 100 serial episodes, ~60 s per
evaluation): fixed reward weights (`reset(Random_weight_type="eval")`, env:1262), no reward scaling, greedy decoding, and
— because its BatchNorms are always in training mode — statistics over the rows of the single instance.  Here the B
instances run side by side: `Encoder.set_bn_mode(True)` gives every instance its own BatchNorm statistics
(`k_gin_inst` / `k_gat_inst`, one workgroup per instance), the raw (unscaled) rewards of `mtfjsp_obs_t.raw` are summed
on the device, and the final costs are read from the `*_previous_step` state exactly as validate.py:277-287 does.
"""
import numpy as np
import torch

from . import capi
from .batch_env import DeviceBatchEnv


def validate_cost_batched(weights, t, p, tt, edge, args, greedy=True, device=0, obs_dtype="f32", actor=None,
                          forced_actions=None, on_step=None, on_action=None):
    """weights: (job_actor_state_dict, machine_actor_state_dict) with the reference's key names (or an `ActorPair` via
    `actor=`); t, p [B,T,M], tt [B,M,M], edge [B,E,M/E]: the evaluation instances; args: the reference's config dict
    (n_job, n_machine, n_edge, weight_mk, weight_ec, weight_tt).
    forced_actions [B,T,2] (task, machine): replay these decisions (teacher forcing) — the forwards still run and
    on_step(s, job_prob, mch_prob) sees the policy's distributions in every visited state; used by the parity tests, because the
    reference's own greedy choice among machines whose scores tie is decided by f32 round-off (validate.py has many
    exactly-uniform machine distributions in its first steps).
    Returns, per instance and in the reference's order (validate.py:297):
      cost_dict_cumsum  dict of [B] arrays: opr_Gt, opr_mk, opr_idleT, opr_pt, opr_transT (sums of the raw step rewards)
      Final_4cost       [B,4]: makespan, mean processing energy, transport time, idle time of the finished schedule
      Objective         [B]:   w_mk*mk + w_ec*(pt + idle) + w_tt*transT
    """
    from . import encoder as enc_mod
    J, M, E = int(args["n_job"]), int(args["n_machine"]), int(args["n_edge"])
    T = J * M
    t = np.asarray(t, np.float64)
    B = t.shape[0]
    scal = args.get("reward_scaling", {}) or {}
    env = DeviceBatchEnv(J, M, E, B, obs_dtype=obs_dtype, device=device,
                         w_cfg=(float(args["weight_mk"]), float(args["weight_ec"]), float(args["weight_tt"])),
                         scaling_divisor=float(scal.get("scaling_divisor", 1.0)))
    env.load_instances(t, np.asarray(p, np.float64), np.asarray(tt, np.float64), edge=np.asarray(edge))
    env.scaler_init()                                               # the scaled components are produced but not used here
    dev = env.device
    if actor is None:
        actor = enc_mod.ActorPair(J, M, B, device=device, obs_dtype=obs_dtype, weights=weights, greedy=greedy, seed=0)
    actor.enc.set_bn_mode(True)
    try:
        w3 = torch.tensor([[args["weight_mk"], args["weight_ec"], args["weight_tt"]]], dtype=torch.float64, device=dev).repeat(B, 1)
        env.reset(w3)                                               # env:1262 Random_weight_type="eval"
        actor.begin_episode()
        task = torch.zeros(B, dtype=torch.int32, device=dev); mach = torch.zeros_like(task); job = torch.zeros_like(task)
        cum = torch.zeros(B, 5, dtype=torch.float64, device=dev)
        fa = None if forced_actions is None else torch.as_tensor(np.asarray(forced_actions), dtype=torch.int32, device=dev)
        for s in range(T):
            if fa is None:
                actor.act(env, s, task, mach, job)
            else:
                # the forwards run on the forced trajectory's states: job_prob is the policy's distribution in this state,
                # mch_prob its machine distribution for the FORCED task
                actor.act(env, s, task, mach, job, force=(fa[:, s, 0].contiguous(), fa[:, s, 1].contiguous()))
                if on_step is not None:
                    on_step(s, actor.enc.job_prob, actor.enc.mch_prob)
            if on_action is not None:
                on_action(s, task, mach)                            # the decisions about to be applied (tests)
            env.step(task, mach)
            cum += env.raw                                          # reward, r_mk, r_idle, r_pt, r_tt (env:1051-1171), unscaled
        torch.cuda.synchronize(dev)
        st = env.status
        if int((st & capi.ST_INVALID).sum().item()) != 0:
            raise RuntimeError("greedy evaluation produced an invalid action")
        assert bool(env.info[:, 1].all().item()), "evaluation episode did not finish"
        prev = env.read_state(capi.STATE_PREV_COSTS)                # mk, e1, transT, idle of the finished schedule (validate.py:277-281)
    finally:
        actor.enc.set_bn_mode(False)
    c = cum.cpu().numpy()
    cost = {"opr_Gt": c[:, 0], "opr_mk": c[:, 1], "opr_idleT": c[:, 2], "opr_pt": c[:, 3], "opr_transT": c[:, 4]}
    final4 = np.stack([prev[:, 0], prev[:, 1] / T, prev[:, 2], prev[:, 3]], 1)
    obj = args["weight_mk"] * final4[:, 0] + args["weight_ec"] * (final4[:, 1] + final4[:, 3]) + args["weight_tt"] * final4[:, 2]
    return cost, final4, obj
