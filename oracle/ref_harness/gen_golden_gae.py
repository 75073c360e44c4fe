#!/usr/bin/env python3
"""Golden fixture for the rollout -> update hand-off: a REAL reference rollout, the loop of
Run.py:229-661 driven with the reference's own objects (Parallel_env, PPOAlgorithm incl. its three networks,
ReplayBuffer), followed by the value sampling and advantage code of the reference's update
(ppo_algorithm.py:628-703: global critic on the buffer; :437-489 cal_local_job_machine_reward_GAE;
:491-536 separate_cal_4_reward_GAE).

Recorded (tests/golden/rollout_gae_j6m6e2_b4.npz):
  instances, weights of the three networks (seeded random init, BatchNorm affine perturbed), the sampled decisions
  (so the device rollout can be teacher-forced), per step job_v / machine_v / log-probs / probabilities, the
  post-terminal extra forward pair of Run.py:455-475 (job_v_, machine_v_ of the terminal step), the reference
  ReplayBuffer's 27-tuple, multi_v / multi_v_ of the global critic, and the 4 local + 4 global normalised advantages
  plus their value targets.

ORACLE HARNESS ONLY (build container; imports /root/reference, copies nothing of it).
Usage: python oracle/ref_harness/gen_golden_gae.py
"""
import contextlib
import copy
import io
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from bootstrap import bootstrap, default_config  # noqa: E402

REF = bootstrap(models=True)
import torch  # noqa: E402

torch.set_num_threads(1)
with contextlib.redirect_stdout(io.StringIO()):
    from trainer.parallel_env import Parallel_env  # noqa: E402
    from trainer.replaybuffer import ReplayBuffer  # noqa: E402
    from algorithm.ppo_algorithm import PPOAlgorithm  # noqa: E402
    from algorithm.agent_func import select_machine_action  # noqa: E402
    from model.gcn_mlp import g_pool_cal  # noqa: E402
    from model.actor_critic import Operation_Actor_JointAction_selfCritic as JA  # noqa: E402
    from model.actor_critic import Machine_Actor_JointAction_selfGAT_selfCritic as MA  # noqa: E402
    from model.actor_critic import Global_Critic_JointAction_GAT as GC  # noqa: E402
from gen_golden import quiet, ref_generate  # noqa: E402

GOLDEN = os.path.join(HERE, "..", "..", "tests", "golden")
NAMES = ["adj", "tasks_fea", "candidate", "mask_operation", "a_operation", "a_logprob_operation",
         "adj_", "tasks_fea_", "candidate_", "mask_operation_", "r_operation", "done_operation",
         "machine_fea2", "a", "a_logprob", "machine_fea2_", "mask_machine_",
         "mk", "pt", "tt", "it", "machine_fea1", "rw", "job_v", "machine_v", "job_v_", "machine_v_"]


def main(J=6, M=6, E=2, B=4, episodes=2, seed=31, out_name="rollout_gae_j6m6e2_b4"):
    T = J * M
    cfg = default_config(J, M, E, B, buffer_size=episodes)
    ins = ref_generate(B, J, M, E, 21)
    t_all, p_all, tt_all, edge_all = [np.asarray(x[:B]) for x in ins]
    ppo = quiet(PPOAlgorithm, cfg, False)
    torch.manual_seed(seed)
    ppo.job_actor, ppo.machine_actor_gcn, ppo.global_critic = quiet(JA, cfg), quiet(MA, cfg), quiet(GC, cfg)
    with torch.no_grad():
        for net in (ppo.job_actor, ppo.machine_actor_gcn, ppo.global_critic):
            for mod in net.modules():
                if isinstance(mod, torch.nn.BatchNorm1d):
                    mod.weight.uniform_(0.5, 1.5)
                    mod.bias.uniform_(-0.5, 0.5)
    penv = Parallel_env(cfg)
    instance_bs_dict = {"t": torch.tensor(t_all), "p": torch.tensor(p_all), "transT": torch.tensor(tt_all), "edge": torch.tensor(edge_all)}
    try:
        quiet(penv.get_batch, instance_bs_dict)
    except ValueError:
        pass
    penv.init_RewardScaling_sameBATCH(shape=4)
    rb = ReplayBuffer({"n_job": J, "n_machine": M, "buffer_size": episodes, "env_batch": B, "gcn_input_dim": 12})
    gpool = g_pool_cal("average", B, T, torch.device("cpu"))
    random.seed(seed)
    rec = {k: [] for k in ("w3", "task", "mach", "job", "job_prob", "mch_prob", "h_o", "h_m", "term_job_v_", "term_machine_v_")}
    for ep in range(episodes):
        # Run.py:229-265
        adj_batch, mfea2_batch, tfea_batch = quiet(penv.init_DGFJSPEnv_state0)
        rec["w3"].append(np.array([e.reward_random_weight for e in penv.paral_env_DG]))
        candidate_batch = np.array([list(d.values()) for d in ppo.pool_task_dict_batch]) - 1
        mask_operation_batch = ppo.mask_new_batch.bool()
        mask_machine_batch0 = ~torch.tensor(instance_bs_dict["t"].numpy() >= 0)
        h_mch_pooled = None
        for s in penv.paral_Rscaling_instance:                                  # Run.py:283-284
            s.reset()
        step_flag_for_v_ = 0
        while True:
            with torch.no_grad():                                               # Run.py:316-376
                task_index, action_index, log_a, prob, h_g_o_pooled, job_v = ppo.job_actor(
                    x_fea=tfea_batch, graph_pool_avg=gpool, padded_nei=None, adj=adj_batch, candidate=candidate_batch,
                    h_g_m_pooled=h_mch_pooled, mask_operation=mask_operation_batch, use_greedy=False)
                mask_machine_batch_ = torch.gather(mask_machine_batch0, 1, task_index.unsqueeze(-1).unsqueeze(-1).expand(
                    mask_machine_batch0.size(0), -1, mask_machine_batch0.size(2)))
                mfea1_batch = penv.cal_cur_task_machine_feature(task_index=task_index, m_mask=mask_machine_batch_, all_task_fea=tfea_batch)
                mch_prob, h_mch_pooled, machine_v = ppo.machine_actor_gcn(
                    machine_fea_1=mfea1_batch, machine_fea_2=mfea2_batch, h_pooled_o=h_g_o_pooled, machine_mask=mask_machine_batch_)
                m_action, m_action_logprob = select_machine_action(mch_prob)
            for k, v in (("task", task_index), ("mach", m_action), ("job", action_index), ("job_prob", prob), ("mch_prob", mch_prob),
                         ("h_o", h_g_o_pooled), ("h_m", h_mch_pooled)):
                rec[k].append(v.numpy().copy())
            joint_actions = [x for x in zip(task_index.tolist(), m_action.tolist())]                    # Run.py:411-412
            adj_batch_, oenv_step_info, mfea2_batch_, tfea_batch_ = quiet(penv.DGFJSPEnv_paral_step, joint_actions)
            candidate_batch_, mask_operation_batch_ = ppo.esa_update_chosenTaskID_CandidateTaskIDx_JobMask(
                paralenv=penv, action_batch=action_index, mask_value=cfg["mask_value"])                   # Run.py:427
            o_r = [copy.deepcopy(info[0]) for info in oenv_step_info]
            mk = [copy.deepcopy(info[2]) for info in oenv_step_info]
            it = [copy.deepcopy(info[3]) for info in oenv_step_info]
            pt = [copy.deepcopy(info[4]) for info in oenv_step_info]
            tt = [copy.deepcopy(info[5]) for info in oenv_step_info]
            check_done = [info[1] for info in oenv_step_info]
            step_flag_for_v_ += 1                                                                        # Run.py:451-475
            if step_flag_for_v_ > 1:
                rb.store_v_next(j_v_=job_v, m_v_=machine_v)
            if all(check_done):
                with torch.no_grad():
                    _, _, _, _, h_g_o_pooled_, job_v_ = ppo.job_actor(
                        x_fea=tfea_batch_, graph_pool_avg=gpool, padded_nei=None, adj=adj_batch_, candidate=candidate_batch_,
                        h_g_m_pooled=h_mch_pooled, mask_operation=mask_operation_batch, use_greedy=False)
                    _, _, machine_v_ = ppo.machine_actor_gcn(
                        machine_fea_1=mfea1_batch, machine_fea_2=mfea2_batch_, h_pooled_o=h_g_o_pooled_, machine_mask=mask_machine_batch_)
                    rb.store_v_next(j_v_=job_v_, m_v_=machine_v_)
                    rec["term_job_v_"].append(job_v_.numpy().copy()); rec["term_machine_v_"].append(machine_v_.numpy().copy())
                    step_flag_for_v_ = 0
            rw = [copy.deepcopy(penv.paral_env_DG[i].reward_random_weight) for i in range(B)]          # Run.py:478
            rb.store_operation(adj=adj_batch, fea=tfea_batch, candidate=candidate_batch, mask=mask_operation_batch,
                               a_o=action_index, a_o_logprob=log_a, r=o_r, adj_=adj_batch_, fea_=tfea_batch_,
                               candidate_=candidate_batch_, mask_=mask_operation_batch_, mch_fea1=mfea1_batch,
                               mch_fea2=mfea2_batch, mch_fea2_=mfea2_batch_, a_m=m_action, a_m_logprob=m_action_logprob,
                               dw=None, done=check_done, mask_machine_=mask_machine_batch_, mk=mk, pt=pt, tt=tt, it=it,
                               rw=rw, j_v=job_v, m_v=machine_v)                                          # Run.py:489-514
            adj_batch, tfea_batch, candidate_batch = adj_batch_, tfea_batch_, candidate_batch_          # Run.py:585-589
            mask_operation_batch, mfea2_batch = mask_operation_batch_, mfea2_batch_
            if all(check_done):                                                                          # Run.py:615-661
                ppo.set_to_0(None)
                for e in penv.paral_env_DG:
                    quiet(e.reset)
                penv.reset_data()
                break
    assert rb.count_operation == episodes * T == rb.count_operation_
    out = rb.numpy_to_tensor_operation()
    d = {"meta": np.array([J, M, E, B, episodes], np.int32), "gamma_lambda": np.array([cfg["GAMMA"], cfg["LAMDA"]]),
         "t": t_all, "p": p_all, "tt": tt_all, "edge": edge_all.astype(np.int32)}
    tup = dict(zip(NAMES, out))
    for n, v in tup.items():
        a = v.cpu().numpy()
        if n in ("adj", "adj_"):
            assert np.array_equal(a, a.astype(np.int16))
            a = a.astype(np.int16)
        d["out_" + n] = a
    for k, v in rec.items():
        d[k] = np.stack(v)
    # ---- value sampling + advantages of the update (ppo:628-703), on the reference's own functions
    with torch.no_grad():
        multi_v = ppo.step_for_net_out_Critic_GAT(net_model=ppo.global_critic, task_fea=tup["tasks_fea"], graph_pool_avg=gpool,
                                                  adj=tup["adj"], candidate=tup["candidate"], machine_fea1=tup["machine_fea1"],
                                                  machine_fea2=tup["machine_fea2"])
        mf1_ = copy.deepcopy(tup["machine_fea1"])                                   # ppo:640-645
        for i in range(mf1_.shape[0]):
            mf1_[i] = tup["machine_fea1"][i] if i == mf1_.shape[0] - 1 else tup["machine_fea1"][i + 1]
        multi_v_ = ppo.step_for_net_out_Critic_GAT(net_model=ppo.global_critic, task_fea=tup["tasks_fea_"], graph_pool_avg=gpool,
                                                   adj=tup["adj_"], candidate=tup["candidate_"], machine_fea1=mf1_,
                                                   machine_fea2=tup["machine_fea2_"])
        local_adv = ppo.cal_local_job_machine_reward_GAE(mk_r=tup["mk"], pt_r=tup["pt"], tt_r=tup["tt"], it_r=tup["it"],
                                                         jv=tup["job_v"], jv_=tup["job_v_"], mv=tup["machine_v"],
                                                         mv_=tup["machine_v_"], done_operation=tup["done_operation"])
        global_adv = ppo.separate_cal_4_reward_GAE(mk_r=tup["mk"], pt_r=tup["pt"], tt_r=tup["tt"], it_r=tup["it"],
                                                   v=multi_v, v_=multi_v_, done_operation=tup["done_operation"])
    d["multi_v"], d["multi_v_"] = multi_v.numpy(), multi_v_.numpy()
    d["local_adv"] = np.stack([a.numpy() for a in local_adv])          # order mk, pt, tt, it (ppo:441-443)
    d["global_adv"] = np.stack([a.numpy() for a in global_adv])
    jv, mv = tup["job_v"], tup["machine_v"]
    d["local_target"] = np.stack([(local_adv[0] + jv[:, :, 0]).numpy(), (local_adv[1] + mv[:, :, 0]).numpy(),
                                  (local_adv[2] + mv[:, :, 1]).numpy(), (local_adv[3] + jv[:, :, 1]).numpy()])   # ppo:668-671
    d["global_target"] = np.stack([(global_adv[i] + multi_v[:, :, i]).numpy() for i in range(4)])               # ppo:689
    for name, net in (("ja", ppo.job_actor), ("ma", ppo.machine_actor_gcn), ("gc", ppo.global_critic)):
        for k, v in net.state_dict().items():
            if "num_batches_tracked" in k or "running_" in k:
                continue
            d[f"w_{name}.{k}"] = v.detach().numpy().astype(np.float32)
    path = os.path.join(GOLDEN, out_name + ".npz")
    np.savez_compressed(path, **d)
    print("wrote", path, os.path.getsize(path), "bytes; terminal v_ vs next-episode v differ by",
          float(np.abs(d["term_job_v_"][0] - d["out_job_v"][T]).max()))


if __name__ == "__main__":
    main()
