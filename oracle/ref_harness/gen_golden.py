#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

ORACLE HARNESS ONLY (runs in the build container where /root/reference is
mounted; never on the GPU box).  Nothing of the reference is copied: this
script imports it (oracle/ref_harness/bootstrap.py), drives its public API the
way Run.py:190-665 does, and stores inputs + observed outputs as .npz data.

What is pinned:
  A1-A7  Parallel_env.init_DGFJSPEnv_state0 / DGFJSPEnv_paral_step outputs and
         the per-env graph state (machine, start/finish, routes, *_previous_step)
  A8     RewardScaling outputs + state across consecutive episodes
  A9     Parallel_env.cal_cur_task_machine_feature
  A11    PPOAlgorithm.esa_update_chosenTaskID_CandidateTaskIDx_JobMask
  A12    env.valid_action_mask
  E2-E4  job-actor / machine-actor forward (shipped top1 weights + seeded
         random-init weights), training-mode BatchNorm
  N4     Instance_Dataset generator stream (legacy numpy RNG)

Usage:  python oracle/ref_harness/gen_golden.py [--only NAME] [--out DIR]
"""
import argparse
import io
import contextlib
import os
import pickle
import random
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from bootstrap import bootstrap, default_config, AttrDict  # noqa: E402

REF = bootstrap(models=True)
import torch  # noqa: E402

torch.set_num_threads(1)

with contextlib.redirect_stdout(io.StringIO()):
    from trainer.parallel_env import Parallel_env  # noqa: E402
    from algorithm.ppo_algorithm import PPOAlgorithm  # noqa: E402
    from instance.generate_allsize_mofjsp_dataset import Instance_Dataset  # noqa: E402
    from model.gcn_mlp import g_pool_cal  # noqa: E402

ABILITY_SCOPE = {  # values of the reference's instance/config_ins.json (data)
    "t_low": 1, "t_high": 99, "p_low": 1, "p_high": 20,
    "transT_in_low": 1, "transT_in_high": 10, "transT_out_low": 1,
    "transT_out_high": 20, "equal_edge": True, "weight_low": 0.8,
    "weight_high": 1.2, "e1_low": 1, "e1_high": 99,
}


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


# ------------------------------------------------------------------ instances
def ref_generate(samples, n_job, n_machine, n_edge, seed):
    """Run the reference generator, return [t, p, tt, edge] numpy arrays."""
    with tempfile.TemporaryDirectory() as d:
        pkl = os.path.join(d, "ins.pkl")
        quiet(Instance_Dataset, samples=samples, n_job=n_job, n_machine=n_machine,
              n_edge=n_edge, ability_scope=ABILITY_SCOPE, use_PT=0, seed=seed,
              generate_true=1, csv_pth="/nonexistent_dir/x.csv", pkl_pth=pkl)
        with open(pkl, "rb") as f:
            ins = pickle.load(f)
    return [np.asarray(x) for x in ins]


def load_shipped(name):
    with open(os.path.join(REF, "instance", name), "rb") as f:
        ins = pickle.load(f)
    return [np.asarray(x) for x in ins]


# ------------------------------------------------------------------ env traces
def env_state(env, T, M):
    """Snapshot of the integer + float scheduling state of one reference env."""
    mach = np.full(T, -1, np.int32)
    sched = np.zeros(T, np.uint8)
    st = np.full(T, np.nan)
    ft = np.full(T, np.nan)
    dur = np.zeros(T)
    for k in range(1, T + 1):
        nd = env.G.nodes[k]
        mach[k - 1] = nd["machine"]
        dur[k - 1] = nd["duration"]
        if nd["scheduled"]:
            sched[k - 1] = 1
            st[k - 1] = nd["start_time"]
            ft[k - 1] = nd["finish_time"]
    routes = np.full((M, T), -1, np.int16)
    for m in range(M):
        r = np.asarray(env.machine_routes[m]).astype(np.int64)
        routes[m, :len(r)] = r - 1          # 0-based task index
    prev = np.array([env.makespan_previous_step, env.total_e1_previous_step,
                     env.trans_t_previous_step, env.idle_t_previous_step], np.float64)
    return mach, sched, st, ft, dur, routes, prev


def scaler_state(s):
    ms = s.running_ms
    return np.concatenate([np.asarray(s.R, np.float64).ravel(), [float(ms.n)],
                           np.asarray(ms.mean, np.float64).ravel(),
                           np.asarray(ms.S, np.float64).ravel(),
                           np.asarray(ms.std, np.float64).ravel()])  # 4+1+4+4+4 = 17


def run_trace(ins, n_job, n_machine, n_edge, B, episodes, policy, left_shift=True,
              act_seed=0, w_seed=0, keep_every=1):
    """Drive the reference exactly like Run.py does, with a scripted policy."""
    J, M, T = n_job, n_machine, n_job * n_machine
    cfg = default_config(J, M, n_edge, B)
    t_all, p_all, tt_all, edge_all = [np.asarray(x[:B]) for x in ins]
    penv = Parallel_env(cfg)
    batch = {"t": torch.tensor(t_all), "p": torch.tensor(p_all),
             "transT": torch.tensor(tt_all), "edge": torch.tensor(edge_all)}
    try:
        quiet(penv.get_batch, batch)
    except ValueError:      # numpy>=1.24 ragged np.array in the log line; state is set
        pass
    assert len(penv.ability_instance) == B
    penv.init_RewardScaling_sameBATCH(shape=4)
    ppo = quiet(PPOAlgorithm, cfg, False)
    if not left_shift:
        # PDR-harness mode (tester/pdrs.py:669): same env class, flag off.
        import trainer.parallel_env as pe_mod
        orig = pe_mod.DisjunctiveGraphJspEnv_singleStep

        def no_ls(*a, **k):
            k["perform_left_shift_if_possible"] = False
            return orig(*a, **k)
        pe_mod.DisjunctiveGraphJspEnv_singleStep = no_ls
    random.seed(w_seed)
    rs = np.random.RandomState(act_seed)
    feas = t_all >= 0                                   # [B,T,M]
    rec = {k: [] for k in (
        "w3", "adj0", "tfea0", "mfea2_0", "cand0", "mask0", "actions", "job_actions",
        "mfea1", "adj", "tfea", "mfea2", "info", "raw_rewards", "cand", "mask", "mach", "sched",
        "st", "ft", "routes", "prev", "scaler", "vmask", "kept_steps")}
    t_step = 0.0
    n_step = 0
    for ep in range(episodes):
        adj, mfea2, tfea = quiet(penv.init_DGFJSPEnv_state0)
        rec["w3"].append(np.array([e.reward_random_weight for e in penv.paral_env_DG]))
        rec["adj0"].append(adj.astype(np.int16)); assert np.array_equal(adj, adj.astype(np.int16))
        rec["tfea0"].append(tfea.copy()); rec["mfea2_0"].append(mfea2.copy())
        cand = np.array([list(d.values()) for d in ppo.pool_task_dict_batch]) - 1
        mask = ppo.mask_new_batch.bool().numpy().copy()
        rec["cand0"].append(cand.astype(np.int32)); rec["mask0"].append(mask.astype(np.uint8))
        for s in penv.paral_Rscaling_instance:
            s.reset()
        ep_rec = {k: [] for k in ("actions", "job_actions", "mfea1", "adj", "tfea", "mfea2", "info",
                                  "raw_rewards", "cand", "mask", "mach", "sched", "st", "ft", "routes", "prev",
                                  "scaler", "vmask")}
        kept = []
        for step in range(T):
            # ---- scripted policy
            remaining = np.array([[d[j] for j in range(J)] for d in ppo.remaining_m_batch])
            job_a = np.zeros(B, np.int64)
            m_a = np.zeros(B, np.int64)
            for b in range(B):
                if policy == "free":
                    ok = np.flatnonzero(remaining[b] > 0)
                else:
                    ok = np.flatnonzero(~mask[b])
                j = int(ok[rs.randint(len(ok))])
                a = int(cand[b, j])
                fm = np.flatnonzero(feas[b, a])
                m = int(fm[rs.randint(len(fm))])
                if policy == "sticky" and a % M != 0:
                    pm = int(penv.paral_env_DG[b].G.nodes[a]["machine"])   # node id a == task a-1
                    if feas[b, a, pm]:
                        m = pm
                job_a[b], m_a[b] = j, m
            task_idx = cand[np.arange(B), job_a]
            mmask = ~feas[np.arange(B), task_idx][:, None, :]                    # [B,1,M]
            mfea1 = penv.cal_cur_task_machine_feature(torch.tensor(task_idx), torch.tensor(mmask), tfea)
            joint = list(zip(task_idx.tolist(), m_a.tolist()))
            t0 = time.perf_counter()
            # env.step's raw (unscaled) 5 rewards are not returned by Parallel_env; tap them
            raw = []
            for e in penv.paral_env_DG:
                if not hasattr(e, "_orig_get_reward"):
                    e._orig_get_reward = e.get_reward

                    def tap(*a, _e=e, **k):
                        out = _e._orig_get_reward(*a, **k)
                        _e._last_raw = out
                        return out
                    e.get_reward = tap
            adj_, info, mfea2_, tfea_ = quiet(penv.DGFJSPEnv_paral_step, joint)
            t_step += time.perf_counter() - t0
            n_step += B
            raw = np.array([e._last_raw for e in penv.paral_env_DG], np.float64)   # reward,r_mk,r_idle,r_pt,r_tt
            cand_, mask_t = ppo.esa_update_chosenTaskID_CandidateTaskIDx_JobMask(
                paralenv=penv, action_batch=torch.tensor(job_a), mask_value=1)
            mask_ = mask_t.numpy().copy()
            keep = (step % keep_every == 0) or step == T - 1
            ep_rec["actions"].append(np.stack([task_idx, m_a], 1).astype(np.int32))
            ep_rec["job_actions"].append(job_a.astype(np.int32))
            if keep:
                kept.append(step)
                ep_rec["mfea1"].append(mfea1.copy())
                assert np.array_equal(adj_, adj_.astype(np.int16))
                ep_rec["adj"].append(adj_.astype(np.int16))
                ep_rec["tfea"].append(tfea_.copy()); ep_rec["mfea2"].append(mfea2_.copy())
                ep_rec["info"].append(np.array([[float(x) for x in row] for row in info]))
                ep_rec["raw_rewards"].append(raw)
                ep_rec["cand"].append(cand_.astype(np.int32)); ep_rec["mask"].append(mask_.astype(np.uint8))
                sn = [env_state(e, T, M) for e in penv.paral_env_DG]
                for i, k in enumerate(("mach", "sched", "st", "ft", None, "routes", "prev")):
                    if k:
                        ep_rec[k].append(np.stack([x[i] for x in sn]))
                ep_rec["scaler"].append(np.stack([scaler_state(s) for s in penv.paral_Rscaling_instance]))
                vm = []
                for e in penv.paral_env_DG:
                    e.env_transform = "mask"     # suppress the RuntimeError at the terminal state
                    vm.append(np.array(e.valid_action_mask(), np.uint8))
                    e.env_transform = None
                ep_rec["vmask"].append(np.stack(vm))
            adj, mfea2, tfea, cand, mask = adj_, mfea2_, tfea_, cand_, mask_
        assert all(row[1] for row in info)
        for k, v in ep_rec.items():
            rec[k].append(np.stack(v))
        rec["kept_steps"].append(np.array(kept, np.int32))
        # Run.py:653-661
        ppo.set_to_0(penv)
        for e in penv.paral_env_DG:
            quiet(e.reset)
        penv.reset_data()
    if not left_shift:
        pe_mod.DisjunctiveGraphJspEnv_singleStep = orig
    out = {k: np.stack(v) for k, v in rec.items()}
    out.update(t=t_all, p=p_all, tt=tt_all, edge=edge_all.astype(np.int32),
               meta=np.array([J, M, n_edge, B, episodes, int(left_shift), keep_every], np.int32),
               cfg_w=np.array([cfg["weight_mk"], cfg["weight_ec"], cfg["weight_tt"],
                               cfg["reward_scaling"]["scaling_divisor"], cfg["GAMMA"]], np.float64))
    return out, n_step / t_step


# ------------------------------------------------------------------ encoder vectors
def encoder_vectors(ins, weights, B=16, steps=(0, 17), seed=0, size=(6, 6, 2), keep_h_nodes=True):
    """Job-actor / machine-actor forwards of the reference modules on observations
    produced by the reference env (greedy decoding so no sampling RNG is involved).
    The modules are size-generic (gcn_mlp.py:109-197, actor_critic.py:104-296,359-498): `size` = (J, M, E)."""
    J, M, E = size
    T = J * M
    cfg = default_config(J, M, E, B)
    ppo = quiet(PPOAlgorithm, cfg, False)
    if weights == "top1":
        base = os.path.join(REF, "trained_model", "can_use", "No_lr_decay")
        ppo.job_actor.load_state_dict(torch.load(os.path.join(base, "PPO_job_actor_J6M6E2_top1.pth"), map_location="cpu"))
        ppo.machine_actor_gcn.load_state_dict(torch.load(os.path.join(base, "PPO_machine_actor_J6M6E2_top1.pth"), map_location="cpu"))
        ppo.global_critic.load_state_dict(torch.load(os.path.join(base, "PPO_global_critic_J6M6E2_top1.pth"), map_location="cpu"))
    else:
        torch.manual_seed(seed)
        from model.actor_critic import Operation_Actor_JointAction_selfCritic as JA, \
            Machine_Actor_JointAction_selfGAT_selfCritic as MA
        from model.actor_critic import Global_Critic_JointAction_GAT as GC
        ppo.job_actor = quiet(JA, cfg)
        ppo.machine_actor_gcn = quiet(MA, cfg)
        ppo.global_critic = quiet(GC, cfg)
        # make BN affine parameters non-trivial so the test sees gamma/beta
        with torch.no_grad():
            for mod in list(ppo.job_actor.modules()) + list(ppo.machine_actor_gcn.modules()) + list(ppo.global_critic.modules()):
                if isinstance(mod, torch.nn.BatchNorm1d):
                    mod.weight.uniform_(0.5, 1.5)
                    mod.bias.uniform_(-0.5, 0.5)
    t_all, p_all, tt_all, edge_all = [np.asarray(x[:B]) for x in ins]
    penv = Parallel_env(cfg)
    try:
        quiet(penv.get_batch, {"t": torch.tensor(t_all), "p": torch.tensor(p_all),
                               "transT": torch.tensor(tt_all), "edge": torch.tensor(edge_all)})
    except ValueError:
        pass
    penv.init_RewardScaling_sameBATCH(shape=4)
    random.seed(seed)
    adj, mfea2, tfea = quiet(penv.init_DGFJSPEnv_state0)
    cand = np.array([list(d.values()) for d in ppo.pool_task_dict_batch]) - 1
    mask = ppo.mask_new_batch.bool()
    gpool = g_pool_cal("average", B, T, torch.device("cpu"))
    feas = torch.tensor(t_all >= 0)
    h_m = None
    out = {}
    for step in range(max(steps) + 1):
        with torch.no_grad():
            task_index, action_index, log_a, prob, h_o, job_v = ppo.job_actor(
                x_fea=tfea, graph_pool_avg=gpool, padded_nei=None, adj=adj, candidate=cand,
                h_g_m_pooled=h_m, mask_operation=mask, use_greedy=True)
            mm = torch.gather(~feas, 1, task_index.view(-1, 1, 1).expand(B, 1, M))
            mfea1 = penv.cal_cur_task_machine_feature(task_index, mm, tfea)
            h_m_prev = h_m
            mch_prob, h_m, mach_v = ppo.machine_actor_gcn(machine_fea_1=mfea1, machine_fea_2=mfea2,
                                                          h_pooled_o=h_o, machine_mask=mm)
            m_action = mch_prob.argmax(1)
            gv = ppo.global_critic(x_fea=tfea, graph_pool_avg=gpool, adj=adj, candidate=cand, machine_fea1=mfea1, machine_fea2=mfea2)
        if step in steps:
            # node embeddings are not returned by the actor; recompute through the encoder
            with torch.no_grad():
                from model.gcn_mlp import aggr_obs
                a_sp = aggr_obs(torch.from_numpy(adj.copy()).to_sparse(), T)
                _, h_nodes = ppo.job_actor.encoder(x=torch.from_numpy(tfea.copy()).float(), graph_pool=gpool,
                                                   padded_nei=None, adj=a_sp)
            p = f"s{step}_"
            out.update({
                p + "adj": adj.astype(np.int16), p + "tfea": tfea.copy(), p + "cand": cand.astype(np.int32),
                p + "mask": mask.numpy().astype(np.uint8),
                p + "h_m_in": (np.zeros((0,), np.float32) if h_m_prev is None else h_m_prev.numpy()),
                p + "mfea1": mfea1.copy(), p + "mfea2": mfea2.copy(), p + "mmask": mm.numpy().astype(np.uint8),
                p + "task_index": task_index.numpy().astype(np.int32), p + "job_index": action_index.numpy().astype(np.int32),
                p + "job_logp": log_a.numpy(), p + "job_prob": prob.numpy(), p + "h_o": h_o.numpy(),
                p + "job_v": job_v.numpy(), **({p + "h_nodes": h_nodes.numpy()} if keep_h_nodes else {}),
                p + "mch_prob": mch_prob.numpy(), p + "h_m": h_m.numpy(), p + "mach_v": mach_v.numpy(),
                p + "global_v": gv.numpy(),
            })
        joint = list(zip(task_index.tolist(), m_action.tolist()))
        adj, info, mfea2, tfea = quiet(penv.DGFJSPEnv_paral_step, joint)
        cand, mask = ppo.esa_update_chosenTaskID_CandidateTaskIDx_JobMask(paralenv=penv, action_batch=action_index, mask_value=1)
    for name, net in (("ja", ppo.job_actor), ("ma", ppo.machine_actor_gcn), ("gc", ppo.global_critic)):
        for k, v in net.state_dict().items():
            if "num_batches_tracked" in k or "running_" in k:
                continue
            out[f"w_{name}.{k}"] = v.detach().numpy().astype(np.float32)
    out["meta"] = np.array([J, M, E, B], np.int32)
    out["steps"] = np.array(steps, np.int32)
    return out


# ------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(HERE, "..", "..", "tests", "golden"))
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    out_dir = os.path.abspath(args.out)
    os.makedirs(out_dir, exist_ok=True)

    def want(n):
        return args.only is None or args.only == n

    def save(name, d):
        path = os.path.join(out_dir, name + ".npz")
        np.savez_compressed(path, **d)
        print(f"  wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB)")

    speed = {}
    if want("gen"):
        # N4: generator stream — small sample counts so the fixture stays small
        d = {}
        for tag, (s, j, m, e, seed) in {"j6m6e2_s0": (8, 6, 6, 2, 0), "j10m10e2_s5": (3, 10, 10, 2, 5),
                                        "j20m20e4_s7": (2, 20, 20, 4, 7), "j10m6e2_s2": (4, 10, 6, 2, 2)}.items():
            ins = ref_generate(s, j, m, e, seed)
            for k, a in zip(("t", "p", "tt", "edge"), ins):
                d[f"{tag}_{k}"] = a
            d[f"{tag}_args"] = np.array([s, j, m, e, seed], np.int32)
        ev = load_shipped("eval_Instance_J6M6E2.pkl")          # shipped file == generator(seed=1, samples=100)
        ge = ref_generate(100, 6, 6, 2, 1)
        assert all(np.array_equal(a, b) for a, b in zip(ev, ge)), "generator does not reproduce shipped eval set"
        for k, a in zip(("t", "p", "tt", "edge"), ev):
            d[f"eval100_s1_head_{k}"] = a[:3]
            d[f"eval100_s1_tail_{k}"] = a[-2:]
        d["eval100_s1_args"] = np.array([100, 6, 6, 2, 1], np.int32)
        save("instances_generator", d)

    ev = load_shipped("eval_Instance_J6M6E2.pkl")
    if want("c1"):
        # BASELINE.json configs[0]: first 16 of Instance_Dataset(12800, 6,6,2, seed=0), mask-respecting policy
        tr = ref_generate(12800, 6, 6, 2, 0)
        d, sp = run_trace(tr, 6, 6, 2, 16, episodes=3, policy="mask", act_seed=0, w_seed=0)
        speed["c1_J6M6E2_B16"] = sp
        save("trace_j6m6e2_train16_mask", d)
    if want("free"):
        d, sp = run_trace(ev, 6, 6, 2, 16, episodes=2, policy="free", act_seed=1, w_seed=1)
        speed["eval16_free"] = sp
        save("trace_j6m6e2_eval16_free", d)
    if want("sticky"):
        d, _ = run_trace([x[16:] for x in ev], 6, 6, 2, 8, episodes=1, policy="sticky", act_seed=2, w_seed=2)
        save("trace_j6m6e2_eval8_sticky", d)
    if want("nols"):
        d, _ = run_trace([x[32:] for x in ev], 6, 6, 2, 4, episodes=1, policy="free", left_shift=False, act_seed=3, w_seed=3)
        save("trace_j6m6e2_eval4_noleftshift", d)
    if want("j10m6"):
        ins = ref_generate(4, 10, 6, 2, 11)
        d, _ = run_trace(ins, 10, 6, 2, 3, episodes=1, policy="free", act_seed=4, w_seed=4, keep_every=3)
        save("trace_j10m6e2_b3_free", d)
    if want("j10m10"):
        ins = ref_generate(4, 10, 10, 2, 12)
        d, sp = run_trace(ins, 10, 10, 2, 2, episodes=1, policy="free", act_seed=5, w_seed=5, keep_every=5)
        speed["J10M10E2_B2"] = sp
        save("trace_j10m10e2_b2_free", d)
        d, _ = run_trace(ins[:], 10, 10, 2, 2, episodes=1, policy="mask", act_seed=6, w_seed=6, keep_every=5)
        save("trace_j10m10e2_b2_mask", d)
    if want("j20m20"):
        ins = ref_generate(2, 20, 20, 4, 13)
        d, sp = run_trace(ins, 20, 20, 4, 1, episodes=1, policy="free", act_seed=7, w_seed=7, keep_every=20)
        speed["J20M20E4_B1"] = sp
        save("trace_j20m20e4_b1_free", d)
    if want("enc"):
        save("encoder_j6m6e2_top1", encoder_vectors(ev, "top1"))
        save("encoder_j6m6e2_rand", encoder_vectors([x[50:] for x in ev], "rand", B=8, steps=(0, 9), seed=123))
    if want("enc_big"):
        # BASELINE configs 2 and 4 sizes: seeded random weights with perturbed BatchNorm affine, steps 0 and T/2
        save("encoder_j10m10e2_rand", encoder_vectors(ref_generate(4, 10, 10, 2, 14), "rand", B=4, steps=(0, 50), seed=124, size=(10, 10, 2)))
        save("encoder_j20m20e4_rand", encoder_vectors(ref_generate(2, 20, 20, 4, 15), "rand", B=2, steps=(0, 200), seed=125, size=(20, 20, 4)))
    if want("enc_mid"):
        # the partition the headline runs k_gin_res in: more than 256 instances, i.e. 2 instances per workgroup (B = 320: a
        # ragged last workgroup set; B = 512: every workgroup full), whole-batch BatchNorm over B*T rows (gcn:109-197,
        # ac:104-296); one weight set (seed 126), node embeddings not stored
        ins = ref_generate(512, 6, 6, 2, 16)
        a = encoder_vectors(ins, "rand", B=320, steps=(9,), seed=126, keep_h_nodes=False)
        b = encoder_vectors(ins, "rand", B=512, steps=(17,), seed=126, keep_h_nodes=False)
        assert all(np.array_equal(a[k], b[k]) for k in a if k.startswith("w_")), "same seed, same weights"
        d = {k: v for k, v in a.items() if k.startswith("w_")}
        for tag, src in (("b320_", a), ("b512_", b)):
            for k, v in src.items():
                if not k.startswith("w_"):
                    d[tag + k] = v
        save("encoder_j6m6e2_mid", d)
    if speed:
        import platform
        with open(os.path.join(out_dir, "reference_cpu_speed.txt"), "a") as f:
            f.write(f"# {time.strftime('%Y-%m-%d %H:%M:%S')} reference Parallel_env.DGFJSPEnv_paral_step, 1 core, "
                    f"{platform.processor() or platform.machine()}, py{platform.python_version()}, numpy {np.__version__}\n")
            for k, v in speed.items():
                f.write(f"{k}: {v:.1f} env-steps/s\n")
        print(speed)


if __name__ == "__main__":
    main()
