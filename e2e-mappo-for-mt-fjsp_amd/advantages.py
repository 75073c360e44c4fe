"""Rollout -> update hand-off on the device: the value sampling and advantage computation that open
the reference's update, `PPOAlgorithm.global_update_JointActions_GAT_selfCritic` (algorithm/ppo_algorithm.py:563-703).

  * local advantages  = cal_local_job_machine_reward_GAE (ppo:437-489): the four scaled reward channels against the
    actors' own critic heads; v_ of a step is the value at act time of the next step, and for an episode's last step the
    value of the terminal state from the post-terminal forward pair (Run.py:451-475);
  * global advantages = separate_cal_4_reward_GAE (ppo:491-536) on the global critic's values, sampled over the whole
    buffer with no gradient (ppo:628-655, step_for_net_out_Critic_GAT ppo:422-435; m_fea1 of the NEXT stored step — the
    last step reuses its own — for the next-state pass, ppo:640-645);
  * both: delta = r + GAMMA*v_ - v with NO (1-done) factor, gae = delta + GAMMA*LAMDA*gae*(1-done), then
    (adv - adv.mean()) / (adv.std() + 1e-5) over the WHOLE [S, B_total] tensor (torch's unbiased std) — the one place the
    shards of a multi-GPU run exchange data (all-gather over RCCL / xGMI, dist.py); value target = normalised advantage +
    value at act time (ppo:668-671, 689).

Channel order everywhere: mk, pt, tt, it (ppo:441-443).  The reverse scans run as HIP kernels (`mtfjsp_gae`).
"""
import torch

from . import dist as D


def normalise(advs, group=None, eps=1e-5, timed=False):
    """list of UN-normalised [S,B_local] advantages -> list of normalised ones (statistics over all shards' columns)"""
    res = D.all_gather_advantages(advs, group=group, timed=timed)
    full, info = res if timed else (res, None)
    out = []
    for a_full, a_loc in zip(full, advs):
        mean, std = a_full.mean(), a_full.std()
        out.append((a_loc - mean) / (std + eps))
    return (out, info) if timed else out


def local_advantages(env, r4, job_v, job_v_, machine_v, machine_v_, done, gamma, lam, group=None, timed=False):
    """r4 [S,4,B] in the step kernel's order (mk, idle, pt, tt; pe:255-262); job_v* [S,B,2] = (mk, it), machine_v* [S,B,2] =
    (pt, tt); done [S,B].  -> (advantages[4], value_targets[4], raw[4], gather_info) in the order mk, pt, tt, it."""
    pairs = [(r4[:, 0], job_v[..., 0], job_v_[..., 0]), (r4[:, 2], machine_v[..., 0], machine_v_[..., 0]),
             (r4[:, 3], machine_v[..., 1], machine_v_[..., 1]), (r4[:, 1], job_v[..., 1], job_v_[..., 1])]
    raw = [env.gae(r, v, v_, done, gamma, lam) for r, v, v_ in pairs]
    res = normalise(raw, group=group, timed=timed)
    norm, info = res if timed else (res, None)
    targets = [a + p[1] for a, p in zip(norm, pairs)]
    return norm, targets, raw, info


def sample_global_values(enc, traj):
    """multi_v, multi_v_ [S,B,4] of ppo:628-655 from a device TrajectoryBuffer: the global critic on every stored
    pre-decision state and on every post-decision state (with the next stored step's m_fea1)."""
    S, B = traj.total_step, traj.B
    v = torch.empty(S, B, 4, dtype=torch.float32, device=traj.device)
    v_ = torch.empty_like(v)
    for s in range(S):
        enc.global_critic_forward(traj.tasks_fea[s], traj.ell_col[s], traj.ell_val[s], traj.machine_fea1[s],
                                  traj.machine_fea2[s], out=v[s])
        nxt = s if s == S - 1 else s + 1
        enc.global_critic_forward(traj.tasks_fea_[s], traj.ell_col_[s], traj.ell_val_[s], traj.machine_fea1[nxt],
                                  traj.machine_fea2_[s], out=v_[s])
    return v, v_


def global_advantages(env, r4, multi_v, multi_v_, done, gamma, lam, group=None):
    """separate_cal_4_reward_GAE (ppo:491-536): channel i of the global critic against reward channel i (mk, pt, tt, it).
    -> (advantages[4], value_targets[4], raw[4])"""
    order = (0, 2, 3, 1)                                   # mk, pt, tt, it inside r4's (mk, idle, pt, tt)
    raw = [env.gae(r4[:, order[i]], multi_v[..., i], multi_v_[..., i], done, gamma, lam) for i in range(4)]
    norm = normalise(raw, group=group)
    targets = [norm[i] + multi_v[..., i] for i in range(4)]
    return norm, targets, raw


def full_handoff(env, r4, job_v, job_v_, machine_v, machine_v_, multi_v, multi_v_, done, gamma, lam, group=None, timed=False,
                 gather_values=True):
    """The complete rollout -> update hand-off of ppo:628-703 in ONE collective: the four global-critic advantages
    (separate_cal_4_reward_GAE, ppo:491-536) and the four local-critic advantages (cal_local_job_machine_reward_GAE, ppo:437-489)
    are computed per shard (`mtfjsp_gae` reverse scans) and exchanged as one packed buffer — with gather_values the eight value
    tensors their targets are built from ride along, so that every rank ends with what the reference's single process holds:
    16 tensors x [S, B_total] f32 (SURVEY §8e sizes the exchange so: 47 MB per rank at 8 x 4096 J6M6 instances); a data-parallel
    trainer that only consumes its own columns sets gather_values=False (8 tensors).  Normalisation (adv - mean) / (std + 1e-5)
    per tensor over ALL shards' columns (ppo:485,532), value target = normalised advantage + value at act time (ppo:668-671,689).
    Channel order mk, pt, tt, it everywhere.
    -> dict(global_adv[4], global_targets[4], local_adv[4], local_targets[4], raw_global[4], raw_local[4], gather=info,
            full_adv / full_values: the gathered [S,B_total] tensors when more than one rank took part)"""
    order = (0, 2, 3, 1)                                   # mk, pt, tt, it inside r4's (mk, idle, pt, tt)
    raw_g = [env.gae(r4[:, order[i]], multi_v[..., i], multi_v_[..., i], done, gamma, lam) for i in range(4)]
    pairs = [(r4[:, 0], job_v[..., 0], job_v_[..., 0]), (r4[:, 2], machine_v[..., 0], machine_v_[..., 0]),
             (r4[:, 3], machine_v[..., 1], machine_v_[..., 1]), (r4[:, 1], job_v[..., 1], job_v_[..., 1])]
    raw_l = [env.gae(r, v, v_, done, gamma, lam) for r, v, v_ in pairs]
    vals = [multi_v[..., i] for i in range(4)] + [p[1] for p in pairs]
    packed = raw_g + raw_l + (vals if gather_values else [])
    res = D.all_gather_advantages(packed, group=group, timed=timed)
    full, info = res if timed else (res, None)
    norm = []
    for a_full, a_loc in zip(full[:8], raw_g + raw_l):
        mean, std = a_full.mean(), a_full.std()
        norm.append((a_loc - mean) / (std + 1e-5))
    out = dict(global_adv=norm[:4], local_adv=norm[4:], raw_global=raw_g, raw_local=raw_l,
               global_targets=[norm[i] + vals[i] for i in range(4)], local_targets=[norm[4 + i] + vals[4 + i] for i in range(4)],
               gather=info, full_adv=full[:8], full_values=full[8:] if gather_values else None)
    return out
