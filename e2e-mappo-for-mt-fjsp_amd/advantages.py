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


def _pairs(r4, job_v, job_v_, machine_v, machine_v_):
    """(reward, value, next value) views per local-critic channel in the order mk, pt, tt, it (r4 is mk, idle, pt, tt: pe:255-262)"""
    return [(r4[:, 0], job_v[..., 0], job_v_[..., 0]), (r4[:, 2], machine_v[..., 0], machine_v_[..., 0]),
            (r4[:, 3], machine_v[..., 1], machine_v_[..., 1]), (r4[:, 1], job_v[..., 1], job_v_[..., 1])]


def _device_handoff(env, packed, K, values, group, timed, want_full):
    """packed [K_total,S,B] (first K: raw advantages) -> all-gather (one collective) -> HIP normalisation (mtfjsp_normalize_advantages):
    (norm [K,S,B], targets [K,S,B], full [K_total,S,B_total] or None, gather info).  No torch kernel runs on this path."""
    Kt, S, B = packed.shape
    G, info = D.all_gather_packed(packed, group=group, timed=timed)
    world, rank = info["world"], info["rank"]
    norm = torch.empty(K, S, B, dtype=torch.float32, device=packed.device)
    targets = torch.empty_like(norm)
    full = torch.empty(Kt, S, world * B, dtype=torch.float32, device=packed.device) if (want_full and world > 1) else None
    env.normalize_advantages(G, K, world, rank, values, norm, targets, full)
    return norm, targets, full, info


def local_advantages(env, r4, job_v, job_v_, machine_v, machine_v_, done, gamma, lam, group=None, timed=False):
    """r4 [S,4,B] in the step kernel's order (mk, idle, pt, tt; pe:255-262); job_v* [S,B,2] = (mk, it), machine_v* [S,B,2] =
    (pt, tt); done [S,B].  -> (advantages[4], value_targets[4], raw[4], gather_info) in the order mk, pt, tt, it.
    The four GAE scans write into ONE packed buffer, which is what the all-gather sends and the normalisation kernel reads."""
    pairs = _pairs(r4, job_v, job_v_, machine_v, machine_v_)
    S, B = done.shape
    packed = torch.empty(4, S, B, dtype=torch.float32, device=done.device)
    for k, (r, v, v_) in enumerate(pairs):
        env.gae(r, v, v_, done, gamma, lam, out=packed[k])
    norm, targets, _, info = _device_handoff(env, packed, 4, [p[1] for p in pairs], group, timed, False)
    return list(norm.unbind(0)), list(targets.unbind(0)), list(packed.unbind(0)), (info if timed else None)


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
    are computed per shard (`mtfjsp_gae` reverse scans, straight into the packed buffer) and exchanged as one packed buffer — with
    gather_values the eight value tensors their targets are built from ride along (`mtfjsp_pack_views`), so that every rank ends with
    what the reference's single process holds: 16 tensors x [S, B_total] f32 (SURVEY §8e sizes the exchange so: 47 MB per rank at
    8 x 4096 J6M6 instances); a data-parallel trainer that only consumes its own columns sets gather_values=False (8 tensors).
    Normalisation (adv - mean) / (std + 1e-5) per tensor over ALL shards' columns (ppo:485,532), value target = normalised advantage
    + value at act time (ppo:668-671,689): `mtfjsp_normalize_advantages`.  Channel order mk, pt, tt, it everywhere.
    -> dict(global_adv[4], global_targets[4], local_adv[4], local_targets[4], raw_global[4], raw_local[4], gather=info,
            full_adv / full_values: the gathered [S,B_total] tensors (views of this rank's own when only one rank took part))"""
    order = (0, 2, 3, 1)                                   # mk, pt, tt, it inside r4's (mk, idle, pt, tt)
    pairs = _pairs(r4, job_v, job_v_, machine_v, machine_v_)
    S, B = done.shape
    Kt = 16 if gather_values else 8
    packed = torch.empty(Kt, S, B, dtype=torch.float32, device=done.device)
    for i in range(4):
        env.gae(r4[:, order[i]], multi_v[..., i], multi_v_[..., i], done, gamma, lam, out=packed[i])
    for k, (r, v, v_) in enumerate(pairs):
        env.gae(r, v, v_, done, gamma, lam, out=packed[4 + k])
    vals = [multi_v[..., i] for i in range(4)] + [p[1] for p in pairs]
    if gather_values:
        env.pack_views(vals, packed[8:])
    norm, targets, full, info = _device_handoff(env, packed, 8, vals, group, timed, True)
    src = full if full is not None else packed              # one rank: the gathered tensors ARE this rank's
    out = dict(global_adv=list(norm[:4].unbind(0)), local_adv=list(norm[4:].unbind(0)),
               raw_global=list(packed[:4].unbind(0)), raw_local=list(packed[4:8].unbind(0)),
               global_targets=list(targets[:4].unbind(0)), local_targets=list(targets[4:].unbind(0)),
               gather=(info if timed else None), full_adv=list(src[:8].unbind(0)), full_values=list(src[8:].unbind(0)) if gather_values else None)
    return out
