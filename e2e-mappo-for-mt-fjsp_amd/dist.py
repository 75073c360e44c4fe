"""Multi-GPU: one process per GPU, instances sharded in contiguous blocks, no collective on the env/encoder path.

The only exchange of the rollout->update hand-off is the all-gather of per-shard advantages/returns before the global
advantage normalisation `(adv - mean) / (std + 1e-5)` of ppo:485,532 (RCCL over xGMI on the GPU box — backend "nccl";
gloo in the CPU tests).  GAE itself (ppo:438-536) is a reverse scan over the S stored steps and is per-instance,
so it runs on each shard locally.
"""
import warnings

import torch
import torch.distributed as dist

# Test switch (tests/test_rollout_handoff.py): treat an initialised ONE-rank group as a collective to run, so that the RCCL branches
# execute on a single GPU (a world-size-1 all-gather is a copy, but it goes through the same calls).  A module attribute the tests set
# explicitly — never read from the environment, so nothing that leaks into a production run changes single-rank behaviour.
COLLECT_ON_ONE_RANK = False
_warned_fallback = False


def active(group=None):
    """is there a collective to run?  A process group with more than one rank (or the test switch above)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or COLLECT_ON_ONE_RANK


def device_collectives(group=None):
    """can `group` run collectives on device tensors (RCCL: backend "nccl", also as the cuda half of "cpu:gloo,cuda:nccl" or of a
    group created without a backend name)?  Decided by capability, not by the configured name; gloo-only groups go through host
    copies.  A query that raises is reported once (the host-copy path is correct but slow: it must not be taken silently)."""
    global _warned_fallback
    try:
        if "nccl" in str(dist.get_backend(group)).lower():
            return True
    except Exception as ex:
        if not _warned_fallback:
            _warned_fallback = True
            warnings.warn(f"mtfjsp dist: backend query failed ({ex!r}); collectives go through host copies")
        return False
    try:
        return "nccl" in str((group or dist.group.WORLD)._get_backend(torch.device("cuda")).name()).lower()
    except Exception:
        return False                                      # (a group without a cuda backend: gloo-only — the expected answer, not an error)


def agree_any(flag, group=None):
    """True on every rank iff `flag` is true on at least one (MAX all-reduce of one word; the ranks call it at the same point of
    their step sequence — Rollout.finish_buffer before the hand-off's all-gather)."""
    if not active(group):
        return bool(flag)
    dev = "cuda" if device_collectives(group) else "cpu"
    t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return bool(t.item() > 0.5)


def shard_range(total, rank, world):
    """contiguous block of instances owned by `rank` (rows [B_local*g, B_local*(g+1)))."""
    if total % world:
        raise ValueError("instances must divide evenly over the GPUs")
    per = total // world
    return rank * per, (rank + 1) * per


def gae(rewards, values, next_values, dones, gamma, lam):
    """ppo:444-457 / 500-510: delta = r + gamma*v' - v ; gae_t = delta_t + gamma*lam*gae_{t+1}*(1-done_t).
    All [S,B_local]; returns the UN-normalised advantages [S,B_local]."""
    deltas = rewards + gamma * next_values - values
    adv = torch.empty_like(deltas)
    g = torch.zeros_like(deltas[0])
    for t in range(deltas.shape[0] - 1, -1, -1):
        g = deltas[t] + gamma * lam * g * (1.0 - dones[t])
        adv[t] = g
    return adv


def all_gather_columns(x, group=None):
    """[S,B_local] on every rank -> [S,B_total] (rank-major column blocks) on every rank."""
    if not active(group):
        return x
    world = dist.get_world_size(group)
    xs = x.contiguous()
    out = torch.empty((world * xs.shape[0],) + tuple(xs.shape[1:]), dtype=xs.dtype, device=xs.device)   # dim-0 concat form
    if xs.is_cuda and not device_collectives(group):
        oc = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(oc, xs.cpu(), group=group)
        out.copy_(oc)
    else:
        dist.all_gather_into_tensor(out, xs, group=group)
    out = out.view((world,) + tuple(xs.shape))
    return torch.cat(list(out.unbind(0)), dim=1)


def all_gather_advantages(tensors, group=None, timed=False):
    """One collective for a list of [S,B_local] tensors (4 global + 4 local advantages and their 8 value targets in
    the reference's update): packed into a single [K,S,B_local] buffer so that xGMI sees one large all-gather.
    timed=True: also returns {"world", "bytes_per_rank", "ms"} of the collective (device events around it; ms is None
    when no process group is active)."""
    info = {"world": 1, "bytes_per_rank": 0, "ms": None}
    if not tensors:
        return ([], info) if timed else []
    packed = torch.stack([t.contiguous() for t in tensors], 0)
    info["bytes_per_rank"] = packed.numel() * packed.element_size()
    if not active(group):
        out = list(packed.unbind(0))
        return (out, info) if timed else out
    world = dist.get_world_size(group)
    info["world"] = world
    out = torch.empty((world * packed.shape[0],) + tuple(packed.shape[1:]), dtype=packed.dtype, device=packed.device)
    ev = None
    if timed and packed.is_cuda:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    if packed.is_cuda and not device_collectives(group):          # CPU-only group (gloo in the two-process tests): through host copies
        oc = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(oc, packed.cpu(), group=group)
        out.copy_(oc)
    else:
        dist.all_gather_into_tensor(out, packed, group=group)
    if ev is not None:
        ev[1].record()
        ev[1].synchronize()
        info["ms"] = ev[0].elapsed_time(ev[1])
    out = out.view((world,) + tuple(packed.shape))
    full = torch.cat(list(out.unbind(0)), dim=2)          # [K,S,B_total]
    res = list(full.unbind(0))
    return (res, info) if timed else res


def all_gather_packed(packed, group=None, timed=False):
    """ONE collective for a packed [K,S,B_local] device buffer -> ([world,K,S,B_local] on every rank, info); with no process group
    active the packed buffer itself comes back as [1,K,S,B_local] (no copy).  info = {"world", "rank", "bytes_per_rank", "ms"};
    ms (timed=True) from device events around the collective."""
    info = {"world": 1, "rank": 0, "bytes_per_rank": packed.numel() * packed.element_size(), "ms": None}
    assert packed.is_contiguous()
    if not active(group):
        return packed.unsqueeze(0), info
    world = dist.get_world_size(group)
    info["world"], info["rank"] = world, dist.get_rank(group)
    out = torch.empty((world,) + tuple(packed.shape), dtype=packed.dtype, device=packed.device)
    ev = None
    if timed and packed.is_cuda:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    if packed.is_cuda and not device_collectives(group):          # CPU-only group (gloo in the two-process tests): through host copies
        oc = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(oc.view(-1), packed.cpu().view(-1), group=group)
        out.copy_(oc)
    else:
        dist.all_gather_into_tensor(out.view(-1), packed.view(-1), group=group)
    if ev is not None:
        ev[1].record()
        ev[1].synchronize()
        info["ms"] = ev[0].elapsed_time(ev[1])
    return out, info


def normalize_advantages_global(adv_local, group=None, eps=1e-5):
    """`(adv - adv.mean()) / (adv.std() + 1e-5)` over the GLOBAL [S,B_total] tensor (unbiased std, as torch's default in
    ppo:485), returning this rank's columns."""
    full = all_gather_columns(adv_local, group)
    mean, std = full.mean(), full.std()
    return (adv_local - mean) / (std + eps)


class _DeviceF64:
    """`count` doubles at a raw device address, as something torch.as_tensor(..., device="cuda") can wrap without a copy"""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 3, "strides": None}


def bn_stats_allreduce(group=None):
    """reduction callback for Encoder.set_stats_reduce: the BatchNorm column sums of a layer, summed over the ranks of `group`
    in place (RCCL directly on the device buffer; with a CPU backend such as gloo through a host copy).  SURVEY §8e's
    optional exact big-batch BatchNorm: 7 all-reduces of 2048 doubles per actor-pair forward."""
    import torch
    import torch.distributed as td

    def fn(ptr, count):
        t = torch.as_tensor(_DeviceF64(ptr, count), device="cuda")
        if device_collectives(group):
            td.all_reduce(t, group=group)
        else:
            c = t.cpu()
            td.all_reduce(c, group=group)
            t.copy_(c)
        torch.cuda.synchronize()
    return fn
