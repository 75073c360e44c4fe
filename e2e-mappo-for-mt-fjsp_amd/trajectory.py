"""Device-resident trajectory buffer with the interface of the reference's `ReplayBuffer`
(trainer/replaybuffer.py:18-204).

The reference keeps every transition in host numpy arrays and stores the adjacency DENSE:
`adj`/`adj_` are `[steps, B, T, T]` float64 — 7.6 GB each at B = 4096 J6M6, 2.6 TB at the largest config.
Here every field lives on the device in the layout the kernels produce it in, and the adjacency stays in
ELL form (`<= 2` in-edges per node + implicit self loop, `mtfjsp_obs_t.ell_col/ell_val`): 2 x 8 bytes per
node instead of 8 T bytes.  Field names, `store_operation()` / `store_v_next()` argument order and the
27-tuple order of `numpy_to_tensor_operation()` (replaybuffer.py:141-204) are the reference's; the two
adjacency entries of the tuple are `EllAdjacency` objects whose `.dense(steps)` materialises the
reference's `[n, B, T, T]` float tensor for a minibatch.

Two ways to fill it:
  * `store_operation(...)` — the reference's call (replaybuffer.py:82-139), any mix of numpy / torch inputs,
    dense adjacencies accepted (converted to ELL);
  * `slot()` / `snapshot(env, 'pre')` / `after_step(env, next_pre=...)` — the path of the device rollout: the kernels write
    actions, log-probabilities, critic values, rewards, m_fea1 and the machine mask straight into the slot views, and ONE
    launch per step (`mtfjsp_snapshot_obs2`) copies the environment's post-step observation into the slot's s' fields, into the
    NEXT slot's s fields (the same observation inside an episode) and the scalar reward into `r_operation`.
    (`after_decision()` is the copying fallback for decisions whose m_fea1 comes from `observe_mfea1`, i.e. forced actions.)
"""
import ctypes as C

import numpy as np
import torch

from . import capi


def dense_to_ell(adj):
    """dense adj_wrk [B,T,T] (row = destination node, diagonal = self loop) -> (col [B,T,2] int32, val [B,T,2] float32).
    Slot order = ascending source index (the order is irrelevant to every consumer)."""
    a = torch.as_tensor(np.asarray(adj) if not torch.is_tensor(adj) else adj)
    B, T, _ = a.shape
    off = a.clone()
    idx = torch.arange(T, device=a.device)
    off[:, idx, idx] = 0
    nz = off != 0
    if int(nz.sum(-1).max()) > 2:
        raise ValueError("adjacency has a node with more than 2 in-edges: not a disjunctive-graph adj_wrk")
    # positions of the (up to) two non-zeros of every row: first and last
    pos = torch.arange(T, device=a.device).expand(B, T, T)
    big = torch.where(nz, pos, torch.full_like(pos, T))
    first = big.min(-1).values
    small = torch.where(nz, pos, torch.full_like(pos, -1))
    last = small.max(-1).values
    cnt = nz.sum(-1)
    col = torch.full((B, T, 2), -1, dtype=torch.int32, device=a.device)
    val = torch.zeros((B, T, 2), dtype=torch.float32, device=a.device)
    has0 = cnt >= 1
    has1 = cnt >= 2
    col[..., 0] = torch.where(has0, first, torch.full_like(first, -1)).to(torch.int32)
    col[..., 1] = torch.where(has1, last, torch.full_like(last, -1)).to(torch.int32)
    g0 = torch.gather(off, 2, first.clamp(max=T - 1).unsqueeze(-1)).squeeze(-1)
    g1 = torch.gather(off, 2, last.clamp(min=0).unsqueeze(-1)).squeeze(-1)
    val[..., 0] = torch.where(has0, g0, torch.zeros_like(g0)).to(torch.float32)
    val[..., 1] = torch.where(has1, g1, torch.zeros_like(g1)).to(torch.float32)
    return col, val


class EllAdjacency:
    """Stand-in for the reference's dense `[S, B, T, T]` adjacency: `col`/`val` are `[S, B*T, 2]`."""

    def __init__(self, col, val, B, T):
        self.col, self.val, self.B, self.T = col, val, B, T

    @property
    def shape(self):
        return (self.col.shape[0], self.B, self.T, self.T)

    def dense(self, steps=None, dtype=torch.float32):
        """-> [n, B, T, T] as `torch.tensor(self.adj, dtype=torch.float)` would hold it (replaybuffer.py:142)."""
        col = self.col if steps is None else self.col[steps]
        val = self.val if steps is None else self.val[steps]
        if col.dim() == 2:
            col, val = col.unsqueeze(0), val.unsqueeze(0)
        n, B, T = col.shape[0], self.B, self.T
        out = torch.zeros(n, B, T, T, dtype=dtype, device=col.device)
        idx = torch.arange(T, device=col.device)
        out[:, :, idx, idx] = 1
        c = col.reshape(n, B, T, 2).long()
        v = val.reshape(n, B, T, 2).to(dtype)
        for k in range(2):
            ok = c[..., k] >= 0
            out.scatter_(3, c[..., k].clamp(min=0).unsqueeze(-1), torch.where(ok, v[..., k], out.gather(
                3, c[..., k].clamp(min=0).unsqueeze(-1)).squeeze(-1)).unsqueeze(-1))
        return out


class TrajectoryBuffer:
    """`ReplayBuffer(args)` of the reference (replaybuffer.py:18-80) on the device.  `args` needs n_job, n_machine,
    buffer_size, env_batch, gcn_input_dim (12)."""

    def __init__(self, args, device="cuda", obs_dtype=torch.float64, alias_v_next=False):
        """alias_v_next (the device rollout's layout): the local critic values live in T+1 slots per episode — slot t < T is
        the value at act time of step t, slot T the value of the terminal state written by the post-terminal forward pair
        (run:455-475, `terminal_slot()`); job_v / machine_v are slots 0..T-1 and job_v_ / machine_v_ slots 1..T of every
        episode (v_ of step t = v of step t+1, run:451-454), exactly what the reference's store_v_next sequence produces.
        False: separate arrays filled through the reference's store_operation()/store_v_next() calls."""
        self.J, self.M = int(args["n_job"]), int(args["n_machine"])
        self.total_task = self.J * self.M
        self.total_step = int(args["buffer_size"]) * self.total_task
        self.B = int(args["env_batch"])
        self.F = int(args.get("gcn_input_dim", 12))
        self.device = torch.device(device)
        self.obs_dtype = obs_dtype
        S, B, T, J, M, F = self.total_step, self.B, self.total_task, self.J, self.M, self.F
        z = lambda *shape, dtype=torch.float32: torch.zeros(*shape, dtype=dtype, device=self.device)
        # state before the decision / after it (replaybuffer.py:32-45); adjacency in ELL
        self.ell_col, self.ell_val = z(S, B * T, 2, dtype=torch.int32), z(S, B * T, 2)
        self.tasks_fea = z(S, B * T, F, dtype=obs_dtype)
        self.candidate = z(S, B, J, dtype=torch.int32)
        self.mask_operation = z(S, B, J, dtype=torch.bool)
        self.ell_col_, self.ell_val_ = z(S, B * T, 2, dtype=torch.int32), z(S, B * T, 2)
        self.tasks_fea_ = z(S, B * T, F, dtype=obs_dtype)
        self.candidate_ = z(S, B, J, dtype=torch.int32)
        self.mask_operation_ = z(S, B, J, dtype=torch.bool)
        self.mask_machine_ = z(S, B, 1, M, dtype=torch.bool)
        self.a_operation = z(S, B, dtype=torch.int32)         # cast to long on the way out (what the kernels write is int32)
        self.a_logprob_operation = z(S, B)
        self.r_operation = z(S, B)
        self.r4 = z(S, 4, B)                                       # scaled components in the kernels' order: mk, idle, pt, tt (pe:255-262)
        self.mk, self.it, self.pt, self.tt = self.r4[:, 0], self.r4[:, 1], self.r4[:, 2], self.r4[:, 3]
        self.done_operation = z(S, B)
        self.machine_fea1 = z(S, B, M, 6, dtype=obs_dtype)
        self.machine_fea2 = z(S, B, M, 8, dtype=obs_dtype)
        self.machine_fea2_ = z(S, B, M, 8, dtype=obs_dtype)
        self.a = z(S, B, dtype=torch.int32)
        self.a_logprob = z(S, B)
        self.random_weight = z(S, B, 3)
        self.alias_v_next = alias_v_next
        if alias_v_next:
            E = int(args["buffer_size"])
            self._jv, self._mv = z(E, T + 1, B, 2), z(E, T + 1, B, 2)
        else:
            self.job_v, self.machine_v = z(S, B, 2), z(S, B, 2)
            self.job_v_, self.machine_v_ = z(S, B, 2), z(S, B, 2)
        self.count_operation = 0
        self.count_operation_ = 0

    # ------------------------------------------------------------------ helpers
    def _t(self, x, dtype=None):
        t = x if torch.is_tensor(x) else torch.as_tensor(np.asarray(x))
        return t.to(self.device, dtype) if dtype is not None else t.to(self.device)

    def _put_adj(self, col_buf, val_buf, k, adj):
        if isinstance(adj, (tuple, list)) and len(adj) == 2:
            col, val = adj
        else:
            col, val = dense_to_ell(adj)
        col_buf[k].copy_(self._t(col, torch.int32).reshape(-1, 2))
        val_buf[k].copy_(self._t(val, torch.float32).reshape(-1, 2))

    # ------------------------------------------------------------------ the reference's interface
    def store_operation(self, adj, fea, candidate, mask, a_o, a_o_logprob, r,
                        adj_, fea_, candidate_, mask_,
                        mch_fea1, mch_fea2, mch_fea2_, a_m, a_m_logprob, dw, done, mask_machine_,
                        mk, pt, tt, it, rw, j_v, m_v):
        """= ReplayBuffer.store_operation (replaybuffer.py:82-139), same argument order; `adj`/`adj_` may be dense
        [B,T,T] or an ELL pair (col, val)."""
        k = self.count_operation
        od = self.obs_dtype
        self._put_adj(self.ell_col, self.ell_val, k, adj)
        self.tasks_fea[k].copy_(self._t(fea, od).reshape(-1, self.F))
        self.candidate[k].copy_(self._t(candidate, torch.int32))
        self.mask_operation[k].copy_(self._t(mask, torch.bool).reshape(self.B, self.J))
        self.a_operation[k].copy_(self._t(a_o, torch.int32).reshape(self.B))
        self.a_logprob_operation[k].copy_(self._t(a_o_logprob, torch.float32).reshape(self.B))
        self.r_operation[k].copy_(self._t(r, torch.float32).reshape(self.B))
        for buf, x in ((self.mk, mk), (self.pt, pt), (self.tt, tt), (self.it, it)):
            buf[k].copy_(self._t(x, torch.float32).reshape(self.B))
        self._put_adj(self.ell_col_, self.ell_val_, k, adj_)
        self.tasks_fea_[k].copy_(self._t(fea_, od).reshape(-1, self.F))
        self.candidate_[k].copy_(self._t(candidate_, torch.int32))
        self.mask_operation_[k].copy_(self._t(mask_, torch.bool).reshape(self.B, self.J))
        self.mask_machine_[k].copy_(self._t(mask_machine_, torch.bool).reshape(self.B, 1, self.M))
        self.done_operation[k].copy_(self._t(done, torch.float32).reshape(self.B))
        self.machine_fea1[k].copy_(self._t(mch_fea1, od).reshape(self.B, self.M, 6))
        self.machine_fea2[k].copy_(self._t(mch_fea2, od).reshape(self.B, self.M, 8))
        self.a[k].copy_(self._t(a_m, torch.int32).reshape(self.B))
        self.a_logprob[k].copy_(self._t(a_m_logprob, torch.float32).reshape(self.B))
        self.machine_fea2_[k].copy_(self._t(mch_fea2_, od).reshape(self.B, self.M, 8))
        self.random_weight[k].copy_(self._t(rw, torch.float32).reshape(self.B, 3))
        if self.alias_v_next:                                  # slot t of episode e (the same mapping slot() uses)
            e, t = divmod(k, self.total_task)
            self._jv[e, t].copy_(self._t(j_v, torch.float32).reshape(self.B, 2))
            self._mv[e, t].copy_(self._t(m_v, torch.float32).reshape(self.B, 2))
        else:
            self.job_v[k].copy_(self._t(j_v, torch.float32).reshape(self.B, 2))
            self.machine_v[k].copy_(self._t(m_v, torch.float32).reshape(self.B, 2))
        self.count_operation += 1

    def store_v_next(self, j_v_, m_v_):
        """= ReplayBuffer.store_v_next (replaybuffer.py:131-139)"""
        assert not self.alias_v_next, "alias_v_next: v_ of a step is the shifted view of v, nothing to store"
        k = self.count_operation_
        self.job_v_[k].copy_(self._t(j_v_, torch.float32).reshape(self.B, 2))
        self.machine_v_[k].copy_(self._t(m_v_, torch.float32).reshape(self.B, 2))
        self.count_operation_ += 1

    def numpy_to_tensor_operation(self):
        """= ReplayBuffer.numpy_to_tensor_operation (replaybuffer.py:141-204): the same 27 entries in the same order and
        dtypes (float / long / bool as the reference casts them); entries 0 and 6 are `EllAdjacency`.  No copy is made
        except for the dtype casts the reference performs."""
        f = torch.float32
        adj = EllAdjacency(self.ell_col, self.ell_val, self.B, self.total_task)
        adj_ = EllAdjacency(self.ell_col_, self.ell_val_, self.B, self.total_task)
        return (adj, self.tasks_fea.to(f), self.candidate.long(), self.mask_operation, self.a_operation.long(),
                self.a_logprob_operation,
                adj_, self.tasks_fea_.to(f), self.candidate_.to(f), self.mask_operation_, self.r_operation,
                self.done_operation,
                self.machine_fea2.to(f), self.a.long(), self.a_logprob, self.machine_fea2_.to(f), self.mask_machine_,
                self.mk, self.pt, self.tt, self.it, self.machine_fea1.to(f), self.random_weight) + self.local_values()

    def local_values(self):
        """-> (job_v, machine_v, job_v_, machine_v_), each [S,B,2] (entries 23-26 of the tuple)"""
        if not self.alias_v_next:
            return (self.job_v, self.machine_v, self.job_v_, self.machine_v_)
        S, B, T = self.total_step, self.B, self.total_task
        return (self._jv[:, :T].reshape(S, B, 2), self._mv[:, :T].reshape(S, B, 2),
                self._jv[:, 1:].reshape(S, B, 2), self._mv[:, 1:].reshape(S, B, 2))

    def reset(self):
        """Run.py resets the counters after every update (the arrays are overwritten, never cleared)."""
        self.count_operation = 0
        self.count_operation_ = 0

    @property
    def full(self):
        return self.count_operation == self.total_step

    def nbytes(self):
        return sum(v.numel() * v.element_size() for v in vars(self).values() if torch.is_tensor(v) and v._base is None)

    # ------------------------------------------------------------------ zero-copy path of the device rollout
    def _check_env(self, env):
        """the snapshot kernel copies byte counts derived from the ENVIRONMENT's shape and observation dtype: a mismatching
        buffer would be overrun"""
        if (env.B, env.J, env.M) != (self.B, self.J, self.M):
            raise ValueError(f"trajectory buffer is for B={self.B} J={self.J} M={self.M}, the environment has B={env.B} J={env.J} M={env.M}")
        if env.obs_f32 != (self.obs_dtype == torch.float32):
            raise ValueError("trajectory buffer and environment disagree on the observation dtype (f32 vs f64)")
        if self.count_operation >= self.total_step:
            raise IndexError("trajectory buffer is full: call reset() after the update")

    def _obs_ptrs(self, k, which):
        if which == "pre":
            dst = (self.tasks_fea[k], self.ell_col[k], self.ell_val[k], self.machine_fea2[k], self.candidate[k],
                   self.mask_operation[k])
        else:
            dst = (self.tasks_fea_[k], self.ell_col_[k], self.ell_val_[k], self.machine_fea2_[k], self.candidate_[k],
                   self.mask_operation_[k])
        tf, ec, ev, mf, cand, mask = [d.data_ptr() for d in dst]
        return capi.Obs(tf, ec, ev, mf, 0, 0, cand, mask, 0)

    def snapshot(self, env, which):
        """copy the environment's CURRENT observation into slot `count_operation`: which='pre' (state the decision is
        taken in: adj, fea, candidate, mask, mch_fea2) or 'post' (adj_, fea_, candidate_, mask_, mch_fea2_).
        One kernel launch (mtfjsp_snapshot_obs)."""
        self._check_env(env)
        obs = self._obs_ptrs(self.count_operation, which)
        capi.check(env.L.mtfjsp_snapshot_obs(env.h, C.byref(obs)), env.h)

    def slot(self):
        """views of the current slot the rollout kernels write straight into (no copies): job index / log-prob, machine
        index / log-prob (int32 / f32 [B]), critic values [B,2] x 2, scaled reward components [4,B], done [B], m_fea1 [B,M,6]
        (observation dtype) and the machine mask [B,1,M] (as bytes)"""
        k = self.count_operation
        if self.alias_v_next:
            e, t = divmod(k, self.total_task)
            jv, mv = self._jv[e, t], self._mv[e, t]
        else:
            jv, mv = self.job_v[k], self.machine_v[k]
        return dict(job_idx=self.a_operation[k], job_logp=self.a_logprob_operation[k], mach_idx=self.a[k],
                    mach_logp=self.a_logprob[k], job_v=jv, mach_v=mv, r4=self.r4[k], done=self.done_operation[k],
                    m_fea1=self.machine_fea1[k], mmask=self.mask_machine_[k].view(torch.uint8))

    def terminal_slot(self):
        """(job_v_, machine_v_) [B,2] views of the CURRENT slot's episode that receive the value of the terminal state from
        the post-terminal forward pair (run:455-475); alias_v_next layout only"""
        assert self.alias_v_next
        e = self.count_operation // self.total_task
        return self._jv[e, self.total_task], self._mv[e, self.total_task]

    def begin_episode(self, w3):
        """the episode's reward weights are the same for all of its T slots (Run.py:477-478)"""
        k = self.count_operation
        self.random_weight[k:k + self.total_task].copy_(w3.to(torch.float32).unsqueeze(0).expand(self.total_task, -1, -1))

    def after_decision(self, env):
        """m_fea1 and the machine mask of the chosen task (pe:152-214), produced between the two actor forwards"""
        self._check_env(env)
        k = self.count_operation
        self.machine_fea1[k].copy_(env.m_fea1.reshape(self.B, self.M, 6))
        self.mask_machine_[k].copy_(env.mmask.reshape(self.B, 1, self.M))

    def after_step(self, env, next_pre=False):
        """scalar reward (info[:,0], pe:255-262) and the post-decision observation, in ONE launch; advances the slot.
        next_pre: the same launch also fills the NEXT slot's pre-decision fields (inside an episode s of step k+1 is s' of
        step k), so the next step needs no snapshot(env, 'pre')."""
        self._check_env(env)
        k = self.count_operation
        post = self._obs_ptrs(k, "post")
        nxt = self._obs_ptrs(k + 1, "pre") if next_pre and k + 1 < self.total_step else None
        capi.check(env.L.mtfjsp_snapshot_obs2(env.h, C.byref(post), C.byref(nxt) if nxt is not None else None,
                                              self.r_operation[k].data_ptr()), env.h)
        self.count_operation += 1
        self.count_operation_ = self.count_operation
        return nxt is not None
