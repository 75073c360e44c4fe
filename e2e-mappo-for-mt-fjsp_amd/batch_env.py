"""DeviceBatchEnv — device-resident batch of MT-FJSP instances over the C ABI.

torch is used ONLY as plumbing (device buffers + stream handle); every computation
is a HIP kernel in libmtfjsp.so.  Method names follow the reference's
trainer/parallel_env.py so the parity tests read like the reference's call sites.
"""
import ctypes as C

import numpy as np
import torch

from . import capi


def shop_of_machine(edge):
    """reference `edge` table [B,E,M/E] (machine ids per shop, generate…py:219-233) -> [B,M] shop index."""
    edge = np.asarray(edge)
    B, E, W = edge.shape
    out = np.zeros((B, E * W), np.int32)
    for e in range(E):
        np.put_along_axis(out, edge[:, e, :].astype(np.int64), e, axis=1)
    return out


class DeviceBatchEnv:
    def __init__(self, n_job, n_machine, n_edge, batch, left_shift=True, obs_dtype="f64", device=0,
                 gamma=0.99, w_cfg=(0.4, 0.4, 0.2), scaling_divisor=1.0):
        if not torch.cuda.is_available():
            raise RuntimeError("DeviceBatchEnv needs a GPU (MI355X); there is no CPU fallback")
        self.L = capi.lib()
        self.J, self.M, self.E, self.B = int(n_job), int(n_machine), int(n_edge), int(batch)
        self.T = self.J * self.M
        self.obs_f32 = obs_dtype in ("f32", torch.float32, np.float32)
        self.device = torch.device("cuda", device)
        cfg = capi.Config(self.J, self.M, self.E, self.B, int(bool(left_shift)), capi.OBS_F32 if self.obs_f32 else capi.OBS_F64,
                          device, 0, gamma, w_cfg[0], w_cfg[1], w_cfg[2], scaling_divisor)
        h = C.c_void_p()
        capi.check(self.L.mtfjsp_create(C.byref(cfg), C.byref(h)))
        self.h = h
        odt = torch.float32 if self.obs_f32 else torch.float64
        B, T, M, J = self.B, self.T, self.M, self.J
        dev = self.device
        # observation buffers are torch tensors so the rollout / PPO update can consume them zero-copy
        self.tasks_fea = torch.zeros(B * T, 12, dtype=odt, device=dev)
        self.ell_col = torch.full((B * T, 2), -1, dtype=torch.int32, device=dev)
        self.ell_val = torch.zeros(B * T, 2, dtype=torch.float32, device=dev)
        self.m_fea2 = torch.zeros(B, M, 8, dtype=odt, device=dev)
        self.info = torch.zeros(B, 6, dtype=torch.float64, device=dev)
        self.raw = torch.zeros(B, 5, dtype=torch.float64, device=dev)
        self.candidate = torch.zeros(B, J, dtype=torch.int32, device=dev)
        self.job_mask = torch.zeros(B, J, dtype=torch.uint8, device=dev)
        self.status = torch.zeros(B, dtype=torch.int32, device=dev)
        self.m_fea1 = torch.zeros(B, M, 6, dtype=odt, device=dev)
        self.mmask = torch.zeros(B, M, dtype=torch.uint8, device=dev)
        obs = capi.Obs(*[x.data_ptr() for x in (self.tasks_fea, self.ell_col, self.ell_val, self.m_fea2, self.info,
                                                  self.raw, self.candidate, self.job_mask, self.status)])
        capi.check(self.L.mtfjsp_bind_obs(self.h, C.byref(obs)), self.h)
        self._sp_buf = None
        self.use_current_stream()

    def use_current_stream(self):
        s = torch.cuda.current_stream(self.device).cuda_stream
        capi.check(self.L.mtfjsp_set_stream(self.h, C.c_void_p(s)), self.h)

    def close(self):
        if getattr(self, "h", None):
            self.L.mtfjsp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- instances (= get_batch, pe:39-66)
    def load_instances(self, t, p, tt, edge=None, shop=None):
        t = np.ascontiguousarray(t, np.float64); p = np.ascontiguousarray(p, np.float64)
        tt = np.ascontiguousarray(tt, np.float64)
        assert t.shape == (self.B, self.T, self.M) and p.shape == t.shape and tt.shape == (self.B, self.M, self.M)
        shop = shop_of_machine(edge) if shop is None else np.ascontiguousarray(shop, np.int32)
        self._host_t = t
        capi.check(self.L.mtfjsp_load_instances_host(self.h, t.ctypes.data, p.ctypes.data, tt.ctypes.data, shop.ctypes.data), self.h)

    def load_instances_device(self, t, p, tt, shop):
        for x in (t, p, tt):
            assert x.is_cuda and x.dtype == torch.float64 and x.is_contiguous()
        assert shop.dtype == torch.int32
        capi.check(self.L.mtfjsp_load_instances(self.h, t.data_ptr(), p.data_ptr(), tt.data_ptr(), shop.data_ptr()), self.h)

    def generate_instances(self, seed, first_instance=0, scope=None):
        """draw this batch's instances ON the device: the distribution of the reference's Instance_Dataset
        (instances.DEFAULT_SCOPE = instance/config_ins.json), Philox keyed by (seed, first_instance + b); nothing is uploaded"""
        from .instances import DEFAULT_SCOPE
        sc = dict(DEFAULT_SCOPE)
        if scope:
            sc.update(scope)
        s9 = np.array([sc["t_low"], sc["t_high"], sc["p_low"], sc["p_high"], sc["weight_low"], sc["weight_high"],
                       sc["transT_in_low"], sc["transT_in_high"], sc["transT_out_high"]], np.float64)
        capi.check(self.L.mtfjsp_generate_instances(self.h, int(seed), int(first_instance), s9.ctypes.data), self.h)

    def read_instances(self):
        """-> (t, p [B,T,M], tt [B,M,M], edge [B,E,M/E]) host arrays of the loaded / generated instances"""
        B, T, M, E = self.B, self.T, self.M, self.E
        t = np.zeros((B, T, M)); p = np.zeros((B, T, M)); tt = np.zeros((B, M, M)); shop = np.zeros((B, M), np.int32)
        capi.check(self.L.mtfjsp_read_instances_host(self.h, t.ctypes.data, p.ctypes.data, tt.ctypes.data, shop.ctypes.data), self.h)
        edge = np.stack([np.stack([np.flatnonzero(shop[b] == e) for e in range(E)]) for b in range(B)]).astype(np.int64)
        return t, p, tt, edge

    def scaler_init(self):
        capi.check(self.L.mtfjsp_scaler_init(self.h), self.h)

    def scaler_reset_returns(self):
        capi.check(self.L.mtfjsp_scaler_reset_returns(self.h), self.h)

    def scaler_reset_returns_masked(self, mask):
        m = np.ascontiguousarray(mask, np.uint8)
        assert m.shape == (self.B,)
        capi.check(self.L.mtfjsp_scaler_reset_returns_masked_host(self.h, m.ctypes.data), self.h)

    # ---- reset / step
    def reset(self, w3):
        if torch.is_tensor(w3):
            assert w3.is_cuda and w3.dtype == torch.float64 and w3.is_contiguous()
            self._w3 = w3
            capi.check(self.L.mtfjsp_reset(self.h, w3.data_ptr()), self.h)
        else:
            w3 = np.ascontiguousarray(w3, np.float64)
            assert w3.shape == (self.B, 3)
            capi.check(self.L.mtfjsp_reset_host(self.h, w3.ctypes.data), self.h)

    def reset_episode(self, seed, episode, out=None, reset_returns=True):
        """scaler_reset_returns() + draw_reward_weights(seed, episode) + reset(those weights) in ONE launch -> the weights [B,3] f64"""
        if out is None:
            out = torch.empty(self.B, 3, dtype=torch.float64, device=self.device)
        assert out.is_cuda and out.dtype == torch.float64 and out.is_contiguous()
        self._w3 = out
        capi.check(self.L.mtfjsp_reset_episode(self.h, int(seed), int(episode), out.data_ptr(), 1 if reset_returns else 0), self.h)
        return out

    def draw_reward_weights(self, seed, episode, out=None):
        """reward weights of one episode drawn on the device (env:1253-1259 type "01", Philox keyed by (seed, episode, b)) -> [B,3] f64"""
        if out is None:
            out = torch.empty(self.B, 3, dtype=torch.float64, device=self.device)
        assert out.is_cuda and out.dtype == torch.float64 and out.is_contiguous()
        capi.check(self.L.mtfjsp_draw_reward_weights(self.h, int(seed), int(episode), out.data_ptr()), self.h)
        return out

    def step(self, task_idx, mach_idx):
        """device tensors (int32) -> asynchronous launch; numpy/list -> host variant (raises on invalid actions)."""
        if torch.is_tensor(task_idx):
            assert task_idx.is_cuda and task_idx.dtype == torch.int32 and mach_idx.dtype == torch.int32
            capi.check(self.L.mtfjsp_step(self.h, task_idx.data_ptr(), mach_idx.data_ptr()), self.h)
        else:
            a = np.ascontiguousarray(task_idx, np.int32); m = np.ascontiguousarray(mach_idx, np.int32)
            assert a.shape == (self.B,) and m.shape == (self.B,)
            capi.check(self.L.mtfjsp_step_host(self.h, a.ctypes.data, m.ctypes.data), self.h)

    def step_record(self, task_idx, mach_idx, r4_out, done_out):
        """device step that also writes this step's f32 trajectory entries: r4_out [4,B], done_out [B] (contiguous views)"""
        assert r4_out.is_contiguous() and done_out.is_contiguous() and r4_out.dtype == torch.float32
        capi.check(self.L.mtfjsp_step_record(self.h, task_idx.data_ptr(), mach_idx.data_ptr(), r4_out.data_ptr(), done_out.data_ptr()), self.h)

    def step_params(self, task_idx, mach_idx, r4_out=None, done_out=None):
        """the parameter block of exactly the step that step() / step_record() with these arguments would launch, for
        Encoder.arm_env_step (the step then rides in the machine actor's heads launch) — or None when this environment's step
        cannot (shape, kernel-time recording, diagnostic overrides): then call step() / step_record() as usual"""
        if self._sp_buf is None:
            self._sp_buf = C.create_string_buffer(int(self.L.mtfjsp_step_params_bytes()))
        rc = self.L.mtfjsp_step_params(self.h, task_idx.data_ptr(), mach_idx.data_ptr(), r4_out.data_ptr() if r4_out is not None else None,
                                       done_out.data_ptr() if done_out is not None else None, self._sp_buf, len(self._sp_buf))
        if rc < 0:
            capi.check(rc, self.h)
        return self._sp_buf if rc == 1 else None

    def gae(self, r, v, v_next, done, gamma, lam, out=None):
        """un-normalised GAE advantages [S,B] for one reward channel (views with arbitrary (s,b) strides allowed for r/v/v_next)"""
        S = done.shape[0]
        if out is None:
            out = torch.empty(S, self.B, dtype=torch.float32, device=self.device)
        assert done.is_contiguous() and out.is_contiguous()
        capi.check(self.L.mtfjsp_gae(self.h, S, r.data_ptr(), r.stride(0), r.stride(1), v.data_ptr(), v.stride(0), v.stride(1),
                                     v_next.data_ptr(), v_next.stride(0), v_next.stride(1), done.data_ptr(), gamma, lam, out.data_ptr()), self.h)
        return out

    @staticmethod
    def _view_table(views):
        """K f32 [S,B] views (any element strides) -> (pointer array, stride_s array, stride_b array) for the C ABI"""
        K = len(views)
        ptrs = (C.c_void_p * K)(*[(v.data_ptr() if v is not None else None) for v in views])
        ss = (C.c_int64 * K)(*[(v.stride(0) if v is not None else 0) for v in views])
        sb = (C.c_int64 * K)(*[(v.stride(1) if v is not None else 0) for v in views])
        return ptrs, ss, sb

    def pack_views(self, views, out):
        """K strided f32 [S,B] views -> out[k] (out: contiguous [K,S,B] or a contiguous slice of a larger packed buffer); one launch"""
        K, S = len(views), views[0].shape[0]
        assert out.is_contiguous() and tuple(out.shape) == (K, S, self.B) and all(v.dtype == torch.float32 for v in views)
        ptrs, ss, sb = self._view_table(views)
        capi.check(self.L.mtfjsp_pack_views(self.h, K, S, ptrs, ss, sb, out.data_ptr()), self.h)
        return out

    def normalize_advantages(self, gathered, K, world, rank, values, norm_out, targets_out=None, full_out=None, eps=1e-5):
        """(adv - mean) / (std + eps) of ppo:485,532 for the first K tensors of gathered [world, K_total, S, B] (statistics over all
        shards' columns), value targets = normalised advantage + values[k] ([S,B] views), optional reference layout
        full_out [K_total, S, world*B]; two launches, no torch kernels (include/mtfjsp.h)"""
        Kt, S = gathered.shape[-3], gathered.shape[-2]
        assert gathered.is_contiguous() and gathered.dtype == torch.float32 and gathered.numel() == world * Kt * S * self.B
        assert norm_out.is_contiguous() and (targets_out is None or targets_out.is_contiguous()) and (full_out is None or full_out.is_contiguous())
        ptrs = ss = sb = None
        if targets_out is not None:
            ptrs, ss, sb = self._view_table(list(values) + [None] * (K - len(values)))
        capi.check(self.L.mtfjsp_normalize_advantages(self.h, K, Kt, world, rank, S, gathered.data_ptr(), eps, ptrs, ss, sb, norm_out.data_ptr(),
                                                      targets_out.data_ptr() if targets_out is not None else None,
                                                      full_out.data_ptr() if full_out is not None else None), self.h)

    def observe_mfea1(self, task_idx, mmask=None):
        if not torch.is_tensor(task_idx):
            task_idx = torch.as_tensor(np.ascontiguousarray(task_idx, np.int32), device=self.device)
        mm = 0
        if mmask is not None:
            if not torch.is_tensor(mmask):
                mmask = torch.as_tensor(np.ascontiguousarray(np.asarray(mmask).reshape(self.B, self.M), np.uint8), device=self.device)
            mm = mmask.data_ptr()
        self._keep = (task_idx, mmask)
        capi.check(self.L.mtfjsp_observe_mfea1(self.h, task_idx.data_ptr(), mm, self.m_fea1.data_ptr(), self.mmask.data_ptr()), self.h)
        return self.m_fea1

    def mfea1_context(self):
        """device-pointer recipe of observe_mfea1 (-> self.m_fea1, self.mmask) for the encoder's heads kernel to execute right
        after it has selected the task (capi.Mfea1Ctx; valid while the instances stay loaded)"""
        ctx = capi.Mfea1Ctx()
        capi.check(self.L.mtfjsp_get_mfea1_context(self.h, self.m_fea1.data_ptr(), self.mmask.data_ptr(), C.byref(ctx)), self.h)
        return ctx

    def random_actions(self, seed, counter, task_idx, mach_idx, job_idx=None):
        capi.check(self.L.mtfjsp_random_actions(self.h, seed, counter, task_idx.data_ptr(), mach_idx.data_ptr(),
                                                job_idx.data_ptr() if job_idx is not None else 0), self.h)

    # ---- exports
    def dense_adj(self):
        out = torch.empty(self.B, self.T, self.T, dtype=torch.float64, device=self.device)
        capi.check(self.L.mtfjsp_export_dense_adj(self.h, out.data_ptr()), self.h)
        return out

    def valid_action_mask(self):
        out = torch.empty(self.B, self.T, dtype=torch.uint8, device=self.device)
        capi.check(self.L.mtfjsp_valid_action_mask(self.h, out.data_ptr()), self.h)
        return out

    def set_scaler_state(self, first, state17):
        """restore the RewardScaling state ([n,17], MTFJSP_STATE_SCALER layout) of instances first..first+n-1"""
        a = np.ascontiguousarray(state17, np.float64).reshape(-1, 17)
        capi.check(self.L.mtfjsp_set_scaler_state_host(self.h, int(first), a.shape[0], a.ctypes.data), self.h)

    def read_state(self, which):
        B, T, M = self.B, self.T, self.M
        shape, dt = {capi.STATE_MACHINE: ((B, T), np.int32), capi.STATE_START: ((B, T), np.float64),
                     capi.STATE_FINISH: ((B, T), np.float64), capi.STATE_ROUTES: ((B, M, T), np.int32),
                     capi.STATE_PREV_COSTS: ((B, 4), np.float64), capi.STATE_SCALER: ((B, 17), np.float64),
                     capi.STATE_W3: ((B, 3), np.float64)}[which]
        out = np.zeros(shape, dt)
        capi.check(self.L.mtfjsp_read_state_host(self.h, which, out.ctypes.data), self.h)
        return out

    def synchronize(self):
        capi.check(self.L.mtfjsp_synchronize(self.h), self.h)

    def footprint_copy(self, read_bytes, write_bytes, access_bytes=16, grid=2048, reps=50):
        """measurement only: (average, minimum) microseconds per launch of a plain streaming kernel with this footprint"""
        a, m = C.c_double(), C.c_double()
        capi.check(self.L.mtfjsp_footprint_copy(self.h, int(read_bytes), int(write_bytes), int(access_bytes), int(grid), int(reps),
                                                C.byref(a), C.byref(m)), self.h)
        return a.value, m.value

    def timing_begin(self):
        capi.check(self.L.mtfjsp_timing_begin(self.h), self.h)

    def timing_end(self):
        ms = C.c_double(); n = C.c_int64()
        capi.check(self.L.mtfjsp_timing_end(self.h, C.byref(ms), C.byref(n)), self.h)
        return ms.value, n.value
