"""ctypes front-end of oracle/mtfjsp_oracle.c (TEST INFRASTRUCTURE ONLY).

`OracleBatch` mirrors trainer/parallel_env.py::Parallel_env semantics (pe:19-282)
on top of the C restatement; every method cites the reference in the C file.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmtfjsp_oracle.so")
_lib = None

_dp = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_bp = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def build(force=False):
    src = os.path.join(_HERE, "mtfjsp_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "_build/libmtfjsp_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.or_batch_create.restype = C.c_void_p
        L.or_batch_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _ip,
                                      C.c_double, C.c_double, C.c_double, C.c_double, C.c_double]
        L.or_batch_destroy.argtypes = [C.c_void_p]
        L.or_batch_scaler_init.argtypes = [C.c_void_p]
        L.or_batch_scaler_reset_returns.argtypes = [C.c_void_p]
        L.or_batch_reset.argtypes = [C.c_void_p, _dp]
        L.or_batch_step.argtypes = [C.c_void_p, _ip, _ip, _dp, _dp, _ip]
        L.or_batch_observe_ell.argtypes = [C.c_void_p, _ip, _dp, _dp, _dp]
        L.or_batch_observe_dense_adj.argtypes = [C.c_void_p, _dp]
        L.or_batch_mfea1.argtypes = [C.c_void_p, _ip, _bp, _dp, _dp]
        L.or_batch_job_mask_update.argtypes = [C.c_void_p, _ip, _ip, _bp]
        L.or_batch_job_mask_state.argtypes = [C.c_void_p, _ip, _bp]
        L.or_batch_state.argtypes = [C.c_void_p, _ip, _bp, _dp, _dp, _ip, _dp, _dp]
        L.or_batch_valid_action_mask.argtypes = [C.c_void_p, _bp]
        L.or_batch_bench.restype = C.c_long
        L.or_batch_bench.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.or_batch_bench_blocks.restype = C.c_long
        L.or_batch_bench_blocks.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.or_max_threads.restype = C.c_int
        L.or_np_sum.restype = C.c_double
        L.or_np_sum.argtypes = [_dp, C.c_long]
        _lib = L
    return _lib


def max_threads():
    return lib().or_max_threads()


def shop_of_machine(edge):
    """edge [B,E,M/E] (machine ids per shop) -> shop index per machine [B,M] (pe:211)."""
    edge = np.asarray(edge)
    B, E, W = edge.shape
    out = np.zeros((B, E * W), np.int32)
    for b in range(B):
        for e in range(E):
            out[b, edge[b, e]] = e
    return out


class OracleBatch:
    def __init__(self, t, p, tt, edge, left_shift=True, w_cfg=(0.4, 0.4, 0.2), divisor=1.0, gamma=0.99, n_job=None):
        t = np.ascontiguousarray(t, np.float64)
        self.B, self.T, self.M = t.shape
        self.J = n_job if n_job is not None else self.T // self.M
        assert self.J * self.M == self.T
        self.t, self.p = t, np.ascontiguousarray(p, np.float64)
        self.tt = np.ascontiguousarray(tt, np.float64)
        self.shop = shop_of_machine(edge)
        self.L = lib()
        self.h = self.L.or_batch_create(self.B, self.J, self.M, int(left_shift), self.t, self.p, self.tt, self.shop,
                                        w_cfg[0], w_cfg[1], w_cfg[2], divisor, gamma)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.or_batch_destroy(self.h)
            self.h = None

    def scaler_init(self):
        self.L.or_batch_scaler_init(self.h)

    def scaler_reset_returns(self):
        self.L.or_batch_scaler_reset_returns(self.h)

    def reset(self, w3):
        self.L.or_batch_reset(self.h, np.ascontiguousarray(w3, np.float64))
        return self.observe()

    def step(self, task_idx, mach_idx):
        B = self.B
        info = np.zeros((B, 6)); raw = np.zeros((B, 5)); paths = np.zeros(B, np.int32)
        self.L.or_batch_step(self.h, np.ascontiguousarray(task_idx, np.int32), np.ascontiguousarray(mach_idx, np.int32), info, raw, paths)
        return info, raw, paths

    def observe(self, dense=True):
        B, T, M = self.B, self.T, self.M
        col = np.zeros((B, T, 2), np.int32); val = np.zeros((B, T, 2))
        tfea = np.zeros((B * T, 12)); mfea2 = np.zeros((B, M, 8))
        self.L.or_batch_observe_ell(self.h, col, val, tfea, mfea2)
        out = dict(ell_col=col, ell_val=val, tfea=tfea, mfea2=mfea2)
        if dense:
            adj = np.zeros((B, T, T))
            self.L.or_batch_observe_dense_adj(self.h, adj)
            out["adj"] = adj
        return out

    def mfea1(self, task_idx, mmask, tfea):
        out = np.zeros((self.B, self.M, 6))
        self.L.or_batch_mfea1(self.h, np.ascontiguousarray(task_idx, np.int32),
                              np.ascontiguousarray(np.asarray(mmask).reshape(self.B, self.M), np.uint8),
                              np.ascontiguousarray(tfea, np.float64), out)
        return out

    def job_mask_update(self, job_action):
        cand = np.zeros((self.B, self.J), np.int32); mask = np.zeros((self.B, self.J), np.uint8)
        self.L.or_batch_job_mask_update(self.h, np.ascontiguousarray(job_action, np.int32), cand, mask)
        return cand, mask

    def job_mask_state(self):
        cand = np.zeros((self.B, self.J), np.int32); mask = np.zeros((self.B, self.J), np.uint8)
        self.L.or_batch_job_mask_state(self.h, cand, mask)
        return cand, mask

    def state(self):
        B, T, M = self.B, self.T, self.M
        mach = np.zeros((B, T), np.int32); sched = np.zeros((B, T), np.uint8)
        st = np.zeros((B, T)); ft = np.zeros((B, T)); routes = np.zeros((B, M, T), np.int32)
        prev = np.zeros((B, 4)); sc = np.zeros((B, 17))
        self.L.or_batch_state(self.h, mach, sched, st, ft, routes, prev, sc)
        return dict(mach=mach, sched=sched, st=st, ft=ft, routes=routes, prev=prev, scaler=sc)

    def bench(self, episodes, nthreads, w3):
        """CPU baseline loop in C (random valid actions; step + reward scaling + job mask + ELL observation per env and step,
        OpenMP over envs): -> (env_steps, seconds of the step loops, seconds incl. the per-episode resets)"""
        s1, s2 = C.c_double(), C.c_double()
        n = self.L.or_batch_bench(self.h, int(episodes), int(nthreads), np.ascontiguousarray(w3, np.float64), C.byref(s1), C.byref(s2))
        if n < 0:
            raise ValueError("or_batch_bench: n_job > 64")
        return n, s1.value, s2.value

    def bench_blocks(self, episodes, nthreads, w3):
        """the multi-core CPU baseline: every thread owns a contiguous block of envs (thread-local copies, first-touch) for
        whole episodes, no barrier between steps -> (env_steps, wall seconds incl. resets, max per-thread seconds in the step loops)"""
        s1, s2 = C.c_double(), C.c_double()
        n = self.L.or_batch_bench_blocks(self.h, int(episodes), int(nthreads), np.ascontiguousarray(w3, np.float64), C.byref(s1), C.byref(s2))
        if n < 0:
            raise ValueError("or_batch_bench_blocks: n_job > 64")
        return n, s1.value, s2.value

    def valid_action_mask(self):
        m = np.zeros((self.B, self.T), np.uint8)
        self.L.or_batch_valid_action_mask(self.h, m)
        return m
