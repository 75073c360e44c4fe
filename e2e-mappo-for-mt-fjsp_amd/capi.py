"""ctypes binding of include/mtfjsp.h (thin: no logic, only prototypes).

Fails loudly when libmtfjsp.so is missing — there is no CPU fallback for the product path.
"""
import ctypes as C
import os

from . import _build

_LIB = None

OBS_F64, OBS_F32 = 0, 1
OK, ERR_ARG, ERR_STATE, ERR_HIP, ERR_ACTION, ERR_RETRY = 0, -1, -2, -3, -4, -5
PATH_MASK, ST_INVALID, ST_INFEASIBLE = 0x7, 0x100, 0x200
STATE_MACHINE, STATE_START, STATE_FINISH, STATE_ROUTES, STATE_PREV_COSTS, STATE_SCALER, STATE_W3 = range(7)


class Config(C.Structure):
    _fields_ = [("n_job", C.c_int32), ("n_machine", C.c_int32), ("n_edge", C.c_int32), ("batch", C.c_int32),
                ("left_shift", C.c_int32), ("obs_dtype", C.c_int32), ("device_id", C.c_int32), ("reserved", C.c_int32),
                ("gamma", C.c_double), ("w_mk", C.c_double), ("w_ec", C.c_double), ("w_tt", C.c_double),
                ("scaling_divisor", C.c_double)]


class Obs(C.Structure):
    _fields_ = [("tasks_fea", C.c_void_p), ("ell_col", C.c_void_p), ("ell_val", C.c_void_p), ("m_fea2", C.c_void_p),
                ("info", C.c_void_p), ("raw", C.c_void_p), ("candidate", C.c_void_p), ("job_mask", C.c_void_p),
                ("status", C.c_void_p)]


class Mfea1Ctx(C.Structure):
    _fields_ = [("t", C.c_void_p), ("p", C.c_void_p), ("tt", C.c_void_p), ("mean3", C.c_void_p), ("shop", C.c_void_p),
                ("link", C.c_void_p), ("m_fea1_out", C.c_void_p), ("mmask_out", C.c_void_p),
                ("T", C.c_int32), ("M", C.c_int32), ("obs_f32", C.c_int32), ("m_fea2", C.c_void_p)]


class EncoderConfig(C.Structure):
    _fields_ = [("n_job", C.c_int32), ("n_machine", C.c_int32), ("batch", C.c_int32), ("hidden", C.c_int32),
                ("obs_dtype", C.c_int32), ("device_id", C.c_int32)]


_VP, _I, _U64, _SZ = C.c_void_p, C.c_int, C.c_uint64, C.c_size_t
PROTOTYPES = {
    # name: (restype, argtypes)
    "mtfjsp_create": (_I, [C.POINTER(Config), C.POINTER(_VP)]),
    "mtfjsp_destroy": (_I, [_VP]),
    "mtfjsp_last_error": (C.c_char_p, [_VP]),
    "mtfjsp_set_stream": (_I, [_VP, _VP]),
    "mtfjsp_synchronize": (_I, [_VP]),
    "mtfjsp_alloc_obs": (_I, [_VP, C.POINTER(Obs)]),
    "mtfjsp_bind_obs": (_I, [_VP, C.POINTER(Obs)]),
    "mtfjsp_snapshot_obs": (_I, [_VP, C.POINTER(Obs)]),
    "mtfjsp_snapshot_obs2": (_I, [_VP, C.POINTER(Obs), C.POINTER(Obs), _VP]),
    "mtfjsp_generate_instances": (_I, [_VP, _U64, _U64, _VP]),
    "mtfjsp_read_instances_host": (_I, [_VP, _VP, _VP, _VP, _VP]),
    "mtfjsp_load_instances": (_I, [_VP, _VP, _VP, _VP, _VP]),
    "mtfjsp_load_instances_host": (_I, [_VP, _VP, _VP, _VP, _VP]),
    "mtfjsp_scaler_init": (_I, [_VP]),
    "mtfjsp_scaler_reset_returns": (_I, [_VP]),
    "mtfjsp_scaler_reset_returns_masked_host": (_I, [_VP, _VP]),
    "mtfjsp_reset": (_I, [_VP, _VP]),
    "mtfjsp_draw_reward_weights": (_I, [_VP, _U64, _U64, _VP]),
    "mtfjsp_reset_episode": (_I, [_VP, _U64, _U64, _VP, C.c_int32]),
    "mtfjsp_reset_host": (_I, [_VP, _VP]),
    "mtfjsp_step": (_I, [_VP, _VP, _VP]),
    "mtfjsp_step_host": (_I, [_VP, _VP, _VP]),
    "mtfjsp_step_record": (_I, [_VP, _VP, _VP, _VP, _VP]),
    "mtfjsp_step_params_bytes": (C.c_int32, []),
    "mtfjsp_step_params": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, C.c_int32]),
    "mtfjsp_gae": (_I, [_VP, C.c_int32, _VP, C.c_int64, C.c_int64, _VP, C.c_int64, C.c_int64, _VP, C.c_int64, C.c_int64, _VP, C.c_float, C.c_float, _VP]),
    "mtfjsp_normalize_advantages": (_I, [_VP, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _VP, C.c_float, _VP, _VP, _VP, _VP, _VP, _VP]),
    "mtfjsp_pack_views": (_I, [_VP, C.c_int32, C.c_int32, _VP, _VP, _VP, _VP]),
    "mtfjsp_observe_mfea1": (_I, [_VP, _VP, _VP, _VP, _VP]),
    "mtfjsp_random_actions": (_I, [_VP, _U64, _U64, _VP, _VP, _VP]),
    "mtfjsp_export_dense_adj": (_I, [_VP, _VP]),
    "mtfjsp_export_dense_adj_host": (_I, [_VP, _VP]),
    "mtfjsp_valid_action_mask": (_I, [_VP, _VP]),
    "mtfjsp_read_state_host": (_I, [_VP, _I, _VP]),
    "mtfjsp_set_scaler_state_host": (_I, [_VP, _I, _I, _VP]),
    "mtfjsp_copy_to_host": (_I, [_VP, _VP, _VP, _SZ]),
    "mtfjsp_footprint_copy": (_I, [_VP, _SZ, _SZ, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "mtfjsp_timing_begin": (_I, [_VP]),
    "mtfjsp_timing_end": (_I, [_VP, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "mtfjsp_encoder_create": (_I, [C.POINTER(EncoderConfig), C.POINTER(_VP)]),
    "mtfjsp_encoder_destroy": (_I, [_VP]),
    "mtfjsp_encoder_last_error": (C.c_char_p, [_VP]),
    "mtfjsp_encoder_set_stream": (_I, [_VP, _VP]),
    "mtfjsp_encoder_load_weight_host": (_I, [_VP, C.c_char_p, _VP, C.c_int64]),
    "mtfjsp_encoder_weights_ready": (_I, [_VP]),
    "mtfjsp_job_actor_forward": (_I, [_VP] * 11),
    "mtfjsp_machine_actor_forward": (_I, [_VP] * 8),
    "mtfjsp_global_critic_forward": (_I, [_VP] * 7),
    "mtfjsp_sample_categorical": (_I, [_VP, _VP, C.c_int32, C.c_int32, _U64, _U64, _VP, _VP, _VP, _VP]),
    "mtfjsp_encoder_set_bn_mode": (_I, [_VP, C.c_int32]),
    "mtfjsp_encoder_set_product_mode": (_I, [_VP, C.c_int32]),
    "mtfjsp_encoder_set_stats_reduce": (_I, [_VP, C.c_void_p, _VP, C.c_int64]),
    "mtfjsp_encoder_set_deferred_poll": (_I, [_VP, C.c_int32]),
    "mtfjsp_get_mfea1_context": (_I, [_VP, _VP, _VP, C.POINTER(Mfea1Ctx)]),
    "mtfjsp_encoder_arm_mfea1": (_I, [_VP, C.POINTER(Mfea1Ctx)]),
    "mtfjsp_encoder_arm_machine_heads": (_I, [_VP, _VP, _VP, _VP]),
    "mtfjsp_encoder_fused_launches": (_I, [_VP, C.POINTER(C.c_int64)]),
    "mtfjsp_encoder_arm_env_step": (_I, [_VP, _VP, C.c_int32]),
    "mtfjsp_encoder_arm_values_only": (_I, [_VP]),
    "mtfjsp_encoder_env_step_fused": (_I, [_VP]),
    "mtfjsp_encoder_arm_selection": (_I, [_VP, C.c_int32, C.c_int32, _U64, _U64, _VP, _VP, _VP, _VP]),
    "mtfjsp_hostgen_infeasible": (_I, [_VP, C.POINTER(C.c_int32), C.c_int64, _I, _I, C.c_int64, C.c_int64, _VP]),
    "mtfjsp_hostgen_transport": (_I, [_VP, C.POINTER(C.c_int32), C.c_int64, _I, _VP, C.c_double, C.c_double, C.c_double, C.c_int64, C.c_int64, _VP]),
    "mtfjsp_encoder_check": (_I, [_VP, C.POINTER(C.c_int32)]),
    "mtfjsp_encoder_resident_failures": (_I, [_VP, C.POINTER(C.c_int64)]),
    "mtfjsp_encoder_range_fallbacks": (_I, [_VP, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "mtfjsp_encoder_peek_nodes_host": (_I, [_VP, _VP, C.c_int64]),
    "mtfjsp_encoder_timing_begin": (_I, [_VP]),
    "mtfjsp_encoder_timing_end": (_I, [_VP, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "mtfjsp_encoder_timing_query": (_I, [_VP, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
}


def lib_path():
    return os.environ.get("MTFJSP_LIB") or _build.LIB          # MTFJSP_LIB: diagnostic builds only (tools/)


def lib():
    """Load libmtfjsp.so.  Raises if it has not been built: the product never falls back to CPU."""
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: build the HIP extension first (python __graft_entry__.py build, "
                "or python e2e-mappo-for-mt-fjsp_amd/_build.py). There is no CPU fallback.")
        L = C.CDLL(path)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)          # AttributeError if the library does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        _LIB = L
    return _LIB


class MtfjspError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libmtfjsp error {code}: {msg}")
        self.code = code


def check(rc, handle=None, enc=False):
    if rc != 0:
        L = lib()
        msg = (L.mtfjsp_encoder_last_error(handle) if enc else L.mtfjsp_last_error(handle)) or b""
        raise MtfjspError(rc, msg.decode())
