"""ctypes view of the C ABI in include/fortattack.h (csrc/libfortattack_hip.so).

There is no CPU fallback: if the HIP library is missing or a call fails, this raises.
"""
import ctypes as C
import os

from . import build as _build

c_p = C.c_void_p


class FaError(RuntimeError):
    pass


class WorldConsts(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "agent_size", "accel", "max_speed", "max_rot", "fort_dim", "door_x", "door_y", "dt", "damping",
        "contact_force", "contact_margin", "wall_xmin", "wall_xmax", "wall_ymin", "wall_ymax",
        "shoot_rad", "shoot_win")]


class Config(C.Structure):
    _fields_ = [("num_envs", C.c_int32), ("num_guards", C.c_int32), ("num_attackers", C.c_int32),
                ("max_time_steps", C.c_int32), ("device_id", C.c_int32), ("rng_mode", C.c_int32),
                ("base_seed", C.c_uint64), ("env_offset", C.c_int64), ("rng_skip_doubles", C.c_int32),
                ("track_counters", C.c_int32), ("step_kernel", C.c_int32), ("world", WorldConsts)]


class StepIO(C.Structure):
    _fields_ = [("actions", c_p), ("act_stride_env", C.c_int64), ("act_stride_agent", C.c_int64),
                ("obs_f32", c_p), ("reward_f32", c_p), ("mask_f32", c_p), ("done", c_p),
                ("obs_f64", c_p), ("reward_f64", c_p), ("hit", c_p), ("was_hit", c_p),
                ("auto_reset", C.c_int32), ("num_steps", C.c_int32), ("act_stride_step", C.c_int64)]


class Storage(C.Structure):
    _fields_ = [("num_steps", C.c_int32), ("obs", c_p), ("recurrent_hidden_states", c_p),
                ("rewards", c_p), ("value_preds", c_p), ("returns", c_p), ("action_log_probs", c_p),
                ("actions", c_p), ("masks", c_p), ("done", c_p)]


class PolicyIO(C.Structure):
    _fields_ = [("obs", c_p), ("weights", c_p * 2), ("value", c_p), ("action", c_p), ("log_prob", c_p),
                ("counter", c_p), ("seed", C.c_uint64), ("step", C.c_int32), ("deterministic", C.c_int32),
                ("value_only", C.c_int32), ("attacker_pool", c_p), ("pool_size", C.c_int32), ("env_strategy", c_p)]


class PPOGradIO(C.Structure):
    _fields_ = [("obs", c_p), ("action", c_p), ("value_pred", c_p), ("ret", c_p), ("old_log_prob", c_p), ("adv", c_p),
                ("idx", c_p), ("weights", c_p), ("weights_t", c_p), ("scale", c_p), ("slabs", c_p), ("hsave", c_p), ("out", c_p),
                ("B", C.c_int32), ("num_guards", C.c_int32), ("num_attackers", C.c_int32), ("team", C.c_int32),
                ("clip_param", C.c_float), ("value_loss_coef", C.c_float), ("entropy_coef", C.c_float),
                ("clipped_value_loss", C.c_int32), ("normalize", C.c_int32), ("share_cu", C.c_int32),
                ("adv_mean", c_p), ("adv_std", c_p)]


class Task(C.Structure):
    _fields_ = [("C", c_p), ("A", c_p), ("B", c_p), ("ldc", C.c_int32), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("a_rs", C.c_int32), ("a_cs", C.c_int32), ("b_rs", C.c_int32), ("b_cs", C.c_int32), ("alpha", C.c_float),
                ("type", C.c_int32)]


class StateHost(C.Structure):
    _fields_ = [(n, c_p) for n in (
        "pos_x", "pos_y", "vel_x", "vel_y", "ang", "prev_dist", "alive", "time_step", "num_hit",
        "num_was_hit", "game_result", "result_count", "episode_reward_sum", "alive_at_end")]


FA_RNG_MT19937, FA_RNG_PHILOX = 0, 1
# fa_config.step_kernel (include/fortattack.h FA_KERNEL_*)
STEP_KERNELS = {"auto": 0, "pipe": 1, "pipe3": 2, "waves1": 3, "waves2": 4, "waves3": 5}
# csrc/experiments/fa_step_experiments.h: kernels of variant libraries only (tools/build_variant.py); fa_create of the
# product library refuses these values
EXPERIMENT_STEP_KERNELS = {"pairs": 64, "chain": 65}

# every symbol include/fortattack.h declares
EXPORTS = {
    "fa_config_default": (C.c_int, [C.POINTER(Config)]),
    "fa_create": (C.c_int, [C.POINTER(Config), C.POINTER(c_p)]),
    "fa_destroy": (None, [c_p]),
    "fa_last_error": (C.c_char_p, []),
    "fa_num_agents": (C.c_int, [c_p]),
    "fa_num_envs": (C.c_int, [c_p]),
    "fa_reset": (C.c_int, [c_p, c_p, c_p, c_p, c_p]),
    "fa_step": (C.c_int, [c_p, C.POINTER(StepIO), c_p]),
    "fa_set_reset_choice": (C.c_int, [c_p, C.c_int32, c_p]),
    "fa_bind_storage": (C.c_int, [c_p, C.POINTER(Storage)]),
    "fa_collect_step": (C.c_int, [c_p, C.c_int32, C.c_int32, c_p]),
    "fa_collect_rollout": (C.c_int, [c_p, C.c_int32, C.c_int32, C.c_int32, c_p]),
    "fa_collect_reset": (C.c_int, [c_p, c_p]),
    "fa_gae": (C.c_int, [c_p, C.c_double, C.c_double, c_p]),
    "fa_gae_moments": (C.c_int, [c_p, C.c_double, C.c_double, c_p, c_p, c_p, c_p]),
    "fa_gae_normalize": (C.c_int, [c_p, C.c_double, C.c_double, c_p, c_p, c_p, c_p, c_p]),
    "fa_adv_moments_onepass": (C.c_int, [c_p, c_p, c_p, c_p, c_p]),
    "fa_adv_stats": (C.c_int, [c_p, C.c_int32, c_p, c_p, c_p]),
    "fa_adv_mean_std": (C.c_int, [c_p, c_p, c_p, c_p]),
    "fa_adv_moments": (C.c_int, [c_p, c_p, c_p]),
    "fa_adv_merge": (C.c_int, [c_p, c_p, C.c_int32, c_p, c_p, c_p]),
    "fa_adv_normalize": (C.c_int, [c_p, c_p, c_p, c_p, c_p]),
    "fa_adv_merge_normalize": (C.c_int, [c_p, c_p, C.c_int32, c_p, c_p, c_p, c_p]),
    "fa_after_update": (C.c_int, [c_p, c_p]),
    "fa_policy_act": (C.c_int, [c_p, C.POINTER(PolicyIO), c_p]),
    "fa_collect_act": (C.c_int, [c_p, C.c_int32, C.POINTER(PolicyIO), c_p]),
    "fa_policy_weight_floats": (C.c_int64, []),
    "fa_policy_plain_floats": (C.c_int64, []),
    "fa_attend_forward": (C.c_int, [c_p, c_p, c_p, c_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_p]),
    "fa_attend_backward": (C.c_int, [c_p, c_p, c_p, c_p, c_p, c_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_p]),
    "fa_ppo_grad": (C.c_int, [C.POINTER(PPOGradIO), c_p]),
    "fa_ppo_grad_floats": (C.c_int64, []),
    "fa_ppo_grad_scratch": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "fa_policy_weight_t_floats": (C.c_int64, []),
    "fa_adam_step": (C.c_int, [c_p, c_p, c_p, c_p, c_p, c_p, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float,
                               C.c_float, c_p, c_p]),
    "fa_adam_step_dev": (C.c_int, [c_p, c_p, c_p, c_p, c_p, c_p, C.c_int32, C.c_int32, c_p, c_p, c_p]),
    "fa_adam_scratch_floats": (C.c_int64, []),
    "fa_run_tasks": (C.c_int, [c_p, C.c_int32, c_p]),
    "fa_pack_weights": (C.c_int, [c_p, c_p, c_p, c_p]),
    "fa_rccl_available": (C.c_int, []),
    "fa_rccl_library": (C.c_char_p, []),
    "fa_rccl_unique_id": (C.c_int, [c_p]),
    "fa_rccl_comm_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, c_p, C.c_int32, C.c_int32]),
    "fa_rccl_comm_destroy": (C.c_int, [c_p]),
    "fa_rccl_comm_ranks": (C.c_int, [c_p]),
    "fa_adv_allreduce": (C.c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "fa_gae_allreduce_normalize": (C.c_int, [c_p, C.c_double, C.c_double, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "fa_grad_allreduce": (C.c_int, [c_p, C.c_int64, c_p, c_p]),
    "fa_get_state": (C.c_int, [c_p, C.POINTER(StateHost)]),
    "fa_set_state": (C.c_int, [c_p, C.POINTER(StateHost)]),
    "fa_selftest_math": (C.c_int, [c_p, C.c_uint64, C.c_uint64, c_p]),
    "fa_step_variant": (C.c_char_p, [c_p, C.c_int32]),
    "fa_policy_variant": (C.c_char_p, [c_p]),
    "fa_rng_peek": (C.c_int, [c_p, C.c_int32, C.c_int32, c_p]),
}

_lib = None


def lib_path():
    return _build.LIB


def load():
    """Load libfortattack_hip.so (must have been built: __graft_entry__.build())."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.isfile(path):
            raise FaError("HIP engine not built: %s is missing (run __graft_entry__.build() or "
                          "python emergent-multiagent-strategies_amd/build.py)" % path)
        lib = C.CDLL(path)
        for name, (res, args) in EXPORTS.items():
            fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        msg = load().fa_last_error()
        raise FaError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))


def default_config():
    cfg = Config()
    check(load().fa_config_default(C.byref(cfg)), "fa_config_default")
    return cfg
