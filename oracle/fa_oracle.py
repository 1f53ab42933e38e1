"""ctypes binding of oracle/libfa_oracle.so (CPU restatement of the reference env).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FA_ORACLE_LIB: another build of the same source (`make asan`: libfa_oracle_asan.so under LD_PRELOAD=libasan.so)
_LIB_PATH = os.environ.get("FA_ORACLE_LIB") or os.path.join(_HERE, "libfa_oracle.so")
_lib = None


def build(force=False):
    """Compile the oracle with gcc (seconds)."""
    if os.environ.get("FA_ORACLE_LIB"):
        return _LIB_PATH
    if force or not os.path.isfile(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "fa_oracle.c"))):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


class _Cfg(C.Structure):
    _fields_ = [("num_envs", C.c_int32), ("num_guards", C.c_int32),
                ("num_attackers", C.c_int32), ("max_time_steps", C.c_int32)]


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.fao_create.argtypes = [C.POINTER(_Cfg), C.POINTER(C.c_void_p)]
        _lib.fao_destroy.argtypes = [C.c_void_p]
        _lib.fao_destroy.restype = None
        _lib.fao_seed.argtypes = [C.c_void_p, C.c_uint64, C.c_int]
        _lib.fao_rng_doubles.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _lib.fao_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.fao_step.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int] + [C.c_void_p] * 6
        _lib.fao_rollout.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.fao_set_choice.argtypes = [C.c_void_p, C.c_int]
        _lib.fao_get_choice.argtypes = [C.c_void_p, C.c_void_p]
        _lib.fao_get_state.argtypes = [C.c_void_p] + [C.c_void_p] * 11
        _lib.fao_set_state.argtypes = [C.c_void_p] + [C.c_void_p] * 8
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleEnv(object):
    """E independent FortAttack worlds stepped on the CPU (OpenMP over envs).

    Env e is the reference run under ``np.random.seed(base_seed + e)`` after
    ``skip_doubles`` random_sample() draws (construction).
    """

    def __init__(self, num_envs, num_guards, num_attackers, max_time_steps, base_seed=0,
                 skip_doubles=None):
        lib = _load()
        self.E, self.G, self.A = num_envs, num_guards, num_attackers
        self.N = num_guards + num_attackers
        cfg = _Cfg(num_envs, num_guards, num_attackers, max_time_steps)
        h = C.c_void_p()
        if lib.fao_create(C.byref(cfg), C.byref(h)) != 0:
            raise ValueError("fao_create failed")
        self._h = h
        if skip_doubles is None:
            skip_doubles = 2 * self.N  # FortAttackEnvV1.__init__ -> reset_world (fortattack_env_v1.py:45)
        lib.fao_seed(self._h, C.c_uint64(base_seed), int(skip_doubles))

    def __del__(self):
        if getattr(self, "_h", None) is not None and _lib is not None:
            _lib.fao_destroy(self._h)
            self._h = None

    def set_choice(self, k):
        """Every reset is followed by np.random.choice(k) on the env's stream (learner.py:119-121)."""
        _lib.fao_set_choice(self._h, int(k))

    def get_choice(self):
        out = np.empty(self.E, np.int32)
        _lib.fao_get_choice(self._h, _p(out))
        return out

    def rng_doubles(self, e, count):
        out = np.empty(count, np.float64)
        _lib.fao_rng_doubles(self._h, int(e), int(count), _p(out))
        return out

    def reset(self, mask=None):
        obs = np.empty((self.E, self.N, 6), np.float64)
        if mask is not None:
            mask = np.ascontiguousarray(mask, np.uint8)
            obs[:] = np.nan
        _lib.fao_reset(self._h, _p(mask), _p(obs))
        return obs

    def step(self, actions, auto_reset=False, want_flags=True):
        """actions: int array (E, N).  Returns dict of arrays."""
        a = np.ascontiguousarray(actions, np.int64).reshape(self.E, self.N)
        out = dict(obs=np.empty((self.E, self.N, 6), np.float64),
                   reward=np.empty((self.E, self.N), np.float64),
                   done=np.empty(self.E, np.uint8),
                   alive_before=np.empty((self.E, self.N), np.uint8))
        if want_flags:
            out["hit"] = np.empty((self.E, self.N), np.uint8)
            out["was_hit"] = np.empty((self.E, self.N), np.uint8)
        _lib.fao_step(self._h, _p(a), self.N, 1, int(bool(auto_reset)), _p(out["obs"]),
                      _p(out["reward"]), _p(out["done"]), _p(out["alive_before"]),
                      _p(out.get("hit")), _p(out.get("was_hit")))
        return out

    def step_noout(self, actions, auto_reset=True):
        """Timing helper: step without materialising outputs (a must be int64 (E,N) contiguous)."""
        _lib.fao_step(self._h, _p(actions), self.N, 1, int(bool(auto_reset)), None, None, None, None,
                      None, None)

    def rollout_noout(self, actions, auto_reset=True):
        """Timing helper: T steps of every env in one OpenMP region (fao_rollout).  actions: int64
        (T, E, N) contiguous."""
        assert actions.dtype == np.int64 and actions.flags.c_contiguous and actions.shape[1:] == (self.E, self.N)
        _lib.fao_rollout(self._h, _p(actions), int(actions.shape[0]), int(bool(auto_reset)), None, None, None)

    def rollout(self, actions, auto_reset=True):
        """fao_rollout returning the last step's (obs, reward, done)."""
        a = np.ascontiguousarray(actions, np.int64)
        obs = np.empty((self.E, self.N, 6), np.float64)
        rew = np.empty((self.E, self.N), np.float64)
        done = np.empty(self.E, np.uint8)
        _lib.fao_rollout(self._h, _p(a), int(a.shape[0]), int(bool(auto_reset)), _p(obs), _p(rew), _p(done))
        return obs, rew, done

    def get_state(self):
        E, N = self.E, self.N
        s = dict(pos_x=np.empty((E, N)), pos_y=np.empty((E, N)), vel_x=np.empty((E, N)),
                 vel_y=np.empty((E, N)), ang=np.empty((E, N)), prev_dist=np.empty((E, N)),
                 alive=np.empty((E, N), np.uint8), time_step=np.empty(E, np.int32),
                 num_hit=np.empty((E, N), np.int32), num_was_hit=np.empty((E, N), np.int32),
                 game_result=np.empty((E, 3), np.uint8))
        _lib.fao_get_state(self._h, *[_p(s[k]) for k in (
            "pos_x", "pos_y", "vel_x", "vel_y", "ang", "prev_dist", "alive", "time_step",
            "num_hit", "num_was_hit", "game_result")])
        return s

    def set_state(self, s):
        arrs = [np.ascontiguousarray(s[k], np.float64) for k in
                ("pos_x", "pos_y", "vel_x", "vel_y", "ang", "prev_dist")]
        arrs.append(np.ascontiguousarray(s["alive"], np.uint8))
        arrs.append(np.ascontiguousarray(s["time_step"], np.int32))
        _lib.fao_set_state(self._h, *[_p(a) for a in arrs])
