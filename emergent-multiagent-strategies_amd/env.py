"""FortAttack on the MI355X: batched engine + the reference's single-env surface.

* ``BatchedFortAttack``  -- E independent worlds resident in HBM, one HIP launch per
  env-step, torch tensors in/out (device pointers go straight to the C ABI).
* ``FortAttackGlobalEnv`` / ``make_fortattack_env`` -- the reference's Python boundary
  (gym_fortattack/fortattack.py:17-27, :31-225) over the same engine with E = 1, so a
  copy of train_fortattack.py's loop runs unchanged.

Everything here calls the HIP library; there is no CPU path.
"""
import ctypes as C
import types

import numpy as np
import torch

from . import _lib
from .spaces import Box, Discrete, MASpace


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class BatchedFortAttack(object):
    """E FortAttack worlds on one GPU (fa_env handle of include/fortattack.h).

    Env e reproduces the reference run under ``np.random.seed(base_seed + env_offset + e)``
    (``rng="mt19937"``); ``skip_doubles`` = random_sample() draws the reference consumed
    before its first ``env.reset()`` (default 2N: FortAttackEnvV1.__init__ calls
    reset_world once, fortattack_env_v1.py:45).
    """

    def __init__(self, num_envs, num_guards=3, num_attackers=3, max_time_steps=100, base_seed=0,
                 env_offset=0, skip_doubles=None, rng="mt19937", device=0, track_counters=True,
                 step_kernel="auto"):
        lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.FaError("BatchedFortAttack needs a ROCm GPU (torch.cuda.is_available() is False)")
        self.device = torch.device("cuda", device if isinstance(device, int) else torch.device(device).index or 0)
        cfg = _lib.default_config()
        cfg.num_envs, cfg.num_guards, cfg.num_attackers = int(num_envs), int(num_guards), int(num_attackers)
        cfg.max_time_steps = int(max_time_steps)
        cfg.device_id = self.device.index
        cfg.rng_mode = {"mt19937": _lib.FA_RNG_MT19937, "philox": _lib.FA_RNG_PHILOX}[rng]
        cfg.base_seed, cfg.env_offset = int(base_seed), int(env_offset)
        cfg.rng_skip_doubles = -1 if skip_doubles is None else int(skip_doubles)
        cfg.track_counters = int(bool(track_counters))
        cfg.step_kernel = {**_lib.STEP_KERNELS, **_lib.EXPERIMENT_STEP_KERNELS}[step_kernel]   # "auto" | "pipe" | "pipe3" | "waves1" | "waves2" | "waves3"
        self.cfg = cfg
        self.E, self.G, self.A = cfg.num_envs, cfg.num_guards, cfg.num_attackers
        self.N = self.G + self.A
        self.max_time_steps = cfg.max_time_steps
        h = C.c_void_p()
        _lib.check(lib.fa_create(C.byref(cfg), C.byref(h)), "fa_create")
        self._h, self._lib = h, lib
        self.storage = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.fa_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- env API -------------------------------------------------------------------
    def _new(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def reset(self, env_mask=None, obs_f32=None, obs_f64=None):
        """fa_reset.  env_mask: uint8/bool (E) device tensor or None (all envs)."""
        if env_mask is not None:
            env_mask = env_mask.to(self.device, torch.uint8).contiguous()
        if obs_f32 is None and obs_f64 is None:
            obs_f32 = self._new((self.E, self.N, 6), torch.float32)
        _lib.check(self._lib.fa_reset(self._h, _ptr(env_mask), _ptr(obs_f32), _ptr(obs_f64), _stream()),
                   "fa_reset")
        return obs_f32 if obs_f32 is not None else obs_f64

    def step(self, actions, auto_reset=True, out=None, want=("obs_f32", "reward_f32", "mask_f32", "done")):
        """fa_step.  actions: int64 device tensor (E, N) (any strides).  Returns dict."""
        assert actions.dtype == torch.int64 and actions.is_cuda and tuple(actions.shape) == (self.E, self.N)
        shapes = dict(obs_f32=((self.E, self.N, 6), torch.float32), reward_f32=((self.E, self.N), torch.float32),
                      mask_f32=((self.E, self.N), torch.float32), done=((self.E,), torch.uint8),
                      obs_f64=((self.E, self.N, 6), torch.float64), reward_f64=((self.E, self.N), torch.float64),
                      hit=((self.E, self.N), torch.uint8), was_hit=((self.E, self.N), torch.uint8))
        out = dict(out or {})
        for k in want:
            if k not in out:
                out[k] = self._new(*shapes[k])
        io = _lib.StepIO()
        io.actions = _ptr(actions)
        io.act_stride_env, io.act_stride_agent = actions.stride(0), actions.stride(1)
        for k in shapes:
            t = out.get(k)
            if t is not None:
                assert t.is_contiguous() and t.dtype == shapes[k][1] and tuple(t.shape) == shapes[k][0], k
            setattr(io, k, _ptr(t))
        io.auto_reset = int(bool(auto_reset))
        _lib.check(self._lib.fa_step(self._h, C.byref(io), _stream()), "fa_step")
        return out

    def step_many(self, actions, auto_reset=True, want=("obs_f32", "reward_f32", "mask_f32", "done")):
        """fa_step with fa_step_io.num_steps = T: T env-steps in ONE launch, open loop.  actions: int64
        device tensor (T, E, N) (any strides); every output gets a leading T dimension."""
        assert actions.dtype == torch.int64 and actions.is_cuda and tuple(actions.shape[1:]) == (self.E, self.N)
        T = actions.shape[0]
        shapes = dict(obs_f32=((T, self.E, self.N, 6), torch.float32), reward_f32=((T, self.E, self.N), torch.float32),
                      mask_f32=((T, self.E, self.N), torch.float32), done=((T, self.E), torch.uint8),
                      obs_f64=((T, self.E, self.N, 6), torch.float64), reward_f64=((T, self.E, self.N), torch.float64),
                      hit=((T, self.E, self.N), torch.uint8), was_hit=((T, self.E, self.N), torch.uint8))
        out = {k: self._new(*shapes[k]) for k in want}
        io = _lib.StepIO()
        io.actions = _ptr(actions)
        io.act_stride_step, io.act_stride_env, io.act_stride_agent = actions.stride(0), actions.stride(1), actions.stride(2)
        for k in shapes:
            setattr(io, k, _ptr(out.get(k)))
        io.auto_reset = int(bool(auto_reset))
        io.num_steps = T
        _lib.check(self._lib.fa_step(self._h, C.byref(io), _stream()), "fa_step")
        return out

    def set_reset_choice(self, k, out=None):
        """fa_set_reset_choice: np.random.choice(k) on the env's stream after every reset (the ensemble
        path's sample_attacker, learner.py:119-121).  Returns the (E,) int32 device tensor the chosen
        indices are written to (k = 0 switches it off)."""
        if k and out is None:
            out = torch.zeros(self.E, dtype=torch.int32, device=self.device)
        if k:
            assert out.dtype == torch.int32 and out.is_contiguous() and out.numel() == self.E and out.device == self.device
        _lib.check(self._lib.fa_set_reset_choice(self._h, int(k), _ptr(out) if k else None), "fa_set_reset_choice")
        self._choice_out = out if k else None   # keep it alive: the library holds the pointer
        return self._choice_out

    # -- collector -------------------------------------------------------------------
    def bind_storage(self, storage):
        """Attach a JointRolloutStorage (storage.py) -- fa_bind_storage."""
        st = _lib.Storage()
        st.num_steps = storage.num_steps
        for k in ("obs", "recurrent_hidden_states", "rewards", "value_preds", "returns",
                  "action_log_probs", "actions", "masks", "done"):
            t = getattr(storage, k)
            assert t.is_contiguous() and t.device == self.device, k
            setattr(st, k, _ptr(t))
        _lib.check(self._lib.fa_bind_storage(self._h, C.byref(st)), "fa_bind_storage")
        self.storage = storage

    def collect_reset(self):
        _lib.check(self._lib.fa_collect_reset(self._h, _stream()), "fa_collect_reset")

    def collect_step(self, step, auto_reset=True):
        _lib.check(self._lib.fa_collect_step(self._h, int(step), int(bool(auto_reset)), _stream()),
                   "fa_collect_step")

    def collect_rollout(self, step_begin, num_steps, auto_reset=True):
        """Open-loop rollout of `num_steps` env-steps in ONE launch (actions already in storage)."""
        _lib.check(self._lib.fa_collect_rollout(self._h, int(step_begin), int(num_steps),
                                                int(bool(auto_reset)), _stream()), "fa_collect_rollout")

    def gae(self, gamma=0.99, tau=0.95):
        _lib.check(self._lib.fa_gae(self._h, float(gamma), float(tau), _stream()), "fa_gae")

    def gae_moments(self, gamma=0.99, tau=0.95):
        """fa_gae_moments: GAE, then the per-agent advantage moments in one pass.  Returns (moments (N,3)
        {n, mean, M2}, mean (N,), std (N,)) of this handle's samples, float64 on device."""
        if not hasattr(self, "_gae_mom"):
            self._gae_mom = torch.zeros((self.N, 3), dtype=torch.float64, device=self.device)
        if not hasattr(self, "_adv_ms"):
            self._adv_ms = torch.zeros((2, self.N), dtype=torch.float64, device=self.device)
        _lib.check(self._lib.fa_gae_moments(self._h, float(gamma), float(tau), _ptr(self._gae_mom),
                                            _ptr(self._adv_ms[0]), _ptr(self._adv_ms[1]), _stream()), "fa_gae_moments")
        return self._gae_mom, self._adv_ms[0], self._adv_ms[1]

    def gae_normalize(self, gamma=0.99, tau=0.95, out=None):
        """fa_gae_normalize: the collector tail of one rank in two launches -- GAE + moment partials, fold + normalisation
        (ppo.py:121-124).  Returns (adv (T, E, N, 1) float32, moments (N,3), mean (N,), std (N,)); == gae_moments() +
        adv_normalize() bit for bit."""
        if not hasattr(self, "_gae_mom"):
            self._gae_mom = torch.zeros((self.N, 3), dtype=torch.float64, device=self.device)
        if not hasattr(self, "_adv_ms"):
            self._adv_ms = torch.zeros((2, self.N), dtype=torch.float64, device=self.device)
        if out is None:
            out = self._new((self.storage.num_steps, self.E, self.N, 1), torch.float32)
        _lib.check(self._lib.fa_gae_normalize(self._h, float(gamma), float(tau), _ptr(out), _ptr(self._gae_mom),
                                              _ptr(self._adv_ms[0]), _ptr(self._adv_ms[1]), _stream()), "fa_gae_normalize")
        return out, self._gae_mom, self._adv_ms[0], self._adv_ms[1]

    def adv_moments_onepass(self):
        """fa_adv_moments_onepass: the statistics half of gae_moments alone (same buffers, same values), on the current stream."""
        if not hasattr(self, "_gae_mom"):
            self._gae_mom = torch.zeros((self.N, 3), dtype=torch.float64, device=self.device)
        if not hasattr(self, "_adv_ms"):
            self._adv_ms = torch.zeros((2, self.N), dtype=torch.float64, device=self.device)
        _lib.check(self._lib.fa_adv_moments_onepass(self._h, _ptr(self._gae_mom), _ptr(self._adv_ms[0]), _ptr(self._adv_ms[1]),
                                                    _stream()), "fa_adv_moments_onepass")
        return self._gae_mom, self._adv_ms[0], self._adv_ms[1]

    def adv_stats(self, pass_, mean=None, out=None):
        """fa_adv_stats.  pass 0 fills out[i] = {n, sum(A), 0}; pass 1 (needs `mean`) writes only
        out[i][2] = sum((A-mean)^2) -- hand it pass 0's buffer to get the full triple."""
        if out is None:
            out = torch.zeros((self.N, 3), dtype=torch.float64, device=self.device)
        _lib.check(self._lib.fa_adv_stats(self._h, int(pass_), _ptr(mean), _ptr(out), _stream()), "fa_adv_stats")
        return out

    def adv_mean_std(self):
        """fa_adv_mean_std: per-agent mean / unbiased std of this handle's advantages (N,), on device."""
        if not hasattr(self, "_adv_ms"):
            self._adv_ms = torch.zeros((2, self.N), dtype=torch.float64, device=self.device)
        _lib.check(self._lib.fa_adv_mean_std(self._h, _ptr(self._adv_ms[0]), _ptr(self._adv_ms[1]), _stream()),
                   "fa_adv_mean_std")
        return self._adv_ms[0], self._adv_ms[1]

    def adv_moments(self, out=None):
        """fa_adv_moments: (N,3) {n, mean, M2} of this handle's advantages, on device."""
        if out is None:
            out = torch.zeros((self.N, 3), dtype=torch.float64, device=self.device)
        _lib.check(self._lib.fa_adv_moments(self._h, _ptr(out), _stream()), "fa_adv_moments")
        return out

    def adv_merge(self, gathered):
        """fa_adv_merge: gathered (world, N, 3) -> global (mean, std), each (N,)."""
        assert gathered.is_contiguous() and gathered.dtype == torch.float64 and gathered.shape[1:] == (self.N, 3)
        if not hasattr(self, "_adv_ms"):
            self._adv_ms = torch.zeros((2, self.N), dtype=torch.float64, device=self.device)
        _lib.check(self._lib.fa_adv_merge(self._h, _ptr(gathered), int(gathered.shape[0]), _ptr(self._adv_ms[0]),
                                          _ptr(self._adv_ms[1]), _stream()), "fa_adv_merge")
        return self._adv_ms[0], self._adv_ms[1]

    def adv_normalize(self, mean, std, out=None):
        if out is None:
            out = self._new((self.storage.num_steps, self.E, self.N, 1), torch.float32)
        _lib.check(self._lib.fa_adv_normalize(self._h, _ptr(mean), _ptr(std), _ptr(out), _stream()),
                   "fa_adv_normalize")
        return out

    def adv_merge_normalize(self, gathered, out=None):
        """fa_adv_merge_normalize: the several-rank tail behind the all-gather in one launch -- merge of the gathered (W, N, 3)
        moments + normalisation.  Returns (adv, mean, std); == adv_merge() + adv_normalize() bit for bit."""
        if not hasattr(self, "_adv_ms"):
            self._adv_ms = torch.zeros((2, self.N), dtype=torch.float64, device=self.device)
        if out is None:
            out = self._new((self.storage.num_steps, self.E, self.N, 1), torch.float32)
        _lib.check(self._lib.fa_adv_merge_normalize(self._h, _ptr(gathered), int(gathered.shape[0]), _ptr(out),
                                                    _ptr(self._adv_ms[0]), _ptr(self._adv_ms[1]), _stream()), "fa_adv_merge_normalize")
        return out, self._adv_ms[0], self._adv_ms[1]

    def after_update(self):
        _lib.check(self._lib.fa_after_update(self._h, _stream()), "fa_after_update")

    # -- fused policy (csrc/fa_policy.hip) ----------------------------------------------------
    def _policy_io(self, w_guards, w_attackers, seed, counter, step, deterministic, value_only, pool, env_strategy):
        nfl = self._lib.fa_policy_weight_floats()
        io = _lib.PolicyIO()
        for w in (w_guards, w_attackers):
            assert w is None or (w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() and w.numel() == nfl)
        io.weights[0], io.weights[1] = w_guards.data_ptr(), (w_attackers.data_ptr() if w_attackers is not None else None)
        io.counter = _ptr(counter)
        io.seed, io.step, io.deterministic, io.value_only = int(seed), int(step), int(bool(deterministic)), int(bool(value_only))
        if pool is not None:   # (K, weight_floats) packed attacker strategies + per-env strategy index
            assert pool.is_cuda and pool.dtype == torch.float32 and pool.is_contiguous() and pool.shape[1] == nfl
            assert env_strategy.dtype == torch.int32 and env_strategy.is_contiguous() and env_strategy.numel() == self.E
            io.attacker_pool, io.pool_size, io.env_strategy = _ptr(pool), int(pool.shape[0]), _ptr(env_strategy)
        return io

    def policy_act(self, obs, w_guards, w_attackers, seed=0, counter=None, step=0, deterministic=False,
                   value_only=False, out=None, pool=None, env_strategy=None):
        """fa_policy_act: both teams' MPNN forward + sampling on an observation row obs (E, N, 6) float32.
        w_*: packed weight buffers (mpnn_pack.pack_policy); with `pool` (K, floats) + `env_strategy` (E) int32
        env e's attackers run strategy env_strategy[e] of the pool.  Returns (value, action, log_prob), each
        (E, N) (action / log_prob are None with value_only)."""
        assert obs.is_cuda and obs.dtype == torch.float32 and obs.is_contiguous() and tuple(obs.shape) == (self.E, self.N, 6)
        value, action, logp = out if out is not None else (
            self._new((self.E, self.N), torch.float32),
            None if value_only else self._new((self.E, self.N), torch.int64),
            None if value_only else self._new((self.E, self.N), torch.float32))
        io = self._policy_io(w_guards, w_attackers, seed, counter, step, deterministic, value_only, pool, env_strategy)
        io.obs, io.value, io.action, io.log_prob = _ptr(obs), _ptr(value), _ptr(action), _ptr(logp)
        _lib.check(self._lib.fa_policy_act(self._h, C.byref(io), _stream()), "fa_policy_act")
        return value, action, logp

    def collect_act(self, step, w_guards, w_attackers, seed=0, counter=None, deterministic=False, value_only=False,
                    pool=None, env_strategy=None):
        """fa_collect_act: the policies act on storage.obs[step] and write value_preds / actions /
        action_log_probs[step] (value_only: only value_preds[step], the V(obs[T]) of wrap_horizon)."""
        io = self._policy_io(w_guards, w_attackers, seed, counter, step, deterministic, value_only, pool, env_strategy)
        _lib.check(self._lib.fa_collect_act(self._h, int(step), C.byref(io), _stream()), "fa_collect_act")

    # -- state snapshot ----------------------------------------------------------------
    _F64 = ("pos_x", "pos_y", "vel_x", "vel_y", "ang", "prev_dist")

    def get_state(self):
        E, N = self.E, self.N
        s = {k: np.empty((E, N), np.float64) for k in self._F64}
        s.update(alive=np.empty((E, N), np.uint8), time_step=np.empty(E, np.int32),
                 num_hit=np.empty((E, N), np.int32), num_was_hit=np.empty((E, N), np.int32),
                 game_result=np.empty((E, 3), np.uint8), result_count=np.empty((E, 3), np.int64),
                 episode_reward_sum=np.empty((E, N), np.float64), alive_at_end=np.empty((E, N), np.int64))
        sh = _lib.StateHost()
        for k, v in s.items():
            setattr(sh, k, v.ctypes.data_as(C.c_void_p))
        _lib.check(self._lib.fa_get_state(self._h, C.byref(sh)), "fa_get_state")
        return s

    def eval_stats(self):
        """The row the reference's evaluation prints per attacker strategy
        (test_fortattack_v2.py:94-101, mean over finished episodes):
        [all attackers dead, timeout, guards win (sum of the two), attacker reached fort,
         guards alive at the end, attackers alive at the end, mean guard return, mean attacker return]
        -- computed from device-side counters over every env of this handle."""
        s = self.get_state()
        rc = s["result_count"].sum(0).astype(np.float64)
        n_ep = rc.sum()
        if n_ep == 0:
            return np.zeros(8), 0
        G = self.G
        alive = s["alive_at_end"].sum(0) / n_ep
        rew = s["episode_reward_sum"].sum(0) / n_ep
        row = np.array([rc[0] / n_ep, rc[1] / n_ep, (rc[0] + rc[1]) / n_ep, rc[2] / n_ep,
                        alive[:G].sum(), alive[G:].sum(), rew[:G].mean(), rew[G:].mean()])
        return row, int(n_ep)

    def set_state(self, s):
        keep = []
        sh = _lib.StateHost()
        for k in self._F64 + ("alive", "time_step", "num_hit", "num_was_hit", "game_result"):
            if k in s:
                dt = np.float64 if k in self._F64 else (np.uint8 if k in ("alive", "game_result") else np.int32)
                a = np.ascontiguousarray(s[k], dt)
                assert a.shape == ((self.E,) if k == "time_step" else (self.E, 3) if k == "game_result" else (self.E, self.N)), k
                keep.append(a)
                setattr(sh, k, a.ctypes.data_as(C.c_void_p))
        _lib.check(self._lib.fa_set_state(self._h, C.byref(sh)), "fa_set_state")

    def selftest_math(self, samples=1 << 26, seed=1):
        """fa_selftest_math -> (mismatching quotients, mismatching roots, max sin/cos deviation from
        the device libm in ulp); the first two must be 0."""
        out = np.zeros(3, np.uint64)
        _lib.check(self._lib.fa_selftest_math(self._h, int(samples), int(seed), out.ctypes.data_as(C.c_void_p)),
                   "fa_selftest_math")
        return int(out[0]), int(out[1]), float(out[2]) / 1000.0

    def step_variant(self, num_steps=1):
        """fa_step_variant: name of the step kernel a launch of `num_steps` env-steps uses."""
        return self._lib.fa_step_variant(self._h, int(num_steps)).decode()

    def policy_variant(self):
        """fa_policy_variant: the fa_policy_kernel shape policy_act / collect_act launch (no attacker pool)."""
        return self._lib.fa_policy_variant(self._h).decode()

    def rng_peek(self, e, count):
        out = np.empty(count, np.float64)
        _lib.check(self._lib.fa_rng_peek(self._h, int(e), int(count), out.ctypes.data_as(C.c_void_p)),
                   "fa_rng_peek")
        return out


# =====================================================================================
# The reference's single-env surface
# =====================================================================================
class _AgentView(object):
    """Read-only stand-in for gym_fortattack.core.Agent as seen through env.world."""

    def __init__(self, env, i, attacker):
        self._env, self._i, self.attacker = env, i, attacker
        self.name = "agent %d" % (i + 1)  # fortattack_env_v1.py:28
        self.action_callback = None
        self.movable, self.silent, self.collide = True, True, True
        self.size, self.accel, self.max_speed, self.max_rot = 0.05, 3, 3, 0.17
        self.shootRad, self.shootWin = 0.8, np.pi / 4

    @property
    def alive(self):
        return bool(self._env._state()["alive"][0, self._i])

    @property
    def numHit(self):
        return int(self._env._state()["num_hit"][0, self._i])

    @property
    def numWasHit(self):
        return int(self._env._state()["num_was_hit"][0, self._i])

    @property
    def state(self):
        s = self._env._state()
        i = self._i
        return types.SimpleNamespace(p_pos=np.array([s["pos_x"][0, i], s["pos_y"][0, i]]),
                                     p_vel=np.array([s["vel_x"][0, i], s["vel_y"][0, i]]),
                                     p_ang=float(s["ang"][0, i]))


class _WorldView(object):
    """The attributes of World / the scenario that callers read."""

    def __init__(self, env):
        self._env = env
        self.agents = [_AgentView(env, i, i >= env._eng.G) for i in range(env._eng.N)]
        self.numGuards, self.numAttackers = env._eng.G, env._eng.A
        self.numAgents = env._eng.N
        self.max_time_steps = env._eng.max_time_steps
        self.fortDim, self.doorLoc = 0.15, np.array([0, 0.8])
        self.wall_pos = [-1, 1, -0.8, 0.8]
        self.dim_p, self.dim_c = 3, 0

    @property
    def policy_agents(self):
        return self.agents

    @property
    def numAliveGuards(self):
        return int(self._env._state()["alive"][0, :self.numGuards].sum())

    @property
    def numAliveAttackers(self):
        return int(self._env._state()["alive"][0, self.numGuards:].sum())

    @property
    def time_step(self):
        return int(self._env._state()["time_step"][0])

    @property
    def gameResult(self):
        return self._env._state()["game_result"][0].astype(np.int64)


class FortAttackGlobalEnv(object):
    """gym_fortattack/fortattack.py:31 FortAttackGlobalEnv, one env on the GPU engine.

    ``reset() -> (N,6) float64``; ``step(action_n) -> (obs (N,6) f64, reward_n list,
    done bool, {'n': [{}]*N})``.

    RNG.  The reference draws its reset positions from numpy's GLOBAL stream
    (fortattack_env_v1.py:66,70), seeded by the training script (train_fortattack.py:200).
    ``seed=None`` (default) does exactly that: every reset -- and the one the scenario's constructor
    runs (fortattack_env_v1.py:45) -- calls ``np.random.uniform`` in the reference's order and hands the
    positions to the engine, so the global stream advances as under the reference and anything else
    that uses it (``Learner.sample_attacker``'s ``np.random.choice``, learner.py:120) stays in step:
    ``train_fortattack.py`` runs with the one-line import swap and no extra arguments.
    ``seed=<int>`` makes the env own a private stream instead (the engine's device-side MT19937,
    == the reference under ``np.random.seed(seed)`` whose construction consumed ``skip_doubles`` draws).
    Nothing is printed at episode end (the reference prints, fortattack.py:208,214,220).
    """
    metadata = {"render.modes": ["human", "rgb_array"]}

    def __init__(self, num_steps, num_guards=5, num_attackers=5, seed=None, skip_doubles=None, device=0):
        self._np_global = seed is None
        self._eng = BatchedFortAttack(1, num_guards, num_attackers, num_steps, base_seed=0 if seed is None else seed,
                                      skip_doubles=skip_doubles, device=device)
        e = self._eng
        self.n = e.N                                              # fortattack.py:47
        self.agent_num = e.N
        self.ob_rms = None                                        # fortattack.py:42
        self.discrete_action_space = True
        self.discrete_action_input = True
        self.shared_reward = False
        self.action_space = [Discrete(8) for _ in range(e.N)]     # fortattack.py:73,94
        self.observation_space = [Box(-np.inf, np.inf, (6,), np.float32) for _ in range(e.N)]  # :98
        self.action_spaces = MASpace(tuple(Box(0., 1., (8,)) for _ in range(e.N)))              # :107
        self.observation_spaces = MASpace(tuple(Box(-np.inf, np.inf, (6,)) for _ in range(e.N)))
        self.action_range = [0., 1.]
        self._cache = None
        self._obs64 = torch.empty((1, e.N, 6), dtype=torch.float64, device=e.device)
        # step(): actions are read from, and the (N, 6) observation, the N rewards and the done flag are written to,
        # PINNED HOST memory by the kernel itself (one launch + one stream synchronisation per env.step: no staging copies).
        # One packed block of doubles: obs | reward | done (first byte of the last slot).
        N = e.N
        self._h_act = torch.zeros((1, N), dtype=torch.int64).pin_memory()
        self._h_out = torch.zeros(7 * N + 1, dtype=torch.float64).pin_memory()
        self._np_act = self._h_act.numpy()
        out = self._h_out.numpy()
        self._np_obs, self._np_rew = out[:6 * N].reshape(N, 6), out[6 * N:7 * N]
        self._np_done = out[7 * N:].view(np.uint8)
        io = _lib.StepIO()
        io.actions = C.c_void_p(self._h_act.data_ptr())
        io.act_stride_env, io.act_stride_agent = N, 1
        base = self._h_out.data_ptr()
        io.obs_f64, io.reward_f64, io.done = C.c_void_p(base), C.c_void_p(base + 48 * N), C.c_void_p(base + 56 * N)
        io.auto_reset = 0
        self._io = io
        self._alive_before = np.ones(e.N, bool)
        self.world = _WorldView(self)
        self.agents = self.world.policy_agents
        if self._np_global:
            self._draw_reset_positions()   # FortAttackEnvV1.__init__ -> reset_world (fortattack_env_v1.py:45)

    def _state(self):
        if self._cache is None:
            self._cache = self._eng.get_state()
        return self._cache

    def seed(self, seed=None):  # gym.Env default: a no-op in the reference too (eval.py:26)
        return []

    def _draw_reset_positions(self):
        """The RNG calls of reset_world in the reference's order (fortattack_env_v1.py:56-70): per agent,
        guards first, two np.random.uniform(lo, hi, 1) draws from the global stream."""
        w = self._eng.cfg.world
        xMin, xMax, yMin, yMax = w.wall_xmin, w.wall_xmax, w.wall_ymin, w.wall_ymax
        pos = np.empty((self.n, 2))
        for i in range(self.n):
            if i >= self._eng.G:   # attackers start from far away (:66)
                pos[i] = np.concatenate((np.random.uniform(xMin, xMax, 1), np.random.uniform(yMin, 0.8 * yMin, 1)))
            else:                  # guards start near the door (:70)
                pos[i] = np.concatenate((np.random.uniform(-0.8 * w.fort_dim / 2, 0.8 * w.fort_dim / 2, 1),
                                         np.random.uniform(0.8 * yMax, yMax, 1)))
        return pos

    def reset(self):
        self._cache = None
        self._stepped = False   # render(): no lasers of the previous episode's last action on the first frame
        e = self._eng
        if self._np_global:
            pos = self._draw_reset_positions()
            ang = np.where(np.arange(self.n) >= e.G, np.pi / 2, 3 * np.pi / 2)            # :59
            z = np.zeros((1, self.n))
            # everything reset_world sets; prevDist is NOT reset (quirk Q1): left alone
            e.set_state(dict(pos_x=pos[None, :, 0], pos_y=pos[None, :, 1], vel_x=z, vel_y=z, ang=ang[None],
                             alive=np.ones((1, self.n), np.uint8), time_step=np.zeros(1, np.int32),
                             num_hit=z.astype(np.int32), num_was_hit=z.astype(np.int32),
                             game_result=np.zeros((1, 3), np.uint8)))
            obs = np.stack([np.ones(self.n), pos[:, 0], pos[:, 1], ang, np.zeros(self.n), np.zeros(self.n)], 1)
        else:
            e.reset(obs_f64=self._obs64)
            obs = self._obs64[0].cpu().numpy()
        self._alive_before = obs[:, 0] != 0
        return obs

    def step(self, action_n):
        a = np.asarray(action_n).reshape(-1)
        if a.shape[0] != self.n:
            raise AssertionError("expected %d actions, got %d" % (self.n, a.shape[0]))
        self._cache = None
        self._stepped = True
        self._np_act[0, :] = a
        e = self._eng
        stream = torch.cuda.current_stream(e.device)
        _lib.check(e._lib.fa_step(e._h, C.byref(self._io), C.c_void_p(stream.cuda_stream)), "fa_step")
        stream.synchronize()
        obs = self._np_obs.copy()
        rew = self._np_rew
        # fortattack_env_v1.py:87-92: an agent that is neither alive nor justDied gets the int literal 0;
        # (alive or justDied) after the step == alive before it
        reward_n = [rew[i] if self._alive_before[i] else 0 for i in range(self.n)]
        self._alive_before = obs[:, 0] != 0
        done = bool(self._np_done[0])
        return obs, reward_n, done, {"n": [{} for _ in range(self.n)]}

    def render(self, attn_list=None, mode="human", close=False, size=350, viz_dead=False):
        """fortattack.py:368-600.  The pyglet window is out of scope: mode "human"
        stays a no-op returning [] as the reference's trainers expect; mode "rgb_array" returns [frame], the same
        scene rasterised headlessly from the device state (render.py), lasers for the agents whose last action was 7."""
        if mode != "rgb_array":
            return []
        from .render import render_state
        st = self._eng.get_state()
        shoot = (self._np_act.reshape(1, -1) == 7) if getattr(self, "_stepped", False) else None
        return [render_state(st, 0, self._eng.G, shoot=shoot, size=size, viz_dead=viz_dead)]

    def terminate(self):
        pass

    def close(self):
        self._eng.close()


def make_fortattack_env(num_steps, benchmark=False, num_guards=5, num_attackers=5, seed=None,
                        skip_doubles=None, device=0):
    """gym_fortattack/fortattack.py:17-27.  Defaults = the reference's 5v5 scenario drawing its reset
    positions from numpy's global stream (seed=None); see FortAttackGlobalEnv for `seed`."""
    return FortAttackGlobalEnv(num_steps, num_guards, num_attackers, seed=seed,
                               skip_doubles=skip_doubles, device=device)


def ppo_grad(obs, action, value_pred, ret, old_logp, adv, w, wt, scale, team, G, A, clip, c_value, c_entropy,
             clipped_value_loss=True, scratch=None, out=None, idx=None, normalize=True, share_cu=False, adv_stats=None):
    """fa_ppo_grad: one team's PPO minibatch forward + losses + backward in one fused launch (+ reduction).
    obs (rows, N, 6) float32; action (rows, N[, 1]) int64; value_pred / ret / old_logp / adv (rows, N[, 1]) float32;
    idx: int64 row indices of the minibatch (None: every row); w / wt the packed weights and their transposes
    (mpnn_pack.FlatPolicy.fold_pack / pack_from_params); scale: device float32[2], or None -> the library takes the
    alive-mask mean itself and (normalize) divides by it.  adv_stats = (mean, std) float64 (N,) device tensors: the kernel
    normalises ret - value_pred itself (ppo.py:121-124) and `adv` may be None.  Returns (out, scratch): out = FA_SLAB floats
    (plain-layout gradients + loss sums), scratch reusable."""
    lib = _lib.load()
    N = obs.shape[1]
    B = obs.shape[0] if idx is None else idx.numel()
    for t in (obs, value_pred, ret, old_logp, w, wt) + (() if scale is None else (scale,)) + (() if adv is None else (adv,)):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
    assert adv is not None or adv_stats is not None
    if adv_stats is not None:
        assert all(t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and t.numel() == N for t in adv_stats)
    assert action.dtype == torch.int64 and action.is_contiguous() and N == G + A
    assert idx is None or (idx.is_cuda and idx.dtype == torch.int64 and idx.is_contiguous())
    sf, hf = C.c_int64(), C.c_int64()
    _lib.check(lib.fa_ppo_grad_scratch(B, G, A, C.byref(sf), C.byref(hf)), "fa_ppo_grad_scratch")
    if scratch is None:
        scratch = (torch.empty(sf.value, device=obs.device), torch.empty(hf.value, device=obs.device))
    elif scratch[0].numel() < sf.value or scratch[1].numel() < hf.value or any(
            not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()) for t in scratch):
        # (a scratch sized for another minibatch / team shape or an older record layout: the kernels would write past it)
        raise ValueError("ppo_grad: the scratch passed in holds (%d, %d) floats, this call needs (%d, %d) (fa_ppo_grad_scratch)"
                         % (scratch[0].numel(), scratch[1].numel(), sf.value, hf.value))
    if out is None:
        out = torch.empty(lib.fa_ppo_grad_floats(), device=obs.device)
    io = _lib.PPOGradIO()
    io.obs, io.action, io.value_pred, io.ret, io.old_log_prob = [_ptr(t) for t in (obs, action, value_pred, ret, old_logp)]
    io.adv = None if adv is None else _ptr(adv)
    io.adv_mean, io.adv_std = (None, None) if adv_stats is None else (_ptr(adv_stats[0]), _ptr(adv_stats[1]))
    io.weights, io.weights_t, io.slabs, io.hsave, io.out = [_ptr(t) for t in (w, wt, scratch[0], scratch[1], out)]
    io.scale = None if scale is None else _ptr(scale)
    io.idx = None if idx is None else _ptr(idx)
    io.B, io.num_guards, io.num_attackers, io.team = B, G, A, team
    io.clip_param, io.value_loss_coef, io.entropy_coef = clip, c_value, c_entropy
    io.clipped_value_loss, io.normalize = int(bool(clipped_value_loss)), int(bool(normalize))
    io.share_cu = int(bool(share_cu))    # (kept in the ABI; the 32-row tile kernel shares a CU two ways always and ignores it)
    _lib.check(lib.fa_ppo_grad(C.byref(io), _stream()), "fa_ppo_grad")
    return out, scratch
