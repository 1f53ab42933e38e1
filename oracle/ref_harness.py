"""Import harness for the *reference* implementation (build container only).

TEST INFRASTRUCTURE -- never imported by the product package, by bench.py's timed
path, or on the GPU box (``/root/reference`` does not exist there).

The reference (``/root/reference``, Python) needs ``gym``, ``pygame``, ``pyglet``,
``cv2``, ``gym_vecenv`` and ``tensorboardX`` at import time
(``gym_fortattack/__init__.py:1-8``, ``gym_fortattack/fortattack.py:1-12``,
``utils.py:4``, ``train_fortattack.py:11``).  None of them is installed in this
image, so this module registers minimal stand-ins in ``sys.modules`` (the recipe
of SURVEY.md Appendix B.1) and then imports the reference unmodified.  It is used
by ``oracle/gen_golden.py`` to produce the fixtures under ``tests/golden/`` and by
the optional live cross-check tests (skipped when the reference is absent).

Nothing here restates reference logic: the stand-ins carry only the names the
reference touches while being imported / constructed (spaces are metadata).
"""
import contextlib
import io
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = os.environ.get("FA_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "gym_fortattack", "core.py"))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _install_stubs():
    if "gym" in sys.modules and getattr(sys.modules["gym"], "_fa_stub", False):
        return
    registry = {}

    class Space(object):
        def __init__(self, shape=None, dtype=None):
            self.shape = shape
            self.dtype = dtype

    class Discrete(Space):
        def __init__(self, n):
            Space.__init__(self, (), np.int64)
            self.n = n

    class Box(Space):
        def __init__(self, low=None, high=None, shape=None, dtype=np.float32):
            Space.__init__(self, tuple(shape) if shape is not None else np.shape(low), dtype)
            self.low, self.high = low, high

    class Tuple(Space):
        def __init__(self, spaces):
            Space.__init__(self)
            self.spaces = spaces

    class Dict(Space):
        def __init__(self, spaces=None):
            Space.__init__(self)
            self.spaces = spaces

    class Env(object):
        metadata = {}

        def seed(self, seed=None):
            return []

    class EnvSpec(object):
        def __init__(self, id, entry_point=None, **kw):
            self.id, self.entry_point = id, entry_point

    def register(id, entry_point=None, **kw):
        registry[id] = entry_point

    def make(id, **kw):
        import importlib
        mod, cls = registry[id].split(":")
        return getattr(importlib.import_module(mod), cls)(**kw)

    gym = _mod("gym", Env=Env, Space=Space, make=make, _fa_stub=True)
    spaces = _mod("gym.spaces", Space=Space, Discrete=Discrete, Box=Box, Tuple=Tuple, Dict=Dict)
    gym.spaces = spaces
    gym.error = _mod("gym.error", Error=Exception)
    seeding = _mod("gym.utils.seeding", np_random=lambda seed=None: (np.random.RandomState(seed), seed))
    gym.utils = _mod("gym.utils", seeding=seeding)
    registration = _mod("gym.envs.registration", register=register, EnvSpec=EnvSpec)
    gym.envs = _mod("gym.envs", registration=registration)
    gym.wrappers = _mod("gym.wrappers", Monitor=object)

    music = types.SimpleNamespace(load=lambda *a, **k: None, play=lambda *a, **k: None)
    mixer = _mod("pygame.mixer", init=lambda *a, **k: None, music=music)
    _mod("pygame", mixer=mixer)
    gl = _mod("pyglet.gl")
    gl.__all__ = []
    window = _mod("pyglet.window", key=types.SimpleNamespace())
    _mod("pyglet", gl=gl, window=window)
    _mod("cv2")
    _mod("gym_vecenv")
    _mod("imageio")

    class SummaryWriter(object):
        def __init__(self, *a, **k):
            pass

        def add_scalar(self, *a, **k):
            pass

        def close(self):
            pass

    _mod("tensorboardX", SummaryWriter=SummaryWriter)


def import_reference():
    """Put the stand-ins and /root/reference on the import path; return nothing."""
    if not reference_available():
        raise RuntimeError("reference tree not found at %s (build container only)" % REFERENCE_ROOT)
    _install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


@contextlib.contextmanager
def quiet():
    """The reference prints on every episode end (fortattack.py:208,214,220)."""
    with contextlib.redirect_stdout(io.StringIO()):
        yield


def make_reference_env(num_guards, num_attackers, max_time_steps):
    """Construct the reference env; team sizes other than 5v5 by list truncation.

    The reference hard-codes 5 guards + 5 attackers (fortattack_env_v1.py:18-19).
    For other sizes the recipe of SURVEY.md 8(c) is used: build the 5v5 scenario,
    truncate ``world.agents``, call ``reset_world`` again, wrap in
    ``FortAttackGlobalEnv``.  No reference source is modified.

    Returns (env, rng_doubles_consumed_by_construction).
    """
    import_reference()
    import gym  # the stand-in
    import gym_fortattack  # noqa: F401  (registers 'fortattack-v1')
    from gym_fortattack.fortattack import FortAttackGlobalEnv, make_fortattack_env

    with quiet():
        if num_guards == 5 and num_attackers == 5:
            env = make_fortattack_env(max_time_steps)
            return env, 20
        assert 1 <= num_guards <= 5 and 1 <= num_attackers <= 5
        sc = gym.make("fortattack-v1")
        w = sc.world
        w.agents = w.agents[:num_guards] + w.agents[5:5 + num_attackers]
        w.numGuards, w.numAttackers = num_guards, num_attackers
        w.numAgents = num_guards + num_attackers
        w.max_time_steps = max_time_steps
        sc.reset_world()
        env = FortAttackGlobalEnv(w, sc.reset_world, sc.reward, sc.observation)
    return env, 20 + 2 * (num_guards + num_attackers)
