"""MI355X-native FortAttack rollout engine (hot path of Emergent-Multiagent-Strategies).

HIP kernels + C ABI in ``csrc/`` (include/fortattack.h); this package is the host-side
mirror of the reference's Python boundary for that path.
"""
from . import _lib  # noqa: F401
from ._lib import FaError  # noqa: F401
from .spaces import Box, Discrete, MASpace  # noqa: F401
from .storage import JointRolloutStorage, RolloutStorage  # noqa: F401
from .env import BatchedFortAttack, FortAttackGlobalEnv, make_fortattack_env  # noqa: F401
from .mpnn import MPNN  # noqa: F401
from .learner import BatchedLearner  # noqa: F401
from .rlagent import JointPPO, Neo  # noqa: F401

__all__ = ["BatchedFortAttack", "FortAttackGlobalEnv", "make_fortattack_env", "JointRolloutStorage",
           "RolloutStorage", "FaError", "Box", "Discrete", "MASpace", "MPNN", "BatchedLearner", "Neo", "JointPPO"]
from . import render  # noqa: F401
