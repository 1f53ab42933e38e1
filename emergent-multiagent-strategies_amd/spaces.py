"""Space metadata the reference exposes on the env (gym / malib are not dependencies).

Only what callers read: ``Discrete.n``, ``Box.shape`` (MPNN takes ``action_space.shape[0]``
as its number of actions, mpnn.py:73; learner.py:47-62 reads ``observation_space[i].shape[0]``
and ``action_spaces[i]``).
"""
import numpy as np


class Discrete(object):
    def __init__(self, n):
        self.n, self.shape, self.dtype = int(n), (), np.int64

    def __repr__(self):
        return "Discrete(%d)" % self.n


class Box(object):
    def __init__(self, low, high, shape, dtype=np.float32):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

    def __repr__(self):
        return "Box%s" % (self.shape,)


class MASpace(object):
    """malib/spaces/space.py MASpace: a tuple of per-agent spaces."""

    def __init__(self, spaces):
        self.spaces, self.agent_num = tuple(spaces), len(spaces)

    def __getitem__(self, i):
        return self.spaces[i]

    def __len__(self):
        return self.agent_num

    @property
    def shape(self):
        return tuple(s.shape for s in self.spaces)
