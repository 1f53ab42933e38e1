"""Rollout buffers with the reference's RolloutStorage layout (rlcore/storage.py:9-31).

``JointRolloutStorage`` owns one set of joint device tensors (T[+1], P, N, ...): the step
kernel writes a whole (P, N, ...) row per launch, fully coalesced.  ``RolloutStorage`` is
agent i's reference-shaped window onto them: ``joint.obs[:, :, i]`` has shape (T+1, P, 6)
and strides that still satisfy the reference's own ``.view(-1, ...)`` calls
(storage.py:83-90, rlcore/algo/ppo.py:224-234) because stride_T == P * stride_P.
"""
import torch


class JointRolloutStorage(object):
    FIELDS = ("obs", "recurrent_hidden_states", "rewards", "value_preds", "returns",
              "action_log_probs", "actions", "masks")

    def __init__(self, num_steps, num_processes, num_agents, obs_dim=6, device="cpu"):
        T, P, N = num_steps, num_processes, num_agents
        z = lambda *s, **k: torch.zeros(*s, device=device, **k)
        self.obs = z(T + 1, P, N, obs_dim)                 # storage.py:11
        self.recurrent_hidden_states = z(T + 1, P, N, 1)   # :12 (size 1, rlagent.py:14)
        self.rewards = z(T, P, N, 1)                       # :13
        self.value_preds = z(T + 1, P, N, 1)               # :14
        self.returns = z(T + 1, P, N, 1)                   # :15
        self.action_log_probs = z(T, P, N, 1)              # :16
        self.actions = z(T, P, N, 1, dtype=torch.int64)    # :17-18
        self.masks = torch.ones(T + 1, P, N, 1, device=device)  # :19
        self.done = z(T, P, dtype=torch.uint8)             # per-env end_pts as flags
        self.num_steps, self.num_processes, self.num_agents = T, P, N
        self.step = 0

    def agent_view(self, i):
        return RolloutStorage.view_of(self, i)

    def agent_views(self):
        return [self.agent_view(i) for i in range(self.num_agents)]


class RolloutStorage(object):
    """rlcore/storage.py:9 -- same constructor signature, field names, shapes and methods.

    Built standalone (the reference's constructor) it owns contiguous tensors; built with
    ``view_of`` it is agent i's window onto a JointRolloutStorage and every method writes
    through to the joint tensors.
    """
    # field -> (rows beyond num_steps, trailing shape key, dtype, fill)   (storage.py:11-19)
    _SPEC = (("obs", 1, "obs", torch.float32, 0.0), ("recurrent_hidden_states", 1, "hid", torch.float32, 0.0),
             ("rewards", 0, "one", torch.float32, 0.0), ("value_preds", 1, "one", torch.float32, 0.0),
             ("returns", 1, "one", torch.float32, 0.0), ("action_log_probs", 0, "one", torch.float32, 0.0),
             ("actions", 0, "one", torch.int64, 0), ("masks", 1, "one", torch.float32, 1.0))

    def __init__(self, num_steps, num_processes, obs_shape, action_space, recurrent_hidden_state_size):
        tail = {"obs": tuple(obs_shape), "hid": (recurrent_hidden_state_size,), "one": (1,)}
        for name, extra, key, dtype, fill in self._SPEC:
            setattr(self, name, torch.full((num_steps + extra, num_processes) + tail[key], fill, dtype=dtype))
        self.num_steps, self.step, self._joint = num_steps, 0, None

    @classmethod
    def view_of(cls, joint, i):
        self = cls.__new__(cls)
        for name in JointRolloutStorage.FIELDS:
            setattr(self, name, getattr(joint, name)[:, :, i])
        self.num_steps, self.step, self._joint = joint.num_steps, 0, joint
        return self

    def to(self, device):  # storage.py:23-31
        if self._joint is not None:
            if torch.device(device) != self.obs.device:
                raise ValueError("a joint-storage view lives on the joint tensors' device")
            return
        for name in JointRolloutStorage.FIELDS:
            setattr(self, name, getattr(self, name).to(device))

    def insert(self, obs, recurrent_hidden_states, actions, action_log_probs, value_preds, rewards, masks):
        """storage.py:33-43: rows s+1 get what the step produced, rows s what the policy chose."""
        s = self.step
        for dst, row, src in ((self.obs, s + 1, obs), (self.recurrent_hidden_states, s + 1, recurrent_hidden_states),
                              (self.actions, s, actions), (self.action_log_probs, s, action_log_probs),
                              (self.value_preds, s, value_preds), (self.rewards, s, rewards),
                              (self.masks, s + 1, masks)):
            dst[row].copy_(src)
        self.step = (s + 1) % self.num_steps

    def reset(self):  # storage.py:48-49
        self.step = 0

    def after_update(self):  # storage.py:51-56
        for t in (self.obs, self.recurrent_hidden_states, self.masks):
            t[0].copy_(t[-1])
        self.obs[1:] = 0
        self.step = 0

    def compute_returns(self, next_value, use_gae, gamma, tau, start_pt, end_pt):
        """storage.py:59-71 on torch tensors, host-sequenced (the batched path is fa_gae)."""
        r, v, m = self.rewards, self.value_preds, self.masks
        if not use_gae:
            self.returns[end_pt] = next_value
            for t in range(end_pt - 1, start_pt - 1, -1):
                self.returns[t] = self.returns[t + 1] * gamma * m[t + 1] + r[t]
            return
        v[end_pt] = next_value
        gae = 0
        for t in range(end_pt - 1, start_pt - 1, -1):
            delta = r[t] + gamma * v[t + 1] * m[t + 1] - v[t]
            gae = delta + gamma * tau * m[t + 1] * gae
            self.returns[t] = gae + v[t]

    def feed_forward_generator(self, advantages, num_mini_batch, sampler=None):
        """storage.py:74-96: random minibatches of flattened (T*P) samples, in the reference's
        tuple order (obs, hidden, actions, value_preds, returns, masks, old log-probs, adv)."""
        from torch.utils.data.sampler import BatchSampler, SubsetRandomSampler
        T, P = self.rewards.shape[:2]
        batch_size = T * P
        assert batch_size >= num_mini_batch, (
            "PPO requires the number of processes ({}) * number of steps ({}) = {} to be greater than "
            "or equal to the number of PPO mini batches ({}).".format(P, T, batch_size, num_mini_batch))
        if sampler is None:
            sampler = BatchSampler(SubsetRandomSampler(range(batch_size)), batch_size // num_mini_batch,
                                   drop_last=False)
        flat = [self.obs[:-1], self.recurrent_hidden_states[:-1], self.actions, self.value_preds[:-1],
                self.returns[:-1], self.masks[:-1], self.action_log_probs, advantages]
        flat = [t.view(batch_size, *t.shape[2:]) for t in flat]
        for indices in sampler:
            yield tuple(t[indices] for t in flat)
