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
    """rlcore/storage.py:9 -- same constructor, fields, shapes and methods.

    Standalone (reference constructor) it owns contiguous tensors; as ``view_of`` a
    JointRolloutStorage its tensors are strided windows and every method writes through.
    """

    def __init__(self, num_steps, num_processes, obs_shape, action_space, recurrent_hidden_state_size):
        self.obs = torch.zeros(num_steps + 1, num_processes, *obs_shape)
        self.recurrent_hidden_states = torch.zeros(num_steps + 1, num_processes, recurrent_hidden_state_size)
        self.rewards = torch.zeros(num_steps, num_processes, 1)
        self.value_preds = torch.zeros(num_steps + 1, num_processes, 1)
        self.returns = torch.zeros(num_steps + 1, num_processes, 1)
        self.action_log_probs = torch.zeros(num_steps, num_processes, 1)
        self.actions = torch.zeros(num_steps, num_processes, 1).long()
        self.masks = torch.ones(num_steps + 1, num_processes, 1)
        self.num_steps = num_steps
        self.step = 0
        self._joint = None

    @classmethod
    def view_of(cls, joint, i):
        self = cls.__new__(cls)
        for k in JointRolloutStorage.FIELDS:
            setattr(self, k, getattr(joint, k)[:, :, i])
        self.num_steps, self.step, self._joint = joint.num_steps, 0, joint
        return self

    def to(self, device):  # storage.py:23-31
        if self._joint is not None:
            if torch.device(device) != self.obs.device:
                raise ValueError("a joint-storage view lives on the joint tensors' device")
            return
        for k in JointRolloutStorage.FIELDS:
            setattr(self, k, getattr(self, k).to(device))

    def insert(self, obs, recurrent_hidden_states, actions, action_log_probs, value_preds, rewards, masks):
        s = self.step  # storage.py:33-43
        self.obs[s + 1].copy_(obs)
        self.recurrent_hidden_states[s + 1].copy_(recurrent_hidden_states)
        self.actions[s].copy_(actions)
        self.action_log_probs[s].copy_(action_log_probs)
        self.value_preds[s].copy_(value_preds)
        self.rewards[s].copy_(rewards)
        self.masks[s + 1].copy_(masks)
        self.step = (s + 1) % self.num_steps

    def reset(self):  # storage.py:48-49
        self.step = 0

    def after_update(self):  # storage.py:51-56
        self.obs[0].copy_(self.obs[-1])
        self.obs[1:] = 0
        self.recurrent_hidden_states[0].copy_(self.recurrent_hidden_states[-1])
        self.masks[0].copy_(self.masks[-1])
        self.step = 0

    def compute_returns(self, next_value, use_gae, gamma, tau, start_pt, end_pt):
        """storage.py:59-71, torch ops (host-sequenced).  The batched path is fa_gae."""
        if use_gae:
            self.value_preds[end_pt] = next_value
            gae = 0
            for step in reversed(range(start_pt, end_pt)):
                delta = self.rewards[step] + gamma * self.value_preds[step + 1] * self.masks[step + 1] \
                    - self.value_preds[step]
                gae = delta + gamma * tau * self.masks[step + 1] * gae
                self.returns[step] = gae + self.value_preds[step]
        else:
            self.returns[end_pt] = next_value
            for step in reversed(range(start_pt, end_pt)):
                self.returns[step] = self.returns[step + 1] * gamma * self.masks[step + 1] + self.rewards[step]

    def feed_forward_generator(self, advantages, num_mini_batch, sampler=None):
        """storage.py:74-96."""
        from torch.utils.data.sampler import BatchSampler, SubsetRandomSampler
        num_steps, num_processes = self.rewards.size()[0:2]
        batch_size = num_processes * num_steps
        assert batch_size >= num_mini_batch, (
            "PPO requires the number of processes ({}) * number of steps ({}) = {} to be greater than "
            "or equal to the number of PPO mini batches ({}).".format(
                num_processes, num_steps, batch_size, num_mini_batch))
        mini_batch_size = batch_size // num_mini_batch
        if sampler is None:
            sampler = BatchSampler(SubsetRandomSampler(range(batch_size)), mini_batch_size, drop_last=False)
        for indices in sampler:
            yield (self.obs[:-1].view(-1, *self.obs.size()[2:])[indices],
                   self.recurrent_hidden_states[:-1].view(-1, self.recurrent_hidden_states.size(-1))[indices],
                   self.actions.view(-1, self.actions.size(-1))[indices],
                   self.value_preds[:-1].view(-1, 1)[indices],
                   self.returns[:-1].view(-1, 1)[indices],
                   self.masks[:-1].view(-1, 1)[indices],
                   self.action_log_probs.view(-1, 1)[indices],
                   advantages.view(-1, 1)[indices])
