"""The reference's per-agent holder ``Neo`` (rlagent.py:7-51) and its team trainer ``JointPPO``
(rlcore/algo/ppo.py:98-204) over this package's RolloutStorage and joint_ppo_update: the same
constructor arguments and attributes, and every method the reference's Learner calls (the three it never
calls -- act, before_update, update: dead code there -- exist and say so), so that code written against the reference's
``Learner`` internals (``agent.rollouts``, ``agent.actor_critic``, ``trainer.update(rollouts_list,
opp_rollouts_list)``) runs on the repo's classes.  The batched trainer (learner.BatchedLearner) does
not go through these -- it works on the joint tensors directly; a ``Neo`` built with
``rollouts=joint.agent_view(i)`` is agent i's window onto them.
"""
import torch

from .learner import joint_ppo_update
from .storage import RolloutStorage


class JointPPO(object):
    """rlcore/algo/ppo.py:98-204: one optimizer for the policy shared by a team."""

    def __init__(self, actor_critic, clip_param, ppo_epoch, num_mini_batch, value_loss_coef, entropy_coef, lr=None,
                 max_grad_norm=None, use_clipped_value_loss=False):
        self.actor_critic = actor_critic
        self.clip_param, self.ppo_epoch, self.num_mini_batch = clip_param, ppo_epoch, num_mini_batch
        self.value_loss_coef, self.entropy_coef = value_loss_coef, entropy_coef
        self.max_grad_norm, self.use_clipped_value_loss = max_grad_norm, use_clipped_value_loss
        self.optimizer = torch.optim.Adam(actor_critic.parameters(), lr=lr)   # ppo.py:114

    def update(self, rollouts_list, opp_rollouts_list, sampler=None, group=None):
        """-> (value_loss, action_loss, dist_entropy) floats, as the reference."""
        n, m = len(rollouts_list), len(opp_rollouts_list)
        flat = lambda t: t.reshape(-1, t.shape[-1])
        adv = []
        for r in rollouts_list:                                            # ppo.py:121-124
            a = r.returns[:-1] - r.value_preds[:-1]
            adv.append((a - a.mean()) / (a.std() + 1e-5))
        stack = lambda ts: torch.stack([flat(t) for t in ts], 1)           # (T*P, agents, .)
        rows = (stack([r.obs[:-1] for r in list(rollouts_list) + list(opp_rollouts_list)]),
                stack([r.actions for r in rollouts_list]), stack([r.value_preds[:-1] for r in rollouts_list]),
                stack([r.returns[:-1] for r in rollouts_list]), stack([r.action_log_probs for r in rollouts_list]),
                stack(adv))
        out = joint_ppo_update(self.actor_critic, self.optimizer, slice(0, n), slice(n, n + m), rows, self.clip_param,
                               self.ppo_epoch, self.num_mini_batch, self.value_loss_coef, self.entropy_coef,
                               self.max_grad_norm, self.use_clipped_value_loss, group, sampler)
        return tuple(float(v) for v in out)


class Neo(object):
    """rlagent.py:7-51.  ``args`` needs num_steps, num_processes, gamma, tau (+ the PPO fields when
    ``update`` is used)."""

    def __init__(self, args, policy, obs_shape, action_space, rollouts=None):
        self.obs_shape = obs_shape
        self.action_space = action_space
        self.actor_critic = policy
        self.rollouts = rollouts if rollouts is not None else RolloutStorage(
            args.num_steps, args.num_processes, self.obs_shape, self.action_space, recurrent_hidden_state_size=1)
        self.args = args
        self.trainer = None            # the reference builds a single-agent PPO here that its Learner never uses
        self.alive = True

    def load_model(self, policy_state):                          # :20-21
        self.actor_critic.load_state_dict(policy_state)

    def initialize_obs(self, obs):                                # :23-26
        self.rollouts.reset()
        self.rollouts.obs[0].copy_(obs)

    def initialize_new_episode(self, step, obs, masks):           # :28-31
        self.rollouts.obs[step].copy_(obs)
        self.rollouts.masks[step].copy_(masks)

    def update_rollout(self, obs, reward, mask):                  # :33-34
        self.rollouts.insert(obs, self.states, self.action, self.action_log_prob, self.value, reward, mask)

    def act(self, step, deterministic=False):                     # :36-39
        """Never called by the reference's Learner (learner.py:160 runs the TEAM policy on the concatenated
        observations) and not callable there either: it passes (obs, hidden, masks) where MPNN.act expects
        (inp, state, oppInp).  A single agent's holder has no opponents' observations to give the MPNN."""
        raise NotImplementedError("Neo.act: the MPNN needs the opponents' observations; act through the team policy "
                                  "(BatchedLearner.step, or actor_critic.act(own, opp)) as learner.py:143-172 does")

    def wrap_horizon(self, next_value, start_pt, end_pt):         # :41-42
        self.rollouts.compute_returns(next_value, True, self.args.gamma, self.args.tau, start_pt, end_pt)

    def before_update(self):                                      # :44-45 (storage.py:45 has it commented out)
        fn = getattr(self.rollouts, "before_update", None)
        if fn is not None:
            fn()

    def after_update(self):                                       # :47-48
        self.rollouts.after_update()

    def update(self):                                             # :50-51
        """The reference's single-agent PPO over one agent's rollouts, unused by its Learner (teams are trained
        by JointPPO, learner.py:175-188).  Set `trainer` to an object with update(rollouts) to use it."""
        if self.trainer is None:
            raise NotImplementedError("Neo.update: no single-agent trainer; teams are trained by JointPPO.update "
                                      "(rlagent.JointPPO / BatchedLearner.update)")
        return self.trainer.update(self.rollouts)
