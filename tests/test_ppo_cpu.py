"""CPU: the batched PPO minibatch loss (learner.ppo_losses) against the reference's own
JointPPO.update (rlcore/algo/ppo.py:116-204), live, when the reference tree is present:
same weights, same rollout data, one full-batch minibatch (order independent) -> same three
losses and the same weights after the Adam step."""
import copy

import numpy as np
import pytest
import torch

import ref_harness as rh

pytestmark = pytest.mark.skipif(not rh.reference_available(), reason="reference tree not present")


class _Sp(object):
    shape = (8,)


@pytest.mark.parametrize("clipped", [True, False])   # False: the reference's scalar-MSE branch (ppo.py:178-182)
def test_losses_and_adam_step_match_reference_jointppo(clipped):
    rh.import_reference()
    from mpnn import MPNN as RefMPNN
    from rlcore.algo.ppo import JointPPO
    from rlcore.storage import RolloutStorage as RefStorage
    from emergent_multiagent_strategies_amd.mpnn import MPNN
    from emergent_multiagent_strategies_amd.learner import ppo_losses

    torch.manual_seed(3)
    G, A, T, P = 3, 2, 12, 4
    ref_pol = RefMPNN(action_space=_Sp(), num_agents=G, num_opp_agents=A, num_entities=0, input_size=6,
                      pos_index=2, mask_dist=None, entity_mp=False, policy_layers=1)
    ours = MPNN(num_agents=G, num_opp_agents=A, num_actions=8)
    ours.load_state_dict(copy.deepcopy(ref_pol.state_dict()))
    clip, vcoef, ecoef, lr, gnorm = 0.2, 0.5, 0.01, 1e-3, 0.5
    ppo = JointPPO(ref_pol, clip, 1, 1, vcoef, ecoef, lr=lr, max_grad_norm=gnorm, use_clipped_value_loss=clipped)

    def storage():
        s = RefStorage(T, P, (6,), None, 1)
        s.obs.copy_(torch.randn(T + 1, P, 6))
        s.obs[:, :, 0] = (torch.rand(T + 1, P) > 0.3).float()       # alive flag
        s.actions.copy_(torch.randint(0, 8, (T, P, 1)))
        s.action_log_probs.copy_(-torch.rand(T, P, 1) * 2)
        s.value_preds.copy_(torch.randn(T + 1, P, 1))
        s.returns.copy_(torch.randn(T + 1, P, 1))
        return s

    own_st, opp_st = [storage() for _ in range(G)], [storage() for _ in range(A)]
    # ---- ours, on the same data laid out env-major (T*P, n, .) ----
    cat = lambda sts, k, sl: torch.stack([getattr(s, k)[sl].reshape(T * P, -1) for s in sts], 1)
    own_obs, opp_obs = cat(own_st, "obs", slice(0, T)), cat(opp_st, "obs", slice(0, T))
    acts, olp = cat(own_st, "actions", slice(None)), cat(own_st, "action_log_probs", slice(None))
    vps, rets = cat(own_st, "value_preds", slice(0, T)), cat(own_st, "returns", slice(0, T))
    advs = []
    for s in own_st:                                                 # ppo.py:121-123
        a = s.returns[:-1] - s.value_preds[:-1]
        advs.append(((a - a.mean()) / (a.std() + 1e-5)).reshape(T * P, 1))
    adv = torch.stack(advs, 1)
    opt = torch.optim.Adam(ours.parameters(), lr=lr)
    vl, al, ent = ppo_losses(ours, own_obs, opp_obs, acts, vps, rets, olp, adv, clip, clipped)
    opt.zero_grad()
    (vl * vcoef + al - ent * ecoef).backward()
    torch.nn.utils.clip_grad_norm_(ours.parameters(), gnorm)
    opt.step()
    # ---- reference ----
    with rh.quiet():
        rvl, ral, rent = ppo.update(own_st, opp_st)
    assert abs(float(vl) - rvl) < 1e-5 and abs(float(al) - ral) < 1e-5 and abs(float(ent) - rent) < 1e-5
    rsd, osd = ref_pol.state_dict(), ours.state_dict()
    for k in rsd:
        assert (rsd[k] - osd[k]).abs().max() < 2e-5, k
    assert any((rsd[k] - v).abs().max() > 1e-4 for k, v in MPNN(num_agents=G, num_opp_agents=A, num_actions=8)
               .state_dict().items())  # sanity: weights did move / are not trivially equal


def test_all_dead_minibatch_gives_zero_losses():
    from emergent_multiagent_strategies_amd.mpnn import MPNN
    from emergent_multiagent_strategies_amd.learner import ppo_losses
    torch.manual_seed(0)
    pol = MPNN(num_agents=2, num_opp_agents=2, num_actions=8)
    own = torch.randn(5, 2, 6)
    own[:, :, 0] = 0                                                   # every agent dead
    z = torch.zeros(5, 2, 1)
    vl, al, ent = ppo_losses(pol, own, torch.randn(5, 2, 6), z.long(), z, z, z, z, 0.2)
    assert float(vl) == 0 and float(al) == 0 and float(ent) == 0 and torch.isfinite(vl)


def test_reference_sampling_draws_the_reference_minibatches():
    """BatchedLearner(reference_sampling=True)'s index sets == the ones magent_feed_forward_generator (ppo.py:207-246) draws
    from the same torch seed: rollouts whose observation column 0 holds the flat sample index t * P + p make the reference's
    generator reveal its indices."""
    rh.import_reference()
    from rlcore.algo.ppo import magent_feed_forward_generator
    from rlcore.storage import RolloutStorage as RefStorage
    from emergent_multiagent_strategies_amd.learner import BatchedLearner
    T, P, nmb = 6, 5, 4                                      # batch 30, minibatches of 7, 7, 7, 7, 2 (drop_last=False)
    st = RefStorage(T, P, (6,), None, 1)
    st.obs[:-1, :, 0] = torch.arange(T * P, dtype=torch.float32).view(T, P)
    adv = [torch.zeros(T, P, 1)]
    torch.manual_seed(123)
    want = [[b[0][:, 0].long() for b in magent_feed_forward_generator([st], [st], adv, nmb)] for _ in range(3)]
    L = BatchedLearner.__new__(BatchedLearner)               # the sampler needs T, E, num_mini_batch and a device only
    L.T, L.E, L.num_mini_batch, L.device = T, P, nmb, torch.device("cpu")
    torch.manual_seed(123)
    sampler = L._reference_sampler()
    got = [sampler(ep) for ep in range(3)]
    assert [len(e) for e in got] == [len(e) for e in want] == [5, 5, 5]
    for ge, we in zip(got, want):
        for g, w in zip(ge, we):
            assert torch.equal(g, w)
