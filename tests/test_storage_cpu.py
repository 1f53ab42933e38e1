"""CPU: the rollout-buffer layout.  Per-agent views of the joint (T[+1], P, N, ...) tensors
behave exactly like the reference's standalone RolloutStorage (rlcore/storage.py:9-96):
same shapes, same results from insert / after_update / compute_returns, and the reference's
own `.view(-1, ...)` flattening (storage.py:83-90, ppo.py:224-234) still works on them."""
import numpy as np
import pytest
import torch

import collector_oracle as co


@pytest.fixture(scope="module")
def fa():
    import emergent_multiagent_strategies_amd as m
    return m


def test_shapes_and_init_match_reference_storage(fa):
    T, P, N = 16, 5, 6
    joint = fa.JointRolloutStorage(T, P, N)
    ref = co.StorageOracle(T, P)
    for i in range(N):
        v = joint.agent_view(i)
        for k in ("obs", "recurrent_hidden_states", "rewards", "value_preds", "returns", "action_log_probs",
                  "actions", "masks"):
            assert tuple(getattr(v, k).shape) == getattr(ref, k).shape, k
            assert np.array_equal(getattr(v, k).numpy(), getattr(ref, k)), k
        assert v.actions.dtype == torch.int64 and v.obs.dtype == torch.float32
        # the reference flattens with .view(-1, ...): must not raise on the strided window
        assert v.obs[:-1].view(-1, 6).shape == (T * P, 6)
        assert v.actions.view(-1, 1).shape == (T * P, 1)
        assert v.returns[:-1].view(-1, 1).shape == (T * P, 1)
    assert joint.obs[3].view(P * N, 6).shape == (P * N, 6)  # one contiguous row per step for the policy


def test_views_write_through_and_match_oracle(fa):
    T, P, N = 12, 3, 4
    rng = np.random.RandomState(0)
    joint = fa.JointRolloutStorage(T, P, N)
    views = joint.agent_views()
    alone = [fa.RolloutStorage(T, P, (6,), None, 1) for _ in range(N)]
    orc = [co.StorageOracle(T, P) for _ in range(N)]
    for upd in range(2):
        for s in range(T):
            for i in range(N):
                args = [rng.randn(P, 6).astype(np.float32), np.zeros((P, 1), np.float32),
                        rng.randint(0, 8, (P, 1)), rng.randn(P, 1).astype(np.float32),
                        rng.randn(P, 1).astype(np.float32), rng.randn(P, 1).astype(np.float32),
                        (rng.rand(P, 1) > 0.3).astype(np.float32)]
                orc[i].insert(*args)
                for st in (views[i], alone[i]):
                    st.insert(*[torch.from_numpy(np.asarray(a)) for a in args])
        nv = rng.randn(N, P, 1).astype(np.float32)
        for i in range(N):
            orc[i].compute_returns(nv[i], 0.99, 0.95, 0, 5)
            orc[i].compute_returns(nv[i], 0.99, 0.95, 6, T)
            for st in (views[i], alone[i]):
                st.compute_returns(torch.from_numpy(nv[i]), True, 0.99, 0.95, 0, 5)
                st.compute_returns(torch.from_numpy(nv[i]), True, 0.99, 0.95, 6, T)
            for k in ("obs", "rewards", "value_preds", "returns", "masks", "actions", "action_log_probs"):
                assert np.array_equal(getattr(views[i], k).numpy(), getattr(orc[i], k)), (k, i)
                assert np.array_equal(getattr(alone[i], k).numpy(), getattr(orc[i], k)), (k, i)
            assert torch.equal(joint.returns[:, :, i], views[i].returns)  # wrote through
        for i in range(N):
            orc[i].after_update()
            views[i].after_update()
            alone[i].after_update()
            assert np.array_equal(views[i].obs.numpy(), orc[i].obs) and views[i].step == 0
            assert np.array_equal(alone[i].masks.numpy(), orc[i].masks)


def test_feed_forward_generator_on_views(fa):
    T, P, N = 8, 4, 3
    joint = fa.JointRolloutStorage(T, P, N)
    joint.obs.copy_(torch.arange(joint.obs.numel(), dtype=torch.float32).view_as(joint.obs))
    joint.returns.copy_(torch.randn_like(joint.returns))
    v = joint.agent_view(1)
    adv = torch.randn(T, P, 1)
    seen = 0
    for batch in v.feed_forward_generator(adv, num_mini_batch=4):
        obs_b, hid_b, act_b, val_b, ret_b, mask_b, logp_b, adv_b = batch
        assert obs_b.shape == (T * P // 4, 6) and adv_b.shape == (T * P // 4, 1)
        seen += obs_b.shape[0]
    assert seen == T * P
    with pytest.raises(AssertionError):
        next(v.feed_forward_generator(adv, num_mini_batch=T * P + 1))
    with pytest.raises(ValueError):
        v.to("meta")
