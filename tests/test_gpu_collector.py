"""GPU parity of the rollout collector: fused storage writes of fa_collect_step, fa_gae,
fa_adv_stats / fa_adv_normalize, fa_after_update -- against the golden capture of the
reference's own Learner / RolloutStorage / JointPPO (tests/golden/collector_3v3.npz) and
against the numpy collector oracle on batched random data.

float32 storage tensors: bit exact.  Normalised advantages: 2e-6 (torch's float32
mean/std reduction order is not restated; the kernels accumulate in fp64).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fa():
    import emergent_multiagent_strategies_amd as m
    assert torch.cuda.is_available()
    m._lib.load()
    return m


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("name", ["collector_3v3",
                                  # the reference's own training shape (arguments.py:23 --num-steps 1000; marlsave/tmp_2/params.json:
                                  # 5v5, 100-step episodes), the published ep1240 policies acting, two updates, 105 episode ends
                                  "collector_5v5_T1000"])
def test_collector_golden(fa, golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    compact = "after_obs" not in g.files
    G, A, max_t, T, n_upd, seed, skip = [int(v) for v in g["meta"]]
    N = G + A
    gamma, tau = [float(v) for v in g["gamma_tau"]]
    eng = fa.BatchedFortAttack(1, G, A, max_t, base_seed=seed, skip_doubles=skip)
    st = fa.JointRolloutStorage(T, 1, N, device="cuda")
    eng.bind_storage(st)
    views = st.agent_views()
    assert views[2].obs.shape == (T + 1, 1, 6) and views[2].actions.shape == (T, 1, 1)
    eng.collect_reset()                                    # env.reset + initialize_obs
    assert np.array_equal(st.obs[0, 0].cpu().numpy(), g["obs0"].astype(np.float32))
    for j in range(n_upd):
        # policy-side fields come from the capture (the reference's MPNN sampled them)
        st.actions[:, 0, :, 0] = _t(g["actions"][j].astype(np.int64))
        st.action_log_probs[:, 0, :, :] = _t(g["action_log_probs"][j][:, :, 0].transpose(1, 0, 2))
        vp = _t(g["value_preds"][j][:, :, 0].transpose(1, 0, 2))       # (T+1, N, 1)
        st.value_preds[:T, 0] = vp[:T]
        for s in range(T):
            eng.collect_step(s, auto_reset=True)
        assert np.array_equal(st.done[:, 0].cpu().numpy(), g["done"][j])
        end_pts = [int(v) for v in g["end_pts"][j] if v >= 0]
        st.value_preds[T, 0] = _t(g["next_values"][j][:, len(end_pts) - 1])[:, None]
        eng.gae(gamma, tau)
        for i in range(N):
            v = views[i]
            for k in ("obs", "rewards", "masks", "returns", "value_preds"):
                assert np.array_equal(getattr(v, k).cpu().numpy(), g[k][j, i]), (k, j, i)
        # advantage statistics + normalisation (ppo.py:121-124)
        s0 = eng.adv_stats(0)
        mean = (s0[:, 1] / s0[:, 0]).contiguous()
        s1 = eng.adv_stats(1, mean=mean, out=s0.clone())
        std = torch.sqrt(s1[:, 2] / (s1[:, 0] - 1)).contiguous()
        adv = eng.adv_normalize(mean, std)
        for i in range(N):
            assert np.abs(adv[:, :, i].cpu().numpy() - g["adv"][j, i]).max() < 2e-6
        eng.after_update()
        for i in range(N):
            if compact:
                assert np.array_equal(views[i].obs[0].cpu().numpy(), g["after_obs_row0"][j, i])
                assert np.array_equal(views[i].masks[0].cpu().numpy(), g["after_masks_row0"][j, i])
                assert float(views[i].obs[1:].abs().sum()) == g["after_rest_abs_sum"][j, i] == 0.0
            else:
                assert np.array_equal(views[i].obs.cpu().numpy(), g["after_obs"][j, i])
                assert np.array_equal(views[i].masks.cpu().numpy(), g["after_masks"][j, i])


@pytest.mark.parametrize("E,G,A,T", [(300, 3, 3, 64), (64, 5, 5, 128),
                                     # the three GAE kernels of fa_launch_gae (csrc/fa_collect.hip): <= 131 072 columns
                                     # fa_gae_coop_kernel (above); between, or misaligned -> fa_gae_kernel; >= 262 144
                                     # columns in multiples of four -> fa_gae4_kernel.  T = 40: two 16-step chunks + a tail
                                     (32768, 3, 3, 40),        # 196 608 columns: fa_gae_kernel
                                     (49152, 3, 3, 40),        # 294 912 columns: fa_gae4_kernel
                                     (26215, 5, 5, 19),        # 262 150 columns, not a multiple of 4: fa_gae_kernel
                                     (37, 2, 5, 33)])          # N = 7: the non-vectorised statistics kernels
def test_gae_and_stats_vs_numpy_oracle(fa, E, G, A, T):
    import collector_oracle as co
    N = G + A
    rng = np.random.RandomState(E)
    eng = fa.BatchedFortAttack(E, G, A, 50)
    st = fa.JointRolloutStorage(T, E, N, device="cuda")
    eng.bind_storage(st)
    rewards = rng.randn(T, E, N, 1).astype(np.float32)
    values = rng.randn(T + 1, E, N, 1).astype(np.float32)
    masks = (rng.rand(T + 1, E, N, 1) > 0.2).astype(np.float32)
    stale = rng.randn(T + 1, E, N, 1).astype(np.float32)   # previous update's returns (quirk Q7)
    done = (rng.rand(T, E) < 0.05).astype(np.uint8)
    for k, v in (("rewards", rewards), ("value_preds", values), ("masks", masks), ("returns", stale),
                 ("done", done)):
        getattr(st, k).copy_(_t(v))
    eng.gae(0.99, 0.95)
    ep_start = np.zeros((T, E), bool)
    ep_start[1:] = done[:-1] != 0
    want = stale.copy()
    for i in range(N):
        co.gae_single_pass(rewards[:, :, i], values[:, :, i], masks[:, :, i], want[:, :, i], ep_start, 0.99, 0.95)
    got = st.returns.cpu().numpy()
    assert np.array_equal(got, want)
    assert np.array_equal(got[T], stale[T])                 # row T untouched
    s0 = eng.adv_stats(0)
    mean = (s0[:, 1] / s0[:, 0]).contiguous()
    s1 = eng.adv_stats(1, mean=mean, out=s0.clone()).cpu().numpy()
    for i in range(N):
        n, sm, ssd = co.adv_moments(want[:, :, i], values[:, :, i])
        assert s1[i, 0] == n and abs(s1[i, 1] - sm) <= 1e-9 * max(1, abs(sm)) and abs(s1[i, 2] - ssd) <= 1e-9 * ssd
    std = torch.sqrt(torch.from_numpy(s1[:, 2] / (s1[:, 0] - 1))).cuda().contiguous()
    adv = eng.adv_normalize(mean, std).cpu().numpy()
    for i in range(N):
        assert np.abs(adv[:, :, i] - co.normalized_advantages(want[:, :, i], values[:, :, i])).max() < 2e-6
    m2, s2 = eng.adv_mean_std()                              # single-GPU fused form
    assert torch.equal(m2, mean) and torch.allclose(s2, std, rtol=1e-14, atol=0)
    # run-to-run determinism of the two-stage reduction
    assert torch.equal(eng.adv_stats(0), s0)


def test_collect_step_matches_plain_step(fa):
    """The storage-writing launch and the plain fa_step launch are the same kernel."""
    E, G, A, T = 200, 3, 3, 40
    N = G + A
    a_eng = fa.BatchedFortAttack(E, G, A, 15, base_seed=5)
    b_eng = fa.BatchedFortAttack(E, G, A, 15, base_seed=5)
    st = fa.JointRolloutStorage(T, E, N, device="cuda")
    a_eng.bind_storage(st)
    a_eng.collect_reset()
    ob = b_eng.reset()
    assert torch.equal(st.obs[0], ob) and bool((st.masks[0] == 1).all())
    gen = torch.Generator(device="cuda").manual_seed(1)
    for s in range(T):
        act = torch.randint(0, 8, (E, N), device="cuda", generator=gen)
        st.actions[s, :, :, 0] = act
        a_eng.collect_step(s)
        o = b_eng.step(act)
        assert torch.equal(st.obs[s + 1], o["obs_f32"])
        assert torch.equal(st.rewards[s, :, :, 0], o["reward_f32"])
        assert torch.equal(st.masks[s + 1, :, :, 0], o["mask_f32"])
        assert torch.equal(st.done[s], o["done"])
    # lagged masks (quirk Q2): masks[s+1] == obs[s][..., 0]; 1 where the env was reset at step s
    want = torch.where(st.done[:, :, None] != 0, torch.ones_like(st.masks[1:, :, :, 0]), st.obs[:-1, :, :, 0])
    assert torch.equal(st.masks[1:, :, :, 0], want)


@pytest.mark.parametrize("G,A,E,T,chunk,rng,max_t", [(3, 3, 500, 64, 64, "mt19937", 25), (5, 5, 100, 48, 16, "mt19937", 25),
                                                      (3, 3, 4096, 128, 128, "mt19937", 25),
                                                      (3, 3, 300, 64, 32, "philox", 25),   # counter-based reset stream
                                                      (5, 5, 50, 40, 8, "philox", 1),      # ... a reset every step
                                                      (3, 3, 200, 48, 12, "mt19937", 3)])
def test_fused_rollout_equals_per_step_launches(fa, G, A, E, T, chunk, rng, max_t):
    """fa_collect_rollout (K env-steps in one launch, state in registers; the pipelined kernel for
    K >= 8) is bit-identical to K fa_collect_step launches (the classic kernel) -- storage rows,
    world state and RNG stream position."""
    N = G + A
    gen = torch.Generator(device="cuda").manual_seed(3)
    acts = torch.randint(0, 8, (T, E, N, 1), device="cuda", generator=gen)
    res = []
    for fused in (False, True):
        eng = fa.BatchedFortAttack(E, G, A, max_t, base_seed=99, rng=rng)
        st = fa.JointRolloutStorage(T, E, N, device="cuda")
        eng.bind_storage(st)
        eng.collect_reset()
        st.actions.copy_(acts)
        if fused:
            for s0 in range(0, T, chunk):
                eng.collect_rollout(s0, chunk)
        else:
            for s in range(T):
                eng.collect_step(s)
        # the next reset must draw the same positions: reset once more and compare the observation
        nxt = eng.rng_peek(E - 1, 4) if rng == "mt19937" else eng.reset().cpu().numpy()
        res.append((st, eng.get_state() if rng == "mt19937" else None, nxt))
    (sa, xa, ra), (sb, xb, rb) = res
    for k in ("obs", "rewards", "masks", "done"):
        assert torch.equal(getattr(sa, k), getattr(sb, k)), k
    if xa is not None:
        for k in xa:
            assert np.array_equal(xa[k], xb[k], equal_nan=True), k
    assert np.array_equal(ra, rb)
    assert int(sa.done.sum()) > 0


@pytest.mark.parametrize("E,G,A,T,shift", [(300, 3, 3, 64, 0.0), (64, 5, 5, 128, 0.0), (4096, 3, 3, 128, 0.0),
                                             (37, 2, 5, 33, 0.0),           # ragged: 259 columns, N = 7
                                             (32768, 3, 3, 24, 0.0),        # fa_gae_kernel (196 608 columns)
                                             (65536, 3, 3, 24, 0.0),        # fa_gae4_kernel (393 216 columns)
                                             (4097, 3, 3, 9, 0.0),          # odd row count: fa_adv_onepass_kernel (non-vec)
                                             (500, 3, 3, 48, 1000.0)])      # |mean| >> std: no cancellation
def test_gae_moments_one_pass_equals_gae_plus_two_pass(fa, E, G, A, T, shift):
    """fa_gae_moments (GAE, then one-pass advantage moments) == fa_gae followed
    by the two-pass statistics: returns bit for bit, moments to fp64 rounding, run-to-run identical."""
    import collector_oracle as co
    N = G + A
    rng = np.random.RandomState(E + T)
    eng = fa.BatchedFortAttack(E, G, A, 50)
    st = fa.JointRolloutStorage(T, E, N, device="cuda")
    eng.bind_storage(st)
    data = dict(rewards=rng.randn(T, E, N, 1).astype(np.float32),
                value_preds=(rng.randn(T + 1, E, N, 1) - shift).astype(np.float32),
                masks=(rng.rand(T + 1, E, N, 1) > 0.2).astype(np.float32),
                returns=rng.randn(T + 1, E, N, 1).astype(np.float32),   # stale entries of the previous update (Q7)
                done=(rng.rand(T, E) < 0.05).astype(np.uint8))

    def load():
        for k, v in data.items():
            getattr(st, k).copy_(_t(v))

    load()
    eng.gae(0.99, 0.95)
    ret_ref = st.returns.clone()
    mom_ref = eng.adv_moments().clone()
    mean_ref, std_ref = [x.clone() for x in eng.adv_mean_std()]
    load()
    mom, mean, std = [x.clone() for x in eng.gae_moments(0.99, 0.95)]
    assert torch.equal(st.returns, ret_ref)
    assert torch.equal(mom[:, 0], mom_ref[:, 0]) and float(mom[0, 0]) == T * E
    assert torch.allclose(mean, mean_ref, rtol=1e-12, atol=1e-13)
    assert torch.allclose(std, std_ref, rtol=1e-12, atol=0)
    assert torch.allclose(mom[:, 2], mom_ref[:, 2], rtol=1e-12, atol=0)
    for i in range(N):                                                    # and against numpy in fp64
        a = (ret_ref[:-1, :, i].cpu().numpy() - data["value_preds"][:-1, :, i]).astype(np.float32).astype(np.float64)
        assert abs(float(mean[i]) - a.mean()) <= 1e-12 * max(1.0, abs(a.mean()))
        assert abs(float(std[i]) - a.std(ddof=1)) <= 1e-11 * a.std(ddof=1)
    for _ in range(3):
        load()
        m2, mean2, std2 = eng.gae_moments(0.99, 0.95)
        assert torch.equal(m2, mom) and torch.equal(mean2, mean) and torch.equal(std2, std)   # reproducible


def test_moments_and_merge_equal_global_two_pass(fa):
    """fa_adv_moments + fa_adv_merge (the one-collective multi-GPU form): three 'ranks' worth of
    data merged on the device == the two-pass statistics of the concatenation."""
    import collector_oracle as co
    from emergent_multiagent_strategies_amd.dist import merge_moments
    E, G, A, T, W = 128, 3, 3, 32, 3
    N = G + A
    rng = np.random.RandomState(5)
    eng = fa.BatchedFortAttack(E, G, A, 50)
    st = fa.JointRolloutStorage(T, E, N, device="cuda")
    eng.bind_storage(st)
    parts, gathered = [], torch.zeros((W, N, 3), dtype=torch.float64, device="cuda")
    for r in range(W):
        ret = (rng.randn(T + 1, E, N, 1) * (1 + r) + 0.3 * r).astype(np.float32)
        val = rng.randn(T + 1, E, N, 1).astype(np.float32)
        st.returns.copy_(_t(ret)); st.value_preds.copy_(_t(val))
        eng.adv_moments(out=gathered[r])
        m, s_ = eng.adv_mean_std()
        assert torch.allclose(gathered[r, :, 1], m, rtol=0, atol=1e-15)
        assert torch.allclose(torch.sqrt(gathered[r, :, 2] / (gathered[r, :, 0] - 1)), s_, rtol=1e-14, atol=0)
        parts.append((ret[:-1] - val[:-1]).astype(np.float32).astype(np.float64))
    mean, std = eng.adv_merge(gathered)
    allp = np.concatenate(parts, axis=1)                       # (T, W*E, N, 1)
    for i in range(N):
        assert abs(float(mean[i]) - allp[:, :, i].mean()) < 1e-12
        assert abs(float(std[i]) - allp[:, :, i].std(ddof=1)) < 1e-12
    hm, hs = merge_moments(gathered)
    assert torch.allclose(hm, mean.cpu(), rtol=0, atol=1e-15) and torch.allclose(hs, std.cpu(), rtol=1e-15, atol=0)
    # the several-rank tail behind the all-gather as ONE launch: merge + normalisation == the two separate calls, bit for bit
    mean, std = mean.clone(), std.clone()
    adv_ref = eng.adv_normalize(mean, std).clone()
    out = torch.full((T, E, N, 1), float("nan"), device="cuda")
    adv, m2, s2 = eng.adv_merge_normalize(gathered, out=out)
    assert adv.data_ptr() == out.data_ptr() and torch.equal(adv, adv_ref)
    assert torch.equal(m2, mean) and torch.equal(s2, std)


@pytest.mark.parametrize("E,G,A,T", [(4096, 3, 3, 128),          # the headline shape: fa_gae_mom_kernel, 384 workgroups, fold of 8
                                     (300, 3, 3, 64), (64, 5, 5, 128),
                                     (37, 2, 5, 33),             # ragged last workgroup, T*E*N not a multiple of 4
                                     (5461, 3, 3, 24),           # 32 766 columns: the largest fused shape (512 workgroups, ragged)
                                     (4096, 5, 5, 40),           # 40 960 columns: beyond the fused scan, the separate kernels
                                     (32768, 3, 3, 24)])         # ... and fa_gae_kernel behind the same call
def test_gae_normalize_equals_gae_moments_plus_normalize(fa, E, G, A, T):
    """fa_gae_normalize (scan + moment partials, then fold + normalisation in every workgroup) == fa_gae_moments +
    fa_adv_normalize bit for bit -- returns, moments, mean, std, normalised advantages -- and the numpy oracle's
    normalised advantages (ppo.py:121-124) to 2e-6; the stale episode-end entries belong to the statistics."""
    import collector_oracle as co
    N = G + A
    rng = np.random.RandomState(7 * E + T)
    eng = fa.BatchedFortAttack(E, G, A, 50)
    st = fa.JointRolloutStorage(T, E, N, device="cuda")
    eng.bind_storage(st)
    data = dict(rewards=rng.randn(T, E, N, 1).astype(np.float32),
                value_preds=rng.randn(T + 1, E, N, 1).astype(np.float32),
                masks=(rng.rand(T + 1, E, N, 1) > 0.2).astype(np.float32),
                returns=(3.0 * rng.randn(T + 1, E, N, 1)).astype(np.float32),   # stale entries of the previous update (Q7)
                done=(rng.rand(T, E) < 0.05).astype(np.uint8))

    def load():
        for k, v in data.items():
            getattr(st, k).copy_(_t(v))

    load()
    mom_ref, mean_ref, std_ref = [x.clone() for x in eng.gae_moments(0.99, 0.95)]
    ret_ref = st.returns.clone()
    adv_ref = eng.adv_normalize(mean_ref, std_ref).clone()
    load()
    out = torch.full((T, E, N, 1), float("nan"), device="cuda")
    adv, mom, mean, std = eng.gae_normalize(0.99, 0.95, out=out)
    assert adv.data_ptr() == out.data_ptr()
    assert torch.equal(st.returns, ret_ref)
    assert torch.equal(mom, mom_ref) and torch.equal(mean, mean_ref) and torch.equal(std, std_ref)
    assert torch.equal(adv, adv_ref)
    # the oracle: GAE with the stale entries kept, then (A - mean) / (std + 1e-5) per agent
    ep_start = np.zeros((T, E), bool)
    ep_start[1:] = data["done"][:-1] != 0
    want = data["returns"].copy()
    a = adv.cpu().numpy()
    for i in range(N):
        co.gae_single_pass(data["rewards"][:, :, i], data["value_preds"][:, :, i], data["masks"][:, :, i], want[:, :, i],
                           ep_start, 0.99, 0.95)
        assert np.abs(a[:, :, i] - co.normalized_advantages(want[:, :, i], data["value_preds"][:, :, i])).max() < 2e-6
    assert np.array_equal(ret_ref.cpu().numpy(), want)
    assert int(ep_start.sum()) > 0
    load()
    adv2, _, _, _ = eng.gae_normalize(0.99, 0.95)
    assert torch.equal(adv2, adv)                                              # run-to-run identical
