"""GPU: BASELINE configs 4 and 5 AS STATED -- 32 768 envs sharded 8 ways (rank r owns global envs
[4096 r, 4096 (r + 1)), env e of the job is the reference run under np.random.seed(base_seed + e):
utils.py:17's `seed + rank` convention) -- run shard by shard on the ONE GPU and
compared with the CPU oracle, not with another HIP handle:

  (a) config 4, open loop: shards r in {1, 3, 7} (`env_offset = 4096 r`) through the fused COLLECT launch at
      4096 x 128 against OracleEnv(base_seed = B + 4096 r); GAE returns and advantage moments against numpy;
  (b) + (d) one 3v3 x 32 768 x 128 single-handle fused rollout against the oracle (3 277 workgroups: the classic
      fused loop at the config's own size), then ALL EIGHT shards: rows and returns == the big handle's slices bit
      for bit; the eight (n, mean, M2) triples merged in rank order (fa_adv_merge: what follows the RCCL all-gather
      of ppo.py:121-124's statistics) against the big handle's and against numpy float64; fa_adv_merge_normalize of
      every shard against fa_gae_normalize of the big handle;
  (c) config 5: 5v5, shards r in {1, 7}, the five PUBLISHED attacker policies (tests/golden/attackers_tmp1.npz) as
      frozen strategies drawn by np.random.choice(5) on each env's reset stream (learner.py:119-121), closed loop from
      the hipGraph, against OracleEnv(...).set_choice(5) driven by the sampled actions;
  (e) config 4 closed loop: the MPNN-in-the-loop rollout of one 32 768-env learner against the oracle driven by the
      sampled actions, and shards r in {2, 7} of it (same policies, same sampling seed: the Philox key holds the GLOBAL
      env index) bit for bit -- actions, log-probs, values, env rows.

What this leaves untested of configs 4 / 5: RCCL with more than one rank on hardware (the exchange itself:
tests/test_gpu_rccl.py in a world of one rank, tests/test_dist_cpu.py / test_gpu_two_ranks.py over gloo).

Bar: done / masks / alive flags / strategy ids / reset-stream cursor bit exact; float rows == float32(oracle) bit for bit
(and <= 1e-5 of max(1, |value|) from the oracle's float64); returns bit exact; moments <= 1e-12 relative; normalised advantages <= 2e-6.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHARD, WORLD, T = 4096, 8, 128
BASE_SEED = 20260


@pytest.fixture(scope="module")
def fa():
    import emergent_multiagent_strategies_amd as m
    assert torch.cuda.is_available()
    m._lib.load()
    return m


def _job_inputs(E_total, N, seed):
    """Open-loop actions (a shot every fourth decision) and critic values of the WHOLE job; a shard takes its slice."""
    g = torch.Generator().manual_seed(seed)
    acts = torch.randint(0, 8, (T, E_total, N, 1), generator=g)
    acts = torch.where(torch.rand(T, E_total, N, 1, generator=g) < 0.25, torch.full_like(acts, 7), acts)
    vals = torch.randn(T + 1, E_total, N, 1, generator=g)
    return acts, vals


def _rows_vs_oracle(st, orc, acts, n_steps, ids_at=None):
    """storage rows of a rollout vs the oracle stepped n_steps times with the actions in `acts` (T, E, N)."""
    obs = st.obs.cpu().numpy()
    rew = st.rewards.cpu().numpy()[..., 0]
    msk = st.masks.cpu().numpy()[..., 0]
    done = st.done.cpu().numpy()
    E = done.shape[1]
    n_diff, worst, ends, deaths = 0, 0.0, 0, 0
    ep_start = np.zeros((n_steps, E), bool)
    for s in range(n_steps):
        if ids_at is not None:
            assert np.array_equal(ids_at[s], orc.get_choice()), s
        ref = orc.step(acts[s], auto_reset=True)
        assert np.array_equal(done[s], ref["done"]), s
        want_mask = np.where(ref["done"][:, None] != 0, 1, ref["alive_before"]).astype(np.float32)
        assert np.array_equal(msk[s + 1], want_mask), s
        assert np.array_equal(obs[s + 1, :, :, 0], ref["obs"][:, :, 0].astype(np.float32)), s   # alive column
        o32, r32 = ref["obs"].astype(np.float32), ref["reward"].astype(np.float32)
        n_diff += int((obs[s + 1] != o32).sum() + (rew[s] != r32).sum())
        # float rows: stored float32 vs the oracle's float64, relative to max(1, |value|): the heading is not wrapped
        # (core.py:333 adds u[2] % 2 pi, i.e. +6.11 for a right turn) and passes 250 within an episode of the published
        # attackers, where one float32 ulp is 1.5e-5 -- `n_diff` (bit equality with float32(oracle)) is the sharp statement
        worst = max(worst, float((np.abs(obs[s + 1] - ref["obs"]) / np.maximum(1.0, np.abs(ref["obs"]))).max()),
                    float((np.abs(rew[s] - ref["reward"]) / np.maximum(1.0, np.abs(ref["reward"]))).max()))
        ends += int(ref["done"].sum())
        deaths += int(ref["was_hit"].sum())
        if s + 1 < n_steps:
            ep_start[s + 1] = ref["done"] != 0
    return n_diff, worst, ends, deaths, ep_start


def _state_and_stream_equal(eng, orc, N, envs):
    so, sg = orc.get_state(), eng.get_state()
    for k in ("alive", "time_step", "num_hit", "num_was_hit"):
        assert np.array_equal(so[k], sg[k]), k
    for k in ("pos_x", "pos_y", "vel_x", "vel_y", "ang", "prev_dist"):
        assert np.array_equal(so[k], sg[k], equal_nan=True), k
    for e in envs:                                                      # position on the env's MT19937 stream
        assert np.array_equal(eng.rng_peek(e, 2 * N), orc.rng_doubles(e, 2 * N)), e


def _numpy_returns_and_moments(st, ep_start):
    import collector_oracle as co
    rew, vals, msk = [getattr(st, k).cpu().numpy() for k in ("rewards", "value_preds", "masks")]
    N = rew.shape[2]
    want = np.zeros_like(vals)
    mom = np.zeros((N, 3))
    for i in range(N):
        co.gae_single_pass(rew[:, :, i], vals[:, :, i], msk[:, :, i], want[:, :, i], ep_start, 0.99, 0.95)
        a = (want[:-1, :, i] - vals[:-1, :, i]).astype(np.float32).astype(np.float64)
        mom[i] = a.size, a.mean(), ((a - a.mean()) ** 2).sum()
    return want, mom


def _assert_moments(got, want, tag):
    """(n, mean, M2) triples: n exact; mean to 1e-12 of the advantages' scale; M2 to 1e-12 relative."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert np.array_equal(got[:, 0], want[:, 0]), tag
    scale = np.sqrt(want[:, 2] / (want[:, 0] - 1))
    assert (np.abs(got[:, 1] - want[:, 1]) <= 1e-12 * np.maximum(scale, np.abs(want[:, 1]))).all(), (tag, got[:, 1], want[:, 1])
    assert (np.abs(got[:, 2] / want[:, 2] - 1) <= 1e-12).all(), (tag, got[:, 2], want[:, 2])


# ------------------------------------------------------------------------------------------------------------------
# (a) config 4, open loop, one shard at a time against the oracle
@pytest.mark.parametrize("r", [1, 3, 7])
def test_config4_shard_open_loop_vs_oracle(fa, r):
    from fa_oracle import OracleEnv
    G, A, max_t = 3, 3, 100
    N, lo = G + A, SHARD * r
    acts, vals = _job_inputs(SHARD * WORLD, N, 404)
    acts, vals = acts[:, lo:lo + SHARD].contiguous(), vals[:, lo:lo + SHARD].contiguous()
    orc = OracleEnv(SHARD, G, A, max_t, base_seed=BASE_SEED + lo)      # global env seeds 4096 r ... 4096 r + 4095
    eng = fa.BatchedFortAttack(SHARD, G, A, max_t, base_seed=BASE_SEED, env_offset=lo)
    assert eng.step_variant(T) == "fa_step_pipe_kernel"                # the bench's launch
    st = fa.JointRolloutStorage(T, SHARD, N, device="cuda")
    eng.bind_storage(st)
    eng.collect_reset()
    assert np.array_equal(st.obs[0].cpu().numpy(), orc.reset().astype(np.float32))
    st.actions.copy_(acts.cuda())
    st.value_preds.copy_(vals.cuda())
    eng.collect_rollout(0, T)
    mom = eng.gae_moments(0.99, 0.95)[0].clone()
    torch.cuda.synchronize()
    n_diff, worst, ends, deaths, ep_start = _rows_vs_oracle(st, orc, acts[..., 0].numpy(), T)
    print("config 4 shard %d (global envs %d..%d): episodes=%d deaths=%d differing f32 values=%d worst=%.2e"
          % (r, lo, lo + SHARD - 1, ends, deaths, n_diff, worst))
    assert ends >= SHARD and deaths > SHARD // 4 and n_diff == 0 and worst <= 1e-5
    _state_and_stream_equal(eng, orc, N, (0, SHARD // 2, SHARD - 1))
    want_ret, want_mom = _numpy_returns_and_moments(st, ep_start)
    assert np.array_equal(st.returns.cpu().numpy(), want_ret)
    _assert_moments(mom.cpu().numpy(), want_mom, "shard %d" % r)


# ------------------------------------------------------------------------------------------------------------------
# (b) + (d) one 32 768-env handle against the oracle; the eight shards against it; the merged statistics
def test_config4_eight_shards_equal_one_32768_env_handle_and_the_oracle(fa):
    from fa_oracle import OracleEnv
    G, A, max_t = 3, 3, 100
    N, E = G + A, SHARD * WORLD
    acts, vals = _job_inputs(E, N, 404)
    big = fa.BatchedFortAttack(E, G, A, max_t, base_seed=BASE_SEED)
    assert big.step_variant(T).startswith("fa_step_kernel")             # 3 277 workgroups: the classic fused loop
    st = fa.JointRolloutStorage(T, E, N, device="cuda")
    big.bind_storage(st)
    big.collect_reset()
    st.actions.copy_(acts.cuda())
    st.value_preds.copy_(vals.cuda())
    big.collect_rollout(0, T)
    big_adv, big_mom, big_mean, big_std = big.gae_normalize(0.99, 0.95)
    torch.cuda.synchronize()
    # (d) the single handle at the config's own size vs the oracle
    orc = OracleEnv(E, G, A, max_t, base_seed=BASE_SEED)
    assert np.array_equal(st.obs[0].cpu().numpy(), orc.reset().astype(np.float32))
    n_diff, worst, ends, deaths, ep_start = _rows_vs_oracle(st, orc, acts[..., 0].numpy(), T)
    print("config 4, one handle of %d envs (%s): episodes=%d deaths=%d differing f32 values=%d worst=%.2e"
          % (E, big.step_variant(T), ends, deaths, n_diff, worst))
    assert ends >= E and n_diff == 0 and worst <= 1e-5
    _state_and_stream_equal(big, orc, N, (0, SHARD, E - 1))
    want_ret, want_mom = _numpy_returns_and_moments(st, ep_start)
    assert np.array_equal(st.returns.cpu().numpy(), want_ret)
    _assert_moments(big_mom.cpu().numpy(), want_mom, "one handle")
    big_rows = {k: getattr(st, k).cpu().numpy() for k in ("obs", "rewards", "masks", "done", "returns")}
    big_adv_h = big_adv.cpu().numpy()
    big_mean_h, big_std_h = big_mean.cpu().numpy().copy(), big_std.cpu().numpy().copy()
    del st, big, big_adv
    torch.cuda.empty_cache()
    # (b) the eight shards, one after another; their moments gathered in rank order
    gathered = torch.zeros((WORLD, N, 3), dtype=torch.float64, device="cuda")
    shards = []
    for r in range(WORLD):
        lo = SHARD * r
        eng = fa.BatchedFortAttack(SHARD, G, A, max_t, base_seed=BASE_SEED, env_offset=lo)
        s = fa.JointRolloutStorage(T, SHARD, N, device="cuda")
        eng.bind_storage(s)
        eng.collect_reset()
        s.actions.copy_(acts[:, lo:lo + SHARD].cuda())
        s.value_preds.copy_(vals[:, lo:lo + SHARD].cuda())
        eng.collect_rollout(0, T)
        gathered[r].copy_(eng.gae_moments(0.99, 0.95)[0])
        torch.cuda.synchronize()
        for k, v in big_rows.items():                                   # rows and GAE: bit for bit
            assert np.array_equal(getattr(s, k).cpu().numpy(), v[:, lo:lo + SHARD]), (r, k)
        shards.append((eng, s))
    # what every rank computes behind the all-gather: Chan-Golub-LeVeque merge in rank order, then normalisation
    mean0 = std0 = None
    for r, (eng, s) in enumerate(shards):
        adv, mean, std = eng.adv_merge_normalize(gathered)
        torch.cuda.synchronize()
        mean, std = mean.cpu().numpy().copy(), std.cpu().numpy().copy()
        if r == 0:
            mean0, std0 = mean, std
            m2, s2 = eng.adv_merge(gathered)                            # the stand-alone merge: same bits
            assert np.array_equal(m2.cpu().numpy(), mean) and np.array_equal(s2.cpu().numpy(), std)
        assert np.array_equal(mean, mean0) and np.array_equal(std, std0), r     # every rank: the same bits
        lo = SHARD * r
        assert np.abs(adv.cpu().numpy() - big_adv_h[:, lo:lo + SHARD]).max() <= 2e-6, r
    # merged == the one handle's statistics == numpy float64 (different summation trees: 1e-12, not bits)
    sd = np.sqrt(want_mom[:, 2] / (want_mom[:, 0] - 1))
    assert (np.abs(mean0 - big_mean_h) <= 1e-12 * np.maximum(sd, np.abs(big_mean_h))).all()
    assert (np.abs(std0 / big_std_h - 1) <= 1e-12).all()
    assert (np.abs(mean0 - want_mom[:, 1]) <= 1e-12 * np.maximum(sd, np.abs(want_mom[:, 1]))).all()
    assert (np.abs(std0 / sd - 1) <= 1e-12).all()
    n_tot = gathered[:, :, 0].sum(0).cpu().numpy()
    assert np.array_equal(n_tot, want_mom[:, 0]) and float(n_tot[0]) == T * E


# ------------------------------------------------------------------------------------------------------------------
# (c) config 5: 5v5 shards with the published attacker ensemble, closed loop from the hipGraph
@pytest.mark.parametrize("r", [1, 7])
def test_config5_shard_published_ensemble_closed_loop_vs_oracle(fa, golden_dir, r):
    from fa_oracle import OracleEnv
    from test_mpnn_cpu import attacker_pool_from_golden
    z, pool, G, A = attacker_pool_from_golden(fa.MPNN, golden_dir)
    assert (G, A) == (5, 5)
    N, K, max_t, lo = G + A, len(pool), 60, SHARD * r
    torch.manual_seed(50)
    eng = fa.BatchedFortAttack(SHARD, G, A, max_t, base_seed=BASE_SEED, env_offset=lo)
    orc = OracleEnv(SHARD, G, A, max_t, base_seed=BASE_SEED + lo)
    orc.set_choice(K)                                                   # np.random.choice(K) after every reset
    L = fa.BatchedLearner(eng, num_steps=T, use_graph=True, sample_seed=77)
    assert L.policy_backend == "hip"
    L.load_attacker_ensemble([{"models": [None] * G + [p.state_dict()] * A, "ob_rms": (None, None)} for p in pool])
    L.reset()
    assert L._graphs is not None
    L.collect()
    torch.cuda.synchronize()
    st = L.storage
    ids = L.attacker_id_rows.cpu().numpy()                              # (T, E): the strategy in force at every step
    assert np.array_equal(st.obs[0].cpu().numpy(), orc.reset().astype(np.float32))
    acts = st.actions.cpu().numpy()[..., 0]
    n_diff, worst, ends, deaths, ep_start = _rows_vs_oracle(st, orc, acts, T, ids_at=ids)
    print("config 5 shard %d: episodes=%d deaths=%d differing f32 values=%d worst=%.2e strategies seen=%s"
          % (r, ends, deaths, n_diff, worst, np.bincount(ids.ravel(), minlength=K).tolist()))
    assert ends >= SHARD and n_diff == 0 and worst <= 1e-5
    assert np.array_equal(L.attacker_id.cpu().numpy(), orc.get_choice())
    assert len(np.unique(ids)) == K
    _state_and_stream_equal(eng, orc, N, (0, SHARD // 2, SHARD - 1))
    for s in range(T - 1):                                              # an id moves only where an episode ended
        assert ((ids[s + 1] != ids[s]) <= (st.done[s].cpu().numpy() != 0)).all(), s
    # every env's attacker rows come from the PUBLISHED policy its stream drew
    att = slice(G, N)
    dev_pool = [p.cuda() for p in pool]
    with torch.no_grad():
        for s in (0, T // 2, T - 1):
            obs = st.obs[s]
            outs = [p.evaluate_actions(obs[:, att], obs[:, :G], st.actions[s, :, att]) for p in dev_pool]
            sel = torch.from_numpy(ids[s]).long().cuda()
            ar = torch.arange(SHARD, device="cuda")
            want_lp = torch.stack([o[1] for o in outs])[sel, ar]
            want_v = torch.stack([o[0] for o in outs])[sel, ar]
            assert (want_lp - st.action_log_probs[s, :, att]).abs().max() < 2e-4, s
            assert (want_v - st.value_preds[s, :, att]).abs().max() < 1e-5 * float(want_v.abs().max()) + 1e-5, s
    want_ret, want_mom = _numpy_returns_and_moments(st, ep_start)
    assert np.array_equal(st.returns.cpu().numpy(), want_ret)
    _assert_moments(L.eng.adv_moments().cpu().numpy(), want_mom, "config 5 shard %d" % r)
    L.close()


# ------------------------------------------------------------------------------------------------------------------
# (e) config 4 closed loop: one 32 768-env learner vs the oracle, and shards of it bit for bit
def test_config4_closed_loop_32768_vs_oracle_and_shards_bit_for_bit(fa):
    from fa_oracle import OracleEnv
    G, A, max_t = 3, 3, 100
    N, E = G + A, SHARD * WORLD
    torch.manual_seed(4)
    big_eng = fa.BatchedFortAttack(E, G, A, max_t, base_seed=BASE_SEED)
    Lb = fa.BatchedLearner(big_eng, num_steps=T, use_graph=True, sample_seed=99)
    assert Lb.policy_backend == "hip"
    Lb.reset()
    Lb.collect()
    torch.cuda.synchronize()
    st = Lb.storage
    orc = OracleEnv(E, G, A, max_t, base_seed=BASE_SEED)
    assert np.array_equal(st.obs[0].cpu().numpy(), orc.reset().astype(np.float32))
    n_diff, worst, ends, deaths, ep_start = _rows_vs_oracle(st, orc, st.actions.cpu().numpy()[..., 0], T)
    print("config 4 closed loop, one learner of %d envs: episodes=%d deaths=%d differing f32 values=%d worst=%.2e"
          % (E, ends, deaths, n_diff, worst))
    assert ends >= E and n_diff == 0 and worst <= 1e-5
    want_ret, want_mom = _numpy_returns_and_moments(st, ep_start)
    assert np.array_equal(st.returns.cpu().numpy(), want_ret)
    mean, std = Lb._adv_mean_std
    sd = np.sqrt(want_mom[:, 2] / (want_mom[:, 0] - 1))
    assert (np.abs(mean.cpu().numpy() - want_mom[:, 1]) <= 1e-12 * np.maximum(sd, np.abs(want_mom[:, 1]))).all()
    assert (np.abs(std.cpu().numpy() / sd - 1) <= 1e-11).all()
    keys = ("obs", "rewards", "masks", "done", "actions", "action_log_probs", "value_preds", "returns")
    big_rows = {k: getattr(st, k).cpu().numpy() for k in keys}
    state = [{k: v.detach().clone() for k, v in d.items()} for d in Lb.state_dicts()]
    Lb.close()
    del Lb, big_eng, st
    torch.cuda.empty_cache()
    for r in (2, 7):
        lo = SHARD * r
        eng = fa.BatchedFortAttack(SHARD, G, A, max_t, base_seed=BASE_SEED, env_offset=lo)
        L = fa.BatchedLearner(eng, num_steps=T, use_graph=True, sample_seed=99)
        L.load_models(state)                                            # the job's policies (replicated on every rank)
        L.reset()
        L.collect()
        torch.cuda.synchronize()
        for k in keys:
            assert np.array_equal(getattr(L.storage, k).cpu().numpy(), big_rows[k][:, lo:lo + SHARD]), (r, k)
        L.close()
