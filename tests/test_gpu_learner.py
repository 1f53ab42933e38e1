"""GPU: the closed-loop batched rollout (MPNN forward -> sample -> fa_collect_step, eager and
replayed from hipGraphs), checked for consistency against the CPU oracle driven by the
actions the policies actually sampled, then GAE / PPO update / after_update / checkpoint."""
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


# The policy-side rows (values, log-probs) of the fused forward are judged against the module evaluated in FLOAT64 on the same
# observations and actions -- trajectory independent: the bound is the float32 module's own distance from float64 on the very
# same rows, times ERR_FACTOR, plus a float32 resolution term.  (Rounds 3-5 compared the fused rows with the float32 module
# under constants taken from training soaks -- 6e-4 absolute for log-probs, 2e-4 relative for values -- which could not tell
# float32 noise of the folded algebra from a regression of that size feeding the PPO ratio.)
# POLICY_ROW_ERRORS keeps the largest errors seen in this process and the largest fused / module ratio: tools/soak_closed_loop.py
# reports them (profiles/r06_soak_closed_loop.jsonl).
ERR_FACTOR, ERR_FLOOR = 2.0, 1e-6
POLICY_ROW_ERRORS = {"value_fused": 0.0, "value_module_f32": 0.0, "logp_fused": 0.0, "logp_module_f32": 0.0,
                     "value_ratio": 0.0, "logp_ratio": 0.0, "checks": 0}


def _policy_rows_vs_float64(pol, st, own, opp, T, E):
    """max |fused - f64| and max |module_f32 - f64| for the value rows (incl. V(obs[T])) and the log-prob rows of one team."""
    p64 = type(pol)(num_agents=pol.num_agents, num_opp_agents=pol.num_opp_agents, hidden_dim=pol.h_dim, num_actions=8)
    p64.load_state_dict(pol.state_dict())
    p64 = p64.to(st.obs.device).double()
    p64.refresh_fused_weights()
    obs = st.obs[:-1].flatten(0, 1)
    act = st.actions.flatten(0, 1)[:, own]
    with torch.no_grad():
        v32, lp32, _ = pol.evaluate_actions(obs[:, own], obs[:, opp], act)
        v64, lp64, _ = p64.evaluate_actions(obs[:, own].double(), obs[:, opp].double(), act)
        vT32 = pol.get_value(st.obs[T][:, own], st.obs[T][:, opp])
        vT64 = p64.get_value(st.obs[T][:, own].double(), st.obs[T][:, opp].double())
    fv = st.value_preds[:-1, :, own].reshape(v64.shape).double()
    flp = st.action_log_probs[:, :, own].reshape(lp64.shape).double()
    fvT = st.value_preds[T, :, own].reshape(vT64.shape).double()
    mx = lambda t: float(t.abs().max())
    return {"value_fused": max(mx(fv - v64), mx(fvT - vT64)), "value_module_f32": max(mx(v32.double() - v64), mx(vT32.double() - vT64)),
            "logp_fused": mx(flp - lp64), "logp_module_f32": mx(lp32.double() - lp64),
            "value_scale": max(1.0, mx(v64), mx(vT64)), "logp_scale": max(1.0, mx(lp64))}


def _check_rollout_against_oracle(fa, learner, orc, first):
    import collector_oracle as co
    st, T, E, N, G = learner.storage, learner.T, learner.E, learner.N, learner.G
    obs, rew, msk, done, acts = [getattr(st, k).cpu().numpy() for k in ("obs", "rewards", "masks", "done", "actions")]
    if first:
        assert np.array_equal(obs[0], orc.reset().astype(np.float32))
    ep_start = np.zeros((T, E), bool)
    for s in range(T):
        ref = orc.step(acts[s, :, :, 0], auto_reset=True)
        assert np.array_equal(done[s], ref["done"]), s
        assert np.array_equal(obs[s + 1], ref["obs"].astype(np.float32)), s
        assert np.array_equal(rew[s, :, :, 0], ref["reward"].astype(np.float32)), s
        want_mask = np.where(ref["done"][:, None] != 0, 1, ref["alive_before"]).astype(np.float32)
        assert np.array_equal(msk[s + 1, :, :, 0], want_mask), s
        if s + 1 < T:
            ep_start[s + 1] = ref["done"] != 0
    # policy-side rows are what the policies compute from the stored observations: |fused - f64| <= 2 |module_f32 - f64| + 1e-6
    for ti, (own, opp) in enumerate(((slice(0, G), slice(G, N)), (slice(G, N), slice(0, G)))):
        if ti == 1 and learner.attacker_pool:
            continue                                             # (an ensemble's rows are checked per strategy by its own tests)
        e = _policy_rows_vs_float64(learner.policies[ti], st, own, opp, T, E)
        R = POLICY_ROW_ERRORS
        for k in ("value_fused", "value_module_f32", "logp_fused", "logp_module_f32"):
            R[k] = max(R[k], e[k] / (e["value_scale"] if k.startswith("value") else 1.0))
        R["value_ratio"] = max(R["value_ratio"], e["value_fused"] / max(e["value_module_f32"], 1e-30))
        R["logp_ratio"] = max(R["logp_ratio"], e["logp_fused"] / max(e["logp_module_f32"], 1e-30))
        R["checks"] += 1
        assert e["value_fused"] <= ERR_FACTOR * e["value_module_f32"] + ERR_FLOOR * e["value_scale"], (ti, e)
        assert e["logp_fused"] <= ERR_FACTOR * e["logp_module_f32"] + ERR_FLOOR * e["logp_scale"], (ti, e)
    # GAE over the stored rows == numpy oracle, bit for bit
    vals, rets = st.value_preds.cpu().numpy(), st.returns.cpu().numpy()
    return ep_start, rew, vals, msk, rets


@pytest.mark.parametrize("use_graph,hidden,backend", [(False, 32, "torch"), (True, 32, "torch"),
                                                      (False, 128, "hip"), (True, 128, "hip")])
def test_closed_loop_rollout_and_update(fa, use_graph, hidden, backend, tmp_path):
    import collector_oracle as co
    from fa_oracle import OracleEnv
    torch.manual_seed(0)
    E, G, A, T, max_t = 96, 3, 3, 24, 10
    N = G + A
    eng = fa.BatchedFortAttack(E, G, A, max_t, base_seed=21)
    orc = OracleEnv(E, G, A, max_t, base_seed=21)
    L = fa.BatchedLearner(eng, num_steps=T, hidden_dim=hidden, num_mini_batch=4, ppo_epoch=2, use_graph=use_graph)
    assert L.policy_backend == backend
    L.reset()
    stale = np.zeros((T + 1, E, N, 1), np.float32)
    for upd in range(2):
        L.collect()
        ep_start, rew, vals, msk, rets = _check_rollout_against_oracle(fa, L, orc, first=(upd == 0))
        want = stale.copy()
        for i in range(N):
            co.gae_single_pass(rew[:, :, i], vals[:, :, i], msk[:, :, i], want[:, :, i], ep_start, 0.99, 0.95)
        assert np.array_equal(rets, want)
        stale = rets
        assert int(L.storage.done.sum()) > 0
        before = [p.detach().clone() for p in L.policies[0].parameters()]
        losses = L.update()
        assert losses.shape == (2, 3) and bool(torch.isfinite(losses).all())
        assert any(not torch.equal(b, p) for b, p in zip(before, L.policies[0].parameters()))
        # (the fused update normalises the advantages inside the kernel and never writes L.adv: fa_adv_normalize on demand)
        adv = L.eng.adv_normalize(*L._adv_mean_std).cpu().numpy()
        for i in range(N):
            assert np.abs(adv[:, :, i] - co.normalized_advantages(rets[:, :, i], vals[:, :, i])).max() < 2e-5
        last_obs, last_mask = L.storage.obs[T].clone(), L.storage.masks[T].clone()
        L.after_update()
        assert torch.equal(L.storage.obs[0], last_obs) and torch.equal(L.storage.masks[0], last_mask)
        assert float(L.storage.obs[1:].abs().sum()) == 0.0
    # checkpoint in the reference's wire format (train_fortattack.py:123-128)
    path = str(tmp_path / "ep0.pt")
    L.save(path)
    ck = torch.load(path, weights_only=False)
    # the reference's two keys + the sampling position of the fused policy kernel (ignored by the reference's loader)
    assert set(ck) == {"models", "ob_rms", "fa_rollout_counter", "fa_sample_seed"} and len(ck["models"]) == N and ck["ob_rms"] == (None, None)
    assert ck["fa_rollout_counter"] == 2 and ck["fa_sample_seed"] == L.sample_seed
    L2 = fa.BatchedLearner(fa.BatchedFortAttack(8, G, A, max_t), num_steps=4, hidden_dim=hidden, sample_seed=L.sample_seed + 5)
    with pytest.warns(UserWarning, match="sample_seed"):
        L2.load(path)
    assert int(L2._rollout_counter.item()) == 2 and L2.sample_seed == L.sample_seed   # the sampling stream continues
    for a, b in zip(L.policies[1].parameters(), L2.policies[1].parameters()):
        assert torch.equal(a, b)
    only_guards = L.update(train_guards_only=True)      # train_fortattack_v2 path (learner.py:177)
    assert only_guards.shape == (1, 3)


def test_bench_closed_loop_launch_at_full_size_vs_oracle(fa):
    """The exact launch bench.py's `closed_loop` record times (bench.py closed_loop(): BatchedLearner(use_graph=True)
    at 3v3 x 4096 envs x 128 steps, max_time_steps 100, base_seed 0 -- ONE hipGraph of 128 x (fa_policy_kernel<3, 8> +
    fa_step_kernel<3,3,false,true,3>) + V(obs[T])), two rollouts: env rows / masks / done against the oracle driven by
    the sampled actions (bit for bit), the policy-side rows against the PyTorch definition, GAE returns against the
    numpy oracle (bit for bit, including the stale entries carried from the first rollout into the second)."""
    import collector_oracle as co
    from fa_oracle import OracleEnv
    torch.manual_seed(0)
    E, G, A, T, max_t = 4096, 3, 3, 128, 100
    N = G + A
    eng = fa.BatchedFortAttack(E, G, A, max_t, base_seed=0)
    orc = OracleEnv(E, G, A, max_t, base_seed=0)
    L = fa.BatchedLearner(eng, num_steps=T, use_graph=True)
    assert L.policy_backend == "hip" and L._update_graphs["fused"]
    torch.manual_seed(1)
    L.reset()
    assert L._graphs is not None
    stale = np.zeros((T + 1, E, N, 1), np.float32)
    for upd in range(2):
        L.collect()
        torch.cuda.synchronize()
        ep_start, rew, vals, msk, rets = _check_rollout_against_oracle(fa, L, orc, first=(upd == 0))
        want = stale.copy()
        for i in range(N):
            co.gae_single_pass(rew[:, :, i], vals[:, :, i], msk[:, :, i], want[:, :, i], ep_start, 0.99, 0.95)
        assert np.array_equal(rets, want)
        stale = rets
        assert int(L.storage.done.sum()) >= E              # every env ended at least once (max_t = 100 < T)
        mean, std = L._adv_mean_std
        for i in range(N):
            a = (rets[:-1, :, i] - vals[:-1, :, i]).astype(np.float32).astype(np.float64)
            assert abs(float(mean[i]) - a.mean()) <= 1e-12 * max(1.0, abs(a.mean()))
            assert abs(float(std[i]) - a.std(ddof=1)) <= 1e-11 * a.std(ddof=1)
        L.after_update()


def test_one_learner_mixing_update_variants_keeps_every_captured_step_valid(fa):
    """A learner that runs update() (two-chain steps), update(train_guards_only=True) (a second captured step for the
    guards' policy with its own gradient buffer) and update() again: every captured graph keeps replaying its own
    unfold task list (mpnn_pack.FlatPolicy.unfold keeps one list per gradient buffer) -- the repeated update equals
    the first one bit for bit from the same state."""
    torch.manual_seed(2)
    eng = fa.BatchedFortAttack(256, 3, 3, 12, base_seed=8)
    L = fa.BatchedLearner(eng, num_steps=16, num_mini_batch=4, ppo_epoch=2, use_graph=True, update_backend="fused")
    L.reset()
    L.collect()
    fps = L._flat

    def snap():
        return [(fp.pflat.clone(), getattr(fp, "mflat", None) is not None and fp.mflat.clone(),
                 getattr(fp, "vflat", None) is not None and fp.vflat.clone(),
                 getattr(fp, "steps", None) is not None and fp.steps.clone()) for fp in fps]

    def restore(sn):
        for fp, (p, m, v, st) in zip(fps, sn):
            fp.pflat.copy_(p)
            if m is not False:
                fp.mflat.copy_(m); fp.vflat.copy_(v); fp.steps.copy_(st)

    torch.manual_seed(7)
    L.update()                                              # captures; binds the optimizers to the flat buffers
    s0 = snap()
    torch.manual_seed(7)
    l1 = L.update().clone()
    p1 = [fp.pflat.clone() for fp in fps]
    restore(s0)
    L.update(train_guards_only=True)                        # a second GraphedPPOStep of the guards' policy
    assert len(fps[0]._unfold) == 2 and len(fps[1]._unfold) == 1
    for _ in range(3):                                      # churn the allocator: a freed task list would be reused
        torch.empty(1 << 16, device="cuda").random_()
    restore(s0)
    torch.manual_seed(7)
    l2 = L.update().clone()
    torch.cuda.synchronize()
    assert torch.equal(l1, l2)
    assert all(torch.equal(a, fp.pflat) for a, fp in zip(p1, fps))


def test_captured_optimizer_step_follows_the_learning_rate(fa):
    """lr / betas / eps reach fa_adam_step through device memory (fa_adam_step_dev), not as values frozen into the
    hipGraph: halving param_groups[0]['lr'] after the capture halves the next first-step displacement."""
    torch.manual_seed(3)
    eng = fa.BatchedFortAttack(128, 3, 3, 12, base_seed=1)
    L = fa.BatchedLearner(eng, num_steps=8, num_mini_batch=1, ppo_epoch=1, use_graph=True, update_backend="fused", lr=1e-3)
    L.reset()
    L.collect()
    L.update()                                              # capture
    fp = L._flat[0]
    moved = []
    for lr in (1e-3, 5e-4):
        L.optimizers[0].param_groups[0]["lr"] = lr
        fp.mflat.zero_(); fp.vflat.zero_(); fp.steps.zero_()
        before = fp.pflat.clone()
        torch.manual_seed(4)
        L.update()
        moved.append((fp.pflat - before).abs().max().item())
        fp.pflat.copy_(before)
    assert abs(moved[0] - 1e-3) < 2e-5 and abs(moved[1] - 5e-4) < 1e-5, moved   # Adam's first step is lr * sign(g)


def test_fused_update_backend_is_refused_where_it_cannot_run(fa):
    eng = fa.BatchedFortAttack(64, 3, 3, 10)
    with pytest.raises(ValueError, match="update_backend='fused'"):
        fa.BatchedLearner(eng, num_steps=8, use_graph=False, update_backend="fused")
    with pytest.raises(ValueError, match="update_backend='fused'"):
        fa.BatchedLearner(eng, num_steps=8, use_graph=True, hidden_dim=32, update_backend="fused")


@pytest.mark.parametrize("use_graph", [False, True])
def test_ensemble_attackers_per_env_strategy(fa, use_graph):
    """Config-5 path (train_fortattack_v2.py): frozen attacker strategies, one per env, re-drawn
    at that env's episode end; guards only are trained."""
    torch.manual_seed(1)
    E, G, A, T, max_t = 64, 3, 3, 20, 9
    eng = fa.BatchedFortAttack(E, G, A, max_t, base_seed=4)
    L = fa.BatchedLearner(eng, num_steps=T, hidden_dim=32, num_mini_batch=2, ppo_epoch=1, use_graph=use_graph)
    pool_sd = [fa.MPNN(num_agents=A, num_opp_agents=G, hidden_dim=32, num_actions=8).state_dict() for _ in range(3)]
    L.load_attacker_ensemble([{"models": [None] * G + [sd] * A, "ob_rms": (None, None)} for sd in pool_sd],
                             hidden_dim=32)
    L.reset()
    ids_before = L.attacker_id.clone()
    st = L.storage
    if use_graph:       # the whole rollout replays from one graph; attacker_id_rows logs the ids in force per step
        L._graphs.replay()
        ids_at = list(L.attacker_id_rows.clone())
    else:               # step by step (eager path of collect)
        ids_at = []
        for s in range(T):
            ids_at.append(L.attacker_id.clone())
            L.step(s)
        assert torch.equal(torch.stack(ids_at), L.attacker_id_rows)
    att = slice(G, G + A)
    with torch.no_grad():
        for s in range(T):
            obs = st.obs[s]
            lp_all = torch.stack([p.evaluate_actions(obs[:, att], obs[:, :G], st.actions[s, :, att])[1]
                                  for p in L.attacker_pool])            # (K,E,A,1)
            want = lp_all[ids_at[s].long(), torch.arange(E, device="cuda")]
            assert (want - st.action_log_probs[s, :, att]).abs().max() < 1e-4, s
    changed = (L.attacker_id != ids_before)
    ended = st.done.sum(0) > 0
    assert bool((changed <= ended).all()) and int(ended.sum()) > 0   # ids only move where an episode ended
    att_before = [p.detach().clone() for p in L.attacker_pool[0].parameters()]
    L.collect()
    out = L.update(train_guards_only=True)
    assert out.shape == (1, 3)
    assert all(torch.equal(a, b) for a, b in zip(att_before, L.attacker_pool[0].parameters()))


def test_ensemble_rollout_config5_shape_vs_oracle(fa):
    """BASELINE config 5's per-GPU shape: 5v5, 4096 envs, K = 5 frozen attacker strategies (h = 128),
    closed loop.  The env rows must equal the oracle driven by the sampled actions, every env's
    attacker log-probs must come from the strategy assigned to that env at that step, and a strategy
    id may only change where an episode ended."""
    from fa_oracle import OracleEnv
    torch.manual_seed(5)
    E, G, A, T, max_t, K = 4096, 5, 5, 48, 20, 5
    N = G + A
    eng = fa.BatchedFortAttack(E, G, A, max_t, base_seed=55)
    orc = OracleEnv(E, G, A, max_t, base_seed=55)
    orc.set_choice(K)                                   # np.random.choice(K) after every reset (quirk Q14)
    L = fa.BatchedLearner(eng, num_steps=T, num_mini_batch=8, ppo_epoch=1, use_graph=False)
    assert L.policy_backend == "hip"
    pool_sd = [fa.MPNN(num_agents=A, num_opp_agents=G, num_actions=8).state_dict() for _ in range(K)]
    L.load_attacker_ensemble([{"models": [None] * G + [sd] * A, "ob_rms": (None, None)} for sd in pool_sd])
    L.reset()
    st = L.storage
    ids_at = []
    for s in range(T):
        ids_at.append(L.attacker_id.clone())
        L.step(s)
    assert len(torch.unique(torch.stack(ids_at))) == K
    _check_rollout_env_rows(L, orc, ids_at)             # env rows AND the strategy ids follow the oracle's stream
    att = slice(G, N)
    with torch.no_grad():
        for s in (0, T // 2, T - 1):
            obs = st.obs[s]
            lp_all = torch.stack([p.evaluate_actions(obs[:, att], obs[:, :G], st.actions[s, :, att])[1]
                                  for p in L.attacker_pool])
            want = lp_all[ids_at[s].long(), torch.arange(E, device="cuda")]
            assert (want - st.action_log_probs[s, :, att]).abs().max() < 2e-4, s
    for s in range(T - 1):
        moved = ids_at[s + 1] != ids_at[s]
        assert bool((moved <= (st.done[s] != 0)).all()), s
    assert int(st.done.sum()) > E


def _check_rollout_env_rows(learner, orc, ids_at=None):
    st, T = learner.storage, learner.T
    obs, rew, msk, done, acts = [getattr(st, k).cpu().numpy() for k in ("obs", "rewards", "masks", "done", "actions")]
    assert np.array_equal(obs[0], orc.reset().astype(np.float32))
    for s in range(T):
        if ids_at is not None:
            assert np.array_equal(ids_at[s].cpu().numpy(), orc.get_choice()), s
        ref = orc.step(acts[s, :, :, 0], auto_reset=True)
        assert np.array_equal(done[s], ref["done"]), s
        assert np.array_equal(obs[s + 1], ref["obs"].astype(np.float32)), s
        assert np.array_equal(rew[s, :, :, 0], ref["reward"].astype(np.float32)), s
        want_mask = np.where(ref["done"][:, None] != 0, 1, ref["alive_before"]).astype(np.float32)
        assert np.array_equal(msk[s + 1, :, :, 0], want_mask), s


@pytest.mark.parametrize("G,A,clipped", [(3, 3, True), (5, 5, True), (3, 3, False)])
def test_fused_update_follows_the_torch_update(fa, G, A, clipped):
    """BatchedLearner.update with the fused fa_ppo_grad step (update_backend="fused") vs PyTorch autograd
    (update_backend="torch") from the same rollout, same initial policies and the same minibatch permutations:
    the averaged losses agree and the parameters stay together (Adam amplifies rounding differences of
    near-zero gradients to at most one learning-rate step)."""
    E, T = 256, 16
    res = []
    for backend in ("torch", "fused"):
        torch.manual_seed(3)
        eng = fa.BatchedFortAttack(E, G, A, 12, base_seed=9)
        L = fa.BatchedLearner(eng, num_steps=T, num_mini_batch=4, ppo_epoch=2, use_graph=True, update_backend=backend,
                              clipped_value_loss=clipped, lr=1e-4)
        L.reset()
        L.collect()
        torch.manual_seed(77)                       # the minibatch permutations
        losses = L.update()
        torch.cuda.synchronize()
        step = next(iter(v for k, v in L._update_graphs.items() if k != "fused"))
        assert step.fused == (backend == "fused")
        res.append((losses.cpu(), [p.detach().cpu().clone() for pol in L.policies for p in pol.parameters()],
                    L.storage.actions.clone()))
    (l0, p0, a0), (l1, p1, a1) = res
    assert torch.equal(a0, a1)                      # same rollout
    assert (l0 - l1).abs().max() < 2e-4 * max(1.0, float(l0.abs().max())), (l0, l1)
    worst = max(float((x - y).abs().max()) for x, y in zip(p0, p1))
    print("max parameter distance fused vs torch: %.2e" % worst)
    assert worst < 4e-4                             # 8 Adam steps of 1e-4


@pytest.mark.parametrize("G,A", [(3, 3), (5, 5), (2, 4)])
def test_update_of_both_teams_in_one_graph_is_the_sequential_update(fa, G, A):
    """The two teams' updates as concurrent chains on two streams (BatchedLearner._update_teams_together, with the
    register-capped build of the train kernel) against the same steps replayed one team after the other: identical
    parameters and losses, bit for bit (the chains share nothing but the read-only rollout; no atomics anywhere)."""
    res = []
    for together in (False, True):
        torch.manual_seed(5)
        eng = fa.BatchedFortAttack(256, G, A, 12, base_seed=2)
        L = fa.BatchedLearner(eng, num_steps=16, num_mini_batch=4, ppo_epoch=2, use_graph=True, update_backend="fused")
        L._update_graphs["teams_together"] = together
        L.reset()
        for _ in range(2):
            L.collect()
            torch.manual_seed(41)
            losses = L.update()
            L.after_update()
        torch.cuda.synchronize()
        assert ("team_streams" in L._update_graphs) == together
        res.append((losses.clone(), [p.detach().clone() for pol in L.policies for p in pol.parameters()]))
    assert torch.equal(res[0][0], res[1][0])
    assert all(torch.equal(a, b) for a, b in zip(res[0][1], res[1][1]))


def test_fused_update_at_config3_size_follows_the_torch_update_and_is_reproducible(fa):
    """BASELINE config 3's update shape -- 3v3 x 4096 envs x 128 steps, 32 minibatches of 16 384 samples (781 tiles
    per launch) -- one epoch: the fused step against PyTorch autograd from the same rollout and permutations, and
    the fused update twice from the same state (bitwise: no atomics, fixed-order reductions, also with the two teams'
    launches overlapping)."""
    res = []
    for backend in ("torch", "fused", "fused"):
        torch.manual_seed(1)
        eng = fa.BatchedFortAttack(4096, 3, 3, 100, base_seed=0)
        L = fa.BatchedLearner(eng, num_steps=128, num_mini_batch=32, ppo_epoch=1, use_graph=True, update_backend=backend)
        L.reset()
        L.collect()
        torch.manual_seed(9)
        losses = L.update()
        torch.cuda.synchronize()
        res.append((losses.clone(), [p.detach().clone() for pol in L.policies for p in pol.parameters()]))
        del L, eng
    (l_t, p_t), (l_f, p_f), (l_g, p_g) = res
    assert torch.equal(l_f, l_g) and all(torch.equal(a, b) for a, b in zip(p_f, p_g))
    assert bool(torch.isfinite(l_f).all())
    assert (l_t - l_f).abs().max() < 2e-4 * max(1.0, float(l_t.abs().max())), (l_t, l_f)
    worst = max(float((x - y).abs().max()) for x, y in zip(p_t, p_f))
    print("max parameter distance fused vs torch after 32 steps at 16 384 x 3 samples: %.2e" % worst)
    assert worst < 1.5e-3                           # 32 Adam steps of 1e-4, signs of near-zero gradients may differ


def test_detached_parameters_are_refused(fa):
    """The fused kernels read the policies through mpnn_pack.FlatPolicy's flat buffer: re-pointing a parameter
    (instead of loading in place) must fail loudly, not run the rollout on stale weights."""
    eng = fa.BatchedFortAttack(64, 3, 3, 10)
    L = fa.BatchedLearner(eng, num_steps=8, num_mini_batch=2, ppo_epoch=1, use_graph=True)
    L.reset()
    L.collect()
    sd = {k: v.clone() + 0.01 for k, v in L.policies[0].state_dict().items()}
    L.policies[0].load_state_dict(sd)                       # in place: fine
    L.collect()
    p = next(L.policies[0].parameters())
    p.data = p.data.clone()                                 # re-pointed
    with pytest.raises(RuntimeError, match="flat buffer"):
        L.collect()


def test_reference_sampling_mode_replays_the_cpu_generator(fa):
    """BatchedLearner(reference_sampling=True): the minibatch index sets come from torch.randperm on the CPU's default
    generator in the reference's order (ppo.py:213; guards' epochs, then attackers'), the optimizer steps still replay from the
    fused graphs.  Same seed -> same update, bit for bit; the update consumes the CPU generator; and it equals the update
    driven through the `sampler` hook with the very same index sets (PyTorch-autograd path) to the fused-vs-torch tolerance."""
    res = []
    for mode in ("reference", "reference", "hook"):
        torch.manual_seed(4)
        eng = fa.BatchedFortAttack(256, 3, 3, 12, base_seed=6)
        L = fa.BatchedLearner(eng, num_steps=16, num_mini_batch=4, ppo_epoch=2, use_graph=True, update_backend="fused",
                              reference_sampling=(mode == "reference"), lr=1e-4)
        L.reset()
        L.collect()
        torch.manual_seed(99)
        before = torch.get_rng_state().clone()
        if mode == "reference":
            losses = L.update()
            assert any(isinstance(k, tuple) for k in L._update_graphs)      # the captured fused steps ran
        else:
            hook = L._reference_sampler()                                    # the same draws, through the eager autograd path
            losses = L.update(sampler=hook)
        torch.cuda.synchronize()
        assert not torch.equal(before, torch.get_rng_state())               # 2 teams x 2 epochs of randperm on the CPU generator
        res.append((losses.cpu(), [p.detach().cpu().clone() for pol in L.policies for p in pol.parameters()]))
    (l0, p0), (l1, p1), (l2, p2) = res
    assert torch.equal(l0, l1) and all(torch.equal(a, b) for a, b in zip(p0, p1))
    assert (l0 - l2).abs().max() < 2e-4 * max(1.0, float(l0.abs().max()))
    assert max(float((x - y).abs().max()) for x, y in zip(p0, p2)) < 4e-4   # 8 Adam steps of 1e-4
