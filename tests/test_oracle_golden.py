"""CPU: the oracle (oracle/fa_oracle.c, oracle/collector_oracle.py) against the golden
vectors generated from the reference itself (oracle/gen_golden.py).

Bar: bit-exact on every float64 observation / reward and on every alive / hit / done /
gameResult flag; float32 storage tensors bit-exact; normalised
advantages within 1e-6 (torch's float32 mean/std reduction order is not restated).
"""
import os

import numpy as np
import pytest

from fa_oracle import OracleEnv
import collector_oracle as co


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def test_mt19937_known_answers(golden_dir):
    g = _load(golden_dir, "mt19937_kat")
    for i, s in enumerate(g["seeds"]):
        o = OracleEnv(1, 1, 1, 10, base_seed=int(s), skip_doubles=0)
        assert np.array_equal(o.rng_doubles(0, 8), g["first"][i])
        o.rng_doubles(0, 700 - 8)
        assert np.array_equal(o.rng_doubles(0, 8), g["after_700"][i])
    # SURVEY.md B.2 known answer (numpy 2.2.6): seed 123
    o = OracleEnv(1, 1, 1, 10, base_seed=123, skip_doubles=0)
    assert o.rng_doubles(0, 2).tolist() == [0.6964691855978616, 0.28613933495037946]


@pytest.mark.parametrize("name", ["env_3v3", "env_5v5", "env_2v4", "env_1v1", "env_3v3_long",
                                  "env_5v5_long"])
def test_env_trajectories_bit_exact(golden_dir, name):
    g = _load(golden_dir, name)
    G, A, max_t, T, E, base_seed, skip = [int(v) for v in g["meta"]]
    o = OracleEnv(E, G, A, max_t, base_seed=base_seed, skip_doubles=skip)
    assert np.array_equal(o.reset(), g["obs0"])
    term = {tuple(ix): k for k, ix in enumerate(g["term_idx"])}
    for t in range(T):
        out = o.step(g["actions"][t], auto_reset=False)
        assert np.array_equal(out["reward"], g["reward"][t]), t
        assert np.array_equal(out["done"], g["done"][t]), t
        assert np.array_equal(out["alive_before"], g["alive_before"][t]), t
        assert np.array_equal(out["obs"][:, :, 0].astype(np.uint8), g["alive_after"][t]), t
        assert np.array_equal(out["hit"], g["hit"][t]), t
        assert np.array_equal(out["was_hit"], g["was_hit"][t]), t
        obs = out["obs"]
        d = out["done"].astype(bool)
        if d.any():
            gr = o.get_state()["game_result"]
            for e in np.nonzero(d)[0]:
                assert np.array_equal(obs[e], g["term_obs"][term[(t, e)]])
                assert np.array_equal(gr[e], g["game_result"][t, e])
            obs = np.where(d[:, None, None], o.reset(mask=d), obs)
        assert np.array_equal(obs.reshape(E, -1).sum(1), g["obs_sum"][t]) or \
            np.allclose(obs.reshape(E, -1).sum(1), g["obs_sum"][t], rtol=0, atol=1e-12)
        if "obs" in g.files:
            assert np.array_equal(obs, g["obs"][t]), t
    s = o.get_state()
    assert np.array_equal(s["prev_dist"], g["final_prev_dist"], equal_nan=True)
    assert np.array_equal(s["num_hit"], g["final_num_hit"])
    assert np.array_equal(s["num_was_hit"], g["final_num_was_hit"])
    assert np.array_equal(s["time_step"], g["final_time_step"])


def test_env_auto_reset_equals_manual_reset(golden_dir):
    g = _load(golden_dir, "env_3v3")
    G, A, max_t, T, E, base_seed, skip = [int(v) for v in g["meta"]]
    o = OracleEnv(E, G, A, max_t, base_seed=base_seed, skip_doubles=skip)
    o.reset()
    for t in range(T):
        out = o.step(g["actions"][t], auto_reset=True)
        assert np.array_equal(out["obs"], g["obs"][t]), t


def test_survey_known_answer_5v5():
    """np.random.seed(123), 5v5, actions a[t][i] = (t + 3 i) % 8."""
    o = OracleEnv(1, 5, 5, 100, base_seed=123, skip_doubles=20)
    obs = o.reset()[0]
    assert obs[0].tolist() == [1.0, 0.016128115026158532, 0.7759090870524463, 4.71238898038469, 0.0, 0.0]
    assert obs[5].tolist() == [1.0, -0.8157901201098496, -0.7306078123712756, 1.5707963267948966, 0.0, 0.0]
    rsum = np.zeros(10)
    for t in range(100):
        out = o.step(np.array([[(t + 3 * i) % 8 for i in range(10)]]))
        rsum += out["reward"][0]
        if t == 4:
            assert out["obs"][0, 0].tolist() == [1.0, 0.08002140384008218, 0.6181297941142567,
                                                 4.71238898038469, 0.010952066503557161,
                                                 -0.5912904061522642]
            assert out["reward"][0, 9] == -1.0054530212861503
    assert out["done"][0] == 1
    assert o.get_state()["game_result"][0].tolist() == [0, 1, 0]
    assert out["obs"][0, :, 0].tolist() == [1, 1, 1, 1, 1, 1, 1, 1, 0, 1]
    assert abs(out["obs"][0].sum() - 756.049492039794) < 1e-9
    assert abs(rsum[5] - (-12.968857569917889)) < 1e-12


def test_out_files_invariants():
    """The only result trace the reference ships (out_files/1.npy, 36x10x6) pins these
    invariants; check the oracle obeys them on its own rollout."""
    rng = np.random.RandomState(3)
    o = OracleEnv(8, 5, 5, 100, base_seed=50)
    prev = o.reset()
    assert np.all(prev[:, :5, 3] == 3 * np.pi / 2) and np.all(prev[:, 5:, 3] == np.pi / 2)
    for t in range(60):
        a = rng.randint(0, 8, size=(8, 10))
        out = o.step(a)
        cur = out["obs"]
        alive = cur[:, :, 0] == 1
        # pos[t+1] - pos[t] - 0.1 * vel[t+1] == 0 for alive agents
        res = cur[:, :, 1:3] - prev[:, :, 1:3] - 0.1 * cur[:, :, 4:6]
        assert np.abs(res[alive]).max() <= 1e-15
        dang = cur[:, :, 3] - prev[:, :, 3]
        ok = np.isclose(dang, 0, atol=1e-12) | np.isclose(dang, 0.17, atol=1e-12) | \
            np.isclose(dang, 6.113185307179586, atol=1e-12)
        assert ok[alive].all()
        assert (np.hypot(cur[:, :, 4], cur[:, :, 5]) <= 3 + 1e-12).all()
        dead_before = out["alive_before"] == 0
        assert np.array_equal(cur[dead_before], prev[dead_before])  # dead rows freeze
        if out["done"].any():
            break
        prev = cur


@pytest.mark.parametrize("name", ["collector_3v3",
                                  # the reference's own rollout length and team size (arguments.py:23, marlsave/tmp_2/params.json:
                                  # num_steps 1000, 5v5, 100-step episodes), the published ep1240 policies acting: 105 episode ends
                                  "collector_5v5_T1000"])
def test_collector_gae_advnorm(golden_dir, name):
    g = _load(golden_dir, name)
    compact = "after_obs" not in g
    G, A, max_t, T, n_upd, seed, skip = [int(v) for v in g["meta"]]
    N = G + A
    gamma, tau = [float(v) for v in g["gamma_tau"]]
    env = OracleEnv(1, G, A, max_t, base_seed=seed, skip_doubles=skip)
    st = [co.StorageOracle(T, 1) for _ in range(N)]
    obs = env.reset()[0]
    assert np.array_equal(obs, g["obs0"])
    for j in range(n_upd):
        for i in range(N):
            st[i].initialize_obs(obs[i].astype(np.float32))
        ep_start = np.zeros((T, 1), bool)
        for s in range(T):
            masks = obs[:, 0].astype(np.float32)
            out = env.step(g["actions"][j, s][None].astype(np.int64))
            obs = out["obs"][0]
            for i in range(N):
                st[i].insert(obs[i].astype(np.float32), 0.0,
                             np.array([[g["actions"][j, s, i]]], np.int64) if compact else g["actions_st"][j, i, s],
                             g["action_log_probs"][j, i, s], g["value_preds"][j, i, s],
                             np.float32(out["reward"][0, i]), masks[i])
            assert bool(out["done"][0]) == bool(g["done"][j, s])
            if out["done"][0]:
                obs = env.reset()[0]
                if s + 1 < T:
                    ep_start[s + 1] = True
                for i in range(N):
                    st[i].initialize_new_episode(s + 1, obs[i].astype(np.float32), np.float32(obs[i, 0]))
        end_pts = [int(v) for v in g["end_pts"][j] if v >= 0]
        for i in range(N):
            # (a) literal restatement of wrap_horizon / compute_returns
            lit = co.StorageOracle(T, 1)
            for k in ("obs", "rewards", "value_preds", "masks", "returns"):
                setattr(lit, k, getattr(st[i], k).copy())
            co.wrap_horizon(lit, end_pts, [g["next_values"][j, i, k] for k in range(len(end_pts))],
                            gamma, tau)
            # (b) single-pass batched equivalent (per-process episode boundaries)
            st[i].value_preds[T] = g["next_values"][j, i, len(end_pts) - 1]
            co.gae_single_pass(st[i].rewards, st[i].value_preds, st[i].masks, st[i].returns,
                               ep_start, gamma, tau)
            for name, s_ in (("literal", lit), ("single-pass", st[i])):
                for k in ("obs", "rewards", "masks", "returns"):
                    assert np.array_equal(getattr(s_, k), g[k][j, i]), (name, k, j, i)
            assert np.array_equal(st[i].value_preds, g["value_preds"][j, i])
            adv = co.normalized_advantages(st[i].returns, st[i].value_preds)
            assert np.abs(adv - g["adv"][j, i]).max() < 2e-6
        for i in range(N):
            st[i].after_update()
            if compact:
                assert np.array_equal(st[i].obs[0], g["after_obs_row0"][j, i]) and np.array_equal(st[i].masks[0], g["after_masks_row0"][j, i])
                assert float(np.abs(st[i].obs[1:]).sum()) == g["after_rest_abs_sum"][j, i] == 0.0
            else:
                assert np.array_equal(st[i].obs, g["after_obs"][j, i])
                assert np.array_equal(st[i].masks, g["after_masks"][j, i])


def test_oracle_rollout_equals_per_step_calls():
    """fao_rollout (T steps per env inside one OpenMP region: the cpu_baseline's form) == T fao_step
    calls: last-step outputs, world state and reset-stream position, bit for bit."""
    from fa_oracle import OracleEnv
    E, G, A, T, max_t = 96, 3, 3, 70, 11
    rng = np.random.RandomState(4)
    acts = np.where(rng.rand(T, E, G + A) < 0.3, 7, rng.randint(0, 8, size=(T, E, G + A))).astype(np.int64)
    a, b = OracleEnv(E, G, A, max_t, base_seed=9), OracleEnv(E, G, A, max_t, base_seed=9)
    a.reset(), b.reset()
    for t in range(T):
        ref = a.step(acts[t], auto_reset=True)
    obs, rew, done = b.rollout(acts, auto_reset=True)
    assert np.array_equal(obs, ref["obs"]) and np.array_equal(rew, ref["reward"]) and np.array_equal(done, ref["done"])
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k], equal_nan=True), k
    assert np.array_equal(a.rng_doubles(E - 1, 8), b.rng_doubles(E - 1, 8))


def test_oracle_reset_choice_interleaving_golden(golden_dir):
    """Ensemble path (quirk Q14): np.random.choice(attacker_ckpts) after every env.reset() on the same stream
    (learner.py:119-121, train_fortattack_v2.py:29-35,104-111) -- the oracle against the capture of the
    reference env driven that way: chosen checkpoint indices and every observation, bit for bit."""
    from fa_oracle import OracleEnv
    g = np.load(os.path.join(golden_dir, "env_choice_5v5.npz"))
    G, A, max_t, T, E, base_seed, skip, K = [int(v) for v in g["meta"]]
    orc = OracleEnv(E, G, A, max_t, base_seed=base_seed, skip_doubles=skip)
    orc.set_choice(K)
    assert np.array_equal(orc.reset(), g["obs0"]) and np.array_equal(orc.get_choice(), g["choice0"])
    n = 0
    for t in range(T):
        ref = orc.step(g["actions"][t].astype(np.int64), auto_reset=True)
        assert np.array_equal(ref["done"], g["done"][t]) and np.array_equal(ref["obs"], g["obs"][t]), t
        assert np.array_equal(ref["reward"], g["reward"][t]), t
        d = g["done"][t] != 0
        assert np.array_equal(orc.get_choice()[d], g["choice"][t][d]), t
        n += int(d.sum())
    assert n >= 10
