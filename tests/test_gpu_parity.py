"""GPU parity: the HIP engine (through the C ABI) against the golden vectors generated
from the reference and against the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): bit-exact alive / hit / done masks; float obs / rewards
within 1e-5.  The fp64 state is in fact compared at FLOAT_TOL = 1e-9 here (the dynamics
path is +,-,*,/,sqrt in the reference's order, so it is normally bit-identical; only
device cos/sin/exp/log1p may differ from glibc in the last ulp, which enters through the
laser test and a 7.5e-8-wide contact band).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

FLOAT_TOL = 1e-9


@pytest.fixture(scope="module")
def fa():
    import emergent_multiagent_strategies_amd as m
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    m._lib.load()  # fail loudly if the HIP library is not built
    return m


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def _dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t if dtype is None else t.to(dtype)


def test_device_mt19937_stream(fa, golden_dir):
    g = _load(golden_dir, "mt19937_kat")
    for i, s in enumerate(g["seeds"]):
        eng = fa.BatchedFortAttack(1, 1, 1, 10, base_seed=int(s), skip_doubles=0)
        assert np.array_equal(eng.rng_peek(0, 8), g["first"][i])
        assert np.array_equal(eng.rng_peek(0, 708)[700:], g["after_700"][i])
    # the construction skip advances the device-side stream
    eng = fa.BatchedFortAttack(3, 1, 1, 10, base_seed=121, skip_doubles=700)
    assert np.array_equal(eng.rng_peek(2, 8), g["after_700"][2])  # env 2 <-> seed 123


@pytest.mark.parametrize("name", ["env_3v3", "env_5v5", "env_2v4", "env_1v1", "env_3v3_long",
                                  "env_5v5_long"])
def test_env_golden(fa, golden_dir, name):
    g = _load(golden_dir, name)
    G, A, max_t, T, E, base_seed, skip = [int(v) for v in g["meta"]]
    eng = fa.BatchedFortAttack(E, G, A, max_t, base_seed=base_seed, skip_doubles=skip)
    obs64 = torch.empty((E, G + A, 6), dtype=torch.float64, device="cuda")
    eng.reset(obs_f64=obs64)
    assert np.array_equal(obs64.cpu().numpy(), g["obs0"])
    term = {tuple(ix): k for k, ix in enumerate(g["term_idx"])}
    want = ("obs_f64", "reward_f64", "mask_f32", "done", "hit", "was_hit", "obs_f32", "reward_f32")
    worst = 0.0
    for t in range(T):
        out = eng.step(_dev(g["actions"][t], torch.int64), auto_reset=False, want=want)
        o = {k: v.cpu().numpy() for k, v in out.items()}
        # masks: bit exact
        assert np.array_equal(o["done"], g["done"][t]), t
        assert np.array_equal(o["mask_f32"], g["alive_before"][t].astype(np.float32)), t
        assert np.array_equal(o["obs_f64"][:, :, 0], g["alive_after"][t].astype(np.float64)), t
        assert np.array_equal(o["hit"], g["hit"][t]), t
        assert np.array_equal(o["was_hit"], g["was_hit"][t]), t
        # floats
        worst = max(worst, np.abs(o["reward_f64"] - g["reward"][t]).max())
        assert np.array_equal(o["obs_f32"], o["obs_f64"].astype(np.float32))
        assert np.array_equal(o["reward_f32"], o["reward_f64"].astype(np.float32))
        obs = o["obs_f64"]
        d = o["done"].astype(bool)
        if d.any():
            gr = eng.get_state()["game_result"]
            for e in np.nonzero(d)[0]:
                worst = max(worst, np.abs(obs[e] - g["term_obs"][term[(t, e)]]).max())
                assert np.array_equal(gr[e], g["game_result"][t, e])
            eng.reset(env_mask=_dev(d.astype(np.uint8)), obs_f64=out["obs_f64"])
            obs = out["obs_f64"].cpu().numpy()
        worst = max(worst, np.abs(obs.reshape(E, -1).sum(1) - g["obs_sum"][t]).max() / (6 * (G + A)))
        if "obs" in g.files:
            worst = max(worst, np.abs(obs - g["obs"][t]).max())
    s = eng.get_state()
    assert np.array_equal(np.isnan(s["prev_dist"]), np.isnan(g["final_prev_dist"]))
    worst = max(worst, np.nanmax(np.abs(s["prev_dist"] - g["final_prev_dist"]), initial=0.0))
    assert np.array_equal(s["num_hit"], g["final_num_hit"])
    assert np.array_equal(s["num_was_hit"], g["final_num_was_hit"])
    assert np.array_equal(s["time_step"], g["final_time_step"])
    print("%s: max abs float deviation from the reference = %.3e" % (name, worst))
    assert worst <= FLOAT_TOL


def test_env_golden_auto_reset(fa, golden_dir):
    g = _load(golden_dir, "env_3v3")
    G, A, max_t, T, E, base_seed, skip = [int(v) for v in g["meta"]]
    eng = fa.BatchedFortAttack(E, G, A, max_t, base_seed=base_seed, skip_doubles=skip)
    eng.reset()
    for t in range(T):
        out = eng.step(_dev(g["actions"][t], torch.int64), auto_reset=True, want=("obs_f64", "done"))
        assert np.array_equal(out["done"].cpu().numpy(), g["done"][t])
        assert np.abs(out["obs_f64"].cpu().numpy() - g["obs"][t]).max() <= FLOAT_TOL, t


@pytest.mark.parametrize("G,A,E,T,max_t", [(3, 3, 1000, 160, 50), (5, 5, 333, 120, 40), (1, 4, 77, 60, 20),
                                           (8, 8, 50, 60, 25), (4, 2, 129, 80, 30),
                                           (3, 3, 1, 90, 12),       # a single env
                                           (3, 3, 11, 40, 1),       # every step ends an episode
                                           (15, 1, 9, 50, 20),      # 16 lanes per env, lopsided teams
                                           (1, 1, 65, 70, 15),      # 32 envs per wave
                                           (2, 5, 4097, 30, 10)])   # N=7: 9 envs per wave, ragged tail
def test_env_vs_oracle_random(fa, G, A, E, T, max_t):
    """Same seeds, same random actions, auto reset: HIP vs CPU oracle every step."""
    from fa_oracle import OracleEnv
    N = G + A
    rng = np.random.RandomState(G * 100 + A)
    orc = OracleEnv(E, G, A, max_t, base_seed=31337)
    eng = fa.BatchedFortAttack(E, G, A, max_t, base_seed=31337)
    o0 = torch.empty((E, N, 6), dtype=torch.float64, device="cuda")
    eng.reset(obs_f64=o0)
    assert np.array_equal(o0.cpu().numpy(), orc.reset())
    want = ("obs_f64", "reward_f64", "mask_f32", "done", "hit", "was_hit")
    n_float_mismatch, worst, deaths, ends = 0, 0.0, 0, 0
    for t in range(T):
        a = rng.randint(0, 8, size=(E, N))
        a = np.where(rng.rand(E, N) < 0.25, 7, a)  # shoot a lot
        ref = orc.step(a, auto_reset=True)
        out = {k: v.cpu().numpy() for k, v in eng.step(_dev(a, torch.int64), auto_reset=True, want=want).items()}
        assert np.array_equal(out["done"], ref["done"]), t
        want_mask = np.where(ref["done"][:, None] != 0, 1, ref["alive_before"]).astype(np.float32)
        assert np.array_equal(out["mask_f32"], want_mask), t
        assert np.array_equal(out["hit"], ref["hit"]), t
        assert np.array_equal(out["was_hit"], ref["was_hit"]), t
        assert np.array_equal(out["obs_f64"][:, :, 0], ref["obs"][:, :, 0]), t
        n_float_mismatch += int((out["obs_f64"] != ref["obs"]).sum() + (out["reward_f64"] != ref["reward"]).sum())
        worst = max(worst, np.abs(out["obs_f64"] - ref["obs"]).max(), np.abs(out["reward_f64"] - ref["reward"]).max())
        deaths += int(ref["was_hit"].sum())
        ends += int(ref["done"].sum())
    so, sg = orc.get_state(), eng.get_state()
    for k in ("alive", "time_step", "num_hit", "num_was_hit"):
        assert np.array_equal(so[k], sg[k]), k
    print("%dv%d E=%d T=%d: deaths=%d episodes=%d float mismatches=%d worst=%.3e" % (
        G, A, E, T, deaths, ends, n_float_mismatch, worst))
    assert ends > 0 and (deaths > 0 or T * E < 5000)
    assert worst <= FLOAT_TOL


@pytest.mark.parametrize("G,A,E,T,max_t", [(3, 3, 333, 96, 30), (5, 5, 100, 64, 25),
                                           (3, 3, 41, 40, 1),      # every step ends an episode: a reset draw per step
                                           (5, 5, 13, 33, 2),
                                           (3, 3, 1, 50, 9),       # one env, 54 idle lanes
                                           (4, 2, 60, 24, 10)])    # run-time team sizes: the generic kernel
def test_fused_launch_vs_oracle(fa, G, A, E, T, max_t):
    """T env-steps in ONE fa_step launch (3v3 / 5v5: the pipelined multi-wave kernel, whose helper
    waves draw resets ahead, emit the rows one step late and test the laser in the shooter's frame)
    against the CPU oracle stepped T times: flags bit-exact, fp64 values within FLOAT_TOL, and the
    world + RNG state after the launch."""
    from fa_oracle import OracleEnv
    N = G + A
    rng = np.random.RandomState(7 * G + A)
    orc = OracleEnv(E, G, A, max_t, base_seed=2024)
    eng = fa.BatchedFortAttack(E, G, A, max_t, base_seed=2024)
    if (G, A) in ((3, 3), (5, 5)):
        assert eng.step_variant(T) == "fa_step_pipe_kernel"
    o0 = torch.empty((E, N, 6), dtype=torch.float64, device="cuda")
    eng.reset(obs_f64=o0)
    assert np.array_equal(o0.cpu().numpy(), orc.reset())
    acts = rng.randint(0, 8, size=(T, E, N))
    acts = np.where(rng.rand(T, E, N) < 0.25, 7, acts)
    out = {k: v.cpu().numpy() for k, v in eng.step_many(
        _dev(acts, torch.int64), auto_reset=True,
        want=("obs_f64", "reward_f64", "mask_f32", "done", "hit", "was_hit", "obs_f32", "reward_f32")).items()}
    worst, ends = 0.0, 0
    for t in range(T):
        ref = orc.step(acts[t], auto_reset=True)
        assert np.array_equal(out["done"][t], ref["done"]), t
        want_mask = np.where(ref["done"][:, None] != 0, 1, ref["alive_before"]).astype(np.float32)
        assert np.array_equal(out["mask_f32"][t], want_mask), t
        assert np.array_equal(out["hit"][t], ref["hit"]), t
        assert np.array_equal(out["was_hit"][t], ref["was_hit"]), t
        assert np.array_equal(out["obs_f64"][t][:, :, 0], ref["obs"][:, :, 0]), t
        worst = max(worst, np.abs(out["obs_f64"][t] - ref["obs"]).max(), np.abs(out["reward_f64"][t] - ref["reward"]).max())
        assert np.array_equal(out["obs_f32"][t], out["obs_f64"][t].astype(np.float32)), t
        assert np.array_equal(out["reward_f32"][t], out["reward_f64"][t].astype(np.float32)), t
        ends += int(ref["done"].sum())
    assert ends > 0 and worst <= FLOAT_TOL
    so, sg = orc.get_state(), eng.get_state()
    for k in ("alive", "time_step", "num_hit", "num_was_hit"):
        assert np.array_equal(so[k], sg[k]), k
    for k in ("pos_x", "pos_y", "vel_x", "vel_y", "ang"):
        assert np.abs(so[k] - sg[k]).max() <= FLOAT_TOL, k
    # the reset stream is where the oracle's is: the next reset draws the same positions
    assert np.array_equal(eng.rng_peek(E - 1, 2 * N), orc.rng_doubles(E - 1, 2 * N))


def test_fused_launch_without_auto_reset_vs_oracle(fa):
    """auto_reset = 0 through the pipelined kernel: finished envs keep stepping (dead agents frozen,
    done raised every step after the first), nothing is drawn from the reset stream."""
    from fa_oracle import OracleEnv
    G, A, E, T, max_t = 3, 3, 97, 40, 12
    N = G + A
    rng = np.random.RandomState(3)
    orc = OracleEnv(E, G, A, max_t, base_seed=5)
    eng = fa.BatchedFortAttack(E, G, A, max_t, base_seed=5)
    o0 = torch.empty((E, N, 6), dtype=torch.float64, device="cuda")
    eng.reset(obs_f64=o0)
    assert np.array_equal(o0.cpu().numpy(), orc.reset())
    peek = eng.rng_peek(0, 4)
    acts = np.where(rng.rand(T, E, N) < 0.3, 7, rng.randint(0, 8, size=(T, E, N)))
    assert eng.step_variant(T) == "fa_step_pipe_kernel"
    out = {k: v.cpu().numpy() for k, v in eng.step_many(_dev(acts, torch.int64), auto_reset=False,
                                                        want=("obs_f64", "reward_f64", "mask_f32", "done", "hit", "was_hit")).items()}
    for t in range(T):
        ref = orc.step(acts[t], auto_reset=False)
        assert np.array_equal(out["done"][t], ref["done"]), t
        assert np.array_equal(out["mask_f32"][t], ref["alive_before"].astype(np.float32)), t
        assert np.array_equal(out["hit"][t], ref["hit"]) and np.array_equal(out["was_hit"][t], ref["was_hit"]), t
        assert np.abs(out["obs_f64"][t] - ref["obs"]).max() <= FLOAT_TOL and np.abs(out["reward_f64"][t] - ref["reward"]).max() <= FLOAT_TOL
    assert np.array_equal(eng.rng_peek(0, 4), peek)


def test_shards_equal_one_big_batch(fa):
    """Multi-GPU sharding by env_offset: two half handles == one full handle, bit for bit."""
    E, G, A, T = 512, 3, 3, 80
    rng = np.random.RandomState(1)
    full = fa.BatchedFortAttack(E, G, A, 30, base_seed=7)
    lo = fa.BatchedFortAttack(E // 2, G, A, 30, base_seed=7, env_offset=0)
    hi = fa.BatchedFortAttack(E // 2, G, A, 30, base_seed=7, env_offset=E // 2)
    of = full.reset()
    assert torch.equal(of, torch.cat([lo.reset(), hi.reset()]))
    for t in range(T):
        a = _dev(rng.randint(0, 8, size=(E, G + A)), torch.int64)
        f = full.step(a)
        l, h = lo.step(a[:E // 2].contiguous()), hi.step(a[E // 2:])  # hi: non-zero storage offset view
        for k in f:
            assert torch.equal(f[k], torch.cat([l[k], h[k]])), (t, k)


def test_action_strides(fa):
    """Agent-major action tensors (the reference's cat/chunk layout, learner.py:164-170) work."""
    E, G, A = 64, 3, 3
    rng = np.random.RandomState(2)
    e1 = fa.BatchedFortAttack(E, G, A, 30, base_seed=3)
    e2 = fa.BatchedFortAttack(E, G, A, 30, base_seed=3)
    e1.reset(), e2.reset()
    for t in range(40):
        a = _dev(rng.randint(0, 8, size=(E, G + A)), torch.int64)
        a_agent_major = a.t().contiguous().t()  # shape (E,N), strides (1, E)
        assert a_agent_major.stride() == (1, E)
        o1, o2 = e1.step(a), e2.step(a_agent_major)
        for k in o1:
            assert torch.equal(o1[k], o2[k])


def test_philox_mode_properties(fa):
    E, G, A = 256, 3, 3
    eng = fa.BatchedFortAttack(E, G, A, 20, base_seed=11, rng="philox")
    o1 = eng.reset().cpu().numpy()
    o2 = eng.reset().cpu().numpy()
    assert not np.array_equal(o1, o2)           # the counter advances
    for o in (o1, o2):                          # spawn boxes of fortattack_env_v1.py:66,70
        assert (np.abs(o[:, :G, 1]) <= 0.06).all() and (o[:, :G, 2] >= 0.64).all() and (o[:, :G, 2] <= 0.8).all()
        assert (np.abs(o[:, G:, 1]) <= 1).all() and (o[:, G:, 2] >= -0.8).all() and (o[:, G:, 2] <= -0.64).all()
    again = fa.BatchedFortAttack(E, G, A, 20, base_seed=11, rng="philox").reset().cpu().numpy()
    assert np.array_equal(again, o1)            # reproducible


def test_facade_matches_survey_known_answers(fa):
    """make_fortattack_env / env.reset / env.step with the reference's call shapes
    (np.random.seed(123), 5v5, actions a[t][i] = (t + 3 i) % 8)."""
    env = fa.make_fortattack_env(100, seed=123, skip_doubles=20)
    assert env.n == 10 and env.action_spaces[0].shape[0] == 8 and env.ob_rms is None
    assert [a.attacker for a in env.world.policy_agents] == [False] * 5 + [True] * 5
    obs = env.reset()
    assert obs.shape == (10, 6) and obs.dtype == np.float64
    assert obs[0].tolist() == [1.0, 0.016128115026158532, 0.7759090870524463, 4.71238898038469, 0.0, 0.0]
    assert obs[5].tolist() == [1.0, -0.8157901201098496, -0.7306078123712756, 1.5707963267948966, 0.0, 0.0]
    for t in range(100):
        obs, rew, done, info = env.step(np.array([(t + 3 * i) % 8 for i in range(10)]))
        assert isinstance(rew, list) and len(rew) == 10 and isinstance(done, bool) and len(info["n"]) == 10
        if t == 4:
            assert np.abs(obs[0] - [1.0, 0.08002140384008218, 0.6181297941142567, 4.71238898038469,
                                    0.010952066503557161, -0.5912904061522642]).max() <= FLOAT_TOL
            assert abs(rew[9] - (-1.0054530212861503)) <= FLOAT_TOL
    assert done and env.world.gameResult.tolist() == [0, 1, 0]
    assert obs[:, 0].tolist() == [1, 1, 1, 1, 1, 1, 1, 1, 0, 1]
    assert abs(obs.sum() - 756.049492039794) < 1e-6
    assert env.world.numAliveAttackers == 4 and env.world.numGuards == 5
    with pytest.raises(AssertionError):
        env.step(np.zeros(3, dtype=np.int64))


def test_error_reporting(fa):
    with pytest.raises(fa.FaError):
        fa.BatchedFortAttack(0, 3, 3, 10)
    with pytest.raises(fa.FaError):
        fa.BatchedFortAttack(4, 9, 9, 10)       # > 16 agents
    eng = fa.BatchedFortAttack(4, 3, 3, 10)
    with pytest.raises(fa.FaError):
        eng.collect_step(0)                      # no storage bound


def test_full_size_invariants(fa):
    """BASELINE config 2 size (4096 envs x 128 steps, 3v3): size-independent properties."""
    E, G, A, T = 4096, 3, 3, 128
    N = G + A
    eng = fa.BatchedFortAttack(E, G, A, 100, base_seed=0)
    gen = torch.Generator(device="cuda").manual_seed(0)
    prev = eng.reset(obs_f64=torch.empty((E, N, 6), dtype=torch.float64, device="cuda")).clone()
    assert bool((prev[:, :G, 3] == 3 * np.pi / 2).all()) and bool((prev[:, G:, 3] == np.pi / 2).all())
    episodes = 0
    for t in range(T):
        a = torch.randint(0, 8, (E, N), device="cuda", generator=gen)
        out = eng.step(a, auto_reset=True, want=("obs_f64", "mask_f32", "done", "was_hit"))
        cur, alive_before = out["obs_f64"], out["mask_f32"] > 0
        keep = (out["done"] == 0)[:, None] & alive_before & (cur[:, :, 0] == 1)   # alive, no reset
        res = cur[:, :, 1:3] - prev[:, :, 1:3] - 0.1 * cur[:, :, 4:6]
        assert float(res[keep].abs().max()) <= 1e-15        # pos += vel*dt
        dang = (cur[:, :, 3] - prev[:, :, 3])[keep]
        ok = (dang.abs() < 1e-12) | ((dang - 0.17).abs() < 1e-12) | ((dang - 6.113185307179586).abs() < 1e-12)
        assert bool(ok.all())
        assert float(torch.hypot(cur[:, :, 4], cur[:, :, 5]).max()) <= 3 + 1e-12
        frozen = (out["done"] == 0)[:, None] & ~alive_before
        assert torch.equal(cur[frozen], prev[frozen])         # dead rows freeze
        died = (out["done"] == 0)[:, None] & alive_before & (cur[:, :, 0] == 0)
        assert torch.equal(died, (out["done"] == 0)[:, None] & (out["was_hit"] == 1))
        episodes += int(out["done"].sum())
        prev = cur.clone()
    s = eng.get_state()
    assert int(s["result_count"].sum()) == episodes and episodes >= E  # every env finished >= 1 episode


def test_eval_stats_match_oracle_bookkeeping(fa):
    """Device-side evaluation counters (fa_state_host.episode_reward_sum / alive_at_end /
    result_count) == the reference's per-episode bookkeeping (test_fortattack_v2.py:88-101)
    done on the host from the oracle's per-step outputs."""
    from fa_oracle import OracleEnv
    E, G, A, T, max_t = 300, 3, 3, 150, 30
    N = G + A
    rng = np.random.RandomState(9)
    orc = OracleEnv(E, G, A, max_t, base_seed=77)
    eng = fa.BatchedFortAttack(E, G, A, max_t, base_seed=77)
    eng.reset(), orc.reset()
    ep_rew = np.zeros((E, N))
    rows = []
    for t in range(T):
        a = np.where(rng.rand(E, N) < 0.3, 7, rng.randint(0, 8, size=(E, N)))
        eng.step(_dev(a, torch.int64), auto_reset=True, want=("done",))
        ref = orc.step(a, auto_reset=False)
        ep_rew += ref["reward"] * ref["alive_before"]
        d = ref["done"].astype(bool)
        if d.any():
            gr = orc.get_state()["game_result"]
            alive = ref["obs"][:, :, 0]
            for e in np.nonzero(d)[0]:
                rows.append([gr[e, 0], gr[e, 1], gr[e, 0] + gr[e, 1], gr[e, 2], alive[e, :G].sum(),
                             alive[e, G:].sum(), ep_rew[e, :G].mean(), ep_rew[e, G:].mean()])
                ep_rew[e] = 0
            orc.reset(mask=d)
    want = np.array(rows, dtype=np.float64).mean(0)
    got, n_ep = eng.eval_stats()
    assert n_ep == len(rows) and n_ep > 500
    assert np.abs(got - want).max() < 1e-9


def test_device_divide_and_sqrt_sequences_are_correctly_rounded(fa):
    """The step kernel's wrapper-free fp64 divide / sqrt sequences == the compiler's `/` and
    sqrt() bit for bit on 2^27 random operands (magnitudes 1e-17..1e11, plus divisor 1e-10)."""
    eng = fa.BatchedFortAttack(64, 3, 3, 10)
    bad_div, bad_sqrt, sincos_ulp = eng.selftest_math(samples=1 << 27, seed=12345)
    print("heading sin/cos: max deviation from the device libm = %.3f ulp" % sincos_ulp)
    assert (bad_div, bad_sqrt) == (0, 0)
    assert sincos_ulp <= 2.0


def test_large_batch_matches_oracle(fa):
    """131 072 envs (13 108 workgroups, > 2^16 blocks' worth of lanes): every env of a big batch
    equals the oracle -- catches index-width / grid-size mistakes that small batches cannot."""
    from fa_oracle import OracleEnv
    E, G, A, T = 131072, 3, 3, 6
    N = G + A
    rng = np.random.RandomState(11)
    orc = OracleEnv(E, G, A, 4, base_seed=900000)          # max_time_steps 4: resets inside the run
    eng = fa.BatchedFortAttack(E, G, A, 4, base_seed=900000)
    o0 = torch.empty((E, N, 6), dtype=torch.float64, device="cuda")
    eng.reset(obs_f64=o0)
    assert np.array_equal(o0.cpu().numpy(), orc.reset())
    for t in range(T):
        a = rng.randint(0, 8, size=(E, N))
        ref = orc.step(a, auto_reset=True)
        out = eng.step(_dev(a, torch.int64), auto_reset=True, want=("obs_f64", "reward_f64", "done"))
        assert np.array_equal(out["done"].cpu().numpy(), ref["done"]), t
        assert np.array_equal(out["obs_f64"].cpu().numpy(), ref["obs"]), t
        assert np.array_equal(out["reward_f64"].cpu().numpy(), ref["reward"]), t


@pytest.mark.parametrize("kernel", ["auto", "waves1", "waves2", "waves3"])
def test_reset_choice_interleaving_golden(fa, golden_dir, kernel):
    """Ensemble path (quirk Q14): np.random.choice(attacker_ckpts) after every env.reset() on the env's own
    stream (learner.py:119-121, train_fortattack_v2.py:29-35,104-111) -- the engine with
    fa_set_reset_choice against the capture of the reference env driven that way: chosen indices and
    every observation (hence every later reset position) bit for bit."""
    g = _load(golden_dir, "env_choice_5v5")
    G, A, max_t, T, E, base_seed, skip, K = [int(v) for v in g["meta"]]
    eng = fa.BatchedFortAttack(E, G, A, max_t, base_seed=base_seed, skip_doubles=skip, step_kernel=kernel)
    choice = eng.set_reset_choice(K)
    obs64 = torch.empty((E, G + A, 6), dtype=torch.float64, device="cuda")
    eng.reset(obs_f64=obs64)
    assert np.array_equal(obs64.cpu().numpy(), g["obs0"]) and np.array_equal(choice.cpu().numpy(), g["choice0"])
    n = 0
    for t in range(T):
        out = eng.step(_dev(g["actions"][t], torch.int64), auto_reset=True, want=("obs_f64", "reward_f64", "done"))
        assert np.array_equal(out["done"].cpu().numpy(), g["done"][t]), t
        assert np.abs(out["obs_f64"].cpu().numpy() - g["obs"][t]).max() <= FLOAT_TOL, t
        assert np.abs(out["reward_f64"].cpu().numpy() - g["reward"][t]).max() <= FLOAT_TOL, t
        d = g["done"][t] != 0
        assert np.array_equal(choice.cpu().numpy()[d], g["choice"][t][d]), t
        n += int(d.sum())
    assert n >= 10
    # a rollout launch with a choice pending uses fa_step_kernel, never the draw-ahead pipelined kernel
    assert "pipe" not in eng.step_variant(64)
    eng.set_reset_choice(0)
    assert eng.step_variant(64) == ("fa_step_pipe_kernel" if kernel == "auto" else eng.step_variant(64))


def test_facade_default_draws_from_the_global_numpy_stream(fa, golden_dir):
    """make_fortattack_env(num_steps) with NO extra arguments, after the script's np.random.seed(seed)
    (train_fortattack.py:200): construction and every reset consume numpy's global stream exactly as the
    reference does -- the golden capture of the reference 5v5 env (seed 123) is reproduced, reward types
    included (int 0 for agents that are neither alive nor justDied, fortattack_env_v1.py:87-92)."""
    g = _load(golden_dir, "env_5v5")
    G, A, max_t, T, E, base_seed, skip = [int(v) for v in g["meta"]]
    assert (G, A) == (5, 5)
    np.random.seed(base_seed)                                   # env 0 of the fixture <-> seed base_seed
    env = fa.make_fortattack_env(max_t)
    obs = env.reset()
    assert obs.dtype == np.float64 and np.array_equal(obs, g["obs0"][0])
    alive_before = obs[:, 0] != 0
    ints = 0
    for t in range(T):
        obs, rew, done, info = env.step(g["actions"][t, 0].astype(np.int64))
        assert done == bool(g["done"][t, 0])
        assert np.abs(np.array(rew, dtype=np.float64) - g["reward"][t, 0]).max() <= FLOAT_TOL
        for i in range(G + A):
            assert (type(rew[i]) is int and rew[i] == 0) == (not alive_before[i]), (t, i)
            ints += type(rew[i]) is int
        if done:
            assert env.world.gameResult.tolist() == g["game_result"][t, 0].tolist()
            obs = env.reset()
            assert env.world.gameResult.tolist() == [0, 0, 0]  # reset_world clears it (fortattack_env_v1.py:54)
        assert np.abs(obs - g["obs"][t, 0]).max() <= FLOAT_TOL, t
        alive_before = obs[:, 0] != 0
    assert ints > 0 and int(g["done"][:, 0].sum()) >= 1
    # the global stream is where the reference's would be: the next draw equals a reference-side draw
    np.random.seed(base_seed)
    np.random.random_sample(2 * (G + A) * (2 + int(g["done"][:, 0].sum())))   # construction + first reset + one per episode
    expect = np.random.random_sample()
    np.random.seed(base_seed)
    env2 = fa.make_fortattack_env(max_t)
    env2.reset()
    for t in range(T):
        _, _, d, _ = env2.step(g["actions"][t, 0].astype(np.int64))
        if d:
            env2.reset()
    assert np.random.random_sample() == expect
