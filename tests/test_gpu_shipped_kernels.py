"""GPU parity of the EXACT kernel instantiations the bench and the trainers launch, against the
CPU oracle (not against another HIP variant):

  * fa_collect_rollout at BASELINE config 2's full size -- fa_step_pipe_kernel<3,3,COLLECT=true,2,2>,
    4096 envs x 128 steps -- storage rows vs the oracle stepped 128 times;
  * the three-workgroups-per-CU builds of the pipelined kernel (<= 168 VGPRs; grids of 513..768
    workgroups): 5v5 x 4096 x 128 (config 5's per-GPU shape) and 3v3 x 7680 x 64;
  * every step-kernel build (pipelined 2 / 3 per CU, fa_step_kernel with 1 / 2 / 3 waves, with and
    without the COLLECT specialisation) pinned through fa_config.step_kernel at a small size, so that
    the builds the grid-size heuristic only picks for large batches are also compared with the oracle;
  * the exactly-coincident-agents corner (pair force NaN in the reference; a partner shot in the same
    step is skipped by the reference).

Bar: done / masks / alive / hit flags bit exact; float rows within 1e-5 (observed: 0 differing
float32 values).
"""
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


def _shooty_actions(rng, shape, p_shoot=0.3):
    a = rng.randint(0, 8, size=shape)
    return np.where(rng.rand(*shape) < p_shoot, 7, a)


def _check_rows_vs_oracle(st, orc, acts, T):
    """storage rows obs[1:], rewards, masks[1:], done of a rollout vs the oracle stepped T times."""
    obs = st.obs.cpu().numpy()
    rew = st.rewards.cpu().numpy()[..., 0]
    msk = st.masks.cpu().numpy()[..., 0]
    done = st.done.cpu().numpy()
    n_f32_diff, worst, ends, deaths = 0, 0.0, 0, 0
    for s in range(T):
        ref = orc.step(acts[s], auto_reset=True)
        assert np.array_equal(done[s], ref["done"]), s
        want_mask = np.where(ref["done"][:, None] != 0, 1, ref["alive_before"]).astype(np.float32)
        assert np.array_equal(msk[s + 1], want_mask), s
        assert np.array_equal(obs[s + 1, :, :, 0], ref["obs"][:, :, 0].astype(np.float32)), s   # alive column
        o32, r32 = ref["obs"].astype(np.float32), ref["reward"].astype(np.float32)
        n_f32_diff += int((obs[s + 1] != o32).sum() + (rew[s] != r32).sum())
        worst = max(worst, float(np.abs(obs[s + 1] - ref["obs"]).max()), float(np.abs(rew[s] - ref["reward"]).max()))
        ends += int(ref["done"].sum())
        deaths += int(ref["was_hit"].sum())
    return n_f32_diff, worst, ends, deaths


@pytest.mark.parametrize("G,A,E,T,variant,kernel", [
    (3, 3, 4096, 128, "fa_step_pipe_kernel", "auto"),             # the bench's launch (config 2)
    (5, 5, 4096, 128, "fa_step_pipe_kernel/3 per CU", "auto"),    # config 5's per-GPU shape: 683 workgroups
    (3, 3, 7680, 64, "fa_step_pipe_kernel/3 per CU", "auto"),     # 768 workgroups: the largest pipelined grid
])
def test_collect_rollout_full_size_vs_oracle(fa, G, A, E, T, variant, kernel):
    from fa_oracle import OracleEnv
    N, max_t = G + A, 60
    rng = np.random.RandomState(G * 1000 + E)
    orc = OracleEnv(E, G, A, max_t, base_seed=4242)
    eng = fa.BatchedFortAttack(E, G, A, max_t, base_seed=4242, step_kernel=kernel)   # track_counters on, as the bench
    assert eng.step_variant(T) == variant
    st = fa.JointRolloutStorage(T, E, N, device="cuda")
    eng.bind_storage(st)
    eng.collect_reset()
    assert np.array_equal(st.obs[0].cpu().numpy(), orc.reset().astype(np.float32))
    acts = _shooty_actions(rng, (T, E, N))
    st.actions.copy_(torch.from_numpy(acts[..., None]).cuda())
    eng.collect_rollout(0, T)                                               # ONE launch, COLLECT rows
    torch.cuda.synchronize()
    n_diff, worst, ends, deaths = _check_rows_vs_oracle(st, orc, acts, T)
    print("%dv%d E=%d T=%d %s: episodes=%d deaths=%d differing f32 values=%d worst=%.2e" % (
        G, A, E, T, variant, ends, deaths, n_diff, worst))
    assert ends >= E and deaths > E // 4
    assert worst <= 1e-5 and n_diff == 0
    so, sg = orc.get_state(), eng.get_state()
    for k in ("alive", "time_step", "num_hit", "num_was_hit"):   # (game_result: the device keeps the last
        assert np.array_equal(so[k], sg[k]), k                   #  finished episode's, the oracle's reset clears it)
    for k in ("pos_x", "pos_y", "vel_x", "vel_y", "ang", "prev_dist"):
        assert np.array_equal(so[k], sg[k], equal_nan=True), k
    for e in (0, E // 2, E - 1):                                            # reset stream position
        assert np.array_equal(eng.rng_peek(e, 2 * N), orc.rng_doubles(e, 2 * N))


@pytest.mark.parametrize("G,A", [(3, 3), (5, 5)])
@pytest.mark.parametrize("kernel,name", [("pipe", "fa_step_pipe_kernel"), ("pipe3", "fa_step_pipe_kernel/3 per CU"),
                                         ("waves1", "fa_step_kernel/1 wave"), ("waves2", "fa_step_kernel/2 waves"),
                                         ("waves3", "fa_step_kernel/3 waves")])
@pytest.mark.parametrize("collect", [False, True])
def test_every_step_kernel_build_vs_oracle(fa, G, A, kernel, name, collect):
    """fa_config.step_kernel pins the build; T steps in one launch (and, for fa_step_kernel, also as T single-step
    launches) against the oracle.  (The two round-4 experiment kernels are not in the product library: the same test runs on
    them from tests/test_gpu_experiment_kernels.py against the variant library that carries them.)"""
    from fa_oracle import OracleEnv
    if kernel == "pairs" and (G, A) != (3, 3):
        pytest.skip("the pair-per-lane kernel exists for 3v3 only")
    E, T, max_t = 77, 40, 9
    N = G + A
    rng = np.random.RandomState(17 * G + len(kernel))
    acts = _shooty_actions(rng, (T, E, N), 0.25)
    single = kernel.startswith("waves") or kernel == "pairs"
    for per_step in ((False, True) if single else (False,)):
        orc = OracleEnv(E, G, A, max_t, base_seed=606)
        eng = fa.BatchedFortAttack(E, G, A, max_t, base_seed=606, step_kernel=kernel)
        assert eng.step_variant(T) == name and (not single or eng.step_variant(1) == name)
        if collect:
            st = fa.JointRolloutStorage(T, E, N, device="cuda")
            eng.bind_storage(st)
            eng.collect_reset()
            assert np.array_equal(st.obs[0].cpu().numpy(), orc.reset().astype(np.float32))
            st.actions.copy_(torch.from_numpy(acts[..., None]).cuda())
            if per_step:
                for s in range(T):
                    eng.collect_step(s)
            else:
                eng.collect_rollout(0, T)
            n_diff, worst, ends, _ = _check_rows_vs_oracle(st, orc, acts, T)
            assert ends > 0 and n_diff == 0 and worst <= 1e-5
        else:
            o0 = torch.empty((E, N, 6), dtype=torch.float64, device="cuda")
            eng.reset(obs_f64=o0)
            assert np.array_equal(o0.cpu().numpy(), orc.reset())
            want = ("obs_f64", "reward_f64", "mask_f32", "done", "hit", "was_hit")
            if per_step:
                rows = [eng.step(torch.from_numpy(acts[s]).cuda(), auto_reset=True, want=want) for s in range(T)]
                out = {k: np.stack([r[k].cpu().numpy() for r in rows]) for k in want}
            else:
                out = {k: v.cpu().numpy() for k, v in eng.step_many(torch.from_numpy(acts).cuda(), auto_reset=True,
                                                                    want=want).items()}
            ends = 0
            for s in range(T):
                ref = orc.step(acts[s], auto_reset=True)
                assert np.array_equal(out["done"][s], ref["done"]), s
                assert np.array_equal(out["hit"][s], ref["hit"]) and np.array_equal(out["was_hit"][s], ref["was_hit"]), s
                want_mask = np.where(ref["done"][:, None] != 0, 1, ref["alive_before"]).astype(np.float32)
                assert np.array_equal(out["mask_f32"][s], want_mask), s
                assert np.array_equal(out["obs_f64"][s], ref["obs"]), s
                assert np.array_equal(out["reward_f64"][s], ref["reward"]), s
                ends += int(ref["done"].sum())
            assert ends > 0
        so, sg = orc.get_state(), eng.get_state()
        for k in ("alive", "time_step", "num_hit", "num_was_hit"):
            assert np.array_equal(so[k], sg[k]), (k, per_step)
        assert np.array_equal(eng.rng_peek(E - 1, 2 * N), orc.rng_doubles(E - 1, 2 * N))


@pytest.mark.parametrize("kernel", ["pipe", "pipe3", "waves1", "waves2", "waves3"])
@pytest.mark.parametrize("partner_shot", [True, False])
def test_exactly_coincident_agents(fa, kernel, partner_shot):
    """Two agents at the same point: the reference's pair force is 0/0 = NaN for both (core.py:447-455).
    If one of the two is shot in that very step it is dead before the forces are applied and the
    reference never visits the pair (core.py:233-236): the survivor stays finite.  Every kernel build
    must do the same as the oracle in both cases."""
    from fa_oracle import OracleEnv
    E, G, A, T = 12, 3, 3, 3
    N = G + A
    orc = OracleEnv(E, G, A, 50, base_seed=1)
    eng = fa.BatchedFortAttack(E, G, A, 50, base_seed=1, step_kernel=kernel)
    orc.reset(), eng.reset()
    s = orc.get_state()
    # guard 0 and attacker 3 coincide at the origin; attacker 4 stands below them facing up (pi/2, the
    # reset heading): its laser wedge covers the origin.  Everyone else is far away.
    s["pos_x"][:] = np.array([0.0, -0.9, 0.9, 0.0, 0.0, 0.9])[None]
    s["pos_y"][:] = np.array([0.0, 0.7, 0.7, 0.0, -0.4, -0.7])[None]
    s["vel_x"][:] = 0.0
    s["vel_y"][:] = 0.0
    orc.set_state(s)
    eng.set_state({k: s[k] for k in ("pos_x", "pos_y", "vel_x", "vel_y", "ang", "prev_dist", "alive", "time_step")})
    acts = np.zeros((T, E, N), np.int64)
    if partner_shot:
        acts[0, :, 4] = 7            # attacker 4 shoots: guard 0 dies, attacker 3 (same team) does not
    want = ("obs_f64", "reward_f64", "done", "was_hit")
    out = {k: v.cpu().numpy() for k, v in eng.step_many(torch.from_numpy(acts).cuda(), auto_reset=False, want=want).items()}
    for t in range(T):
        ref = orc.step(acts[t], auto_reset=False)
        assert np.array_equal(out["was_hit"][t], ref["was_hit"]), t
        assert np.array_equal(out["done"][t], ref["done"]), t
        assert np.array_equal(out["obs_f64"][t], ref["obs"], equal_nan=True), (t, out["obs_f64"][t][0], ref["obs"][0])
        # (rewards of a world that has gone NaN are not compared: prevDist = NaN is the engine's encoding
        #  of None, so 2 * (prevDist - dist) is 0 here where the reference has NaN)
        if partner_shot or t == 0:
            assert np.array_equal(out["reward_f64"][t], ref["reward"], equal_nan=True), t
        if t == 0:
            assert bool(ref["was_hit"][0, 0]) == partner_shot
            assert np.isnan(ref["obs"][0, 3, 1]) == (not partner_shot)   # the survivor is finite iff its partner died
