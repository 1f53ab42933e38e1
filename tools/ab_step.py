"""Experiment (run ON THE GPU BOX): A/B of step-kernel builds.  For each library (tools/_build/lib_<name>.so,
or "product") in its own process: the fused COLLECT launch at 3v3 x 4096 x 128 / 5v5 x 4096 x 128 / 3v3 x 7680 x 64
checked against the oracle (rows, state, counters), then timed with events.
FA_AB_KERNEL=<pipe|chain|...> pins the step kernel (fa_config.step_kernel) in both parts.
usage: ab_step.py [--no-parity] [--quick] product name1 name2 ..."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(name, parity, quick):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import torch
    import emergent_multiagent_strategies_amd as fa
    if name != "product":
        fa._lib._build.LIB = os.path.join(ROOT, "tools", "_build", "lib_%s.so" % name)
    out = {"lib": name}
    shapes = ((3, 3, 4096, 128),) if quick else ((3, 3, 4096, 128), (5, 5, 4096, 128), (3, 3, 7680, 64))
    for G, A, E, T in shapes:
        N = G + A
        tag = "%dv%d_%d" % (G, A, E)
        if parity:
            from fa_oracle import OracleEnv
            from test_gpu_shipped_kernels import _check_rows_vs_oracle, _shooty_actions
            rng = np.random.RandomState(G * 1000 + E)
            orc = OracleEnv(E, G, A, 60, base_seed=4242)
            eng = fa.BatchedFortAttack(E, G, A, 60, base_seed=4242, step_kernel=os.environ.get('FA_AB_KERNEL', 'auto'))
            st = fa.JointRolloutStorage(T, E, N, device="cuda")
            eng.bind_storage(st)
            eng.collect_reset()
            ok = bool(np.array_equal(st.obs[0].cpu().numpy(), orc.reset().astype(np.float32)))
            acts = _shooty_actions(rng, (T, E, N))
            st.actions.copy_(torch.from_numpy(acts[..., None]).cuda())
            eng.collect_rollout(0, T)
            torch.cuda.synchronize()
            try:
                n_diff, worst, ends, deaths = _check_rows_vs_oracle(st, orc, acts, T)
                so, sg = orc.get_state(), eng.get_state()
                for k in ("alive", "time_step", "num_hit", "num_was_hit"):
                    ok = ok and bool(np.array_equal(so[k], sg[k]))
                for k in ("pos_x", "pos_y", "vel_x", "vel_y", "ang", "prev_dist"):
                    ok = ok and bool(np.array_equal(so[k], sg[k], equal_nan=True))
                ok = ok and n_diff == 0 and bool(np.array_equal(eng.rng_peek(0, 2 * N), orc.rng_doubles(0, 2 * N)))
                for k in ("ep_rew_sum", "alive_end"):
                    if k in so and k in sg:
                        ok = ok and bool(np.allclose(so[k], sg[k]))
            except AssertionError as ex:
                ok = False
                out[tag + "_err"] = "rows differ at step %s" % (ex,)
            out[tag + "_parity"] = ok
            del eng, st, orc
        eng = fa.BatchedFortAttack(E, G, A, 100, base_seed=0, step_kernel=os.environ.get('FA_AB_KERNEL', 'auto'))
        st = fa.JointRolloutStorage(T, E, N, device="cuda")
        eng.bind_storage(st)
        st.actions.copy_(torch.randint(0, 8, st.actions.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(0)))
        eng.collect_reset()
        out[tag + "_variant"] = eng.step_variant(T)
        for _ in range(20):
            eng.collect_rollout(0, T)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(200):
                eng.collect_rollout(0, T)
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) / 200 * 1e3)
        out[tag + "_us"] = round(best, 2)
        del eng, st
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "--one":
        one(args[1], args[2] == "1", args[3] == "1")
    else:
        parity = "--no-parity" not in args
        quick = "--quick" in args
        for n in [a for a in args if not a.startswith("--")]:
            subprocess.call([sys.executable, os.path.abspath(__file__), "--one", n, "1" if parity else "0", "1" if quick else "0"])
