"""Experiment (run ON THE GPU BOX): A/B of fa_train_kernel builds.  For each library (tools/_build/lib_<name>.so, or
"product") in its own process: the fused-gradient / fused-update parity tests (pytest, against torch autograd), then
the time of one eager fa_ppo_grad call (3v3, 16 384 x 3 rows) and of one whole PPO update at config 3.
usage: ab_train.py [--no-tests] product name1 name2 ..."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(name, tests):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import emergent_multiagent_strategies_amd as fa
    if name != "product":
        fa._lib._build.LIB = os.path.join(ROOT, "tools", "_build", "lib_%s.so" % name)
    out = {"lib": name}
    if tests:
        import pytest
        rc = pytest.main(["-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_gpu_policy.py"),
                          os.path.join(ROOT, "tests", "test_gpu_learner.py"), "-k", "ppo_grad or fused_update or one_graph or fold"])
        out["tests_rc"] = int(rc)
    from emergent_multiagent_strategies_amd import mpnn_pack as mp_
    from emergent_multiagent_strategies_amd.env import ppo_grad
    for G, A, B in ((3, 3, 16384), (5, 5, 16384)):
        N = G + A
        torch.manual_seed(0)
        pol = fa.MPNN(num_agents=G, num_opp_agents=A, num_actions=8).cuda()
        obs = torch.randn(B, N, 6, device="cuda")
        obs[:, :, 0] = (torch.rand(B, N, device="cuda") > 0.3).float()
        action = torch.randint(0, 8, (B, N, 1), device="cuda")
        vp, ret, adv = [torch.randn(B, N, 1, device="cuda") for _ in range(3)]
        olp = -torch.rand(B, N, 1, device="cuda") * 2
        P = mp_.kernel_params(pol)
        w = torch.zeros(mp_.WEIGHT_FLOATS, device="cuda")
        wt = torch.zeros(mp_.TRANS_FLOATS, device="cuda")
        mp_.pack_from_params(P, w, wt)
        scale = torch.tensor([1.0 / (B * G), 1.0], device="cuda")
        o, sc = ppo_grad(obs, action, vp, ret, olp, adv, w, wt, scale, 0, G, A, 0.2, 0.5, 0.01, True)
        fn = lambda: ppo_grad(obs, action, vp, ret, olp, adv, w, wt, scale, 0, G, A, 0.2, 0.5, 0.01, True, scratch=sc, out=o)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                fn()
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) * 100)
        out["ppo_grad_%dv%d_us" % (G, A)] = round(best, 1)
        out["grad_checksum_%dv%d" % (G, A)] = float(o.double().abs().sum())
        if os.environ.get("FA_AB_DUMP"):
            import numpy as np
            os.makedirs(os.path.join(ROOT, "gpurun_out", "dump"), exist_ok=True)
            np.save(os.path.join(ROOT, "gpurun_out", "dump", "%s_%s_%dv%d.npy" % (name, os.environ["FA_AB_DUMP"], G, A)), o.cpu().numpy())
    torch.manual_seed(0)
    eng = fa.BatchedFortAttack(4096, 3, 3, 100, track_counters=False)
    L = fa.BatchedLearner(eng, num_steps=128, use_graph=True)
    L.reset()
    L.collect()
    L.update()
    ts = []
    for _ in range(3):
        L.collect()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        L.update()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    out["update_s"] = round(min(ts), 4)
    L.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "--one":
        one(args[1], args[2] == "1")
    else:
        tests = "--no-tests" not in args
        for n in [a for a in args if not a.startswith("--")]:
            subprocess.call([sys.executable, os.path.abspath(__file__), "--one", n, "1" if tests else "0"])
