"""Experiment (run ON THE GPU BOX): A/B of fa_policy_kernel builds.  For each library (tools/_build/lib_<name>.so, or
"product") in its own process: values / log-probs against the PyTorch module (3v3 x 4096, 5v5 x 1000, 2v4 x 333) and the
committed reference golden, then the launch time of policy_act at 3v3 x 4096 and 5v5 x 4096.
usage: ab_policy.py product name1 name2 ..."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(name):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.nn.functional as F
    import emergent_multiagent_strategies_amd as fa
    if name != "product":
        fa._lib._build.LIB = os.path.join(ROOT, "tools", "_build", "lib_%s.so" % name)
    from test_gpu_policy import _policies, _obs, _torch_reference
    out = {"lib": name}
    worst = 0.0
    for G, A, E in ((3, 3, 4096), (5, 5, 1000), (2, 4, 333), (8, 8, 50), (3, 3, 1)):
        N = G + A
        pols, packed = _policies(fa, G, A, 10 * G + A)
        eng = fa.BatchedFortAttack(E, G, A, 20)
        obs = _obs(E, N, E)
        logits, value = _torch_reference(pols, obs, G)
        v, act, lp = eng.policy_act(obs, packed[0], packed[1], deterministic=True)
        logp_all = F.log_softmax(logits, dim=-1)
        err = max(float((v - value).abs().max()), float((lp - logp_all.gather(-1, act.unsqueeze(-1))[..., 0]).abs().max()))
        worst = max(worst, err)
        del eng
    out["max_err_vs_torch"] = worst
    for G, A, E in ((3, 3, 4096), (5, 5, 4096)):
        N = G + A
        pols, packed = _policies(fa, G, A, 1)
        eng = fa.BatchedFortAttack(E, G, A, 20)
        obs = _obs(E, N, 3)
        counter = torch.zeros(1, dtype=torch.int64, device="cuda")
        fn = lambda: eng.policy_act(obs, packed[0], packed[1], seed=5, counter=counter, step=0)
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(100):
                fn()
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) / 100 * 1e3)
        out["%dv%d_%d_us" % (G, A, E)] = round(best, 2)
        del eng
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "--one":
        one(args[1])
    else:
        for n in args:
            subprocess.call([sys.executable, os.path.abspath(__file__), "--one", n])
