#!/usr/bin/env python
"""E-sweep of the step kernel (run on the GPU box): launch duration, env-steps/s and
algorithmic GB/s (837 B/env-step at 3v3, 1389 at 5v5) vs number of envs, for the fused
(K steps per launch) and the single-step launch shape.  Prints one JSON line per point."""
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import emergent_multiagent_strategies_amd as fa


def point(E, G, A, T, fused, iters=5):
    N = G + A
    eng = fa.BatchedFortAttack(E, G, A, 100, base_seed=0, track_counters=False)
    st = fa.JointRolloutStorage(T, E, N, device="cuda")
    eng.bind_storage(st)
    gen = torch.Generator(device="cuda").manual_seed(0)
    st.actions.copy_(torch.randint(0, 8, st.actions.shape, device="cuda", generator=gen))
    eng.collect_reset()
    run = (lambda: eng.collect_rollout(0, T)) if fused else (lambda: [eng.collect_step(s) for s in range(T)])
    run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        run()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / iters
    nl = 1 if fused else T
    bpe = 138 * N + 9
    print(json.dumps({"E": E, "G": G, "A": A, "T": T, "launch": "fused" if fused else "per-step (eager)",
                      "us_per_launch": ms * 1e3 / nl, "us_per_env_step_row": ms * 1e3 / T,
                      "env_steps_per_s": E * T / (ms * 1e-3), "algorithmic_GBps": bpe * E * T / (ms * 1e-3) / 1e9,
                      "frac_of_8TBps": bpe * E * T / (ms * 1e-3) / 8e12}), flush=True)
    del eng, st
    torch.cuda.empty_cache()


if __name__ == "__main__":
    for (G, A) in ((3, 3), (5, 5)):
        for E, T in ((4096, 128), (32768, 128), (262144, 64), (1048576, 32)):
            for fused in (True, False):
                point(E, G, A, T, fused)
