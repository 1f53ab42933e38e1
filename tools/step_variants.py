"""Experiment (GPU box): the fused 128-step rollout launch at 3v3 x 4096 (and other E) under every step-kernel build
(FA_KERNEL_*): microseconds per launch and shader cycles per env-step of a wave's dependent chain.
usage: step_variants.py [E ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import emergent_multiagent_strategies_amd as fa
G = A = 3
T = 128
for E in [int(x) for x in sys.argv[1:]] or [4096]:
    for kern in ("pipe", "pipe3", "waves1", "waves2", "waves3", ):
        try:
            eng = fa.BatchedFortAttack(E, G, A, 100, base_seed=0, step_kernel=kern)
        except Exception as exc:
            print(json.dumps({"E": E, "kernel": kern, "error": str(exc)[:80]})); continue
        st = fa.JointRolloutStorage(T, E, G + A, device="cuda")
        eng.bind_storage(st)
        st.actions.copy_(torch.randint(0, 8, st.actions.shape, device="cuda"))
        eng.collect_reset()
        for _ in range(3):
            eng.collect_rollout(0, T)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            eng.collect_rollout(0, T)
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) / 20 * 1e3
        print(json.dumps({"E": E, "kernel": eng.step_variant(T), "launch_us": round(us, 1), "cycles_per_step_at_2.4GHz": round(us * 2400 / T)}), flush=True)
        del eng, st
