"""Experiment: one-wave vs two-wave step kernel as a function of the number of envs."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import emergent_multiagent_strategies_amd as fa
for G in (3, 5):
    for E in (2048, 4096, 8192, 12288, 16384, 24576, 32768, 65536):
        T = 64
        N = 2 * G
        eng = fa.BatchedFortAttack(E, G, G, 100, track_counters=False)
        st = fa.JointRolloutStorage(T, E, N, device="cuda")
        eng.bind_storage(st)
        st.actions.copy_(torch.randint(0, 8, st.actions.shape, device="cuda"))
        eng.collect_reset()
        for fused in (True, False):
            run = (lambda: eng.collect_rollout(0, T)) if fused else (lambda: [eng.collect_step(s) for s in range(T)])
            run(); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5): run()
            b.record(); torch.cuda.synchronize()
            print(json.dumps({"G": G, "E": E, "fused": fused, "us_per_step_row": a.elapsed_time(b) / 5 / T * 1e3}), flush=True)
        del eng, st
