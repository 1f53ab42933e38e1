"""Experiment: fused launch at large E for two builds of the library (A/B on the same box)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import emergent_multiagent_strategies_amd as fa
if len(sys.argv) > 1 and sys.argv[1] != "product":
    fa._lib._build.LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", sys.argv[1])
for E, T in ((32768, 128), (262144, 64)):
    eng = fa.BatchedFortAttack(E, 3, 3, 100, base_seed=0, track_counters=False)
    st = fa.JointRolloutStorage(T, E, 6, device="cuda")
    eng.bind_storage(st)
    st.actions.copy_(torch.randint(0, 8, st.actions.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(0)))
    eng.collect_reset()
    eng.collect_rollout(0, T); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        eng.collect_rollout(0, T)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    print(json.dumps({"lib": sys.argv[1] if len(sys.argv) > 1 else "product", "E": E, "us": ms * 1e3, "frac": 837 * E * T / (ms * 1e-3) / 8e12, "variant": eng.step_variant(T)}))
    del eng, st
