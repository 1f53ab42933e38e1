"""One fused rollout launch for the ATT thread trace (small T keeps the trace short)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import emergent_multiagent_strategies_amd as fa
E, G, A, T = 4096, 3, 3, 32
eng = fa.BatchedFortAttack(E, G, A, 100, track_counters=False)
st = fa.JointRolloutStorage(T, E, G + A, device="cuda")
eng.bind_storage(st)
st.actions.copy_(torch.randint(0, 8, st.actions.shape, device="cuda"))
eng.collect_reset()
for _ in range(3):
    eng.collect_rollout(0, T)
torch.cuda.synchronize()
