"""Experiment: per-section shader-clock breakdown of the fused step kernel (needs the
-DFA_TIMING build in exp_libs/libfa_timing.so; run with FA_LIB_OVERRIDE pointing at it)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import emergent_multiagent_strategies_amd as fa
E, G, A, T = 4096, 3, 3, 128
eng = fa.BatchedFortAttack(E, G, A, 100, track_counters=False)
st = fa.JointRolloutStorage(T, E, G + A, device="cuda")
eng.bind_storage(st)
st.actions.copy_(torch.randint(0, 8, st.actions.shape, device="cuda"))
eng.collect_reset()
lib = C.CDLL(os.environ["FA_LIB_OVERRIDE"])
buf = (C.c_ulonglong * 16)()
for _ in range(3):
    eng.collect_rollout(0, T)
torch.cuda.synchronize()
lib.fa_dbg_read(buf, 1)
for _ in range(5):
    eng.collect_rollout(0, T)
torch.cuda.synchronize()
lib.fa_dbg_read(buf, 0)
waves = buf[10]
names = ["loop + action read", "decode + stage pos (to barrier 1)", "barrier 1", "sin/cos + triangle stage",
         "laser tests", "ballots (to barrier 2)", "barriers 2+3 (force wave)", "F read + integrate", "reward + done + outputs",
         "reset + obs store"]
tot = sum(buf[k] for k in range(10))
for k, n in enumerate(names):
    print("%-22s %8.1f cycles/step  %5.1f%%" % (n, buf[k] / waves / T, 100.0 * buf[k] / tot))
print("total %.1f cycles/step" % (tot / waves / T))
