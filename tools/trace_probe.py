"""Experiment: absolute timestamps of the pipelined step kernel's marks (tools/fa_probe_trace.h) for a lone workgroup
and one that shares its CU.  usage: FA_PROBE=fa_probe_trace.h FA_TIMING_LIB=libfa_trace.so make_timing_build.py; trace_probe.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import emergent_multiagent_strategies_amd as fa
LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", os.environ.get("FA_TIMING_LIB", "libfa_trace.so"))
fa._lib._build.LIB = LIB
E, G, A, T = 4096, 3, 3, 128
eng = fa.BatchedFortAttack(E, G, A, 100, track_counters=True)
st = fa.JointRolloutStorage(T, E, G + A, device="cuda")
eng.bind_storage(st)
st.actions.copy_(torch.randint(0, 8, st.actions.shape, device="cuda"))
eng.collect_reset()
for _ in range(3):
    eng.collect_rollout(0, T)
torch.cuda.synchronize()
lib = C.CDLL(LIB)
S = 96
buf = (C.c_ulonglong * (2 * S * 24))()
lib.fa_dbg_trace(buf)
import numpy as np
tr = np.array(buf[:], dtype=np.int64).reshape(2, S, 24)
names = {7: "w0 P released", 0: "w0 trig read issued", 1: "w0 laser done (B2 arrive)", 2: "w0 B2 passed (data in)", 3: "w0 integrated", 6: "w0 published (P arrive)",
         20: "w1 loop top", 21: "w1 pair done", 22: "w1 B2 passed", 23: "w1 P arrive",
         10: "w2 loop top", 11: "w2 pair done", 12: "w2 B2 passed", 13: "w2 P arrive",
         16: "w3 loop top", 17: "w3 walls done", 18: "w3 B2 passed", 19: "w3 P arrive"}
for wg, label in ((0, "workgroup 3 (lone on its CU)"), (1, "workgroup 300 (shares its CU)")):
    print(label)
    acc = {k: [] for k in names}
    for s in range(8, S - 1):
        t0 = tr[wg, s - 1, 7]          # wave 0 passed P(s-1)
        for k in names:
            v = tr[wg, s, k] if k != 7 else tr[wg, s, 7]
            acc[k].append(v - t0)
    for k in (0, 1, 2, 3, 6, 7, 20, 21, 22, 23, 10, 11, 12, 13, 16, 17, 18, 19):
        a = np.array(acc[k])
        print("  %-28s mean %7.0f  p10 %7.0f  p90 %7.0f" % (names[k], a.mean(), np.percentile(a, 10), np.percentile(a, 90)))
