"""Experiment: shader-clock phase marks of one workgroup of fa_policy_kernel (needs tools/_build/libfa_timing.so)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import emergent_multiagent_strategies_amd as fa
LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", os.environ.get("FA_TIMING_LIB", "libfa_timing.so"))
fa._lib._build.LIB = LIB
from test_gpu_policy import _policies, _obs
G, A, E = 3, 3, 4096
pols, packed = _policies(fa, G, A, 1)
eng = fa.BatchedFortAttack(E, G, A, 20)
obs = _obs(E, G + A, 3)
counter = torch.zeros(1, dtype=torch.int64, device="cuda")
for _ in range(20):
    eng.policy_act(obs, packed[0], packed[1], seed=5, counter=counter, step=0)
torch.cuda.synchronize()
lib = C.CDLL(LIB)
buf = (C.c_ulonglong * 128)()
lib.fa_dbg_policy(buf)
names = {0: "start (obs staged)", 1: "encoders", 2: "opp: g_o gemm + store", 3: "opp: attention", 28: "opp: e_opp gemm (after round loop start)",
         29: "heads gemm x2 + stores", 30: "W9 gemm", 31: "sampling"}
for r in range(3):
    b = 4 + r * 8
    names.update({b: "  round %d: (e_opp / prev store + barrier)" % r, b + 1: "  round %d: g gemm128" % r, b + 2: "  round %d: g store" % r,
                  b + 3: "  round %d: barrier" % r, b + 4: "  round %d: attention" % r, b + 5: "  round %d: barrier" % r,
                  b + 6: "  round %d: gemm256" % r, b + 7: "  round %d: barrier (old h read)" % r})
order = [0, 1, 2, 3] + list(range(4, 28)) + [28, 29, 30, 31]
for w, off in (("wave 0", 0), ("wave 4", 64)):
    print(w)
    prev = None
    for k in order:
        if not buf[off + k]:
            continue
        if prev is not None:
            print("  %-44s %8d cycles" % (names[k], buf[off + k] - buf[off + prev]))
        prev = k
    print("  %-44s %8d cycles" % ("total", buf[off + 31] - buf[off + 0]))
