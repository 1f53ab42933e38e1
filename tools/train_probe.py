"""Experiment: shader-clock phase marks of one workgroup of fa_train_kernel (needs tools/_build/libfa_timing.so)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import emergent_multiagent_strategies_amd as fa
LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libfa_timing.so")
fa._lib._build.LIB = LIB
from emergent_multiagent_strategies_amd import mpnn_pack as mp_
from emergent_multiagent_strategies_amd.env import ppo_grad
G, A, B = 3, 3, 16384
N = G + A
pol = fa.MPNN(num_agents=G, num_opp_agents=A, num_actions=8).cuda()
obs = torch.randn(B, N, 6, device="cuda"); obs[:, :, 0] = (torch.rand(B, N, device="cuda") > 0.3).float()
action = torch.randint(0, 8, (B, N, 1), device="cuda")
vp, ret, adv = [torch.randn(B, N, 1, device="cuda") for _ in range(3)]
olp = -torch.rand(B, N, 1, device="cuda") * 2
P = mp_.kernel_params(pol)
w = torch.zeros(mp_.WEIGHT_FLOATS, device="cuda"); wt = torch.zeros(mp_.TRANS_FLOATS, device="cuda")
mp_.pack_from_params(P, w, wt)
scale = torch.tensor([1.0 / (B * G), 1.0], device="cuda")
out, sc = ppo_grad(obs, action, vp, ret, olp, adv, w, wt, scale, 0, G, A, 0.2, 0.5, 0.01, True)
for _ in range(3):
    ppo_grad(obs, action, vp, ret, olp, adv, w, wt, scale, 0, G, A, 0.2, 0.5, 0.01, True, scratch=sc, out=out)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    ppo_grad(obs, action, vp, ret, olp, adv, w, wt, scale, 0, G, A, 0.2, 0.5, 0.01, True, scratch=sc, out=out)
b.record(); torch.cuda.synchronize()
print("fa_ppo_grad (train + reduce): %.1f us per call at B = %d" % (a.elapsed_time(b) * 100, B))
lib = C.CDLL(LIB)
buf = (C.c_ulonglong * 64)()
lib.fa_dbg_train(buf)
names = {0: "start", 1: "fwd: encoders + opponent stage", 2: "fwd: 3 rounds", 3: "fwd: heads", 4: "losses", 5: "bwd: heads",
         30: "bwd: rounds done", 31: "bwd: opponent stage", 32: "bwd: encoders", 48: "bwd: shared weight gradients (end phase)"}
for r in range(3):
    base = 6 + r * 8
    names.update({base: "  round %d: (prev tail)" % (2 - r), base + 1: "  round %d: relu mask + bias grad" % (2 - r),
                  base + 2: "  round %d: load h, recompute g, hmix" % (2 - r), base + 3: "  round %d: dW7" % (2 - r),
                  base + 4: "  round %d: dZ W7^T" % (2 - r), base + 5: "  round %d: attention backward" % (2 - r)})
for r in range(3):
    base = 33 + r * 7
    names.update({base: "  fwd round %d: (prev tail: save_tile)" % r, base + 1: "  fwd round %d: g = h A (gemm128 + store)" % r,
                  base + 2: "  fwd round %d: barrier" % r, base + 3: "  fwd round %d: attention" % r, base + 4: "  fwd round %d: barrier" % r,
                  base + 5: "  fwd round %d: gemm256" % r, base + 6: "  fwd round %d: barrier + store + barrier" % r})
order = [0, 1] + list(range(33, 48)) + [2, 3, 4, 5] + list(range(6, 33)) + [48]
keys = [k for k in order if k in names and buf[k]]
prev = None
for k in keys:
    if prev is not None:
        print("%-42s %8d cycles" % (names[k], buf[k] - buf[prev]))
    prev = k
print("%-42s %8d cycles = %.1f us at 2.4 GHz" % ("total", buf[keys[-1]] - buf[keys[0]], (buf[keys[-1]] - buf[keys[0]]) / 2400.0))
