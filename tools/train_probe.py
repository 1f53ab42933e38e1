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
names = {0: "start", 1: "fwd: gather + encoders + opponent stage", 2: "fwd: rounds done", 3: "fwd: heads + last layer", 4: "losses",
         5: "bwd: heads (dW9, d[P|V], save dPV, dh3)", 60: "bwd: rounds done", 61: "bwd: opponent stage", 62: "bwd: encoders"}
for r in range(3):
    b = 10 + 5 * r
    names.update({b: "  fwd round %d: g = h A (gemm128 + store)" % r, b + 1: "  fwd round %d: barrier" % r, b + 2: "  fwd round %d: attention" % r,
                  b + 3: "  fwd round %d: barrier" % r, b + 4: "  fwd round %d: gemm256, barrier, store, barrier, save h" % r})
for k in range(3):
    b = 30 + 8 * k
    names.update({b: "  bwd round %d: request h_in, bias sums, save dZ" % (2 - k), b + 1: "  bwd round %d: dZ W7^T (2 gemm128)" % (2 - k),
                  b + 2: "  bwd round %d: barrier, stores, h_in -> LDS" % (2 - k), b + 3: "  bwd round %d: g -> LDS + barrier" % (2 - k),
                  b + 4: "  bwd round %d: attention backward + barrier" % (2 - k), b + 5: "  bwd round %d: save dg + dg A^T + gated add + barrier" % (2 - k)})
order = [0, 1] + list(range(10, 25)) + [2, 3, 4, 5] + list(range(30, 54)) + [60, 61, 62]
keys = [k for k in order if k in names and buf[k]]
prev = None
for k in keys:
    if prev is not None:
        print("%-56s %8d cycles" % (names[k], buf[k] - buf[prev]))
    prev = k
print("%-56s %8d cycles = %.1f us at 2.4 GHz" % ("total", buf[keys[-1]] - buf[keys[0]], (buf[keys[-1]] - buf[keys[0]]) / 2400.0))
