"""Experiment (GPU box): eager fa_ppo_grad calls at config 3's minibatch shape (rows gathered by a random index set from a
4096 x 128 rollout's worth of rows), for rocprofv3 --kernel-trace --stats.  usage: prof_grad.py [G A] [share]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import emergent_multiagent_strategies_amd as fa
from emergent_multiagent_strategies_amd import mpnn_pack as mp_
from emergent_multiagent_strategies_amd.env import ppo_grad
G = int(sys.argv[1]) if len(sys.argv) > 1 else 3
A = int(sys.argv[2]) if len(sys.argv) > 2 else 3
share = len(sys.argv) > 3 and sys.argv[3] == "share"
N, R, B = G + A, 4096 * 128, 16384
torch.manual_seed(0)
pol = fa.MPNN(num_agents=G, num_opp_agents=A, num_actions=8).cuda()
obs = torch.randn(R, N, 6, device="cuda"); obs[:, :, 0] = (torch.rand(R, N, device="cuda") > 0.3).float()
action = torch.randint(0, 8, (R, N, 1), device="cuda")
vp, ret, adv = [torch.randn(R, N, 1, device="cuda") for _ in range(3)]
olp = -torch.rand(R, N, 1, device="cuda") * 2
P = mp_.kernel_params(pol)
w = torch.zeros(mp_.WEIGHT_FLOATS, device="cuda"); wt = torch.zeros(mp_.TRANS_FLOATS, device="cuda")
mp_.pack_from_params(P, w, wt)
idx = torch.randperm(R, device="cuda")[:B].contiguous()
out, sc = ppo_grad(obs, action, vp, ret, olp, adv, w, wt, None, 0, G, A, 0.2, 0.5, 0.01, True, idx=idx, share_cu=share)
fn = lambda: ppo_grad(obs, action, vp, ret, olp, adv, w, wt, None, 0, G, A, 0.2, 0.5, 0.01, True, scratch=sc, out=out, idx=idx, share_cu=share)
for _ in range(3):
    fn()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(30):
    fn()
b.record(); torch.cuda.synchronize()
print(json.dumps({"shape": "%dv%d, %d rows" % (G, A, B * G), "share_cu": share, "fa_ppo_grad_us": a.elapsed_time(b) / 30 * 1e3}))
