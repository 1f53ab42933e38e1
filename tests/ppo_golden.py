"""Shared by the CPU and GPU tests of tests/golden/ppo_update_h128.npz (oracle/gen_golden.py gen_ppo_update_fixture:
the REFERENCE's JointPPO.update, rlcore/algo/ppo.py:116-204, on seed-constructed full-size policies)."""
import os

import numpy as np
import torch

CASES = ["3v3_g_clip", "3v3_a_noclip", "5v5_g_clip", "5v5_a_clip"]


def load(golden_dir):
    return np.load(os.path.join(golden_dir, "ppo_update_h128.npz"))


def case(MPNN, g, tag, device="cpu"):
    """-> dict(pol, rows, own_sl, opp_sl, hyper..., expected...).  The policy is re-made from the fixture's seed exactly as
    the generator made the reference's (mpnn_h128_setup) and its fingerprints are checked against the reference's."""
    G, A, team, clipped, seed, T, P = [int(v) for v in g[tag + ".meta"]]
    clip, vcoef, ecoef, lr, gnorm = [float(v) for v in g[tag + ".hyper"]]
    N = G + A
    n, m = (G, A) if team == 0 else (A, G)
    torch.manual_seed(seed)
    pol = MPNN(num_agents=n, num_opp_agents=m, num_actions=8)
    for p in pol.parameters():
        if p.dim() == 1:
            p.data.uniform_(-0.3, 0.3)
    pol.dist.linear.weight.data.mul_(3.0)
    fp = []
    for v in pol.state_dict().values():
        a = v.detach().numpy().reshape(-1).astype(np.float64)
        fp.append([a.sum(), np.abs(a).sum(), a[0], a[-1]])
    fp, ref = np.array(fp), g[tag + ".fingerprint"]
    assert (np.abs(fp[:, :2] - ref[:, :2]).max(1) <= 1e-6 * ref[:, 1] + 1e-6).all() and \
        np.abs(fp[:, 2:] - ref[:, 2:]).max() < 1e-5, "seed-constructed weights differ from the reference's"
    pol = pol.to(device)
    t = lambda k, dt=torch.float32: torch.from_numpy(g["%s.%s" % (tag, k)]).to(dt).to(device).contiguous()
    rows = (t("obs"), t("actions", torch.int64), t("value_preds"), t("returns"), t("old_logp"), t("adv"))
    own_sl = slice(0, G) if team == 0 else slice(G, N)
    opp_sl = slice(G, N) if team == 0 else slice(0, G)
    names = [str(k) for k in g[tag + ".param_names"]]
    return dict(pol=pol, rows=rows, own_sl=own_sl, opp_sl=opp_sl, clip=clip, vcoef=vcoef, ecoef=ecoef, lr=lr, gnorm=gnorm,
                clipped=bool(clipped), B=T * P, G=G, A=A, team=team, losses=g[tag + ".losses"], names=names,
                grad_fp=g[tag + ".grad_fingerprint"], delta_fp=g[tag + ".delta_fingerprint"],
                grad={k: g["%s.grad.%s" % (tag, k)] for k in names}, delta={k: g["%s.delta.%s" % (tag, k)] for k in names})


def compare(c, losses, before, grad_tol, lr_slack=0.02):
    """losses (3,), the policy's .grad (clipped) and parameters (stepped) against the reference's.
    -> (max loss deviation, max gradient deviation relative to the tensor's largest entry).
    Gradients: every stored element within grad_tol x the tensor's largest reference entry, and the l2 norms within
    grad_tol.  Adam displacement: the first step is -lr * g / (|g| + eps): compared where |g| is clear of eps-scale
    noise (|g_ref| > 1e-6), to lr_slack * lr."""
    pol = c["pol"]
    dl = float(np.abs(np.asarray(losses, np.float64) - c["losses"]).max())
    params = dict(pol.named_parameters())
    worst = 0.0
    for k, name in enumerate(c["names"]):
        if name not in params or c["grad_fp"][k][1] == 0.0:      # oppUpdate: unused by the forward, no gradient anywhere
            continue
        p = params[name]
        stride = 1 if p.numel() <= 1024 else 61
        got = p.grad.detach().reshape(-1)[::stride].double().cpu().numpy()
        want = c["grad"][name].astype(np.float64)
        scale = float(np.abs(want).max())
        dev = float(np.abs(got - want).max()) / scale
        worst = max(worst, dev)
        assert dev <= grad_tol, "gradient of %s: %.3e of its largest entry" % (name, dev)
        l2 = float(p.grad.detach().double().norm())
        assert abs(l2 - c["grad_fp"][k][4]) <= grad_tol * c["grad_fp"][k][4] + 1e-9, name
        delta = (p.detach() - before[name]).reshape(-1)[::stride].double().cpu().numpy()
        clear = np.abs(want) > 1e-6
        assert np.abs(delta[clear] - c["delta"][name].astype(np.float64)[clear]).max(initial=0.0) <= lr_slack * c["lr"], name
    return dl, worst
