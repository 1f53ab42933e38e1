"""GPU: the fused optimizer step (GraphedPPOStep: fold, fa_ppo_grad, unfold, clip + fa_adam_step, replayed from a hipGraph)
against tests/golden/ppo_update_h128.npz -- the REFERENCE's JointPPO.update (rlcore/algo/ppo.py:116-204) on
seed-constructed hidden_dim-128 policies, one full-batch minibatch: the three losses, the clipped gradient and the Adam
displacement of every parameter.  No hop through this repo's autograd restatement."""
import pytest
import torch

import ppo_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ppo_golden.CASES)
def test_fused_step_matches_the_reference_golden(golden_dir, tag):
    import emergent_multiagent_strategies_amd as fa
    from emergent_multiagent_strategies_amd.learner import GraphedPPOStep
    fa._lib.load()
    g = ppo_golden.load(golden_dir)
    c = ppo_golden.case(fa.MPNN, g, tag, device="cuda")
    pol, B = c["pol"], c["B"]
    opt = torch.optim.Adam(pol.parameters(), lr=c["lr"], capturable=True)
    step = GraphedPPOStep(pol, opt, c["own_sl"], c["opp_sl"], c["rows"], B, c["clip"], c["vcoef"], c["ecoef"], c["gnorm"],
                          c["clipped"], None, fused=True)
    assert step.fused
    before = {k: p.detach().clone() for k, p in pol.named_parameters()}
    losses = step.run(c["rows"], torch.arange(B, device="cuda"))
    torch.cuda.synchronize()
    # the fused kernels sum in another order than autograd: observed <= 1e-5 of a tensor's largest entry (the printed
    # line); the bound is 10 x that
    dl, worst = ppo_golden.compare(c, losses.cpu().numpy(), before, grad_tol=1e-4)
    print("%s: losses within %.1e, gradients within %.1e of each tensor's largest entry" % (tag, dl, worst))
    assert dl < 1e-5
