"""CPU (also on the GPU box, where the reference does not exist): this repo's PyTorch statement of JointPPO.update --
learner.joint_ppo_update over mpnn.MPNN -- against tests/golden/ppo_update_h128.npz, the outputs of the REFERENCE's
JointPPO.update (rlcore/algo/ppo.py:116-204) on seed-constructed hidden_dim-128 policies: three losses, the clipped
gradients and the Adam displacement of every parameter, 3v3 and 5v5, both teams, clipped and un-clipped value loss."""
import numpy as np
import pytest
import torch

import ppo_golden


@pytest.mark.parametrize("tag", ppo_golden.CASES)
def test_torch_update_matches_the_reference_golden(golden_dir, tag):
    from emergent_multiagent_strategies_amd.mpnn import MPNN
    from emergent_multiagent_strategies_amd.learner import joint_ppo_update
    torch.set_num_threads(1)
    g = ppo_golden.load(golden_dir)
    c = ppo_golden.case(MPNN, g, tag)
    pol = c["pol"]
    before = {k: p.detach().clone() for k, p in pol.named_parameters()}
    opt = torch.optim.Adam(pol.parameters(), lr=c["lr"])
    losses = joint_ppo_update(pol, opt, c["own_sl"], c["opp_sl"], c["rows"], c["clip"], 1, 1, c["vcoef"], c["ecoef"], c["gnorm"],
                              clipped_value_loss=c["clipped"])
    dl, worst = ppo_golden.compare(c, losses.numpy(), before, grad_tol=2e-5)      # observed: 1.9e-6
    assert dl < 1e-5
    print("%s: losses within %.1e, gradients within %.1e of each tensor's largest entry" % (tag, dl, worst))
