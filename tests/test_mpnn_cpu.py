"""CPU: the batched MPNN against the reference's mpnn.py -- golden forward (tests/golden/
mpnn_h32.npz: reference weights + inputs -> value / log-probs / entropy / logits), state_dict
compatibility, and (when the reference tree is present) seed-identical construction of the
full-size policy.  float32 torch on both sides; tolerance 1e-5 (matmul association differs:
batched (B,n,d) @ W here vs flattened (B*n,d) @ W there)."""
import os

import numpy as np
import pytest
import torch

TOL = 1e-5


@pytest.fixture(scope="module")
def MPNN():
    from emergent_multiagent_strategies_amd.mpnn import MPNN
    return MPNN


def _golden(golden_dir):
    return np.load(os.path.join(golden_dir, "mpnn_h32.npz"))


@pytest.mark.parametrize("tag", ["g", "a"])
def test_forward_matches_reference_golden(MPNN, golden_dir, tag):
    g = _golden(golden_dir)
    n, m, B, hdim = [int(v) for v in g[tag + ".shape"]]
    net = MPNN(num_agents=n, num_opp_agents=m, hidden_dim=hdim, num_actions=8)
    sd = {k[len(tag) + 4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + ".sd.")}
    missing = net.load_state_dict(sd, strict=True)   # reference key set == ours
    assert not missing.missing_keys and not missing.unexpected_keys
    inp, opp, act = [torch.from_numpy(g["%s.%s" % (tag, k)]) for k in ("inp", "opp", "act")]
    with torch.no_grad():
        value, logp, ent = net.evaluate_actions_agent_major(inp, opp, act)
        own = net._env_major(inp, n)
        logits, _ = net.logits_value(own, net._env_major(opp, m))
    assert np.abs(value.numpy() - g[tag + ".value"]).max() < TOL
    assert np.abs(logp.numpy() - g[tag + ".logp"]).max() < TOL
    assert np.abs(ent.numpy() - g[tag + ".entropy"]).max() < TOL
    ref_logits = torch.from_numpy(g[tag + ".logits"]).view(n, B, 8).transpose(0, 1)
    ours = logits - logits.logsumexp(-1, keepdim=True)      # torch Categorical normalises logits
    assert np.abs(ours.numpy() - ref_logits.numpy()).max() < TOL


def h128_policies(MPNN, golden, tag, device="cpu"):
    """The two full-size policies of tests/golden/mpnn_h128.npz re-made from their seed (oracle/gen_golden.py
    gen_mpnn_h128_fixture does the same to the reference's module), fingerprints checked against the reference's."""
    G, A, seed, B = [int(v) for v in golden[tag + ".meta"]]
    torch.manual_seed(seed)
    pols = []
    for (n, m), fp_key in (((G, A), ".fingerprint_g"), ((A, G), ".fingerprint_a")):
        pol = MPNN(num_agents=n, num_opp_agents=m, num_actions=8)
        for p in pol.parameters():
            if p.dim() == 1:
                p.data.uniform_(-0.3, 0.3)
        pol.dist.linear.weight.data.mul_(3.0)
        fp = []
        for v in pol.state_dict().values():
            a = v.detach().numpy().reshape(-1).astype(np.float64)
            fp.append([a.sum(), np.abs(a).sum(), a[0], a[-1]])
        # (to 1e-6, not to the bit: orthogonal_'s LAPACK QR rounds differently with the thread count / CPU)
        fp, ref = np.array(fp), golden[tag + fp_key]
        assert (np.abs(fp[:, :2] - ref[:, :2]).max(1) <= 1e-6 * ref[:, 1] + 1e-6).all() and \
            np.abs(fp[:, 2:] - ref[:, 2:]).max() < 1e-5, "seed-constructed weights differ from the reference's"
        pols.append(pol.to(device))
    return pols, G, A


@pytest.mark.parametrize("tag", ["3v3", "5v5"])
def test_full_size_forward_matches_reference_golden(MPNN, golden_dir, tag):
    """hidden_dim 128 (the size the kernels implement): same seed -> the reference's weights (fingerprints), same
    observations -> the reference's values and log-softmax logits."""
    g = np.load(os.path.join(golden_dir, "mpnn_h128.npz"))
    pols, G, A = h128_policies(MPNN, g, tag)
    obs = torch.from_numpy(g[tag + ".obs"])
    N = G + A
    with torch.no_grad():
        lg, vg = pols[0].logits_value(obs[:, :G], obs[:, G:])
        la, va = pols[1].logits_value(obs[:, G:], obs[:, :G])
    value = torch.cat((vg, va), 1)[..., 0]
    logp = torch.log_softmax(torch.cat((lg, la), 1), -1)
    assert np.abs(value.numpy() - g[tag + ".value"]).max() < TOL
    assert np.abs(logp.numpy() - g[tag + ".logp_all"]).max() < TOL
    assert float(np.exp(g[tag + ".logp_all"]).max()) > 0.2       # not the near-uniform policy of the 0.01-gain init


def test_full_size_parameter_inventory(MPNN, golden_dir):
    g = _golden(golden_dir)
    net = MPNN(num_agents=3, num_opp_agents=3, num_actions=8)
    assert sum(p.numel() for p in net.parameters()) == int(g["full_param_count"]) == 158153
    assert list(net.state_dict().keys()) == [str(k) for k in g["full_keys"]]


def test_act_shapes_and_sampling(MPNN):
    torch.manual_seed(0)
    net = MPNN(num_agents=3, num_opp_agents=2, num_actions=8)
    own, opp = torch.randn(7, 3, 6), torch.randn(7, 2, 6)
    with torch.no_grad():
        value, action, logp = net.act(own, opp)
        v2, a2, l2 = net.act(own, opp, deterministic=True)
        ve, le, ent = net.evaluate_actions(own, opp, action)
    assert value.shape == (7, 3, 1) and action.shape == (7, 3, 1) and logp.shape == (7, 3, 1)
    assert action.dtype == torch.int64 and int(action.min()) >= 0 and int(action.max()) < 8
    assert torch.allclose(le, logp) and torch.allclose(ve, value) and ent.shape == (7, 3)
    logits, _ = net.logits_value(own, opp)
    assert torch.equal(a2, logits.argmax(-1, keepdim=True))
    # single-agent team: the message round contributes zeros (mpnn.py:266-274)
    solo = MPNN(num_agents=1, num_opp_agents=4, num_actions=8)
    v, a, l = solo.act(torch.randn(5, 1, 6), torch.randn(5, 4, 6))
    assert v.shape == (5, 1, 1) and torch.isfinite(l).all()


def test_seed_identical_construction_vs_live_reference(MPNN):
    """Same torch seed -> same weights as the reference module (construction order and
    initialisers match), checked against the reference itself when it is available."""
    import ref_harness as rh
    if not rh.reference_available():
        pytest.skip("reference tree not present (build container only)")
    rh.import_reference()
    from mpnn import MPNN as RefMPNN

    class _Sp(object):
        shape = (8,)

    torch.manual_seed(11)
    ref = RefMPNN(action_space=_Sp(), num_agents=3, num_opp_agents=3, num_entities=0, input_size=6,
                  pos_index=2, mask_dist=None, entity_mp=False, policy_layers=1)
    torch.manual_seed(11)
    ours = MPNN(action_space=_Sp(), num_agents=3, num_opp_agents=3)
    rsd, osd = ref.state_dict(), ours.state_dict()
    assert list(rsd.keys()) == list(osd.keys())
    for k in rsd:
        assert torch.equal(rsd[k], osd[k]), k
    inp, opp = torch.randn(12, 6), torch.randn(12, 6)
    act = torch.randint(0, 8, (12, 1))
    with torch.no_grad():
        rv, rl, re_, _ = ref.evaluate_actions(inp, None, opp, None, act)
        ov, ol, oe = ours.evaluate_actions_agent_major(inp, opp, act)
    assert (rv - ov).abs().max() < TOL and (rl - ol).abs().max() < TOL and (re_ - oe).abs().max() < TOL


def test_shipped_reference_checkpoints_load(MPNN):
    """The reference's published 5v5 checkpoints (marlsave/tmp_1/ep*.pt) load unchanged; the
    weights do not depend on team size, so they also drive 3v3."""
    import glob
    paths = sorted(glob.glob("/root/reference/marlsave/tmp_1/ep*.pt"))
    if not paths:
        pytest.skip("reference checkpoints not present (build container only)")
    ck = torch.load(paths[0], map_location="cpu", weights_only=False)
    assert set(ck) == {"models", "ob_rms"} and len(ck["models"]) == 10
    guard = MPNN(num_agents=5, num_opp_agents=5, num_actions=8)
    guard.load_state_dict(ck["models"][0])
    att3 = MPNN(num_agents=3, num_opp_agents=3, num_actions=8)
    att3.load_state_dict(ck["models"][-1])
    with torch.no_grad():
        v, a, lp = att3.act(torch.randn(4, 3, 6), torch.randn(4, 3, 6))
    assert torch.isfinite(v).all() and torch.isfinite(lp).all()


def test_twin_forward_equals_two_separate_forwards(MPNN):
    from emergent_multiagent_strategies_amd.mpnn import TwinMPNN
    torch.manual_seed(5)
    for n in (3, 1):
        g = MPNN(num_agents=n, num_opp_agents=n, num_actions=8, hidden_dim=64)
        a = MPNN(num_agents=n, num_opp_agents=n, num_actions=8, hidden_dim=64)
        twin = TwinMPNN(g, a)
        obs = torch.randn(17, 2 * n, 6)
        with torch.no_grad():
            lg, vg = g.logits_value(obs[:, :n], obs[:, n:])
            la, va = a.logits_value(obs[:, n:], obs[:, :n])
            lt, vt = twin.logits_value(obs)
            assert (lt[0] - lg).abs().max() < 1e-5 and (lt[1] - la).abs().max() < 1e-5
            assert (vt[0] - vg).abs().max() < 1e-5 and (vt[1] - va).abs().max() < 1e-5
            value, action, logp = twin.act(obs)
            assert value.shape == (17, 2 * n, 1) and action.dtype == torch.int64
            ve, lpe, _ = g.evaluate_actions(obs[:, :n], obs[:, n:], action[:, :n])
            assert (lpe - logp[:, :n]).abs().max() < 1e-5 and (ve - value[:, :n]).abs().max() < 1e-5
            assert (twin.get_value(obs) - value).abs().max() < 1e-6
            # weights move -> refresh() rewrites the stacked buffers in place
            ptr = twin._w["enc"][0].data_ptr()
            a.encoder[0].weight.add_(0.5)
            twin.refresh()
            assert twin._w["enc"][0].data_ptr() == ptr
            lt2, _ = twin.logits_value(obs)
            la2, _ = a.logits_value(obs[:, n:], obs[:, :n])
            assert (lt2[1] - la2).abs().max() < 1e-5 and (lt2[1] - lt[1]).abs().max() > 1e-4
    with pytest.raises(ValueError):
        TwinMPNN(MPNN(num_agents=2, num_opp_agents=3, num_actions=8), MPNN(num_agents=3, num_opp_agents=2, num_actions=8))


@pytest.mark.parametrize("n,m", [(3, 3), (5, 5), (2, 4), (1, 3), (4, 1)])
def test_packed_weights_reproduce_the_module(n, m):
    """mpnn_pack: the fused kernel's weight buffer (three linear-map pairs multiplied out, dense operands in
    MFMA B-operand lane order) evaluated with plain torch ops == MPNN.logits_value."""
    from emergent_multiagent_strategies_amd.mpnn import MPNN
    from emergent_multiagent_strategies_amd import mpnn_pack as mp_
    torch.manual_seed(n * 10 + m)
    pol = MPNN(num_agents=n, num_opp_agents=m, num_actions=8)
    for p in pol.parameters():                      # biases are zero-initialised: make them count
        if p.dim() == 1:
            p.data.uniform_(-0.3, 0.3)
    flat = mp_.pack_policy(pol)
    assert flat.numel() == mp_.WEIGHT_FLOATS and flat.dtype == torch.float32
    own, opp = torch.randn(50, n, 6), torch.randn(50, m, 6)
    own[:, :, 0], opp[:, :, 0] = (torch.rand(50, n) > 0.3).float(), (torch.rand(50, m) > 0.3).float()
    with torch.no_grad():
        logits, value = pol.logits_value(own, opp)
    l2, v2 = mp_.folded_forward(flat, own, opp)
    assert (logits - l2).abs().max() < 2e-5 and (value - v2).abs().max() < 2e-5
    w = torch.randn(256, 128)
    assert torch.equal(mp_.unpack_gemm(mp_.pack_gemm(w), 256, 128), w)
    # the bf16x3 pack (csrc/fa_policy.h FA_POFF3_*): every float32 weight is EXACTLY hi + mid + lo; the packed buffer carries
    # both forms of the same matrices; lane order: 16-byte word ((cb * K/16 + s) * 3 + term) * 64 + lane, element j =
    # W[(lane >> 5) * K/2 + 8 s + j][32 cb + (lane & 31)]
    w3 = torch.cat((w, w * 1e-7, w * 3e5, torch.tensor([[0.0, 1.0, -1.0, 2.0 ** -100] * 32] * 16)), 0)[:256]
    assert torch.equal(mp_.unpack_gemm3(mp_.pack_gemm3(w3), 256, 128), w3)
    for k, (K, C) in (("AO", (64, 64)), ("AM", (128, 128)), ("W7", (256, 128)), ("W8", (128, 256)), ("W9", (256, 32))):
        a = mp_.unpack_gemm(flat[mp_.POFF[k]:mp_.POFF[k] + K * C], K, C)
        b = mp_.unpack_gemm3(flat[mp_.POFF3[k]:mp_.POFF3[k] + K * C * 3 // 2], K, C)
        assert torch.equal(a, b), k
    p16 = mp_.pack_gemm3(w).view(torch.int16).view(-1, 8)
    hi = lambda x: int(mp_._rne_hi(torch.tensor([x])).view(torch.int32).item()) >> 16
    for cb, s_, lane, j in ((0, 0, 0, 0), (3, 15, 63, 7), (2, 5, 37, 3)):
        want = hi(float(w[(lane >> 5) * 128 + 8 * s_ + j, 32 * cb + (lane & 31)]))
        got = int(p16[((cb * 16 + s_) * 3 + 0) * 64 + lane, j].item()) & 0xFFFF
        assert got == (want & 0xFFFF), (cb, s_, lane, j)
    # lane order: float4 (cb * K/8 + t4) * 64 + lane = W[(lane >> 5) * K/2 + 4*t4 + q][32*cb + (lane & 31)]
    pk, K = mp_.pack_gemm(w).view(-1, 4), 256
    for cb, t4, lane, q in ((0, 0, 0, 0), (3, 31, 63, 3), (1, 7, 40, 2)):
        assert pk[(cb * (K // 8) + t4) * 64 + lane, q] == w[(lane >> 5) * (K // 2) + 4 * t4 + q, 32 * cb + (lane & 31)]
    mp_.pack_policy(pol, out=flat)                  # in-place refresh keeps the storage
    with pytest.raises(ValueError):
        mp_.pack_policy(MPNN(num_agents=3, num_opp_agents=3, hidden_dim=32, num_actions=8))


@pytest.mark.parametrize("n,m", [(3, 3), (5, 5), (1, 2), (2, 1)])
def test_folded_training_trunk_equals_reference_shaped_trunk(n, m):
    """MPNN.trunk_folded (the PPO update's forward: folded linear maps + attend_mix) == MPNN.trunk: outputs and
    parameter gradients (CPU: attend_mix is its plain-torch statement here)."""
    from emergent_multiagent_strategies_amd.mpnn import MPNN
    torch.manual_seed(5 * n + m)
    pol = MPNN(num_agents=n, num_opp_agents=m, num_actions=8)
    for p in pol.parameters():
        if p.dim() == 1:
            p.data.uniform_(-0.3, 0.3)
    own, opp = torch.randn(40, n, 6), torch.randn(40, m, 6)
    w = torch.randn(40, n, 128)
    grads = []
    for fn in (pol.trunk, pol.trunk_folded):
        pol.zero_grad()
        h = fn(own, opp)
        (h * w).sum().backward()
        grads.append((h.detach().clone(), {k: p.grad.clone() for k, p in pol.named_parameters() if p.grad is not None}))
    assert (grads[0][0] - grads[1][0]).abs().max() < 2e-5
    # (a team of one: the reference-shaped trunk never touches messages.W_val / W_out -> no gradient at all;
    #  the folded one multiplies a zero message through them -> a zero gradient)
    for k in set(grads[0][1]) | set(grads[1][1]):
        a, b = [g[1].get(k, torch.zeros_like(dict(pol.named_parameters())[k])) for g in grads]
        assert (a - b).abs().max() <= 1e-4 * max(1.0, float(a.abs().max())), k


def test_flat_policy_layout_on_cpu():
    """mpnn_pack.FlatPolicy (construction needs no GPU): every parameter the kernels use lives in ONE buffer, in
    disjoint slices that cover it exactly, the module keeps working on them, in-place loads keep the aliasing and a
    re-pointed parameter is detected."""
    import torch
    from emergent_multiagent_strategies_amd import mpnn_pack as mp_
    from emergent_multiagent_strategies_amd.mpnn import MPNN
    torch.manual_seed(0)
    pol = MPNN(num_agents=3, num_opp_agents=3, hidden_dim=128, num_actions=8)
    ref = {k: v.clone() for k, v in pol.state_dict().items()}
    x_own, x_opp = torch.randn(5, 3, 6), torch.randn(5, 3, 6)
    before = pol.evaluate_actions(x_own, x_opp, torch.zeros(5, 3, 1, dtype=torch.int64))
    fp = mp_.FlatPolicy.of(pol)
    assert mp_.FlatPolicy.of(pol) is fp and fp.attached()
    spans = sorted((off, off + get(pol).numel()) for _, get, off in mp_._PF)
    assert spans[0][0] == 0 and spans[-1][1] == mp_.PF_FLOATS
    assert all(0 <= b[0] - a[1] < 4 for a, b in zip(spans, spans[1:]))     # disjoint; holes only to keep 16-byte alignment
    assert all(a[0] % 4 == 0 for a in spans)
    for k, v in pol.state_dict().items():
        assert torch.equal(v, ref[k])                                      # the values moved with the parameters
    after = pol.evaluate_actions(x_own, x_opp, torch.zeros(5, 3, 1, dtype=torch.int64))
    assert all(torch.equal(a, b) for a, b in zip(before, after))
    for _, get, off in mp_._PF:                                            # gradients alias the flat gradient buffer
        p = get(pol)
        assert p.grad.data_ptr() == fp.gflat.data_ptr() + 4 * off
    pol.load_state_dict({k: v + 1.0 for k, v in ref.items()})              # in place: still attached
    assert fp.attached() and float(fp.pflat[0]) == float(next(iter(mp_._PF))[1](pol).reshape(-1)[0])
    p = mp_._PF[0][1](pol)
    p.data = p.data.clone()
    assert not fp.attached()
    assert mp_.FlatPolicy.of(pol) is not fp                                # a detached one is rebuilt, not reused


def attacker_pool_from_golden(MPNN, golden_dir, device="cpu"):
    """The reference's published attacker policies (marlsave/tmp_1/ep*.pt, exported as data into
    tests/golden/attackers_tmp1.npz by oracle/gen_golden.py) as this repo's modules."""
    z = np.load(os.path.join(golden_dir, "attackers_tmp1.npz"))
    G, A, B = [int(v) for v in z["meta"]]
    pool = []
    for e in [int(e) for e in z["episodes"]]:
        sd = {k[len("ep%d." % e):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("ep%d." % e) and ".out." not in k}
        pol = MPNN(num_agents=A, num_opp_agents=G, num_actions=8)
        res = pol.load_state_dict(sd, strict=True)
        assert not res.missing_keys and not res.unexpected_keys
        pool.append(pol.to(device).eval())
    return z, pool, G, A


def test_published_attackers_forward_matches_the_reference(MPNN, golden_dir):
    """Config 5's frozen strategies: the five shipped checkpoints' attacker policies through this repo's module
    reproduce the reference mpnn.py's values and log-softmax logits on the fixture's 5v5 observations."""
    z, pool, G, A = attacker_pool_from_golden(MPNN, golden_dir)
    obs = torch.from_numpy(z["obs"])
    for e, pol in zip([int(e) for e in z["episodes"]], pool):
        with torch.no_grad():
            lg, v = pol.logits_value(obs[:, G:], obs[:, :G])
        # (trained policies: logits up to +-40, values up to +-30 -- float32 association differences scale with them)
        want_v, want_lp = z["ep%d.out.value" % e], z["ep%d.out.logp_all" % e]
        assert np.abs(v[..., 0].numpy() - want_v).max() < 2e-5 * max(1.0, np.abs(want_v).max()), e
        assert np.abs(torch.log_softmax(lg, -1).numpy() - want_lp).max() < 1e-4, e
