"""GPU numerics of the fused policy kernel (csrc/fa_policy.hip, through fa_policy_act / fa_collect_act)
against the plain PyTorch fp32 MPNN module (emergent-multiagent-strategies_amd/mpnn.py, itself pinned to
the reference's mpnn.py by tests/test_mpnn_cpu.py).

Tolerance: 2e-4 absolute on values and log-probs (fp32 MFMA chains of length 64..256 over five to six
layers, three linear-map pairs pre-multiplied on the host; observed ~1e-5).  Sampling is checked as a
distribution (it is the engine's own Philox stream, not torch's).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 2e-4


@pytest.fixture(scope="module")
def fa():
    import emergent_multiagent_strategies_amd as m
    assert torch.cuda.is_available()
    m._lib.load()
    return m


def _policies(fa, G, A, seed):
    from emergent_multiagent_strategies_amd import mpnn_pack
    torch.manual_seed(seed)
    pols = [fa.MPNN(num_agents=G, num_opp_agents=A, num_actions=8).cuda(),
            fa.MPNN(num_agents=A, num_opp_agents=G, num_actions=8).cuda()]
    for pol in pols:
        for p in pol.parameters():          # non-zero biases, larger logits than the 0.01-gain init gives
            if p.dim() == 1:
                p.data.uniform_(-0.3, 0.3)
        pol.dist.linear.weight.data.mul_(3.0)
    return pols, [mpnn_pack.pack_policy(p) for p in pols]


def _obs(E, N, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    obs = torch.randn((E, N, 6), device="cuda", generator=g)
    obs[:, :, 0] = (torch.rand((E, N), device="cuda", generator=g) > 0.3).float()
    obs[:, :, 3] = obs[:, :, 3] * 3 + 4.7    # headings are O(1..10)
    return obs.contiguous()


def _torch_reference(pols, obs, G):
    with torch.no_grad():
        lg, vg = pols[0].logits_value(obs[:, :G], obs[:, G:])
        la, va = pols[1].logits_value(obs[:, G:], obs[:, :G])
    return torch.cat((lg, la), 1), torch.cat((vg, va), 1)[..., 0]


@pytest.mark.parametrize("G,A,E", [(3, 3, 4096), (5, 5, 1000), (2, 4, 333), (1, 3, 65), (8, 8, 50), (3, 3, 1), (4, 1, 97),
                                   (5, 5, 4096),    # config 5's per-GPU shape: 2 x 216 workgroups of 19 envs
                                   (5, 5, 3990), (5, 3, 4096), (2, 5, 4096)])
def test_fused_forward_matches_torch_module(fa, G, A, E):
    N = G + A
    pols, packed = _policies(fa, G, A, 10 * G + A)
    eng = fa.BatchedFortAttack(E, G, A, 20)
    assert eng.policy_variant() == ("fa_policy_kernel<3, 8>" if 2 * -(-E // (96 // max(G, A))) >= 192 else "fa_policy_kernel<2, 4>")
    obs = _obs(E, N, E)
    logits, value = _torch_reference(pols, obs, G)
    v, act, lp = eng.policy_act(obs, packed[0], packed[1], deterministic=True)
    assert (v - value).abs().max() < TOL
    logp_all = F.log_softmax(logits, dim=-1)
    assert (lp - logp_all.gather(-1, act.unsqueeze(-1))[..., 0]).abs().max() < TOL
    top2 = logits.topk(2, dim=-1).values
    clear = (top2[..., 0] - top2[..., 1]) > 1e-3                  # argmax is only defined away from ties
    assert torch.equal(act[clear], logits.argmax(-1)[clear]) and float(clear.float().mean()) > 0.9
    assert int(act.min()) >= 0 and int(act.max()) <= 7
    v2, none_a, none_l = eng.policy_act(obs, packed[0], packed[1], value_only=True)   # get_value (mpnn.py:202-205)
    assert torch.equal(v2, v) and none_a is None and none_l is None


@pytest.mark.parametrize("tag", ["3v3", "5v5"])
def test_fused_forward_matches_the_reference_golden_at_full_size(fa, golden_dir, tag):
    """fa_policy_kernel against the REFERENCE's mpnn.py at hidden_dim 128 (tests/golden/mpnn_h128.npz: values and
    log-softmax logits of the reference module; its seed-constructed weights are re-made here and proven equal by
    their fingerprints): no hop through this repo's PyTorch module."""
    import os
    from emergent_multiagent_strategies_amd import mpnn_pack
    from test_mpnn_cpu import h128_policies
    g = np.load(os.path.join(golden_dir, "mpnn_h128.npz"))
    pols, G, A = h128_policies(fa.MPNN, g, tag, device="cuda")
    packed = [mpnn_pack.pack_policy(p) for p in pols]
    obs = torch.from_numpy(g[tag + ".obs"]).cuda().contiguous()
    E = obs.shape[0]
    eng = fa.BatchedFortAttack(E, G, A, 20)
    v, act, lp = eng.policy_act(obs, packed[0], packed[1], deterministic=True)
    want_v = torch.from_numpy(g[tag + ".value"]).cuda()
    want_lp = torch.from_numpy(g[tag + ".logp_all"]).cuda()
    assert (v - want_v).abs().max() < TOL
    assert (lp - want_lp.gather(-1, act.unsqueeze(-1))[..., 0]).abs().max() < TOL
    top2 = want_lp.topk(2, dim=-1).values
    clear = (top2[..., 0] - top2[..., 1]) > 1e-3
    assert torch.equal(act[clear], want_lp.argmax(-1)[clear]) and float(clear.float().mean()) > 0.9
    # sampled actions' log-probs come from the same table
    counter = torch.zeros(1, dtype=torch.int64, device="cuda")
    _, act_s, lp_s = eng.policy_act(obs, packed[0], packed[1], seed=5, counter=counter, step=0)
    assert (lp_s - want_lp.gather(-1, act_s.unsqueeze(-1))[..., 0]).abs().max() < TOL


def test_sampling_is_a_draw_from_softmax_and_reproducible(fa):
    G, A, E = 3, 3, 16384
    N = G + A
    pols, packed = _policies(fa, G, A, 3)
    eng = fa.BatchedFortAttack(E, G, A, 20)
    obs = _obs(1, N, 9).expand(E, N, 6).contiguous()             # every env sees the same observation
    logits, _ = _torch_reference(pols, obs[:1], G)
    probs = F.softmax(logits[0], dim=-1)                          # (N, 8)
    counter = torch.zeros(1, dtype=torch.int64, device="cuda")
    v, act, lp = eng.policy_act(obs, packed[0], packed[1], seed=123, counter=counter, step=5)
    freq = torch.stack([(act == k).float().mean(0) for k in range(8)], -1)          # (N, 8)
    sigma = torch.sqrt(probs * (1 - probs) / E) + 1e-4
    assert float(((freq - probs).abs() / sigma).max()) < 5.5     # every cell within 5.5 sigma
    assert (lp - F.log_softmax(logits[0], -1)[None].expand(E, N, 8).gather(-1, act.unsqueeze(-1))[..., 0]).abs().max() < TOL
    _, act2, lp2 = eng.policy_act(obs, packed[0], packed[1], seed=123, counter=counter, step=5)
    assert torch.equal(act, act2) and torch.equal(lp, lp2)        # same key -> same draws
    for kw in (dict(seed=124, step=5), dict(seed=123, step=6)):
        _, other, _ = eng.policy_act(obs, packed[0], packed[1], counter=counter, **kw)
        assert float((other != act).float().mean()) > 0.3
    counter.add_(1)                                               # the per-rollout counter is part of the key
    _, other, _ = eng.policy_act(obs, packed[0], packed[1], seed=123, counter=counter, step=5)
    assert float((other != act).float().mean()) > 0.3
    # agents of one env and neighbouring envs draw independently
    a0 = act[:, 0].float()
    assert abs(float(torch.corrcoef(torch.stack((a0[:-1], a0[1:])))[0, 1])) < 0.05
    assert abs(float(torch.corrcoef(torch.stack((act[:, 0].float(), act[:, 1].float())))[0, 1])) < 0.05


def test_env_shards_draw_what_one_big_batch_draws(fa):
    """The sampling key holds the GLOBAL env index: two half handles (env_offset) == one full handle."""
    G, A, E = 3, 3, 640
    N = G + A
    pols, packed = _policies(fa, G, A, 4)
    obs = _obs(E, N, 2)
    full = fa.BatchedFortAttack(E, G, A, 20).policy_act(obs, packed[0], packed[1], seed=7, step=3)
    lo = fa.BatchedFortAttack(E // 2, G, A, 20, env_offset=0).policy_act(obs[:E // 2].contiguous(), packed[0], packed[1], seed=7, step=3)
    hi = fa.BatchedFortAttack(E // 2, G, A, 20, env_offset=E // 2).policy_act(obs[E // 2:].contiguous(), packed[0], packed[1], seed=7, step=3)
    for k in range(3):
        assert torch.equal(full[k], torch.cat((lo[k], hi[k])))


@pytest.mark.parametrize("G,A", [(3, 3), (5, 5), (2, 4)])
def test_eight_wave_tile_kernel_draws_what_the_four_wave_kernel_draws(fa, G, A):
    """4096 envs in one handle run fa_policy_kernel<3, 8> (96-row tiles, eight waves, Gumbel noise drawn by idle waves);
    the same envs as shards of 512 run fa_policy_kernel<2, 4> (64-row tiles, four waves, noise drawn inline).  A row's
    arithmetic is the same in both -- K order of the MFMA chains, attention, sampling key -- so values, sampled actions
    and log-probs agree bit for bit."""
    E, S = 4096, 512
    N = G + A
    pols, packed = _policies(fa, G, A, 6)
    obs = _obs(E, N, 9)
    counter = torch.full((1,), 3, dtype=torch.int64, device="cuda")
    full = fa.BatchedFortAttack(E, G, A, 20).policy_act(obs, packed[0], packed[1], seed=11, counter=counter, step=5)
    parts = [fa.BatchedFortAttack(S, G, A, 20, env_offset=o).policy_act(obs[o:o + S].contiguous(), packed[0], packed[1], seed=11,
                                                                         counter=counter, step=5) for o in range(0, E, S)]
    for k in range(3):
        assert torch.equal(full[k], torch.cat([p[k] for p in parts]))
    assert len(torch.unique(full[1])) == 8                                    # every action is drawn somewhere


def test_collect_act_writes_the_policy_rows(fa):
    G, A, E, T = 3, 3, 200, 6
    N = G + A
    pols, packed = _policies(fa, G, A, 5)
    eng = fa.BatchedFortAttack(E, G, A, 20, base_seed=1)
    st = fa.JointRolloutStorage(T, E, N, device="cuda")
    eng.bind_storage(st)
    eng.collect_reset()
    for s in range(T):
        eng.collect_act(s, packed[0], packed[1], seed=11)
        eng.collect_step(s)
    eng.collect_act(T, packed[0], packed[1], value_only=True)
    for s in range(T + 1):
        logits, value = _torch_reference(pols, st.obs[s], G)
        assert (st.value_preds[s, :, :, 0] - value).abs().max() < TOL, s
        if s < T:
            want = F.log_softmax(logits, -1).gather(-1, st.actions[s])[..., 0]
            assert (st.action_log_probs[s, :, :, 0] - want).abs().max() < TOL, s
    assert len(torch.unique(st.actions)) == 8
    with pytest.raises(fa.FaError):
        eng.collect_act(T, packed[0], packed[1])                 # step T has no action row


@pytest.mark.parametrize("B,n,nk,W,skip", [(1000, 3, 3, 128, True), (1000, 3, 3, 64, False), (333, 5, 5, 128, True),
                                            (77, 2, 7, 64, False), (50, 1, 1, 128, True), (16384, 3, 3, 128, True),
                                            (19, 8, 8, 128, True)])
def test_attend_mix_forward_and_backward_match_torch(fa, B, n, nk, W, skip):
    """fa_attend_forward / fa_attend_backward (the PPO update's attention op) against the plain-torch statement
    under autograd: outputs, dg and dkeys."""
    from emergent_multiagent_strategies_amd.mpnn import attend_mix, attend_mix_reference
    if skip:
        nk = n
    g0 = torch.randn(B * n, W, device="cuda")
    k0 = torch.randn(B, nk, W, device="cuda") * 0.3
    w = torch.randn(B * n, W, device="cuda")
    res = []
    for fn in (attend_mix_reference, attend_mix):
        g, k = g0.clone().requires_grad_(True), k0.clone().requires_grad_(True)
        out = fn(g, k, n, skip)
        (out * w).sum().backward()
        res.append((out.detach(), g.grad, k.grad))
    for a, b in zip(*res):
        a = torch.zeros_like(b) if a is None else a          # (a team of one: the torch statement has no key gradient)
        assert (a - b).abs().max() <= 2e-5 * max(1.0, float(a.abs().max()))


@pytest.mark.parametrize("G,A", [(3, 3), (5, 5)])
def test_update_forward_on_gpu_matches_reference_shaped_forward(fa, G, A):
    """evaluate_actions under autograd on the GPU (folded trunk + HIP attention op) vs the reference-shaped
    trunk: value / log-prob / entropy and every parameter gradient of a PPO-like loss."""
    pols, _ = _policies(fa, G, A, 7)
    pol = pols[0]
    obs = _obs(2048, G + A, 3)
    own, opp = obs[:, :G], obs[:, G:]
    act = torch.randint(0, 8, (2048, G, 1), device="cuda")
    res = []
    for fold in (False, True):
        pol.fold_update = fold
        pol.zero_grad()
        v, lp, ent = pol.evaluate_actions(own, opp, act)
        (v.pow(2).mean() + lp.mean() - 0.01 * ent.mean()).backward()
        res.append((v.detach(), lp.detach(), ent.detach(), {k: p.grad.clone() for k, p in pol.named_parameters() if p.grad is not None}))
    pol.fold_update = True
    for k in range(3):
        assert (res[0][k] - res[1][k]).abs().max() < TOL
    for k, ga in res[0][3].items():
        assert (ga - res[1][3][k]).abs().max() <= 2e-4 * max(1e-3, float(ga.abs().max())), k


@pytest.mark.parametrize("G,A,team,B,clipped", [(3, 3, 0, 500, True), (3, 3, 1, 21, True), (3, 3, 0, 43, False),
                                                (5, 5, 1, 200, True), (2, 4, 0, 100, True), (4, 2, 1, 77, True),
                                                (8, 8, 0, 37, True), (7, 8, 1, 26, True), (1, 3, 0, 50, True)])
def test_fused_ppo_grad_matches_torch_autograd(fa, G, A, team, B, clipped):
    """fa_ppo_grad (forward + PPO losses + complete backward of one team's minibatch, one launch) against torch
    autograd through the plain-torch statement of the same computation on the same kernel-facing matrices."""
    from emergent_multiagent_strategies_amd import mpnn_pack as mp_
    from emergent_multiagent_strategies_amd.env import ppo_grad
    N = G + A
    pols, _ = _policies(fa, G, A, 11 + team)
    pol = pols[team]
    own_sl, opp_sl = (slice(0, G), slice(G, N)) if team == 0 else (slice(G, N), slice(0, G))
    n = G if team == 0 else A
    g = torch.Generator(device="cuda").manual_seed(B)
    obs = _obs(B, N, B + 1)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
    action = torch.randint(0, 8, (B, N, 1), device="cuda", generator=g)
    value_pred, ret, adv = rnd(B, N, 1), rnd(B, N, 1), rnd(B, N, 1)
    old_logp = -torch.rand((B, N, 1), device="cuda", generator=g) * 2.5
    clip, c_v, c_e = 0.2, 0.5, 0.01
    P = {k: v.detach().clone().requires_grad_(True) for k, v in mp_.kernel_params(pol).items()}
    loss, vl, al, en, mm = mp_.folded_ppo_reference(P, obs, own_sl, opp_sl, action[:, own_sl], value_pred[:, own_sl], ret[:, own_sl],
                                                    old_logp[:, own_sl], adv[:, own_sl], clip, c_v, c_e, clipped)
    loss.backward()
    w = torch.zeros(mp_.WEIGHT_FLOATS, device="cuda")
    wt = torch.zeros(mp_.TRANS_FLOATS, device="cuda")
    mp_.pack_from_params(P, w, wt)
    scale = torch.tensor([1.0 / (B * n), 1.0], device="cuda")
    out, _ = ppo_grad(obs, action, value_pred, ret, old_logp, adv, w, wt, scale, team, G, A, clip, c_v, c_e, clipped)
    torch.cuda.synchronize()
    sums = out[mp_.PLAIN_FLOATS:mp_.PLAIN_FLOATS + 4] / (B * n)
    for got, want in zip(sums, (vl, al, en, mm)):
        assert abs(float(got) - float(want)) <= 1e-5 * max(1.0, abs(float(want)))
    grads = mp_.split_plain(out)
    worst = {}
    for k, gk in grads.items():
        ref = P[k].grad
        if ref is None:                  # a team of one has no team attention: A_m never enters the loss
            assert float(gk.abs().max()) == 0.0, k
            continue
        if k == "W9":
            # W9 is block diagonal: rows 0..127 x columns 0..7 are dist.linear.weight^T, rows 128..255 x column 8 is
            # value_head.2.weight^T.  The kernel produces the gradients of those PARAMETER entries; the structural zeros
            # have none (the autograd reference, which treats W9 as a dense matrix, does give them one: masked here)
            real = torch.zeros_like(ref, dtype=torch.bool)
            real[:128, :8] = True
            real[128:, 8] = True
            assert float(gk[~real].abs().max()) == 0.0
            ref = ref * real
        if k == "B9":
            gk, ref = gk[:9], ref[:9]
        scale_k = max(float(ref.abs().max()), 1e-6)
        worst[k] = float((gk - ref).abs().max()) / scale_k
    print({k: "%.1e" % v for k, v in worst.items()})
    assert max(worst.values()) < 2e-3, worst


@pytest.mark.parametrize("G,A,B", [(3, 3, 16384), (5, 5, 4096)])
def test_dw_gemm_split_bf16_is_fp32_class_against_an_fp64_gemm(fa, monkeypatch, G, A, B):
    """The weight-gradient GEMM of fa_ppo_grad on the bf16 matrix cores (csrc/fa_train_dw.hip fa_train_dw3_kernel: every
    float32 operand split EXACTLY into three bf16 terms, six of the nine cross products, fp32 accumulate) against the
    fp32-MFMA form (FA_DW_GEMM=f32) -- both against the SAME products summed in float64 from the operand records the tile
    kernel left (config 3's minibatch: 1 639 tiles, 4 917 + 1 639 records).  Bar: the split form's largest error is at most
    2 x the fp32-MFMA kernel's (it is smaller: its products are exact, only the accumulation rounds)."""
    from emergent_multiagent_strategies_amd import mpnn_pack as mp_
    from emergent_multiagent_strategies_amd.env import ppo_grad
    N = G + A
    pols, _ = _policies(fa, G, A, 3)
    g = torch.Generator(device="cuda").manual_seed(B)
    obs = _obs(B, N, 5)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
    action = torch.randint(0, 8, (B, N, 1), device="cuda", generator=g)
    value_pred, ret, adv = rnd(B, N, 1), rnd(B, N, 1), rnd(B, N, 1)
    old_logp = -torch.rand((B, N, 1), device="cuda", generator=g) * 2.5
    w, wt = torch.zeros(mp_.WEIGHT_FLOATS, device="cuda"), torch.zeros(mp_.TRANS_FLOATS, device="cuda")
    mp_.pack_from_params(mp_.kernel_params(pols[0]), w, wt)
    scale = torch.tensor([1.0 / (B * G), 1.0], device="cuda")
    outs = {}
    sc = None
    for mode in ("f32", "bf16x3"):
        if mode == "f32":
            monkeypatch.setenv("FA_DW_GEMM", "f32")
        else:
            monkeypatch.delenv("FA_DW_GEMM", raising=False)
        o, sc = ppo_grad(obs, action, value_pred, ret, old_logp, adv, w, wt, scale, 0, G, A, 0.2, 0.5, 0.01, True, scratch=sc)
        torch.cuda.synchronize()
        outs[mode] = {k: v.clone() for k, v in mp_.split_plain(o).items()}
        o2, _ = ppo_grad(obs, action, value_pred, ret, old_logp, adv, w, wt, scale, 0, G, A, 0.2, 0.5, 0.01, True, scratch=sc)
        torch.cuda.synchronize()
        L = mp_.PLAIN_FLOATS + 4
        assert torch.equal(o[:L], o2[:L]), mode                         # bitwise reproducible, either form
    # the products in float64 from the records (csrc/fa_train.h: FA_RECA_* / FA_RECB_*)
    tiles = -(-B // (32 // max(G, A)))
    hs = sc[1]
    ra = hs[:tiles * 3 * 16384].view(tiles * 3, 4, 32, 128)
    rb = hs[tiles * 3 * 16384:tiles * 3 * 16384 + tiles * 20480].view(tiles, 20480)
    ref = {k: 0 for k in ("W7", "AM", "W8", "BO", "AO")}
    for q in range(0, tiles * 3, 1024):
        c = ra[q:q + 1024].double()
        ref["W7"] = ref["W7"] + torch.einsum("qrk,qrj->kj", torch.cat((c[:, 0], c[:, 1]), -1), c[:, 2])
        ref["AM"] = ref["AM"] + torch.einsum("qrk,qrj->kj", c[:, 0], c[:, 3])
    for q in range(0, tiles, 1024):
        c = rb[q:q + 1024].double()
        ref["W8"] = ref["W8"] + torch.einsum("qrk,qrj->kj", c[:, :4096].view(-1, 32, 128), c[:, 4096:12288].view(-1, 32, 256))
        ref["BO"] = ref["BO"] + torch.einsum("qrk,qrj->kj", c[:, 12288:14336].view(-1, 32, 64), c[:, 14336:16384].view(-1, 32, 64))
        ref["AO"] = ref["AO"] + torch.einsum("qrk,qrj->kj", c[:, 16384:18432].view(-1, 32, 64), c[:, 18432:20480].view(-1, 32, 64))
    err = {m: {k: float((outs[m][k].double() - ref[k]).abs().max() / ref[k].abs().max()) for k in ref} for m in outs}
    print({m: {k: "%.1e" % v for k, v in e.items()} for m, e in err.items()})
    for k in ref:
        assert float(ref[k].abs().max()) > 0
        assert err["f32"][k] < 1e-5, (k, err["f32"][k])                 # sanity of the reference itself
        assert err["bf16x3"][k] <= 2.0 * err["f32"][k] + 1e-8, (k, err)
    # everything that is not a product of this GEMM is the tile kernel's: identical bits in both forms
    for k in outs["f32"]:
        if k not in ref:
            assert torch.equal(outs["f32"][k], outs["bf16x3"][k]), k


def test_fused_ppo_grad_normalises_the_advantages_itself(fa):
    """fa_ppo_grad with (adv_mean, adv_std) instead of an advantage tensor (ppo.py:121-124 inside the kernel): bit for bit
    the result of passing what fa_adv_normalize writes -- (A - (float)mean) / ((float)std + 1e-5f) in float32."""
    from emergent_multiagent_strategies_amd import mpnn_pack as mp_
    from emergent_multiagent_strategies_amd.env import ppo_grad
    G, A, B = 3, 3, 700
    N = G + A
    pols, _ = _policies(fa, G, A, 5)
    g = torch.Generator(device="cuda").manual_seed(1)
    obs = _obs(B, N, 3)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
    action = torch.randint(0, 8, (B, N, 1), device="cuda", generator=g)
    value_pred, ret = rnd(B, N, 1), rnd(B, N, 1) * 3 + 1
    old_logp = -torch.rand((B, N, 1), device="cuda", generator=g) * 2.5
    mean = (torch.rand(N, device="cuda", generator=g).double() - 0.5) * 2
    std = torch.rand(N, device="cuda", generator=g).double() * 3 + 0.1
    adv = ((ret - value_pred) - mean.float().view(1, N, 1)) / (std.float().view(1, N, 1) + 1e-5)
    P = mp_.kernel_params(pols[1])
    w, wt = torch.zeros(mp_.WEIGHT_FLOATS, device="cuda"), torch.zeros(mp_.TRANS_FLOATS, device="cuda")
    mp_.pack_from_params(P, w, wt)
    a, _ = ppo_grad(obs, action, value_pred, ret, old_logp, adv.contiguous(), w, wt, None, 1, G, A, 0.2, 0.5, 0.01, True)
    b, _ = ppo_grad(obs, action, value_pred, ret, old_logp, None, w, wt, None, 1, G, A, 0.2, 0.5, 0.01, True, adv_stats=(mean, std))
    torch.cuda.synchronize()
    used = mp_.PLAIN_FLOATS + 4                        # gradients + the four loss sums (the buffer's tail is scratch)
    assert torch.equal(a[:used], b[:used]) and float(a[:used].abs().sum()) > 0


@pytest.mark.parametrize("G,A", [(3, 3), (5, 2)])
def test_flat_policy_fold_and_unfold_match_torch(fa, G, A):
    """mpnn_pack.FlatPolicy: parameters as one flat buffer; fold_pack (task-list kernel + pack kernel) == the torch
    packing of kernel_params(); unfold == torch autograd's chain rule through kernel_params(); the module still
    computes the same function and optimizers see the same parameters."""
    from emergent_multiagent_strategies_amd import mpnn_pack as mp_
    pols, _ = _policies(fa, G, A, 21)
    pol = pols[0]
    obs = _obs(64, G + A, 5)
    with torch.no_grad():
        before = pol.logits_value(obs[:, :G], obs[:, G:])
    P = mp_.kernel_params(pol)
    w_ref, wt_ref = torch.zeros(mp_.WEIGHT_FLOATS, device="cuda"), torch.zeros(mp_.TRANS_FLOATS, device="cuda")
    mp_.pack_from_params(P, w_ref, wt_ref)
    fp = mp_.FlatPolicy(pol)
    with torch.no_grad():
        after = pol.logits_value(obs[:, :G], obs[:, G:])
    assert torch.equal(before[0], after[0]) and torch.equal(before[1], after[1])
    w, wt = fp.fold_pack()
    PF = mp_.PLAIN_FLOATS
    assert (w[:PF] - w_ref[:PF]).abs().max() <= 2e-6 * max(1.0, float(w_ref[:PF].abs().max()))
    # the bf16x3 half of the pack (what fa_policy_kernel's GEMMs read) is the SAME matrices, exactly: hi + mid + lo == float32
    for k, (K, C) in (("AO", (64, 64)), ("BO", (64, 64)), ("AM", (128, 128)), ("W7", (256, 128)), ("W8", (128, 256)), ("W9", (256, 32))):
        a = mp_.unpack_gemm(w[mp_.POFF[k]:mp_.POFF[k] + K * C], K, C)
        b = mp_.unpack_gemm3(w[mp_.POFF3[k]:mp_.POFF3[k] + K * C * 3 // 2], K, C)
        assert torch.equal(a, b), k
        assert torch.equal(mp_.pack_gemm3(a), w[mp_.POFF3[k]:mp_.POFF3[k] + K * C * 3 // 2]), k    # device pack == host pack, bit for bit
    assert (wt - wt_ref).abs().max() <= 2e-6 * max(1.0, float(wt_ref.abs().max()))
    # chain rule: random plain-layout gradients through both routes
    gplain = torch.randn(mp_.SLAB_FLOATS, device="cuda") * 0.1
    views = mp_.split_plain(gplain)
    views["W9"][:, 9:] = 0
    views["B9"][9:] = 0
    pol.zero_grad(set_to_none=True)
    P = mp_.kernel_params(pol)
    torch.autograd.backward([P[k] for k in mp_.PLAIN_SHAPES], [views[k] for k in mp_.PLAIN_SHAPES])
    ref = {n: p.grad.clone() for n, p in pol.named_parameters() if p.grad is not None}
    fp.attach_grads()
    fp.unfold(gplain)
    torch.cuda.synchronize()
    for n, p in pol.named_parameters():
        if n in ref:
            assert (p.grad - ref[n]).abs().max() <= 5e-6 * max(1.0, float(ref[n].abs().max())), n
    # an optimizer step through the views moves the flat buffer
    opt = torch.optim.SGD(pol.parameters(), lr=0.1)
    snap = fp.pflat.clone()
    opt.step()
    assert not torch.equal(snap, fp.pflat) and pol.update[0].weight.data_ptr() == fp.pflat[mp_._PF[12][2]:].data_ptr()


@pytest.mark.parametrize("normalize", [True, False])
def test_fused_ppo_grad_gathers_rows_and_takes_the_mask_mean_itself(fa, normalize):
    """fa_ppo_grad with idx (the minibatch = rows idx of the rollout, read in place) and scale = NULL (the library's
    own alive-mask mean) == the same call on the gathered rows with the scale pair computed by torch."""
    from emergent_multiagent_strategies_amd import mpnn_pack as mp_
    from emergent_multiagent_strategies_amd.env import ppo_grad
    G, A, team, R, B = 3, 3, 1, 900, 260
    N, n = G + A, A
    pols, _ = _policies(fa, G, A, 5)
    w = mp_.pack_policy(pols[team])
    wt = torch.zeros(mp_.TRANS_FLOATS, device="cuda")
    mp_.pack_from_params(mp_.kernel_params(pols[team]), torch.zeros_like(w), wt)
    g = torch.Generator(device="cuda").manual_seed(3)
    obs = _obs(R, N, 17)
    obs[:, :, 0] = (torch.rand((R, N), device="cuda", generator=g) < 0.6).float()       # alive flags
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
    action = torch.randint(0, 8, (R, N, 1), device="cuda", generator=g)
    value_pred, ret, adv = rnd(R, N, 1), rnd(R, N, 1), rnd(R, N, 1)
    old_logp = -torch.rand((R, N, 1), device="cuda", generator=g) * 2.5
    idx = torch.randperm(R, device="cuda", generator=g)[:B].contiguous()
    mm = obs[idx][:, G:, 0].mean()
    scale = torch.stack((1.0 / (B * n * mm) if normalize else torch.tensor(1.0 / (B * n), device="cuda"), mm)).float()
    sel = lambda t: t[idx].contiguous()
    want, _ = ppo_grad(sel(obs), sel(action), sel(value_pred), sel(ret), sel(old_logp), sel(adv), w, wt, scale, team, G, A,
                       0.2, 0.5, 0.01, True)
    got, _ = ppo_grad(obs, action, value_pred, ret, old_logp, adv, w, wt, None, team, G, A, 0.2, 0.5, 0.01, True, idx=idx,
                      normalize=normalize)
    torch.cuda.synchronize()
    L = mp_.PLAIN_FLOATS
    assert abs(float(got[L + 9]) - float(mm)) < 1e-6
    assert (got[:L + 4] - want[:L + 4]).abs().max() <= 1e-5 * float(want[:L].abs().max())


def test_flat_adam_step_matches_torch_adam_after_clip(fa):
    """fa_adam_step (global-norm clip + Adam over the flat parameter buffer, 2 launches) against
    nn.utils.clip_grad_norm_ + torch.optim.Adam.step on a copy, three steps; the two share one optimizer state."""
    import copy
    from emergent_multiagent_strategies_amd import mpnn_pack as mp_
    pols, _ = _policies(fa, 3, 3, 9)
    pol = pols[0]
    ref = copy.deepcopy(pol)
    opt = torch.optim.Adam(pol.parameters(), lr=1e-3, capturable=True)
    opt_ref = torch.optim.Adam(ref.parameters(), lr=1e-3, capturable=True)
    fp = mp_.FlatPolicy.of(pol)
    fp.bind_adam(opt)
    names = dict(pol.named_parameters())
    g = torch.Generator(device="cuda").manual_seed(1)
    for it in range(4):
        fp.attach_grads()
        fp.gflat[:mp_.PF_FLOATS].copy_(torch.randn(mp_.PF_FLOATS, device="cuda", generator=g) * (0.002 if it == 1 else 0.05))
        for k, p in ref.named_parameters():
            p.grad = names[k].grad.clone() if names[k].grad is not None else None
        torch.nn.utils.clip_grad_norm_([p for p in ref.parameters() if p.grad is not None], 0.5)
        opt_ref.step()
        if it == 2:      # the optimizer's own step on the shared state (what the torch update path does)
            torch.nn.utils.clip_grad_norm_([p for p in pol.parameters() if p.grad is not None], 0.5)
            opt.step()
        else:
            fp.adam_step(opt, 0.5)
        for k, p in ref.named_parameters():
            if p.grad is not None:
                assert (names[k].detach() - p.detach()).abs().max() <= 2e-6 * max(1.0, float(p.detach().abs().max())), (it, k)
                assert (names[k].grad - p.grad).abs().max() <= 5e-5 * max(1e-3, float(p.grad.abs().max())), (it, k)   # (the norms are summed in different orders)
    assert float(fp.steps.min()) == 4.0 and float(fp.steps.max()) == 4.0


def test_published_attackers_through_the_fused_kernel_and_the_ensemble_loader(fa, golden_dir):
    """BASELINE config 5 with the PUBLISHED policies (reference learner.py:119-140, train_fortattack_v2.py:29-35): the
    attacker policies of marlsave/tmp_1/ep{220,650,1240,1600,2520}.pt (tests/golden/attackers_tmp1.npz: weights as data +
    the reference mpnn.py's outputs on a fixed 5v5 batch) loaded by BatchedLearner.load_attacker_ensemble and run by
    fa_policy_kernel's grouped-by-strategy launch: every env's attacker values / log-probs are the reference's for the
    strategy that env plays."""
    from test_mpnn_cpu import attacker_pool_from_golden
    z, pool, G, A = attacker_pool_from_golden(fa.MPNN, golden_dir)
    N, K = G + A, len(pool)
    obs = torch.from_numpy(z["obs"]).cuda().contiguous()
    B = obs.shape[0]
    eng = fa.BatchedFortAttack(B, G, A, 20)
    L = fa.BatchedLearner(eng, num_steps=4, num_mini_batch=1, ppo_epoch=1)
    assert L.policy_backend == "hip"
    L.load_attacker_ensemble([{"models": [None] * G + [p.state_dict()] * A, "ob_rms": (None, None)} for p in pool])
    assert L._packed_pool is not None and L._packed_pool.shape[0] == K and L.policy_backend == "hip"
    eps = [int(e) for e in z["episodes"]]
    for shift in range(K):                                        # every env meets every strategy once
        strat = ((torch.arange(B) + shift) % K).to(torch.int32).cuda()
        v, act, lp = eng.policy_act(obs, L._packed[0], None, deterministic=True, pool=L._packed_pool, env_strategy=strat)
        want_v = torch.stack([torch.from_numpy(z["ep%d.out.value" % e]) for e in eps]).cuda()[strat.long(), torch.arange(B).cuda()]
        want_lp = torch.stack([torch.from_numpy(z["ep%d.out.logp_all" % e]) for e in eps]).cuda()[strat.long(), torch.arange(B).cuda()]
        # trained policies: values up to +-20, logits up to +-40 -- fp32 folded algebra is ~1e-6 RELATIVE to those
        assert (v[:, G:] - want_v).abs().max() < 1e-5 * float(want_v.abs().max()) + 1e-5
        got_lp = lp[:, G:]
        assert (got_lp - want_lp.gather(-1, act[:, G:].unsqueeze(-1))[..., 0]).abs().max() < 2e-4
        top2 = want_lp.topk(2, dim=-1).values
        clear = (top2[..., 0] - top2[..., 1]) > 1e-3
        assert torch.equal(act[:, G:][clear], want_lp.argmax(-1)[clear])
