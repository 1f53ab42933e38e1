"""MPNN actor-critic (the reference's mpnn.py) for batched rollouts on ROCm.

Stays in PyTorch per the north star (its GEMMs go to rocBLAS/hipBLASLt, i.e. MFMA); what
changes relative to the reference module is what made it unusable inside a device-resident
loop, not the math:
  * input is env-major ``(B, n, 6)`` / ``(B, m, 6)`` -- exactly one row ``obs[s][:, team]`` of
    the joint rollout buffer, no cat/chunk per agent (learner.py:150-170) and no
    view/transpose round trip (mpnn.py:132-134, :161);
  * no ``.cpu().numpy()`` export of the attention matrices inside the forward
    (mpnn.py:140, :166: two host syncs per call) -- they are returned as tensors on request;
  * the ``-inf`` diagonal of the team self-attention is a constant buffer, not a Python loop
    over agents (mpnn.py:297-298).
The parameter names, shapes and *initialisation order* are the reference's, so a reference
``state_dict`` loads as is and the same ``torch.manual_seed`` yields the
same weights.  ``oppUpdate`` is created but unused, exactly as in the reference (mpnn.py:44-45).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class _SplitKMatmul(torch.autograd.Function):
    """y = x @ w for a tall-skinny x (tens of thousands of rows, <= 384 columns).

    Forward and dX are ordinary GEMMs.  dW = x^T @ dy reduces over the rows into a tiny
    (K x M) output: hipBLASLt runs that as a handful of workgroups (one per output tile, no
    split-K: ~6 TFLOP/s measured on MI355X), so the reduction is split here into S independent
    batched GEMMs over row blocks plus one sum -- S workgroup-sets instead of one.
    """

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return x @ w

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = gy @ w.t()
        if ctx.needs_input_grad[1]:
            R = x.shape[0]
            S = 1
            while S < 256 and R % (2 * S) == 0 and R // (2 * S) >= 256:
                S *= 2
            if S == 1:
                gw = x.t() @ gy
            else:
                gw = torch.bmm(x.view(S, R // S, x.shape[1]).transpose(1, 2), gy.reshape(S, R // S, gy.shape[1])).sum(0)
        return gx, gw


def _mm(x, w):
    """x (..., K) @ w (K, M); routes the weight gradient through the split-K path when training
    on a large batch."""
    if torch.is_grad_enabled() and w.requires_grad and x.numel() // x.shape[-1] >= 4096:
        lead = x.shape[:-1]
        return _SplitKMatmul.apply(x.reshape(-1, x.shape[-1]), w).view(*lead, w.shape[1])
    return x @ w


def _linear(x, lin):
    """nn.Linear forward through _mm (same math: x @ W^T + b)."""
    return _mm(x, lin.weight.t()) + lin.bias


def attend_mix_reference(g, keys, n, skip_self):
    """Plain-torch statement of csrc/fa_attend.hip: g (B*n, W) rows env-major, keys (B, nk, W) ->
    softmax_j(g_i . k_j) mixed keys, (B*n, W); j == i excluded with skip_self (a team of one gets zeros)."""
    B, nk, W = keys.shape
    s = torch.matmul(g.view(B, n, W), keys.transpose(1, 2))                    # (B, n, nk)
    if skip_self:
        if nk == 1:
            return g * 0.0
        s = s + torch.zeros(n, nk, device=g.device, dtype=g.dtype).fill_diagonal_(-math.inf)
    return torch.matmul(torch.softmax(s, dim=-1), keys).reshape(B * n, W)


class _AttendMix(torch.autograd.Function):
    """fa_attend_forward / fa_attend_backward (one launch each) on the current stream."""

    @staticmethod
    def forward(ctx, g, keys, n, skip_self):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        g, keys = g.contiguous(), keys.contiguous()
        B, nk, W = keys.shape
        out = torch.empty_like(g)
        attn = torch.empty((B * n, nk), device=g.device, dtype=g.dtype)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(lib.fa_attend_forward(g.data_ptr(), keys.data_ptr(), out.data_ptr(), attn.data_ptr(), B, n, nk, W,
                                         int(skip_self), st), "fa_attend_forward")
        ctx.save_for_backward(g, keys, attn)
        ctx.n, ctx.skip_self = n, int(skip_self)
        return out

    @staticmethod
    def backward(ctx, dout):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        g, keys, attn = ctx.saved_tensors
        B, nk, W = keys.shape
        dout = dout.contiguous()
        dg, dkeys = torch.empty_like(g), torch.empty_like(keys)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(lib.fa_attend_backward(g.data_ptr(), keys.data_ptr(), attn.data_ptr(), dout.data_ptr(), dg.data_ptr(),
                                          dkeys.data_ptr(), B, ctx.n, nk, W, ctx.skip_self, st), "fa_attend_backward")
        return dg, dkeys, None, None


def attend_mix(g, keys, n, skip_self):
    """The agents-of-one-env attention: the fused HIP op on the GPU (widths 64 / 128, teams <= 8, float32),
    the plain-torch statement elsewhere."""
    if g.is_cuda and g.dtype == torch.float32 and keys.shape[2] in (64, 128) and n <= 8 and keys.shape[1] <= 8:
        return _AttendMix.apply(g, keys, n, skip_self)
    return attend_mix_reference(g, keys, n, skip_self)


def _weights_init(m):  # mpnn.py:9-14
    name = m.__class__.__name__
    if name.find("Conv") != -1 or name.find("Linear") != -1:
        nn.init.orthogonal_(m.weight.data)
        if m.bias is not None:
            m.bias.data.fill_(0)


class _AttnParams(nn.Module):
    """Parameter container of MultiHeadAttention / MultiHeadOppAttention (mpnn.py:208-249,
    :335-370): W_query, W_key, W_val (heads, in, key) and W_out (heads, key, embed),
    uniform(+-1/sqrt(last dim)) in that order."""

    def __init__(self, n_heads, input_dim, embed_dim):
        super().__init__()
        key_dim = embed_dim // n_heads
        self.n_heads, self.input_dim, self.embed_dim, self.key_dim = n_heads, input_dim, embed_dim, key_dim
        self.norm_factor = 1 / math.sqrt(key_dim)
        self.W_query = nn.Parameter(torch.Tensor(n_heads, input_dim, key_dim))
        self.W_key = nn.Parameter(torch.Tensor(n_heads, input_dim, key_dim))
        self.W_val = nn.Parameter(torch.Tensor(n_heads, input_dim, key_dim))
        self.W_out = nn.Parameter(torch.Tensor(n_heads, key_dim, embed_dim))
        for p in self.parameters():
            stdv = 1.0 / math.sqrt(p.size(-1))
            p.data.uniform_(-stdv, stdv)


class _Categorical(nn.Module):
    """rlcore/distributions.py:19-31 (the linear head; orthogonal gain 0.01 first, then
    overwritten by MPNN.apply(weights_init), quirk Q13)."""

    def __init__(self, num_inputs, num_outputs):
        super().__init__()
        self.linear = nn.Linear(num_inputs, num_outputs)
        nn.init.orthogonal_(self.linear.weight.data, gain=0.01)
        nn.init.constant_(self.linear.bias.data, 0)

    def forward(self, x):
        return self.linear(x)


class MPNN(nn.Module):
    def __init__(self, action_space=None, num_agents=3, num_opp_agents=3, num_entities=0, input_size=6,
                 hidden_dim=128, embed_dim=None, pos_index=2, norm_in=False, nonlin=nn.ReLU, n_heads=1,
                 mask_dist=None, entity_mp=False, policy_layers=1, num_actions=None):
        super().__init__()
        if n_heads != 1 or entity_mp or norm_in or policy_layers != 1:
            raise NotImplementedError("FortAttack uses n_heads=1, entity_mp=False, norm_in=False, "
                                      "policy_layers=1 (learner.py:62-68)")
        self.h_dim = hidden_dim
        self.num_agents, self.num_opp_agents = num_agents, num_opp_agents
        self.K = 3  # message passing rounds (mpnn.py:27)
        self.embed_dim = hidden_dim if embed_dim is None else embed_dim
        self.input_size = input_size
        half = int(hidden_dim / 2)
        # construction order == reference (mpnn.py:37-74): RNG draws line up
        self.encoder = nn.Sequential(nn.Linear(input_size, half), nonlin(inplace=True))
        self.oppEncoder = nn.Sequential(nn.Linear(input_size, half), nonlin(inplace=True))
        self.oppAttn = _AttnParams(n_heads, half, int(self.embed_dim / 2))
        self.oppUpdate = nn.Sequential(nn.Linear(half + int(self.embed_dim / 2), half), nonlin(inplace=True))
        self.messages = _AttnParams(n_heads, hidden_dim, self.embed_dim)
        self.update = nn.Sequential(nn.Linear(hidden_dim + self.embed_dim, hidden_dim), nonlin(inplace=True))
        self.value_head = nn.Sequential(nn.Linear(hidden_dim, hidden_dim), nonlin(inplace=True),
                                        nn.Linear(hidden_dim, 1))
        self.policy_head = nn.Sequential(nn.Linear(hidden_dim, hidden_dim), nonlin(inplace=True))
        if num_actions is None:
            num_actions = action_space.shape[0]  # mpnn.py:73
        self.dist = _Categorical(hidden_dim, num_actions)
        self.is_recurrent = False
        self.fold_update = True    # evaluate_actions under autograd on the GPU uses trunk_folded()
        self.apply(_weights_init)  # mpnn.py:84
        diag = torch.zeros(num_agents, num_agents)
        diag.fill_diagonal_(-math.inf)
        self.register_buffer("_diag", diag, persistent=False)

    def _fused_weights(self):
        """[W_query | W_val] of the opponent attention and [W_query | W_key | W_val] of the team
        attention as single GEMM operands.  Under autograd they are rebuilt every call (so
        gradients flow to the separate parameters); in no-grad rollouts they are cached until a
        parameter changes (optimizer step / load_state_dict bump the version counters)."""
        a, m = self.oppAttn, self.messages
        ps = (a.W_query, a.W_val, m.W_query, m.W_key, m.W_val)
        if torch.is_grad_enabled():
            return (torch.cat((a.W_query[0], a.W_val[0]), dim=1),
                    torch.cat((m.W_query[0], m.W_key[0], m.W_val[0]), dim=1))
        key = tuple((p._version, p.data_ptr()) for p in ps)
        if getattr(self, "_wcache_key", None) != key:
            self.refresh_fused_weights()
        return self._w_qv, self._w_qkv

    @torch.no_grad()
    def refresh_fused_weights(self):
        """Rewrite the cached operands IN PLACE (same storage): hipGraphs captured over a rollout
        step keep pointing at them, so the learner calls this after every optimizer phase."""
        a, m = self.oppAttn, self.messages
        if getattr(self, "_w_qv", None) is None or self._w_qv.device != a.W_query.device:
            self._w_qv = torch.empty(a.input_dim, 2 * a.key_dim, device=a.W_query.device, dtype=a.W_query.dtype)
            self._w_qkv = torch.empty(m.input_dim, 3 * m.key_dim, device=m.W_query.device, dtype=m.W_query.dtype)
        torch.cat((a.W_query[0], a.W_val[0]), dim=1, out=self._w_qv)
        torch.cat((m.W_query[0], m.W_key[0], m.W_val[0]), dim=1, out=self._w_qkv)
        self._wcache_key = tuple((p._version, p.data_ptr()) for p in
                                 (a.W_query, a.W_val, m.W_query, m.W_key, m.W_val))

    # ---- trunk (mpnn.py:117-172 _fwd) ---------------------------------------------------
    def trunk(self, own, opp, return_attn=False):
        """own (B, n, 6), opp (B, m, 6) -> h (B, n, h_dim)."""
        h = torch.relu(_linear(own, self.encoder[0]))
        h_opp = torch.relu(_linear(opp, self.oppEncoder[0]))
        # Projections that share an input are one GEMM (weights concatenated on the fly: the
        # parameters stay separate so reference state_dicts load), and the n x m attention is a
        # broadcast multiply-sum: per-env (3x64)@(64x3) batched GEMMs over thousands of envs are
        # the slowest way to spend MFMA time.
        a = self.oppAttn                                   # mpnn.py:372-443
        kd = a.key_dim
        w_qv, w_qkv = self._fused_weights()
        qv = _mm(h_opp, w_qv)
        q, v = qv[..., :kd], qv[..., kd:]
        k = _mm(h, a.W_key[0])
        scores = (k.unsqueeze(2) * q.unsqueeze(1)).sum(-1)             # (B, n, m)
        opp_attn = F.softmax(a.norm_factor * scores, dim=-1)
        e_opp = _mm((opp_attn.unsqueeze(-1) * v.unsqueeze(1)).sum(2), a.W_out[0])
        h = torch.cat((h, e_opp), dim=2)
        m = self.messages                                  # mpnn.py:250-332
        attn = None
        for _ in range(self.K):
            if h.shape[1] == 1:                            # mpnn.py:266-274
                msg = torch.zeros(h.shape[0], 1, m.embed_dim, device=h.device, dtype=h.dtype)
                attn = torch.zeros(h.shape[0], 1, 1, device=h.device, dtype=h.dtype)
            else:
                qkv = _mm(h, w_qkv)
                qq, kk, vv = qkv[..., :m.key_dim], qkv[..., m.key_dim:2 * m.key_dim], qkv[..., 2 * m.key_dim:]
                comp = m.norm_factor * (qq.unsqueeze(2) * kk.unsqueeze(1)).sum(-1) + self._diag
                attn = F.softmax(comp, dim=-1)
                msg = _mm((attn.unsqueeze(-1) * vv.unsqueeze(1)).sum(2), m.W_out[0])
            h = torch.relu(_linear(torch.cat((h, msg), 2), self.update[0]))
        if return_attn:
            return h, attn, opp_attn
        return h

    # ---- the same trunk with three pairs of consecutive linear maps multiplied out ------------------------
    def folded_weights(self):
        """A_o = norm W_key W_query^T, B_o = W_val W_out (opponent attention); A_m = norm W_query W_key^T and
        the two halves of the update layer, the second with W_val W_out folded in (see mpnn_pack.py).  Built
        from the parameters inside the autograd graph: gradients reach the original tensors."""
        a, m, hd = self.oppAttn, self.messages, self.h_dim
        uw = self.update[0].weight
        return (a.norm_factor * (a.W_key[0] @ a.W_query[0].t()), a.W_val[0] @ a.W_out[0],
                m.norm_factor * (m.W_query[0] @ m.W_key[0].t()), uw[:, :hd].t(),
                (m.W_val[0] @ m.W_out[0]) @ uw[:, hd:].t())

    def trunk_folded(self, own, opp):
        """own (B, n, 6), opp (B, m, 6) -> h (B, n, h_dim): the training forward of the PPO update.  Per round
        one projection GEMM, the agents-of-one-env attention as ONE op (attend_mix: csrc/fa_attend.hip on
        the GPU) and the update layer as two accumulated GEMMs -- a third of the launches and none of the
        (B, n, n, h) broadcast temporaries of trunk().  Same function of the same parameters; float32
        rounding differs at the 1e-6 level."""
        B, n, m = own.shape[0], own.shape[1], opp.shape[1]
        A_o, B_o, A_m, W7a, W7b = self.folded_weights()
        h1 = torch.relu(_linear(own.reshape(B * n, -1), self.encoder[0]))
        ho = torch.relu(_linear(opp.reshape(B * m, -1), self.oppEncoder[0]))
        mix_o = attend_mix(_mm(h1, A_o), ho.view(B, m, -1), n, False)
        h = torch.cat((h1, _mm(mix_o, B_o)), dim=1)
        bu = self.update[0].bias
        for _ in range(self.K):
            mix = attend_mix(_mm(h, A_m), h.view(B, n, -1), n, True)
            h = torch.relu(_mm(h, W7a) + _mm(mix, W7b) + bu)
        return h.view(B, n, -1)

    def _value(self, h):
        return _linear(torch.relu(_linear(h, self.value_head[0])), self.value_head[2])

    def logits_value(self, own, opp):
        # under autograd on the GPU (the PPO update): the folded trunk; rollouts / CPU: the reference-shaped one
        fold = self.fold_update and own.is_cuda and torch.is_grad_enabled()
        h = self.trunk_folded(own, opp) if fold else self.trunk(own, opp)
        return _linear(torch.relu(_linear(h, self.policy_head[0])), self.dist.linear), self._value(h)

    # ---- env-major API used by the batched rollout ---------------------------------------
    def act(self, own, opp, deterministic=False, generator=None):
        """-> value (B,n,1), action (B,n,1) int64, action_log_prob (B,n,1)  (mpnn.py:183-192)."""
        logits, value = self.logits_value(own, opp)
        logp_all = F.log_softmax(logits, dim=-1)
        if deterministic:
            action = logits.argmax(dim=-1, keepdim=True)
        else:
            # Gumbel-max sampling: argmax(logits + G), G = -log(-log U), is a draw from
            # softmax(logits) -- the same categorical distribution as torch.multinomial
            # (FixedCategorical.sample, distributions.py:12-13) without its validity assert,
            # reductions and scan kernels
            u = torch.rand(logits.shape, device=logits.device, dtype=logits.dtype, generator=generator)
            gumbel = -torch.log(-torch.log(u.clamp_(min=1e-20, max=1.0 - 1e-7)))
            action = (logits + gumbel).argmax(dim=-1, keepdim=True)
        return value, action, logp_all.gather(-1, action)

    def get_value(self, own, opp):                         # mpnn.py:202-205
        return self._value(self.trunk(own, opp))

    def evaluate_actions(self, own, opp, action):
        """-> value, log-prob of `action`, per-sample entropy (all (B,n,1)/(B,n))  (mpnn.py:194-200)."""
        logits, value = self.logits_value(own, opp)
        logp_all = F.log_softmax(logits, dim=-1)
        entropy = -(logp_all.exp() * logp_all).sum(-1)
        return value, logp_all.gather(-1, action), entropy

    # ---- the reference's agent-major calling convention (learner.py:150, mpnn.py:132) ------
    def _env_major(self, flat, n):
        return flat.view(n, -1, flat.shape[-1]).transpose(0, 1)

    def evaluate_actions_agent_major(self, inp, opp_inp, action):
        """inp (n*B, 6), opp_inp (m*B, 6), action (n*B, 1), rows ordered agent-major."""
        own = self._env_major(inp, self.num_agents)
        opp = self._env_major(opp_inp, self.num_opp_agents)
        act = self._env_major(action, self.num_agents)
        value, logp, ent = self.evaluate_actions(own, opp, act)
        flat = lambda t: t.transpose(0, 1).reshape(-1, t.shape[-1])
        return flat(value), flat(logp), ent.transpose(0, 1).reshape(-1)


class TwinMPNN(object):
    """Both teams' policies evaluated in ONE batched pass (no-grad rollouts only).

    With equal team sizes the guard and the attacker MPNN have identical shapes, so every
    GEMM becomes a 2-batch ``baddbmm`` over stacked weights and every elementwise op carries a
    leading team dimension: half the kernel launches of two separate forwards, which is what a
    small-network rollout on a big GPU is bound by.  The stacked weights live in persistent
    buffers that ``refresh()`` rewrites IN PLACE (hipGraph-safe); the learner calls it whenever
    the policies' parameters may have moved.  Numerically this is the same network (same
    weights, same operation order per team); results equal the separate forwards to float32
    rounding of the batched GEMM.
    """

    def __init__(self, guard, attacker):
        if guard.num_agents != attacker.num_agents or guard.num_opp_agents != attacker.num_opp_agents \
                or guard.h_dim != attacker.h_dim:
            raise ValueError("TwinMPNN needs equal team sizes and hidden sizes")
        self.pols = (guard, attacker)
        self.n = guard.num_agents
        self.K = guard.K
        self._w = None
        self.refresh()

    @torch.no_grad()
    def refresh(self):
        g, a = self.pols
        lin = lambda f: (torch.stack([f(g).weight.t(), f(a).weight.t()]), torch.stack([f(g).bias, f(a).bias])[:, None, :])
        w = {}
        w["enc"] = lin(lambda p: p.encoder[0])
        w["oenc"] = lin(lambda p: p.oppEncoder[0])
        w["upd"] = lin(lambda p: p.update[0])
        w["v0"] = lin(lambda p: p.value_head[0])
        w["v2"] = lin(lambda p: p.value_head[2])
        w["p0"] = lin(lambda p: p.policy_head[0])
        w["dist"] = lin(lambda p: p.dist.linear)
        st = lambda f: torch.stack([f(g), f(a)])
        w["o_qv"] = (st(lambda p: torch.cat((p.oppAttn.W_query[0], p.oppAttn.W_val[0]), 1)),)
        w["o_k"] = (st(lambda p: p.oppAttn.W_key[0]),)
        w["o_out"] = (st(lambda p: p.oppAttn.W_out[0]),)
        w["m_qkv"] = (st(lambda p: torch.cat((p.messages.W_query[0], p.messages.W_key[0], p.messages.W_val[0]), 1)),)
        w["m_out"] = (st(lambda p: p.messages.W_out[0]),)
        if self._w is None:
            self._w = {k: tuple(t.contiguous().clone() for t in v) for k, v in w.items()}
        else:
            for k, v in w.items():
                for dst, src in zip(self._w[k], v):
                    dst.copy_(src)

    @staticmethod
    def _lin(x, wb):                      # x (2, R, in) -> (2, R, out)
        return torch.baddbmm(wb[1], x, wb[0])

    @torch.no_grad()
    def logits_value(self, obs):
        """obs (E, 2n, 6), guards first -> logits (2, E, n, A), value (2, E, n, 1)."""
        E, n, w = obs.shape[0], self.n, self._w
        own = torch.stack((obs[:, :n], obs[:, n:]))                 # (2, E, n, 6)
        opp = torch.stack((obs[:, n:], obs[:, :n]))
        flat = lambda t: t.reshape(2, E * n, t.shape[-1])
        unflat = lambda t: t.view(2, E, n, t.shape[-1])
        h = torch.relu_(self._lin(flat(own), w["enc"]))
        ho = torch.relu_(self._lin(flat(opp), w["oenc"]))
        a = self.pols[0].oppAttn
        qv = unflat(torch.bmm(ho, w["o_qv"][0]))
        q, v = qv[..., :a.key_dim], qv[..., a.key_dim:]
        k = unflat(torch.bmm(h, w["o_k"][0]))
        att = torch.softmax(a.norm_factor * (k.unsqueeze(3) * q.unsqueeze(2)).sum(-1), dim=-1)   # (2,E,n,m)
        e_opp = torch.bmm(flat((att.unsqueeze(-1) * v.unsqueeze(2)).sum(3)), w["o_out"][0])
        h = torch.cat((h, e_opp), dim=2)                           # (2, E*n, h_dim)
        m = self.pols[0].messages
        diag = self.pols[0]._diag
        for _ in range(self.K):
            if n == 1:
                msg = torch.zeros(2, E, m.embed_dim, device=h.device, dtype=h.dtype)
            else:
                qkv = unflat(torch.bmm(h, w["m_qkv"][0]))
                qq, kk, vv = qkv[..., :m.key_dim], qkv[..., m.key_dim:2 * m.key_dim], qkv[..., 2 * m.key_dim:]
                comp = m.norm_factor * (qq.unsqueeze(3) * kk.unsqueeze(2)).sum(-1) + diag
                att = torch.softmax(comp, dim=-1)
                msg = torch.bmm(flat((att.unsqueeze(-1) * vv.unsqueeze(2)).sum(3)), w["m_out"][0])
            h = torch.relu_(self._lin(torch.cat((h, msg), dim=2), w["upd"]))
        value = self._lin(torch.relu_(self._lin(h, w["v0"])), w["v2"])
        logits = self._lin(torch.relu_(self._lin(h, w["p0"])), w["dist"])
        return unflat(logits), unflat(value)

    @torch.no_grad()
    def act(self, obs, deterministic=False, generator=None):
        """-> value (E, 2n, 1), action (E, 2n, 1) int64, log-prob (E, 2n, 1), guards first."""
        logits, value = self.logits_value(obs)
        logp_all = F.log_softmax(logits, dim=-1)
        if deterministic:
            action = logits.argmax(dim=-1, keepdim=True)
        else:
            u = torch.rand(logits.shape, device=logits.device, dtype=logits.dtype, generator=generator)
            action = (logits - torch.log(-torch.log(u.clamp_(min=1e-20, max=1.0 - 1e-7)))).argmax(dim=-1, keepdim=True)
        logp = logp_all.gather(-1, action)
        team_major = lambda t: t.permute(1, 0, 2, 3).reshape(obs.shape[0], 2 * self.n, t.shape[-1])
        return team_major(value), team_major(action), team_major(logp)

    @torch.no_grad()
    def get_value(self, obs):
        _, value = self.logits_value(obs)
        return value.permute(1, 0, 2, 3).reshape(obs.shape[0], 2 * self.n, 1)
