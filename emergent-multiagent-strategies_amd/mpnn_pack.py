"""Host side of the fused policy kernel (csrc/fa_policy.hip): pack an MPNN's parameters (mpnn.py /
a reference state_dict) into the kernel's weight buffer.

Three pairs of consecutive linear maps of the reference network are multiplied out here, in float64,
and rounded to float32 once (exact in real arithmetic; fp32 rounding differs from evaluating the
factors one after the other by ~1e-6 relative -- the size of a GEMM's own summation-order noise):

    A_o = norm_o * oppAttn.W_key  @ oppAttn.W_query^T        scores_ij = (h1_i A_o) . ho_j
    B_o =          oppAttn.W_val  @ oppAttn.W_out            e_opp_i   = (sum_j a_ij ho_j) B_o
    A_m = norm   * messages.W_query @ messages.W_key^T       comp_ij   = (h_i A_m) . h_j
    W7  = [ update.weight[:, :128]^T ; messages.W_val @ messages.W_out @ update.weight[:, 128:]^T ]
                                                             h'        = relu([h | sum_j a_ij h_j] W7 + b)

Dense operands are stored in the lane order of the 32x32x2 fp32 MFMA's B operand (`pack_gemm`); the
section offsets are csrc/fa_policy.h's FA_POFF_*.  `folded_forward` evaluates the packed buffer with
plain torch ops (CPU or GPU): it is what the CPU tests compare with MPNN.logits_value, and documents
the kernel's arithmetic.
"""
import torch

HIDDEN = 128
# csrc/fa_policy.h
POFF = dict(WE=0, BE=384, WOE=448, BOE=832, AO=896, BO=4992, AM=9088, W7=25472, BU=58240, W8=58368, B8=91136,
            W9=91392, B9=99584)
WEIGHT_FLOATS = 99616
MAX_TEAM = 8


def pack_gemm(w):
    """(K, C) matrix -> flat float32 in B-operand order: float4 index (cb * K/8 + t4) * 64 + lane holds
    W[k = (lane >> 5) * K/2 + 4*t4 + q][col = 32*cb + (lane & 31)], q = 0..3."""
    K, C = w.shape
    assert K % 8 == 0 and C % 32 == 0
    return w.reshape(2, K // 8, 4, C // 32, 32).permute(3, 1, 0, 4, 2).contiguous().reshape(-1)


def unpack_gemm(flat, K, C):
    """Inverse of pack_gemm."""
    return flat.reshape(C // 32, K // 8, 2, 32, 4).permute(2, 1, 4, 0, 3).contiguous().reshape(K, C)


def supported(pol):
    return pol.h_dim == HIDDEN and pol.embed_dim == HIDDEN and pol.input_size == 6 and \
        pol.num_agents <= MAX_TEAM and pol.num_opp_agents <= MAX_TEAM and pol.dist.linear.out_features == 8


@torch.no_grad()
def pack_policy(pol, out=None):
    """MPNN module (mpnn.py) -> packed float32 buffer on the module's device; `out` is rewritten in
    place when given (a captured hipGraph keeps pointing at it)."""
    if not supported(pol):
        raise ValueError("the fused policy kernel needs hidden_dim = 128, 6 inputs, 8 actions and teams of <= 8")
    d = lambda t: t.detach().double()
    a, m = pol.oppAttn, pol.messages
    upd_w = d(pol.update[0].weight)                                   # (128, 256)
    sec = {
        "WE": d(pol.encoder[0].weight).t(), "BE": d(pol.encoder[0].bias),
        "WOE": d(pol.oppEncoder[0].weight).t(), "BOE": d(pol.oppEncoder[0].bias),
        "AO": pack_gemm(a.norm_factor * d(a.W_key[0]) @ d(a.W_query[0]).t()),
        "BO": pack_gemm(d(a.W_val[0]) @ d(a.W_out[0])),
        "AM": pack_gemm(m.norm_factor * d(m.W_query[0]) @ d(m.W_key[0]).t()),
        "W7": pack_gemm(torch.cat((upd_w[:, :HIDDEN].t(), d(m.W_val[0]) @ d(m.W_out[0]) @ upd_w[:, HIDDEN:].t()), 0)),
        "BU": d(pol.update[0].bias),
        "W8": pack_gemm(torch.cat((d(pol.policy_head[0].weight).t(), d(pol.value_head[0].weight).t()), 1)),
        "B8": torch.cat((d(pol.policy_head[0].bias), d(pol.value_head[0].bias))),
    }
    w9 = torch.zeros(2 * HIDDEN, 32, dtype=torch.float64, device=upd_w.device)
    w9[:HIDDEN, :8] = d(pol.dist.linear.weight).t()
    w9[HIDDEN:, 8] = d(pol.value_head[2].weight)[0]
    b9 = torch.zeros(32, dtype=torch.float64, device=upd_w.device)
    b9[:8] = d(pol.dist.linear.bias)
    b9[8] = d(pol.value_head[2].bias)[0]
    sec["W9"], sec["B9"] = pack_gemm(w9), b9
    if out is None:
        out = torch.zeros(WEIGHT_FLOATS, dtype=torch.float32, device=upd_w.device)
    assert out.numel() == WEIGHT_FLOATS and out.dtype == torch.float32 and out.is_contiguous()
    for k, v in sec.items():
        out[POFF[k]:POFF[k] + v.numel()].copy_(v.reshape(-1).float())
    return out


@torch.no_grad()
def folded_forward(flat, own, opp):
    """The kernel's arithmetic on plain tensors: own (B, n, 6), opp (B, m, 6) -> logits (B, n, 8), value (B, n, 1)."""
    sl = lambda k, n: flat[POFF[k]:POFF[k] + n]
    n = own.shape[1]
    h1 = torch.relu(own @ sl("WE", 384).view(6, 64) + sl("BE", 64))
    ho = torch.relu(opp @ sl("WOE", 384).view(6, 64) + sl("BOE", 64))
    g = h1 @ unpack_gemm(sl("AO", 4096), 64, 64)
    att = torch.softmax(g @ ho.transpose(1, 2), dim=-1)              # (B, n, m)
    h = torch.cat((h1, (att @ ho) @ unpack_gemm(sl("BO", 4096), 64, 64)), dim=2)
    am, w7 = unpack_gemm(sl("AM", 16384), 128, 128), unpack_gemm(sl("W7", 32768), 256, 128)
    diag = torch.zeros(n, n, device=own.device).fill_diagonal_(-float("inf"))
    for _ in range(3):
        if n == 1:
            mix = torch.zeros_like(h)
        else:
            mix = torch.softmax((h @ am) @ h.transpose(1, 2) + diag, dim=-1) @ h
        h = torch.relu(torch.cat((h, mix), dim=2) @ w7 + sl("BU", 128))
    pv = torch.relu(h @ unpack_gemm(sl("W8", 32768), 128, 256) + sl("B8", 256))
    out = pv @ unpack_gemm(sl("W9", 8192), 256, 32) + sl("B9", 32)
    return out[..., :8], out[..., 8:9]


# ---- the PPO update's fused kernel (csrc/fa_train.hip) ---------------------------------------------------------
TOFF = dict(AOT=0, BOT=4096, AMT=8192, W7T=24576, W8T=57344, W9T=90112)     # csrc/fa_train.h
TRANS_FLOATS = 98304
SLAB_FLOATS = WEIGHT_FLOATS + 16
PLAIN_SHAPES = dict(WE=(6, 64), BE=(64,), WOE=(6, 64), BOE=(64,), AO=(64, 64), BO=(64, 64), AM=(128, 128), W7=(256, 128),
                    BU=(128,), W8=(128, 256), B8=(256,), W9=(256, 32), B9=(32,))


def kernel_params(pol):
    """The matrices the fused kernels work with, as float32 tensors INSIDE the autograd graph of the module's
    parameters (the folded products of this file's header, the stacked heads, zero-padded W9 / b9): calling
    torch.autograd.backward on them with the kernel's gradients applies the chain rule to the parameters."""
    a, m, hd = pol.oppAttn, pol.messages, pol.h_dim
    uw = pol.update[0].weight
    z = lambda *s: torch.zeros(*s, device=uw.device, dtype=uw.dtype)
    w9 = torch.cat((torch.cat((pol.dist.linear.weight.t(), z(hd, 24)), 1),
                    torch.cat((z(hd, 8), pol.value_head[2].weight.t(), z(hd, 23)), 1)), 0)
    return dict(
        WE=pol.encoder[0].weight.t(), BE=pol.encoder[0].bias, WOE=pol.oppEncoder[0].weight.t(), BOE=pol.oppEncoder[0].bias,
        AO=a.norm_factor * (a.W_key[0] @ a.W_query[0].t()), BO=a.W_val[0] @ a.W_out[0],
        AM=m.norm_factor * (m.W_query[0] @ m.W_key[0].t()),
        W7=torch.cat((uw[:, :hd].t(), (m.W_val[0] @ m.W_out[0]) @ uw[:, hd:].t()), 0), BU=pol.update[0].bias,
        W8=torch.cat((pol.policy_head[0].weight.t(), pol.value_head[0].weight.t()), 1),
        B8=torch.cat((pol.policy_head[0].bias, pol.value_head[0].bias)),
        W9=w9, B9=torch.cat((pol.dist.linear.bias, pol.value_head[2].bias, z(23))))


_GEMMS = ("AO", "BO", "AM", "W7", "W8", "W9")


@torch.no_grad()
def pack_from_params(P, out_fwd, out_t):
    """kernel_params() -> the forward pack (FA_POFF_*) and the transposed pack (FA_TOFF_*), in place."""
    for k, v in P.items():
        flat = pack_gemm(v.detach()) if k in _GEMMS else v.detach().reshape(-1)
        out_fwd[POFF[k]:POFF[k] + flat.numel()].copy_(flat)
    for k in _GEMMS:
        flat = pack_gemm(P[k].detach().t())
        out_t[TOFF[k + "T"]:TOFF[k + "T"] + flat.numel()].copy_(flat)


def split_plain(flat):
    """A FA_SLAB buffer (gradients in plain layout at the POFF offsets) -> dict of views shaped like kernel_params()."""
    out = {}
    for k, shp in PLAIN_SHAPES.items():
        nfl = 1
        for d in shp:
            nfl *= d
        out[k] = flat[POFF[k]:POFF[k] + nfl].view(*shp)
    return out


def folded_ppo_reference(P, obs, own_sl, opp_sl, action, value_pred, ret, old_logp, adv, clip, c_value, c_entropy,
                         clipped_value_loss=True):
    """Plain-torch statement of csrc/fa_train.hip (forward on the kernel-facing matrices P + the losses of
    ppo.py:146-187 WITHOUT the division by the mask mean): returns (loss, value_loss, action_loss, entropy, mask_mean)
    where loss = c_value * value_loss + action_loss - c_entropy * entropy; differentiable in P."""
    own, opp = obs[:, own_sl], obs[:, opp_sl]
    n = own.shape[1]
    h1 = torch.relu(own @ P["WE"] + P["BE"])
    ho = torch.relu(opp @ P["WOE"] + P["BOE"])
    att = torch.softmax((h1 @ P["AO"]) @ ho.transpose(1, 2), dim=-1)
    h = torch.cat((h1, (att @ ho) @ P["BO"]), dim=2)
    diag = torch.zeros(n, n, device=obs.device).fill_diagonal_(-float("inf"))
    for _ in range(3):
        mix = torch.softmax((h @ P["AM"]) @ h.transpose(1, 2) + diag, dim=-1) @ h if n > 1 else torch.zeros_like(h)
        h = torch.relu(torch.cat((h, mix), dim=2) @ P["W7"] + P["BU"])
    out = torch.relu(h @ P["W8"] + P["B8"]) @ P["W9"] + P["B9"]
    logp_all = torch.log_softmax(out[..., :8], dim=-1)
    value = out[..., 8:9]
    mask = own[:, :, 0:1]
    logp = logp_all.gather(-1, action)
    ent = -(logp_all.exp() * logp_all).sum(-1, keepdim=True)
    ratio = mask * torch.exp(logp - old_logp)
    al = (mask * -torch.min(ratio * adv, torch.clamp(ratio, 1 - clip, 1 + clip) * adv)).mean()
    if clipped_value_loss:
        vclip = value_pred + (value - value_pred).clamp(-clip, clip)
        vl = (0.5 * torch.max((value - ret).pow(2), (vclip - ret).pow(2)) * mask).mean()
    else:
        vl = 0.5 * (ret - value).pow(2).mean()
    en = (ent * mask).mean()
    return vl * c_value + al - en * c_entropy, vl, al, en, mask.mean()
