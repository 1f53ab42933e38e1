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
PLAIN_FLOATS = 99616          # end of the float32 sections: the layout of the plain / gradient buffers (FA_POLICY_PLAIN_FLOATS)
# ... followed by the six dense matrices split into three bf16 terms for the bf16 matrix cores (csrc/fa_policy.h FA_POFF3_*)
POFF3 = dict(AO=99616, BO=105760, AM=111904, W7=136480, W8=185632, W9=234784)
WEIGHT_FLOATS = 247072        # the packed buffer (FA_POLICY_WEIGHT_FLOATS, fa_policy_weight_floats())
MAX_TEAM = 8


def _rne_hi(x):
    """float32 tensor -> float32 tensor rounded to nearest-even at 8 significant bits (a bf16 in a float's clothes)."""
    u = x.contiguous().view(torch.int32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) & -65536).view(torch.float32)


def pack_gemm3(w):
    """(K, C) matrix -> the bf16x3 B-operand pack as a flat float32 tensor of 1.5 K C floats: every weight (rounded to float32
    first) is EXACTLY hi + mid + lo, three bf16 numbers; 16-byte index ((cb * K/16 + s) * 3 + term) * 64 + lane holds the eight
    bf16 W[k = (lane >> 5) * K/2 + 8 s + j][col = 32 cb + (lane & 31)], j = 0..7 (csrc/fa_policy.h; fa_pack_weights writes the same)."""
    K, C = w.shape
    assert K % 16 == 0 and C % 32 == 0
    x = w.float().contiguous()
    hi = _rne_hi(x)
    r1 = x - hi
    mid = _rne_hi(r1)
    lo = r1 - mid
    terms = torch.stack([(t.contiguous().view(torch.int32) >> 16).to(torch.int16) for t in (hi, mid, lo)])   # (3, K, C) bf16 bits
    t = terms.reshape(3, 2, K // 16, 8, C // 32, 32).permute(4, 2, 0, 1, 5, 3).contiguous()                  # cb, s, term, hh, li, j
    return t.reshape(-1).view(torch.float32)


def pack_gemm(w):
    """(K, C) matrix -> flat float32 in B-operand order: float4 index (cb * K/8 + t4) * 64 + lane holds
    W[k = (lane >> 5) * K/2 + 4*t4 + q][col = 32*cb + (lane & 31)], q = 0..3."""
    K, C = w.shape
    assert K % 8 == 0 and C % 32 == 0
    return w.reshape(2, K // 8, 4, C // 32, 32).permute(3, 1, 0, 4, 2).contiguous().reshape(-1)


def unpack_gemm(flat, K, C):
    """Inverse of pack_gemm."""
    return flat.reshape(C // 32, K // 8, 2, 32, 4).permute(2, 1, 4, 0, 3).contiguous().reshape(K, C)


def unpack_gemm3(flat3, K, C):
    """Inverse of pack_gemm3: the (K, C) float32 matrix hi + mid + lo (exact: the split loses nothing)."""
    bits = flat3.contiguous().view(torch.int16).reshape(C // 32, K // 16, 3, 2, 32, 8).permute(2, 3, 1, 5, 0, 4).reshape(3, K, C)
    t = (bits.to(torch.int32) << 16).view(torch.float32)
    return (t[0] + t[1]) + t[2]


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
    for k, (K, C) in (("AO", (64, 64)), ("BO", (64, 64)), ("AM", (128, 128)), ("W7", (256, 128)), ("W8", (128, 256)), ("W9", (256, 32))):
        x3 = pack_gemm3(unpack_gemm(out[POFF[k]:POFF[k] + K * C], K, C))
        out[POFF3[k]:POFF3[k] + x3.numel()].copy_(x3)
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
SLAB_FLOATS = PLAIN_FLOATS + 16
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
        if k in _GEMMS and out_fwd.numel() >= WEIGHT_FLOATS:
            x3 = pack_gemm3(v.detach())
            out_fwd[POFF3[k]:POFF3[k] + x3.numel()].copy_(x3)
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


# ---- the module's parameters as ONE flat buffer + the fold / unfold task lists (csrc/fa_fold.hip) -----------------
# (name, accessor, offset) in floats; what the forward uses (oppUpdate is created but never used, mpnn.py:44-45)
_PF = (("ENC_W", lambda p: p.encoder[0].weight, 0), ("ENC_B", lambda p: p.encoder[0].bias, 384),
       ("OENC_W", lambda p: p.oppEncoder[0].weight, 448), ("OENC_B", lambda p: p.oppEncoder[0].bias, 832),
       ("OQ", lambda p: p.oppAttn.W_query, 896), ("OK", lambda p: p.oppAttn.W_key, 4992),
       ("OV", lambda p: p.oppAttn.W_val, 9088), ("OO", lambda p: p.oppAttn.W_out, 13184),
       ("MQ", lambda p: p.messages.W_query, 17280), ("MK", lambda p: p.messages.W_key, 33664),
       ("MV", lambda p: p.messages.W_val, 50048), ("MO", lambda p: p.messages.W_out, 66432),
       ("UW", lambda p: p.update[0].weight, 82816), ("UB", lambda p: p.update[0].bias, 115584),
       ("V0W", lambda p: p.value_head[0].weight, 115712), ("V0B", lambda p: p.value_head[0].bias, 132096),
       ("V2W", lambda p: p.value_head[2].weight, 132224), ("V2B", lambda p: p.value_head[2].bias, 132352),
       ("P0W", lambda p: p.policy_head[0].weight, 132356), ("P0B", lambda p: p.policy_head[0].bias, 148740),
       ("DW", lambda p: p.dist.linear.weight, 148868), ("DB", lambda p: p.dist.linear.bias, 149892))
PF_FLOATS = 149900


class FlatPolicy(object):
    """An MPNN whose (used) parameters live in one flat device buffer `pflat` -- the module's tensors become views
    of it, their .grad views of `gflat` -- plus the task lists that turn `pflat` into the fused kernels' weight
    packs (`fold_pack`: 3 launches) and the kernel's plain-layout gradients into `gflat` (`unfold`: 2 launches):
    the folding algebra of this file's header and its chain rule, without a PyTorch op.  torch optimizers and
    clip_grad_norm_ work on the module's parameters as before."""

    @classmethod
    def of(cls, pol):
        """The FlatPolicy of `pol` (one per module: a second one would re-point the parameters away from the first)."""
        fp = pol.__dict__.get("_flat_policy")
        if fp is None or fp.pol is not pol or not fp.attached():
            fp = cls(pol)
            pol.__dict__["_flat_policy"] = fp
        return fp

    def attached(self):
        """Whether the module's parameters are still views of pflat (a .to() / .data assignment detaches them)."""
        base = self.pflat.data_ptr()
        return all(get(self.pol).data_ptr() == base + 4 * off for _, get, off in _PF)

    def __init__(self, pol):
        import ctypes as C
        from . import _lib
        if not supported(pol):
            raise ValueError("FlatPolicy needs hidden_dim = 128, 6 inputs, 8 actions and teams of <= 8")
        self.pol, self._lib, self._C = pol, _lib, C
        dev = pol.update[0].weight.device
        z = lambda n: torch.zeros(n, device=dev, dtype=torch.float32)
        self.pflat, self.gflat = z(PF_FLOATS), z(PF_FLOATS + 8)     # + 8: room for loss sums in one all-reduce
        self.off = {}
        with torch.no_grad():
            for name, get, off in _PF:
                p = get(pol)
                n = p.numel()
                self.pflat[off:off + n].copy_(p.data.reshape(-1))
                p.data = self.pflat[off:off + n].view(p.shape)
                self.off[name] = off
        self.plain, self.mscr, self.dmscr = z(PLAIN_FLOATS), z(128 * 128), z(128 * 128)
        self.w, self.wt = z(WEIGHT_FLOATS), z(TRANS_FLOATS)
        self._fold = [self._tasks(self._fold_stage1()), self._tasks(self._fold_stage2())]
        self._unfold = {}       # gradient buffer address -> its two task lists (captured graphs keep the pointers)
        self.attach_grads()

    def attach_grads(self):
        """(Re)point every parameter's .grad at its slice of gflat."""
        for name, get, off in _PF:
            p = get(self.pol)
            p.grad = self.gflat[off:off + p.numel()].view(p.shape)

    # -- task lists ---------------------------------------------------------------------------------------------
    def _tasks(self, items):
        C, L = self._C, self._lib
        arr = (L.Task * len(items))()
        for t, d in zip(arr, items):
            t.C, t.A, t.B = d["C"], d["A"], d.get("B", 0)
            t.ldc, t.M, t.N, t.K = d["ldc"], d["M"], d["N"], d.get("K", 1)
            t.a_rs, t.a_cs, t.b_rs, t.b_cs = d["a"][0], d["a"][1], d.get("b", (0, 0))[0], d.get("b", (0, 0))[1]
            t.alpha, t.type = d.get("alpha", 1.0), 0 if "B" in d else 1
        buf = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.pflat.device)
        return buf, len(items)

    def _run(self, tl):
        L, C = self._lib, self._C
        L.check(L.load().fa_run_tasks(tl[0].data_ptr(), tl[1], C.c_void_p(torch.cuda.current_stream().cuda_stream)), "fa_run_tasks")

    @staticmethod
    def _mm(Cp, ldc, M, N, K, Ap, a, Bp, b, alpha=1.0, split=1):
        """C (M x N) = alpha * A B as `split` row chunks (one workgroup each)."""
        out, rows = [], M // split
        for s in range(split):
            out.append(dict(C=Cp + 4 * s * rows * ldc, ldc=ldc, M=rows, N=N, K=K, A=Ap + 4 * s * rows * a[0], a=a, B=Bp, b=b, alpha=alpha))
        return out

    def _fold_stage1(self):
        th, P, o, po = self.pflat.data_ptr(), self.plain.data_ptr(), self.off, POFF
        a = lambda name, extra=0: th + 4 * (o[name] + extra)
        c = lambda name, extra=0: P + 4 * (po[name] + extra)
        pol = self.pol
        cp = lambda Cp, ldc, M, N, Ap, ars, acs: dict(C=Cp, ldc=ldc, M=M, N=N, A=Ap, a=(ars, acs))
        t = [cp(c("WE"), 64, 6, 64, a("ENC_W"), 1, 6), cp(c("BE"), 64, 1, 64, a("ENC_B"), 0, 1),
             cp(c("WOE"), 64, 6, 64, a("OENC_W"), 1, 6), cp(c("BOE"), 64, 1, 64, a("OENC_B"), 0, 1),
             cp(c("W7"), 128, 128, 128, a("UW"), 1, 256), cp(c("BU"), 128, 1, 128, a("UB"), 0, 1),
             cp(c("W8"), 256, 128, 128, a("P0W"), 1, 128), cp(c("W8", 128), 256, 128, 128, a("V0W"), 1, 128),
             cp(c("B8"), 256, 1, 128, a("P0B"), 0, 1), cp(c("B8", 128), 256, 1, 128, a("V0B"), 0, 1),
             cp(c("W9"), 32, 128, 8, a("DW"), 1, 128), cp(c("W9", 128 * 32 + 8), 32, 128, 1, a("V2W"), 1, 0),
             cp(c("B9"), 32, 1, 8, a("DB"), 0, 1), cp(c("B9", 8), 32, 1, 1, a("V2B"), 0, 1)]
        t += self._mm(c("AO"), 64, 64, 64, 64, a("OK"), (64, 1), a("OQ"), (1, 64), pol.oppAttn.norm_factor)
        t += self._mm(c("BO"), 64, 64, 64, 64, a("OV"), (64, 1), a("OO"), (64, 1))
        t += self._mm(c("AM"), 128, 128, 128, 128, a("MQ"), (128, 1), a("MK"), (1, 128), pol.messages.norm_factor, split=4)
        t += self._mm(self.mscr.data_ptr(), 128, 128, 128, 128, a("MV"), (128, 1), a("MO"), (128, 1), split=4)
        return t

    def _fold_stage2(self):   # W7[128:] = (W_val W_out) update.weight[:, 128:]^T
        th, P = self.pflat.data_ptr(), self.plain.data_ptr()
        return self._mm(P + 4 * (POFF["W7"] + 128 * 128), 128, 128, 128, 128, self.mscr.data_ptr(), (128, 1),
                        th + 4 * (self.off["UW"] + 128), (1, 256), split=4)

    def fold_pack(self):
        """parameters -> plain kernel-facing matrices -> forward + transposed packs (self.w, self.wt)."""
        L, C = self._lib, self._C
        self._run(self._fold[0])
        self._run(self._fold[1])
        L.check(L.load().fa_pack_weights(self.plain.data_ptr(), self.w.data_ptr(), self.wt.data_ptr(),
                                         C.c_void_p(torch.cuda.current_stream().cuda_stream)), "fa_pack_weights")
        return self.w, self.wt

    def _build_unfold(self, gp):
        """gp: data pointer of the plain-layout gradient buffer (the fa_ppo_grad slab sum)."""
        th, g, o, po = self.pflat.data_ptr(), self.gflat.data_ptr(), self.off, POFF
        a = lambda name, extra=0: th + 4 * (o[name] + extra)
        gr = lambda name, extra=0: g + 4 * (o[name] + extra)
        d = lambda name, extra=0: gp + 4 * (po[name] + extra)
        pol = self.pol
        no, nm = pol.oppAttn.norm_factor, pol.messages.norm_factor
        cp = lambda Cp, ldc, M, N, Ap, ars, acs: dict(C=Cp, ldc=ldc, M=M, N=N, A=Ap, a=(ars, acs))
        D = d("W7", 128 * 128)                      # dL/d(W7[128:]) (128 x 128)
        s1 = [cp(gr("ENC_W"), 6, 64, 6, d("WE"), 1, 64), cp(gr("ENC_B"), 64, 1, 64, d("BE"), 0, 1),
              cp(gr("OENC_W"), 6, 64, 6, d("WOE"), 1, 64), cp(gr("OENC_B"), 64, 1, 64, d("BOE"), 0, 1),
              cp(gr("UW"), 256, 128, 128, d("W7"), 1, 128), cp(gr("UB"), 128, 1, 128, d("BU"), 0, 1),
              cp(gr("P0W"), 128, 128, 128, d("W8"), 1, 256), cp(gr("V0W"), 128, 128, 128, d("W8", 128), 1, 256),
              cp(gr("P0B"), 128, 1, 128, d("B8"), 0, 1), cp(gr("V0B"), 128, 1, 128, d("B8", 128), 0, 1),
              cp(gr("DW"), 128, 8, 128, d("W9"), 1, 32), cp(gr("V2W"), 128, 1, 128, d("W9", 128 * 32 + 8), 0, 32),
              cp(gr("DB"), 8, 1, 8, d("B9"), 0, 1), cp(gr("V2B"), 1, 1, 1, d("B9", 8), 0, 1)]
        s1 += self._mm(gr("OK"), 64, 64, 64, 64, d("AO"), (64, 1), a("OQ"), (64, 1), no)        # dW_key = norm dA_o W_query
        s1 += self._mm(gr("OQ"), 64, 64, 64, 64, d("AO"), (1, 64), a("OK"), (64, 1), no)        # dW_query = norm dA_o^T W_key
        s1 += self._mm(gr("OV"), 64, 64, 64, 64, d("BO"), (64, 1), a("OO"), (1, 64))            # dW_val = dB_o W_out^T
        s1 += self._mm(gr("OO"), 64, 64, 64, 64, a("OV"), (1, 64), d("BO"), (64, 1))            # dW_out = W_val^T dB_o
        s1 += self._mm(gr("MQ"), 128, 128, 128, 128, d("AM"), (128, 1), a("MK"), (128, 1), nm, split=4)
        s1 += self._mm(gr("MK"), 128, 128, 128, 128, d("AM"), (1, 128), a("MQ"), (128, 1), nm, split=4)
        s1 += self._mm(self.dmscr.data_ptr(), 128, 128, 128, 128, D, (128, 1), a("UW", 128), (256, 1), split=4)   # dM = D Wu2
        s1 += self._mm(gr("UW", 128), 256, 128, 128, 128, D, (1, 128), self.mscr.data_ptr(), (128, 1), split=4)   # dWu2 = D^T M
        dm = self.dmscr.data_ptr()
        s2 = self._mm(gr("MV"), 128, 128, 128, 128, dm, (128, 1), a("MO"), (1, 128), split=4)   # dW_val = dM W_out^T
        s2 += self._mm(gr("MO"), 128, 128, 128, 128, a("MV"), (1, 128), dm, (128, 1), split=4)  # dW_out = W_val^T dM
        return [self._tasks(s1), self._tasks(s2)]

    def unfold(self, grads_plain):
        """plain-layout gradients of the kernel-facing matrices (a FA_SLAB buffer) -> gflat (== every parameter's
        .grad): the chain rule of the fold.  Uses M = W_val W_out of the LAST fold_pack()."""
        tl = self._unfold.get(grads_plain.data_ptr())
        if tl is None:      # never dropped: every GraphedPPOStep's graph replays fa_task_kernel on ITS list
            tl = self._unfold[grads_plain.data_ptr()] = self._build_unfold(grads_plain.data_ptr()) + [grads_plain]
        self._run(tl[0])
        self._run(tl[1])

    # -- optimizer --------------------------------------------------------------------------------------------------
    def bind_adam(self, opt):
        """Make a torch.optim.Adam over this module keep its state in flat buffers (exp_avg / exp_avg_sq of every
        parameter = views of mflat / vflat, the step counters = views of `steps`), so adam_step() below and the
        optimizer's own step() advance the same state."""
        if getattr(self, "_opt", None) is opt:
            return
        g = opt.param_groups[0]
        assert len(opt.param_groups) == 1 and not g.get("amsgrad") and not g.get("weight_decay") and not g.get("maximize"), \
            "fa_adam_step is plain Adam: one group, no amsgrad / weight decay / maximize"
        dev = self.pflat.device
        self.mflat, self.vflat = torch.zeros_like(self.pflat), torch.zeros_like(self.pflat)
        self.steps = torch.zeros(len(_PF), device=dev, dtype=torch.float32)
        seg = []
        for k, (name, get, off) in enumerate(_PF):
            p = get(self.pol)
            n = p.numel()
            seg.append(off)
            st = opt.state[p]
            views = {"step": self.steps[k], "exp_avg": self.mflat[off:off + n].view(p.shape),
                     "exp_avg_sq": self.vflat[off:off + n].view(p.shape)}
            for key, view in views.items():
                if key in st and torch.is_tensor(st[key]):
                    view.copy_(st[key].to(dev))
                st[key] = view
        assert seg == sorted(seg) and seg[0] == 0
        self._seg = torch.tensor(seg + [PF_FLOATS], dtype=torch.int32, device=dev)
        self._coef = torch.zeros(int(self._lib.load().fa_adam_scratch_floats()), device=dev)   # [0]: the clip coefficient
        self._hyper = torch.zeros(8, device=dev)      # (lr, beta1, beta2, eps, max_grad_norm): refresh_hyper()
        self._hyper_host = None
        self._opt = opt

    def refresh_hyper(self, opt, max_grad_norm):
        """Bring the device copy of (lr, beta1, beta2, eps, max_grad_norm) up to date: fa_adam_step_dev reads them at run
        time, so a captured optimizer step follows a change of opt.param_groups[0] (copy only when something moved)."""
        self.bind_adam(opt)
        g = opt.param_groups[0]
        hp = (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(max_grad_norm))
        if hp != self._hyper_host:
            self._hyper[:5].copy_(torch.tensor(hp, dtype=torch.float32))
            self._hyper_host = hp

    def adam_step(self, opt, max_grad_norm):
        """clip_grad_norm_(max_grad_norm) + Adam over the flat buffers: fa_adam_step_dev (two launches)."""
        if not torch.cuda.is_current_stream_capturing():
            self.refresh_hyper(opt, max_grad_norm)
        L, C = self._lib, self._C
        vp = lambda t: C.c_void_p(t.data_ptr())
        L.check(L.load().fa_adam_step_dev(vp(self.pflat), vp(self.gflat), vp(self.mflat), vp(self.vflat), vp(self.steps), vp(self._seg),
                                          len(_PF), PF_FLOATS, vp(self._hyper), vp(self._coef),
                                          C.c_void_p(torch.cuda.current_stream().cuda_stream)), "fa_adam_step_dev")
