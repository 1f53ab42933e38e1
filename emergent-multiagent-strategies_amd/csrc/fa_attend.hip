// fa_attend.hip -- the agents-of-one-env attention of the MPNN (reference mpnn.py:250-332 MultiHeadAttention,
// :372-443 MultiHeadOppAttention, one head) as ONE kernel forward and ONE kernel backward, for the PPO
// update's training forward (learner.joint_ppo_update -> MPNN.evaluate_actions).
//
//   s_ij = g_i . k_j   (j over the env's nk key rows; j == i excluded for the team's self attention)
//   a_i  = softmax_j(s_ij)
//   o_i  = sum_j a_ij k_j
//
// with g = h A already projected by a GEMM (the scale 1/sqrt(d) and W_query W_key^T are folded into A,
// W_val W_out into the GEMM that follows: see mpnn_pack.py for the algebra).  As PyTorch ops this is a
// broadcast multiply over (B, n, nk, W), two reductions, a softmax and another broadcast multiply + sum --
// ~15 launches forward and ~35 backward over tensors n x larger than the activations, two thirds of the
// update's GPU time.  Here one 16-lane sub-group (a DPP row) owns an env: lane q holds W/16 columns of every
// row, scores are reduced with four row-rotate adds, everything else stays in registers.
//
// Backward (do = dL/do):  da_ij = do_i . k_j ;  ds_ij = a_ij (da_ij - sum_l a_il da_il) ;
//   dg_i = sum_j ds_ij k_j ;  dk_j = sum_i (a_ij do_i + ds_ij g_i).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fa_policy.h"

namespace {
__device__ __forceinline__ float row16_sum(float v) { // sum over the 16 lanes of a DPP row, result in every lane
#define FA_ROR_ADD(n) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 | (n), 0xf, 0xf, false))
    FA_ROR_ADD(8);
    FA_ROR_ADD(4);
    FA_ROR_ADD(2);
    FA_ROR_ADD(1);
#undef FA_ROR_ADD
    return v;
}

constexpr int MT = FA_POLICY_MAX_TEAM;

// g (B*n, W) rows env-major; keys (B, nk, W); out (B*n, W); attn (B*n, nk) saved for the backward
template <int W, bool BACKWARD>
__global__ __launch_bounds__(256) void fa_attend_kernel(const float *__restrict__ g, const float *__restrict__ keys,
                                                       float *__restrict__ out, float *__restrict__ attn,
                                                       const float *__restrict__ dout, float *__restrict__ dg,
                                                       float *__restrict__ dkeys, int B, int n, int nk, int skip_self) {
    constexpr int C = W / 16;
    const int q = threadIdx.x & 15;
    const long long b = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (b >= B) return; // whole sub-groups leave together: the DPP rows of the others are complete
    float kv[MT][C];
#pragma unroll
    for (int j = 0; j < MT; ++j)
        if (j < nk) {
#pragma unroll
            for (int c = 0; c < C; c += 4)
                *reinterpret_cast<float4 *>(&kv[j][c]) = *reinterpret_cast<const float4 *>(keys + ((size_t)b * nk + j) * W + q * C + c);
        }
    float dk[MT][C];
    if (BACKWARD) {
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int c = 0; c < C; ++c) dk[j][c] = 0.0f;
    }
    for (int i = 0; i < n; ++i) {
        const size_t r = (size_t)b * n + i;
        float gv[C];
#pragma unroll
        for (int c = 0; c < C; c += 4) *reinterpret_cast<float4 *>(gv + c) = *reinterpret_cast<const float4 *>(g + r * W + q * C + c);
        const int skip = skip_self ? i : -1;
        float a[MT];
        if (!BACKWARD) {
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                a[j] = -INFINITY;
                if (j < nk && j != skip) {
                    float d = 0.0f;
#pragma unroll
                    for (int c = 0; c < C; ++c) d = fmaf(gv[c], kv[j][c], d);
                    a[j] = row16_sum(d);
                    mx = fmaxf(mx, a[j]);
                }
            }
            float den = 0.0f;
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                a[j] = (j < nk && j != skip) ? expf(a[j] - mx) : 0.0f;
                den += a[j];
            }
            const float inv = den > 0.0f ? 1.0f / den : 0.0f; // a team of one: no partner, zero message (mpnn.py:266-274)
            float ov[C];
#pragma unroll
            for (int c = 0; c < C; ++c) ov[c] = 0.0f;
#pragma unroll
            for (int j = 0; j < MT; ++j)
                if (j < nk) {
                    a[j] *= inv;
#pragma unroll
                    for (int c = 0; c < C; ++c) ov[c] = fmaf(a[j], kv[j][c], ov[c]);
                }
#pragma unroll
            for (int c = 0; c < C; c += 4) *reinterpret_cast<float4 *>(out + r * W + q * C + c) = *reinterpret_cast<const float4 *>(ov + c);
            if (q < nk) { // lane q stores a[q]
                float aq = a[0];
#pragma unroll
                for (int j = 1; j < MT; ++j) aq = (j == q) ? a[j] : aq;
                attn[r * nk + q] = aq;
            }
        } else {
            float dov[C];
#pragma unroll
            for (int c = 0; c < C; c += 4) *reinterpret_cast<float4 *>(dov + c) = *reinterpret_cast<const float4 *>(dout + r * W + q * C + c);
            float da[MT], dot = 0.0f;
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                a[j] = 0.0f;
                da[j] = 0.0f;
                if (j < nk) {
                    a[j] = attn[r * nk + j];
                    float d = 0.0f;
#pragma unroll
                    for (int c = 0; c < C; ++c) d = fmaf(dov[c], kv[j][c], d);
                    da[j] = row16_sum(d);
                    dot = fmaf(a[j], da[j], dot);
                }
            }
            float dgv[C];
#pragma unroll
            for (int c = 0; c < C; ++c) dgv[c] = 0.0f;
#pragma unroll
            for (int j = 0; j < MT; ++j)
                if (j < nk) {
                    const float ds = a[j] * (da[j] - dot); // 0 for the excluded self pair (a == 0)
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        dgv[c] = fmaf(ds, kv[j][c], dgv[c]);
                        dk[j][c] = fmaf(a[j], dov[c], fmaf(ds, gv[c], dk[j][c]));
                    }
                }
#pragma unroll
            for (int c = 0; c < C; c += 4) *reinterpret_cast<float4 *>(dg + r * W + q * C + c) = *reinterpret_cast<const float4 *>(dgv + c);
        }
    }
    if (BACKWARD) {
#pragma unroll
        for (int j = 0; j < MT; ++j)
            if (j < nk) {
#pragma unroll
                for (int c = 0; c < C; c += 4)
                    *reinterpret_cast<float4 *>(dkeys + ((size_t)b * nk + j) * W + q * C + c) = *reinterpret_cast<const float4 *>(&dk[j][c]);
            }
    }
}
} // namespace

hipError_t fa_launch_attend(int width, bool backward, const float *g, const float *keys, float *out, float *attn,
                            const float *dout, float *dg, float *dkeys, int B, int n, int nk, int skip_self, hipStream_t st) {
    const dim3 grid((B + 15) / 16), block(256);
#define FA_ATT(W_, BW_) hipLaunchKernelGGL((fa_attend_kernel<W_, BW_>), grid, block, 0, st, g, keys, out, attn, dout, dg, dkeys, B, n, nk, skip_self)
    if (width == 128) { if (backward) FA_ATT(128, true); else FA_ATT(128, false); }
    else if (width == 64) { if (backward) FA_ATT(64, true); else FA_ATT(64, false); }
    else return hipErrorInvalidValue;
#undef FA_ATT
    return hipGetLastError();
}
