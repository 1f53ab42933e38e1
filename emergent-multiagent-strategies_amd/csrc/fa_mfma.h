// fa_mfma.h -- device helpers shared by the fused MPNN kernels (fa_policy.hip: rollout forward; fa_train.hip:
// the PPO update's forward + backward): fp32-MFMA GEMM on LDS-resident activation tiles with lane-ordered
// weights streamed from L2, accumulator stores, the 16-lane attention row.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fa_policy.h"

#ifndef FA_GEMM_CH
#define FA_GEMM_CH 8 // 16-byte weight steps requested ahead per lane (a chunk = FA_GEMM_CH x 4 MFMAs per row block)
#endif
namespace {
constexpr int LDA = 132;           // padded LDS row stride in floats (128 + 4)
typedef float f32x16 __attribute__((ext_vector_type(16)));

// The first weights of a layer are requested well before the layer starts (behind the previous layer's
// stores, barrier and attention phase): an L2 round trip per layer start was a fifth of the kernel.
template <int K>
struct BHead {
    static constexpr int NT = K / 8;           // 16-byte steps per lane half
    static constexpr int CH = NT < FA_GEMM_CH ? NT : FA_GEMM_CH;  // steps per chunk
    float4 v[CH];
};
template <int K>
__device__ __forceinline__ void prefetch_b(const float4 *__restrict__ wp, int lane, BHead<K> &h) {
#pragma unroll
    for (int c = 0; c < BHead<K>::CH; ++c) h.v[c] = wp[c * 64 + lane];
}

// acc[r] += Act[rows of block r][K] * W[K][32 columns of one block]   (fp32 MFMA, exact fmaf chain)
// arow: this lane's first A element -- &Act[(rb0*32 + (lane & 31)) * LDA + <start of the lane half's k range>]
// wp:   packed weights of the column block: [K/8][64] float4;  head: its first chunk, already requested
template <int K, int NRB>
__device__ __forceinline__ void gemm_cb(const float *arow, const float4 *__restrict__ wp, f32x16 (&acc)[NRB], int lane,
                                        const BHead<K> &head) {
    constexpr int NT = BHead<K>::NT, CH = BHead<K>::CH; // weights are fetched a chunk of CH steps ahead
    float4 bq[CH], bn[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) bq[c] = head.v[c];
#pragma unroll
    for (int t0 = 0; t0 < NT; t0 += CH) {
        if (t0 + CH < NT) {
#pragma unroll
            for (int c = 0; c < CH; ++c) bn[c] = wp[(t0 + CH + c) * 64 + lane];
#ifndef FA_NO_SCHED_BARRIER
            // keep the requests HERE: under register pressure the scheduler sinks each load to its first use and
            // waits for an L2 round trip (vmcnt(0)) in front of every four MFMAs
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
#pragma unroll
            for (int r = 0; r < NRB; ++r) {
                const float4 a = *reinterpret_cast<const float4 *>(arow + r * 32 * LDA + (t0 + c) * 4);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq[c].x, acc[r], 0, 0, 0);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq[c].y, acc[r], 0, 0, 0);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq[c].z, acc[r], 0, 0, 0);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq[c].w, acc[r], 0, 0, 0);
            }
        }
        if (t0 + CH < NT) {
#pragma unroll
            for (int c = 0; c < CH; ++c) bq[c] = bn[c];
        }
    }
}

// C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
template <bool RELU>
__device__ __forceinline__ void store_acc(float *dst, int rb, const f32x16 &acc, float bias, int lane) {
    const int col = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int row = rb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
        float v = acc[reg] + bias;
        if (RELU) v = fmaxf(v, 0.0f);
        dst[row * LDA + col] = v;
    }
}

// sum over the 16 lanes of a row's sub-group == one DPP row: four rotate-and-add steps on the VALU
// (row_ror:8,4,2,1), every lane ends with the total.  (__shfl_xor goes through ds_bpermute: an LDS round
// trip per step, 4 dependent ones per score.)
__device__ __forceinline__ float group16_sum(float v) {
#define FA_ROR_ADD(n) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 | (n), 0xf, 0xf, false))
    FA_ROR_ADD(8);
    FA_ROR_ADD(4);
    FA_ROR_ADD(2);
    FA_ROR_ADD(1);
#undef FA_ROR_ADD
    return v;
}

// Attention mix of one row r = (env el, own agent i) by a 16-lane sub-group: scores s_j = g[r] . key[j]
// over the env's `nk` key rows (skipping j == skip), softmax, out[r] = sum_j a_j key[j]  (W floats per row,
// W / 16 per lane).  `g` and `out` may be the same row (g[r] is only read by this sub-group).
template <int W>
__device__ __forceinline__ void attend_row(const float *grow, const float *key0, int nk, int skip, float *orow, int q,
                                           float *attn_row = nullptr) {
    constexpr int C = W / 16; // columns per lane: 4 or 8
    float gv[C];
#pragma unroll
    for (int c = 0; c < C; c += 4) *reinterpret_cast<float4 *>(gv + c) = *reinterpret_cast<const float4 *>(grow + q * C + c);
    float s[FA_POLICY_MAX_TEAM];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < FA_POLICY_MAX_TEAM; ++j) {
        s[j] = -INFINITY;
        if (j < nk && j != skip) {
            float kv[C];
#pragma unroll
            for (int c = 0; c < C; c += 4)
                *reinterpret_cast<float4 *>(kv + c) = *reinterpret_cast<const float4 *>(key0 + j * LDA + q * C + c);
            float d = 0.0f;
#pragma unroll
            for (int c = 0; c < C; ++c) d = fmaf(gv[c], kv[c], d);
            s[j] = group16_sum(d);
            mx = fmaxf(mx, s[j]);
        }
    }
    float den = 0.0f;
#pragma unroll
    for (int j = 0; j < FA_POLICY_MAX_TEAM; ++j) {
        s[j] = (j < nk && j != skip) ? __expf(s[j] - mx) : 0.0f;
        den += s[j];
    }
    const float inv = den > 0.0f ? 1.0f / den : 0.0f; // a team of one has nobody to listen to: msg = 0 (mpnn.py:266-274)
    float ov[C];
#pragma unroll
    for (int c = 0; c < C; ++c) ov[c] = 0.0f;
#pragma unroll
    for (int j = 0; j < FA_POLICY_MAX_TEAM; ++j) {
        if (j < nk && j != skip) {
            const float a = s[j] * inv;
#pragma unroll
            for (int c = 0; c < C; c += 4) {
                const float4 kv = *reinterpret_cast<const float4 *>(key0 + j * LDA + q * C + c);
                ov[c] = fmaf(a, kv.x, ov[c]);
                ov[c + 1] = fmaf(a, kv.y, ov[c + 1]);
                ov[c + 2] = fmaf(a, kv.z, ov[c + 2]);
                ov[c + 3] = fmaf(a, kv.w, ov[c + 3]);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < C; c += 4) *reinterpret_cast<float4 *>(orow + q * C + c) = *reinterpret_cast<const float4 *>(ov + c);
    if (attn_row && q < nk) { // the weights, for a backward pass: lane q keeps a_q (0 for the excluded pair)
        float aq = s[0];
#pragma unroll
        for (int j = 1; j < FA_POLICY_MAX_TEAM; ++j) aq = (j == q) ? s[j] : aq;
        attn_row[q] = aq * inv;
    }
}

// The same attention row with the env's keys held in registers (MT >= nk: 4 for teams of up to four, else
// FA_POLICY_MAX_TEAM) and no divergent control flow -- the excluded pair is masked, not skipped -- and the result
// returned in registers, so that a caller can compute several rows before it stores any (the loads of the next
// row then overlap this row's DPP / exp chains).  Same operations in the same order as attend_row: same bits.
template <int W, int MT>
__device__ __forceinline__ void attend_row_regs(const float *grow, const float *key0, int nk, int skip, int q, float (&ov)[W / 16],
                                                float *attn_row = nullptr) {
    constexpr int C = W / 16;
    float gv[C], kv[MT][C];
#pragma unroll
    for (int c = 0; c < C; c += 4) *reinterpret_cast<float4 *>(gv + c) = *reinterpret_cast<const float4 *>(grow + q * C + c);
#pragma unroll
    for (int j = 0; j < MT; ++j) {
#pragma unroll
        for (int c = 0; c < C; c += 4)
            *reinterpret_cast<float4 *>(&kv[j][c]) = j < nk ? *reinterpret_cast<const float4 *>(key0 + j * LDA + q * C + c) : float4{0, 0, 0, 0};
    }
    float s[MT], mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        float d = 0.0f;
#pragma unroll
        for (int c = 0; c < C; ++c) d = fmaf(gv[c], kv[j][c], d);
        s[j] = group16_sum(d);
    }
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        s[j] = (j < nk && j != skip) ? s[j] : -INFINITY;
        mx = fmaxf(mx, s[j]);
    }
    float den = 0.0f;
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        s[j] = (j < nk && j != skip) ? __expf(s[j] - mx) : 0.0f;
        den += s[j];
    }
    const float inv = den > 0.0f ? 1.0f / den : 0.0f; // a team of one has nobody to listen to: msg = 0 (mpnn.py:266-274)
#pragma unroll
    for (int c = 0; c < C; ++c) ov[c] = 0.0f;
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        if (j < nk) { // (uniform; the excluded pair's weight is an exact 0: fmaf(0, k, o) == o)
            const float a = s[j] * inv;
#pragma unroll
            for (int c = 0; c < C; ++c) ov[c] = fmaf(a, kv[j][c], ov[c]);
        }
    }
    if (attn_row && q < nk) {
        float aq = s[0];
#pragma unroll
        for (int j = 1; j < MT; ++j) aq = (j == q) ? s[j] : aq;
        attn_row[q] = aq * inv;
    }
}
template <int W>
__device__ __forceinline__ void store_row_regs(float *orow, int q, const float (&ov)[W / 16]) {
    constexpr int C = W / 16;
#pragma unroll
    for (int c = 0; c < C; c += 4) *reinterpret_cast<float4 *>(orow + q * C + c) = *reinterpret_cast<const float4 *>(ov + c);
}

} // namespace
