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

// ---- the same GEMM on the bf16 matrix cores, fp32-class accuracy (three-way operand split) ---------------------------------
// v_mfma_f32_32x32x16_bf16 retires 16 x the MACs per cycle of v_mfma_f32_32x32x2_f32.  A float is the exact sum of three bf16
// numbers (hi + mid + lo: 8 significant bits each), so a product is the sum of nine bf16 x bf16 products, each exact in the
// MFMA's fp32 accumulator; the six largest are issued (hi hi, hi mid, mid hi, hi lo, lo hi, mid mid): 6 MFMAs of 32 cycles
// cover K = 16 where the fp32 form needs 8 of 64 -- 2.67 x less matrix-core time.  The WEIGHTS arrive pre-split (fa_policy.h
// FA_POFF3_*: split once per optimizer step by fa_pack_weights, round-to-nearest terms, so the three dropped cross products
// -- <= 2^-24 |a w| together -- carry either sign); the ACTIVATIONS are split here, on the way from LDS into the MFMA, by
// truncation (4 VALU operations per element, exact: a == hi + mid + lo).  Same K permutation as gemm_cb: lane half hh walks
// k in [hh K/2, (hh + 1) K/2), eight consecutive k per 16-byte step.
#ifndef FA_GEMM3_CH
#define FA_GEMM3_CH 1 // K = 16 steps of weights (3 x 16 bytes per lane each) requested ahead (2: 84 us instead of 79 at 3v3 x 4096,
#endif                // 157 instead of 145 at 5v5 -- the 12 more registers per step cost more than the deeper prefetch hides; 4: 123)
typedef __bf16 fa_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned fa_u32x4 __attribute__((ext_vector_type(4)));
template <int K>
struct BHead3 {
    static constexpr int NS = K / 16;
    static constexpr int CH = NS < FA_GEMM3_CH ? NS : FA_GEMM3_CH;
    fa_u32x4 v[CH * 3];
};
template <int K>
__device__ __forceinline__ void prefetch_b3(const fa_u32x4 *__restrict__ wp, int lane, BHead3<K> &h) {
#pragma unroll
    for (int c = 0; c < BHead3<K>::CH * 3; ++c) h.v[c] = wp[c * 64 + lane];
}
// eight floats -> three packed bf16x8 (hi, mid, lo), truncating split
__device__ __forceinline__ void fa_split8(const float4 &a0, const float4 &a1, fa_u32x4 &H, fa_u32x4 &M, fa_u32x4 &L) {
    const float x[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        h[j] = __float_as_uint(x[j]) & 0xffff0000u;
        const float r1 = x[j] - __uint_as_float(h[j]);
        m[j] = __float_as_uint(r1) & 0xffff0000u;
        l[j] = __float_as_uint(r1 - __uint_as_float(m[j]));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        H[j] = __builtin_amdgcn_perm(h[2 * j + 1], h[2 * j], 0x07060302u);
        M[j] = __builtin_amdgcn_perm(m[2 * j + 1], m[2 * j], 0x07060302u);
        L[j] = __builtin_amdgcn_perm(l[2 * j + 1], l[2 * j], 0x07060302u);
    }
}
// acc[r] += Act[rows of block r][K] * W[K][32 columns of one block]; arow as gemm_cb; wp3: the column block's bf16x3 pack,
// [K/16][3][64] 16-byte words; head: its first chunk, already requested
template <int K, int NRB>
__device__ __forceinline__ void gemm_cb3(const float *arow, const fa_u32x4 *__restrict__ wp3, f32x16 (&acc)[NRB], int lane,
                                         const BHead3<K> &head) {
    constexpr int NS = BHead3<K>::NS, CH = BHead3<K>::CH;
    fa_u32x4 bq[CH * 3], bn[CH * 3];
#pragma unroll
    for (int c = 0; c < CH * 3; ++c) bq[c] = head.v[c];
#pragma unroll
    for (int s0 = 0; s0 < NS; s0 += CH) {
        if (s0 + CH < NS) {
#pragma unroll
#ifdef FA_X3_KO_WLOAD // (knock-out builds: where the time goes)
            for (int c = 0; c < CH * 3; ++c) bn[c] = bq[c] ^ (unsigned)s0;
#else
            for (int c = 0; c < CH * 3; ++c) bn[c] = wp3[((s0 + CH) * 3 + c) * 64 + lane];
#endif
#ifndef FA_NO_SCHED_BARRIER
            __builtin_amdgcn_sched_barrier(0); // (as gemm_cb: keep the requests here)
#endif
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const fa_bf16x8 bh = __builtin_bit_cast(fa_bf16x8, bq[c * 3]), bm = __builtin_bit_cast(fa_bf16x8, bq[c * 3 + 1]),
                            bl = __builtin_bit_cast(fa_bf16x8, bq[c * 3 + 2]);
#pragma unroll
            for (int r = 0; r < NRB; ++r) {
                const float *ap = arow + r * 32 * LDA + (s0 + c) * 8;
                fa_u32x4 H, M, L;
#ifdef FA_X3_KO_SPLIT
                H = __builtin_bit_cast(fa_u32x4, *reinterpret_cast<const float4 *>(ap));
                M = __builtin_bit_cast(fa_u32x4, *reinterpret_cast<const float4 *>(ap + 4));
                L = H ^ M;
#else
                fa_split8(*reinterpret_cast<const float4 *>(ap), *reinterpret_cast<const float4 *>(ap + 4), H, M, L);
#endif
                const fa_bf16x8 ah = __builtin_bit_cast(fa_bf16x8, H), am = __builtin_bit_cast(fa_bf16x8, M),
                                al = __builtin_bit_cast(fa_bf16x8, L);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[r], 0, 0, 0);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[r], 0, 0, 0);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[r], 0, 0, 0);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[r], 0, 0, 0);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[r], 0, 0, 0);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[r], 0, 0, 0);
            }
#if !defined(FA_NO_SCHED_BARRIER) && !defined(FA_X3_NO_STEP_FENCE)
            // one K = 16 step at a time: left alone the scheduler hoists the LDS reads and the splits of many steps to the top
            // (the split's temporaries are 40 registers per row block) and spills
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
        if (s0 + CH < NS) {
#pragma unroll
            for (int c = 0; c < CH * 3; ++c) bq[c] = bn[c];
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
