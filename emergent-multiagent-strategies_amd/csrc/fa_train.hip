// fa_train.hip -- one team's PPO minibatch: the MPNN forward (as fa_policy.hip, folded algebra), the alive-masked
// clipped PPO losses of JointPPO.update (reference rlcore/algo/ppo.py:146-187) and the backward pass down to dL/dX of
// every layer, for a tile of 32 (env, agent) rows per workgroup held in four LDS buffers.  The WEIGHT gradients
// dW = X^T dY are not made here: the tile leaves their operands in global memory and fa_train_dw_kernel
// (fa_train_dw.hip) sums them over all rows of the minibatch as one split-K GEMM.
//
// Why: as PyTorch autograd the optimizer step is ~160 launches and 2.65 ms at 16 384 x 3 samples, of which the
// GEMMs are less than half; the rest is bias / relu / add / reduction kernels over (49 152 x 128) tensors.  Here the
// activations of a tile never leave the CU between layers and the elementwise work is folded into the accumulator
// stores.
//
// Shape (round 4).  Rounds 2-3 ran one 64-row tile per CU on eight waves and kept the weight gradients shared by the
// three message-passing rounds (dW7, dA_m: 96 accumulator registers per lane) live across the rounds, each tile writing
// a 390 KB gradient slab: 0.32 of the fp32 MFMA peak, 2 GB of HBM traffic per launch, half of it register spills, and
// every non-GEMM phase (attention, relu masks, tile reloads, barriers) ran with the matrix pipes idle because all eight
// waves march through the same phases.  Now: a 32-row tile on FOUR waves (a wave owns one 32 x 32 output tile of a
// 32 x 128 layer: column block = wave), 81 KB of LDS, so TWO workgroups share a CU and one's attention / store /
// barrier phases run under the other's MFMA chains; no accumulator lives longer than one GEMM; per tile 10 KB of small
// gradients (encoders, biases, the 1 152 real entries of dW9) instead of 390 KB.
//
// Saved for the backward and for fa_train_dw_kernel (fa_train.h FA_RECA_* / FA_RECB_*): per round the hidden state
// entering it, the attention mix, dZ and dg; per tile h3, [dP | dV], and the opponent stage's mix_o, de_opp, h1, dg_o;
// all attention weights stay in LDS.  g = h A of every round is SAVED by the forward too (recG: the backward reloads it instead
// of recomputing it -- one more GEMM per round cost more than the round trip); the opponents' encodings are recomputed.
// Footprint: FA_TR_SAVE_FLOATS = 81 920 floats = 320 KB of records per 32-row tile -- 524 MB per fa_ppo_grad call at config 3
// (1 639 tiles; ~0.9 GB at 5v5), written once by the tile and read back once (backward / fa_train_dw_kernel) -- next to which the
// 47 MB of partial slabs and 10 KB of small gradients per tile are small: the records ARE the update's HBM traffic
// (PMC: 971 MB per tile-kernel launch).  Each captured GraphedPPOStep holds its own scratch.
//
// Buffers (32 x 132 floats each): B0 = h (forward) / h_in of the round being differentiated; B1 = g, hmix, P;
// B2 = ho, V, recomputed g; B3 = the running dL/dh.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fa_train.h"
// weights requested four 16-byte steps ahead per lane (fa_policy.hip: eight): with one accumulator tile per wave and the
// backward's operands in flight the tile kernel is at its register limit, and the shorter chunk spills less
// (A/B: 754 -> 710 us per fa_ppo_grad call, 0.191 -> 0.180 s per update)
#ifndef FA_GEMM_CH
#define FA_GEMM_CH 4
#endif
#include "fa_mfma.h"
#define FA_PROBE_TRAIN_TU
#include "fa_probe.h"

namespace {
constexpr int TR = FA_TR_ROWS;
constexpr int NWV = 4, NTH = NWV * 64;
constexpr int SOW = 32; // row stride of the head-output buffer sO (9 used: 8 logits + value)

// acc (32 x 32) += X[rows][32 cols at X]^T * DY[rows][32 cols at DY] over the tile's 32 rows: both operands
// from LDS, one float each per MFMA (A[i][kk] = X[row kk][i], B[kk][j] = DY[row kk][j]; lane half hh walks
// rows hh*16 .. hh*16+15)
__device__ __forceinline__ void gemm_tn(const float *X, int ldx, const float *DY, int ldy, f32x16 &acc, int lane) {
    const int li = lane & 31, hh = lane >> 5;
    const float *xp = X + (hh * (TR / 2)) * ldx + li, *yp = DY + (hh * (TR / 2)) * ldy + li;
#pragma unroll
    for (int t = 0; t < TR / 2; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xp[t * ldx], yp[t * ldy], acc, 0, 0, 0);
}

// dst += acc  /  dst = (dst > 0 ? acc : 0)   on an LDS tile (the lane that stores an element also owns its old value)
__device__ __forceinline__ void store_acc_add(float *dst, const f32x16 &acc, int lane) {
    const int col = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        float *p = dst + ((reg & 3) + 8 * (reg >> 2) + 4 * hh) * LDA + col;
        *p += acc[reg];
    }
}
// dst = (gate > 0 ? dst + acc : 0): the last contribution to dL/dh of a round together with the relu gate of the round
// below (gate = that round's output h, same tile position) -- no separate masking pass over the tile
__device__ __forceinline__ void store_acc_add_gate(float *dst, const float *gate, const f32x16 &acc, int lane) {
    const int col = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int o = ((reg & 3) + 8 * (reg >> 2) + 4 * hh) * LDA + col;
        dst[o] = gate[o] > 0.0f ? dst[o] + acc[reg] : 0.0f;
    }
}
__device__ __forceinline__ void store_acc_gate(float *dst, const float *gate, const f32x16 &acc, int lane) {
    const int col = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int o = ((reg & 3) + 8 * (reg >> 2) + 4 * hh) * LDA + col;
        dst[o] = gate[o] > 0.0f ? acc[reg] : 0.0f;
    }
}
__device__ __forceinline__ void store_acc_relu_mask(float *dst, const f32x16 &acc, int lane) {
    const int col = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        float *p = dst + ((reg & 3) + 8 * (reg >> 2) + 4 * hh) * LDA + col;
        *p = *p > 0.0f ? acc[reg] : 0.0f;
    }
}

// Backward of the attention of ONE env by a 16-lane sub-group (cf. fa_attend.hip): rows r0 .. r0+n-1 of
// dout / g (g is overwritten by dg), the env's nk key rows at key0, their gradient ADDED into dkey0 rows
// (ADD) or written (!ADD).  W floats per row.
#ifndef FA_ATTN_BWD_STREAM_KEYS
#define FA_ATTN_BWD_STREAM_KEYS 1
#endif
template <int W, bool ADD, int MT>
__device__ __forceinline__ void attend_env_bwd(const float *dout0, float *g0, const float *key0, float *dkey0, const float *attn0,
                                               int n, int nk, int q) {
    constexpr int C = W / 16;
    if constexpr (FA_ATTN_BWD_STREAM_KEYS && MT > 4) {
        // Larger teams: the keys are NOT held in registers (kv[MT][C] + dk[MT][C] = 96 registers at MT = 6 put 47 VGPRs of the
        // 5v5 instantiation into scratch).  One pass over the keys per row, each key read from LDS once:
        //     da_j = dout . k_j,  dot = sum_j a_j da_j,  U = sum_j (a_j da_j) k_j,  V = sum_j a_j k_j
        // then dg = sum_j a_j (da_j - dot) k_j = U - dot V, and the key gradients need scalars only:
        //     dk_j += a_j dout + a_j (da_j - dot) g.
        float dk[MT][C];
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int c = 0; c < C; ++c) dk[j][c] = 0.0f;
        for (int i = 0; i < n; ++i) {
            float gv[C], dov[C], U[C], V[C];
#pragma unroll
            for (int c = 0; c < C; c += 4) {
                *reinterpret_cast<float4 *>(gv + c) = *reinterpret_cast<const float4 *>(g0 + i * LDA + q * C + c);
                *reinterpret_cast<float4 *>(dov + c) = *reinterpret_cast<const float4 *>(dout0 + i * LDA + q * C + c);
            }
#pragma unroll
            for (int c = 0; c < C; ++c) { U[c] = 0.0f; V[c] = 0.0f; }
            float a[MT], da[MT], dot = 0.0f;
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                a[j] = 0.0f;
                da[j] = 0.0f;
                if (j < nk) {
                    float kj[C];
#pragma unroll
                    for (int c = 0; c < C; c += 4)
                        *reinterpret_cast<float4 *>(kj + c) = *reinterpret_cast<const float4 *>(key0 + j * LDA + q * C + c);
                    a[j] = attn0[i * 8 + j];
                    float d = 0.0f;
#pragma unroll
                    for (int c = 0; c < C; ++c) d = fmaf(dov[c], kj[c], d);
                    da[j] = group16_sum(d);
                    const float ada = a[j] * da[j];
                    dot += ada;
#pragma unroll
                    for (int c = 0; c < C; ++c) { U[c] = fmaf(ada, kj[c], U[c]); V[c] = fmaf(a[j], kj[c], V[c]); }
                }
            }
#pragma unroll
            for (int j = 0; j < MT; ++j)
                if (j < nk) {
                    const float ds = a[j] * (da[j] - dot);
#pragma unroll
                    for (int c = 0; c < C; ++c) dk[j][c] = fmaf(a[j], dov[c], fmaf(ds, gv[c], dk[j][c]));
                }
            float dgv[C];
#pragma unroll
            for (int c = 0; c < C; ++c) dgv[c] = fmaf(-dot, V[c], U[c]);
#pragma unroll
            for (int c = 0; c < C; c += 4) *reinterpret_cast<float4 *>(g0 + i * LDA + q * C + c) = *reinterpret_cast<const float4 *>(dgv + c);
        }
#pragma unroll
        for (int j = 0; j < MT; ++j)
            if (j < nk) {
#pragma unroll
                for (int c = 0; c < C; c += 4) {
                    float4 *p = reinterpret_cast<float4 *>(dkey0 + j * LDA + q * C + c);
                    float4 v = *reinterpret_cast<const float4 *>(&dk[j][c]);
                    if (ADD) { const float4 o = *p; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                    *p = v;
                }
            }
        return;
    }
    float kv[MT][C], dk[MT][C];
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int c = 0; c < C; ++c) { kv[j][c] = 0.0f; dk[j][c] = 0.0f; }
#pragma unroll
    for (int j = 0; j < MT; ++j)
        if (j < nk) {
#pragma unroll
            for (int c = 0; c < C; c += 4)
                *reinterpret_cast<float4 *>(&kv[j][c]) = *reinterpret_cast<const float4 *>(key0 + j * LDA + q * C + c);
        }
    for (int i = 0; i < n; ++i) {
        float gv[C], dov[C];
#pragma unroll
        for (int c = 0; c < C; c += 4) {
            *reinterpret_cast<float4 *>(gv + c) = *reinterpret_cast<const float4 *>(g0 + i * LDA + q * C + c);
            *reinterpret_cast<float4 *>(dov + c) = *reinterpret_cast<const float4 *>(dout0 + i * LDA + q * C + c);
        }
        float a[MT], da[MT], dot = 0.0f;
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            a[j] = 0.0f;
            da[j] = 0.0f;
            if (j < nk) {
                a[j] = attn0[i * 8 + j];
                float d = 0.0f;
#pragma unroll
                for (int c = 0; c < C; ++c) d = fmaf(dov[c], kv[j][c], d);
                da[j] = group16_sum(d);
                dot = fmaf(a[j], da[j], dot);
            }
        }
        float dgv[C];
#pragma unroll
        for (int c = 0; c < C; ++c) dgv[c] = 0.0f;
#pragma unroll
        for (int j = 0; j < MT; ++j)
            if (j < nk) {
                const float ds = a[j] * (da[j] - dot);
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    dgv[c] = fmaf(ds, kv[j][c], dgv[c]);
                    dk[j][c] = fmaf(a[j], dov[c], fmaf(ds, gv[c], dk[j][c]));
                }
            }
#pragma unroll
        for (int c = 0; c < C; c += 4) *reinterpret_cast<float4 *>(g0 + i * LDA + q * C + c) = *reinterpret_cast<const float4 *>(dgv + c);
    }
#pragma unroll
    for (int j = 0; j < MT; ++j)
        if (j < nk) {
#pragma unroll
            for (int c = 0; c < C; c += 4) {
                float4 *p = reinterpret_cast<float4 *>(dkey0 + j * LDA + q * C + c);
                float4 v = *reinterpret_cast<const float4 *>(&dk[j][c]);
                if (ADD) { const float4 o = *p; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                *p = v;
            }
        }
}

// GATHER: the minibatch is rows a.idx[.] of the rollout arrays (else rows 0..B)
// MT: the attention backward's key / key-gradient registers are sized for teams of up to MT agents (4, 6 or 8)
template <bool GATHER, int MT>
__device__ __forceinline__ void fa_train_body(const FaTrainArgs &a) {
    __shared__ __attribute__((aligned(16))) float B0[TR * LDA], B1[TR * LDA], B2[TR * LDA], B3[TR * LDA];
    __shared__ float sX[TR * 2 * FA_OBS_DIM];
    __shared__ __attribute__((aligned(16))) float sO[TR * SOW];
    __shared__ float sAttn[4][TR * 8]; // [opponent stage, round 0, 1, 2][row][key]
    __shared__ __attribute__((aligned(16))) float sO2[TR * SOW]; // bias partial sums; the opponent side's [x | 1] rows

    // `wave` as a scalar: conditions on it become scalar branches, not exec-masked regions (cheaper, and see
    // tools/isa_lint.py for why this kernel wants as few exec-masked joins as possible)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hh = lane >> 5, q16 = lane & 15;
    const int cbw = wave; // this wave's column block of a 32 x 128 layer
    const int N = a.G + a.A;
    const int n = a.team == 0 ? a.G : a.A, m = N - n;
    const int own0 = a.team == 0 ? 0 : a.G, opp0 = a.team == 0 ? a.G : 0;
    const int ET = TR / (n > m ? n : m);
    const int tile = blockIdx.x + a.tile0, slot = blockIdx.x + a.rec_tile0;
    const int e0 = tile * ET;
    if (e0 >= a.B) return;
    const int ne = (a.B - e0) < ET ? (a.B - e0) : ET;
    const int RU = ET * n, RO = ET * m; // rows in use (own / opponent side); envs beyond `ne` are zero rows
    const float *W = a.w;
    const float4 *Wq = reinterpret_cast<const float4 *>(a.w), *Tq = reinterpret_cast<const float4 *>(a.wt);
    float *mslab = a.mslab + (size_t)tile * FA_MSLAB_FLOATS;
    float *recA = a.rec_a + (size_t)slot * 3 * FA_RECA_FLOATS, *recB = a.rec_b + (size_t)slot * FA_RECB_FLOATS;
    float *recG = a.rec_g + (size_t)slot * 3 * FA_REC_PLANE;

    // dense tile <-> LDS: W floats per row (128: four 16-byte pieces per thread; 64: two)
    auto save_tile = [&](const float *src, float *dst) { // 32 x 128 LDS -> global
#pragma unroll
        for (int k = tid; k < TR * 32; k += NTH) {
            const int r = k >> 5, c4 = k & 31;
            // (non-temporal: a record is read once, by another kernel -- it need not displace the weights in L2)
            typedef float v4f __attribute__((ext_vector_type(4)));
#if FA_REC_NT
            __builtin_nontemporal_store(*reinterpret_cast<const v4f *>(src + r * LDA + c4 * 4), reinterpret_cast<v4f *>(dst) + k);
#else
            reinterpret_cast<v4f *>(dst)[k] = *reinterpret_cast<const v4f *>(src + r * LDA + c4 * 4);
#endif
        }
    };
    auto save_tile64 = [&](const float *src, float *dst) { // 32 x 64 LDS (at src, row stride LDA) -> global
#pragma unroll
        for (int k = tid; k < TR * 16; k += NTH) {
            const int r = k >> 4, c4 = k & 15;
            reinterpret_cast<float4 *>(dst)[k] = *reinterpret_cast<const float4 *>(src + r * LDA + c4 * 4);
        }
    };
    auto load_tile = [&](float *dst, const float *src) {
#pragma unroll
        for (int k = tid; k < TR * 32; k += NTH) {
            const int r = k >> 5, c4 = k & 31;
            *reinterpret_cast<float4 *>(dst + r * LDA + c4 * 4) = reinterpret_cast<const float4 *>(src)[k];
        }
    };
    // encoders (mpnn.py:37-41) from sX: h1 -> dst_own[:, 0:64] (own rows), ho -> dst_opp[:, 0:64] (opponent rows)
    auto encoders = [&](float *dst_own, float *dst_opp) {
        const int col = tid & 63, grp = tid >> 6;
        float we[FA_OBS_DIM], wo[FA_OBS_DIM];
#pragma unroll
        for (int k = 0; k < FA_OBS_DIM; ++k) { we[k] = W[FA_POFF_WE + k * 64 + col]; wo[k] = W[FA_POFF_WOE + k * 64 + col]; }
        const float be = W[FA_POFF_BE + col], bo = W[FA_POFF_BOE + col];
        for (int r = grp; r < TR; r += NWV) {
            if (dst_own) {
                float v = 0.0f;
                if (r < RU) {
                    const int el = r / n, i = r - el * n;
                    const float *x = sX + (el * N + own0 + i) * FA_OBS_DIM;
                    v = be;
#pragma unroll
                    for (int k = 0; k < FA_OBS_DIM; ++k) v = fmaf(x[k], we[k], v);
                    v = fmaxf(v, 0.0f);
                }
                dst_own[r * LDA + col] = v;
            }
            float v = 0.0f;
            if (r < RO) {
                const int el = r / m, j = r - el * m;
                const float *x = sX + (el * N + opp0 + j) * FA_OBS_DIM;
                v = bo;
#pragma unroll
                for (int k = 0; k < FA_OBS_DIM; ++k) v = fmaf(x[k], wo[k], v);
                v = fmaxf(v, 0.0f);
            }
            dst_opp[r * LDA + col] = v;
        }
    };
    // g_o = h1 A_o -> B1[:, 64:128]  (64 output columns: waves 0, 1 = the two column blocks)
    auto project_opp = [&]() {
        if (wave >= 2) return;
        BHead<64> hd;
        const float4 *wp = Wq + FA_POFF_AO / 4 + wave * 8 * 64;
        prefetch_b<64>(wp, lane, hd);
        f32x16 acc[1] = {};
        gemm_cb<64, 1>(B0 + li * LDA + hh * 32, wp, acc, lane, hd);
        store_acc<false>(B1 + 64 + wave * 32, 0, acc[0], 0.0f, lane);
    };
    // g = h A_m: B0 -> dst (all 128 columns; wave = column block); `hd`: the first weights, requested earlier
    const float4 *wp_am = Wq + FA_POFF_AM / 4 + cbw * 16 * 64;
    auto project_team = [&](float *dst, const BHead<128> &hd) {
        f32x16 acc[1] = {};
        gemm_cb<128, 1>(B0 + li * LDA + hh * 64, wp_am, acc, lane, hd);
        store_acc<false>(dst + cbw * 32, 0, acc[0], 0.0f, lane);
    };

#if defined(FA_TRAIN_STAGGER) && FA_TRAIN_STAGGER > 0
    // (experiment: two workgroups that start together on a CU march through the same phases in lock step -- both in a
    //  GEMM, then both outside one; delay every second first-round workgroup)
    if (blockIdx.x < 512 && ((FA_TRAIN_STAGGER == 1 ? blockIdx.x : (blockIdx.x >> 8)) & 1))
        for (int k = 0; k < FA_TRAIN_STAGGER_SLEEPS; ++k) __builtin_amdgcn_s_sleep(127);
#endif
    // the loss inputs of this lane's row (wave 0: a lane per row), requested now: two dependent HBM round trips (index, then
    // row) that would otherwise stand in front of the single-wave loss phase
    float l_vp = 0.0f, l_rt = 0.0f, l_adv = 0.0f, l_olp = 0.0f;
    int l_act = 0;
    if (wave == 0 && lane < ne * n) {
        const int el = lane / n, i = lane - el * n;
        const size_t o = (size_t)(GATHER ? a.idx[e0 + el] : (int64_t)(e0 + el)) * N + own0 + i;
        l_act = (int)a.action[o]; l_vp = a.value_pred[o]; l_rt = a.ret[o]; l_olp = a.old_logp[o];
        // (the arithmetic of fa_adv_norm_kernel: float32 with float32 mean / std)
        l_adv = a.adv_mean ? ((l_rt - l_vp) - (float)a.adv_mean[own0 + i]) / ((float)a.adv_std[own0 + i] + 1e-5f) : a.adv[o];
    }
    FA_TR_TICK(0)
    // ================================ forward ==========================================================
    if (GATHER) {
        const int ow = N * FA_OBS_DIM;
        for (int k = tid; k < ET * ow; k += NTH) {
            const int el = k / ow;
            sX[k] = el < ne ? a.obs[(size_t)a.idx[e0 + el] * ow + (k - el * ow)] : 0.0f;
        }
    } else {
        for (int k = tid; k < ET * N * FA_OBS_DIM; k += NTH)
            sX[k] = k < ne * N * FA_OBS_DIM ? a.obs[(size_t)e0 * N * FA_OBS_DIM + k] : 0.0f;
    }
    __syncthreads();
    encoders(B0, B2);
    __syncthreads();
    project_opp();
    __syncthreads();
    {   // opponent attention (mpnn.py:372-443): both of a sub-group's rows computed before either is stored
        float ov[2][4];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int r = wave * 4 + (lane >> 4) + k * NWV * 4, rr = r < RU ? r : RU - 1;
            attend_row_regs<64, MT>(B1 + rr * LDA + 64, B2 + ((rr / n) * m) * LDA, m, -1, q16, ov[k], r < RU ? sAttn[0] + r * 8 : nullptr);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int r = wave * 4 + (lane >> 4) + k * NWV * 4;
            if (r < RU) store_row_regs<64>(B1 + r * LDA + 64, q16, ov[k]);
        }
    }
    __syncthreads();
    save_tile64(B1 + 64, recB + FA_RECB_MO); // (an operand of dB_o = mix_o^T de_opp)
    if (wave < 2) {   // e_opp = mix_o B_o -> B0[:, 64:128]
        BHead<64> hd;
        const float4 *wp = Wq + FA_POFF_BO / 4 + wave * 8 * 64;
        prefetch_b<64>(wp, lane, hd);
        f32x16 acc[1] = {};
        gemm_cb<64, 1>(B1 + li * LDA + 64 + hh * 32, wp, acc, lane, hd);
        store_acc<false>(B0 + 64 + wave * 32, 0, acc[0], 0.0f, lane);
    }
    FA_TR_TICK(1)
    BHead<128> hd_am; // the first weights of the next 128-deep layer, requested a phase ahead (an L2 round trip)
    prefetch_b<128>(wp_am, lane, hd_am);
    __syncthreads();
    save_tile(B0, recA + FA_RECA_HIN);
    const float4 *wpp = Wq + FA_POFF_W8 / 4 + cbw * 16 * 64, *wpv = Wq + FA_POFF_W8 / 4 + (4 + cbw) * 16 * 64;
    for (int round = 0; round < 3; ++round) {
        project_team(B1, hd_am);
        BHead<256> hd_u;
        const float4 *wp_u = Wq + FA_POFF_W7 / 4 + cbw * 32 * 64;
        prefetch_b<256>(wp_u, lane, hd_u);
        FA_TR_TICK(10 + 5 * round)
        __syncthreads();
        FA_TR_TICK(11 + 5 * round)
        {   // team attention, self excluded (mpnn.py:250-332); rows beyond the tile's envs: hmix = 0
            float ov[2][8];
            // g is saved for the backward (a GEMM and a barrier less per round there) by the sub-group that is about to
            // overwrite the row with the mix -- no other wave touches it in this phase
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int r = wave * 4 + (lane >> 4) + k * NWV * 4;
                float4 *dst = reinterpret_cast<float4 *>(recG + round * FA_REC_PLANE + r * 128 + q16 * 8);
                dst[0] = *reinterpret_cast<const float4 *>(B1 + r * LDA + q16 * 8);
                dst[1] = *reinterpret_cast<const float4 *>(B1 + r * LDA + q16 * 8 + 4);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int r = wave * 4 + (lane >> 4) + k * NWV * 4, rr = r < RU ? r : RU - 1, el = rr / n;
                attend_row_regs<128, MT>(B1 + rr * LDA, B0 + (el * n) * LDA, n, rr - el * n, q16, ov[k], r < RU ? sAttn[1 + round] + r * 8 : nullptr);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int r = wave * 4 + (lane >> 4) + k * NWV * 4;
                if (r >= RU) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) ov[k][c] = 0.0f;
                }
                store_row_regs<128>(B1 + r * LDA, q16, ov[k]);
            }
        }
        FA_TR_TICK(12 + 5 * round)
        __syncthreads();
        FA_TR_TICK(13 + 5 * round)
        save_tile(B1, recA + round * FA_RECA_FLOATS + FA_RECA_HMIX); // (an operand of dW7; the backward does not need it)
        {   // h' = relu([h | hmix] W7 + bu)
            f32x16 acc[1] = {};
            gemm_cb<256, 1>((hh ? B1 : B0) + li * LDA, wp_u, acc, lane, hd_u);
            prefetch_b<128>(round < 2 ? wp_am : wpp, lane, hd_am); // next: the following round's g, or the policy head
            const float bias = W[FA_POFF_BU + cbw * 32 + li];
            __syncthreads();
            store_acc<true>(B0 + cbw * 32, 0, acc[0], bias, lane);
        }
        __syncthreads();
        save_tile(B0, round < 2 ? recA + (round + 1) * FA_RECA_FLOATS + FA_RECA_HIN : recB + FA_RECB_H3);
        FA_TR_TICK(14 + 5 * round)
    }
    FA_TR_TICK(2)
    {   // heads: P = relu(h Wp0 + b) -> B1, V = relu(h Wv0 + b) -> B2
        BHead<128> hv;
        const BHead<128> &hp = hd_am;
        prefetch_b<128>(wpv, lane, hv);
        f32x16 accp[1] = {}, accv[1] = {};
        gemm_cb<128, 1>(B0 + li * LDA + hh * 64, wpp, accp, lane, hp);
        gemm_cb<128, 1>(B0 + li * LDA + hh * 64, wpv, accv, lane, hv);
        const float bp = W[FA_POFF_B8 + cbw * 32 + li], bv = W[FA_POFF_B8 + 128 + cbw * 32 + li];
        store_acc<true>(B1 + cbw * 32, 0, accp[0], bp, lane);
        store_acc<true>(B2 + cbw * 32, 0, accv[0], bv, lane);
    }
    __syncthreads();
    {   // [logits | value] = [P | V] W9 + b9 -> sO: the K = 256 rows split over the four waves (wave w: columns 32 w ..
        // of P on lane half 0 and of V on half 1), partial 32 x 32 tiles summed through B3 (free until the backward)
        BHead<64> hd;
        const float4 *wp = Wq + FA_POFF_W9 / 4 + wave * 8 * 64;
        prefetch_b<64>(wp, lane, hd);
        f32x16 acc[1] = {};
        gemm_cb<64, 1>((hh ? B2 : B1) + li * LDA + wave * 32, wp, acc, lane, hd);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) B3[wave * 1024 + ((reg & 3) + 8 * (reg >> 2) + 4 * hh) * 32 + li] = acc[0][reg];
    }
    __syncthreads();
    for (int k = tid; k < TR * 32; k += NTH)
        sO[k] = (((B3[k] + B3[1024 + k]) + B3[2048 + k]) + B3[3072 + k]) + W[FA_POFF_B9 + (k & 31)];
    __syncthreads();

    FA_TR_TICK(3)
    // ================================ losses (ppo.py:150-187) and dL/d[logits | value] -> sO ==============
    if (wave == 0) {   // a lane per row (lanes TR .. 63 only take part in the reductions)
        const int r = lane;
        float inv_count, unmask;
        if (a.scale) {
            inv_count = a.scale[0];
            unmask = a.scale[1];
        } else { // the library's own alive-mask mean: fold the partial sums (every workgroup, the same order)
            float t = a.mask_part[lane];
            for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d);
            const float cnt = (float)a.B * (float)n, mm = t / cnt;
            unmask = mm != 0.0f ? mm : 1.0f;
            inv_count = a.normalize ? 1.0f / (cnt * unmask) : 1.0f / cnt;
            if (tile == 0 && lane == 0) { a.scale_out[0] = inv_count; a.scale_out[1] = unmask; }
        }
        float vl = 0.0f, al = 0.0f, en = 0.0f, mk = 0.0f;
        float dlg[FA_NUM_ACTIONS], dval = 0.0f;
#pragma unroll
        for (int k = 0; k < FA_NUM_ACTIONS; ++k) dlg[k] = 0.0f;
        if (r < ne * n) {
            const int el = r / n, i = r - el * n;
            const float *lo = sO + r * SOW;
            mk = sX[(el * N + own0 + i) * FA_OBS_DIM]; // the alive flag (ppo.py:224)
            float lg[FA_NUM_ACTIONS], mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < FA_NUM_ACTIONS; ++k) { lg[k] = lo[k]; mx = fmaxf(mx, lg[k]); }
            float se = 0.0f;
#pragma unroll
            for (int k = 0; k < FA_NUM_ACTIONS; ++k) se += expf(lg[k] - mx);
            const float lse = mx + logf(se);
            const int act = l_act;
            float p[FA_NUM_ACTIONS], lpk[FA_NUM_ACTIONS], ent = 0.0f, lp = 0.0f;
#pragma unroll
            for (int k = 0; k < FA_NUM_ACTIONS; ++k) {
                lpk[k] = lg[k] - lse;
                p[k] = expf(lpk[k]);
                ent -= p[k] * lpk[k];
                lp = (k == act) ? lpk[k] : lp;
            }
            const float value = lo[8], vp = l_vp, rt = l_rt, adv = l_adv;
            const float ratio = mk * expf(lp - l_olp);
            const float s1 = ratio * adv, rc = fminf(fmaxf(ratio, 1.0f - a.clip), 1.0f + a.clip), s2 = rc * adv;
            al = mk * -fminf(s1, s2);
            // d(-min(s1, s2))/dlp: through s1 when it is the smaller (ties: both paths agree), else through the
            // clamp, which passes the gradient only strictly inside the clip range
            const bool in_range = ratio > 1.0f - a.clip && ratio < 1.0f + a.clip;
            const float g_lp = mk * ((s1 <= s2) ? -adv * ratio : (in_range ? -adv * ratio : 0.0f));
            float g_val;
            if (a.clipped_value_loss) {
                const float dv = value - vp, vc = vp + fminf(fmaxf(dv, -a.clip), a.clip);
                const float l1 = (value - rt) * (value - rt), l2 = (vc - rt) * (vc - rt);
                vl = 0.5f * fmaxf(l1, l2) * mk;
                const bool inside = dv > -a.clip && dv < a.clip;
                g_val = mk * (l1 >= l2 ? (value - rt) : (inside ? (vc - rt) : 0.0f));
            } else { // the reference's scalar-MSE branch (ppo.py:178-182): every sample counts, mask or not
                vl = 0.5f * (rt - value) * (rt - value);
                g_val = -(rt - value) * unmask;
            }
            en = ent * mk;
            dval = a.c_value * g_val * inv_count;
#pragma unroll
            for (int k = 0; k < FA_NUM_ACTIONS; ++k)
                dlg[k] = (g_lp * ((k == act ? 1.0f : 0.0f) - p[k]) + a.c_entropy * mk * p[k] * (lpk[k] + ent)) * inv_count;
        }
        if (r < TR) {
            float *dst = sO + r * SOW;
#pragma unroll
            for (int k = 0; k < FA_NUM_ACTIONS; ++k) dst[k] = dlg[k];
            dst[8] = dval;
#pragma unroll
            for (int k = 9; k < SOW; ++k) dst[k] = 0.0f;
        }
        for (int off = 32; off > 0; off >>= 1) {
            vl += __shfl_down(vl, off); al += __shfl_down(al, off); en += __shfl_down(en, off); mk += __shfl_down(mk, off);
        }
        if (lane == 0) { mslab[FA_MSLAB_LOSS] = vl; mslab[FA_MSLAB_LOSS + 1] = al; mslab[FA_MSLAB_LOSS + 2] = en; mslab[FA_MSLAB_LOSS + 3] = mk; }
    }
    __syncthreads();

    FA_TR_TICK(4)
    // ================================ backward =========================================================
    // ---- heads --------------------------------------------------------------------------------------
    {   // dW9 = [P | V]^T dOUT: W9 is block diagonal -- only P^T dlogits (128 x 8) and V^T dvalue (128 x 1) are gradients
        // of parameters.  Eight 32 x 32 MFMA tiles, two per wave (k-blocks wave, wave + 4), the real entries to mslab.
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int kb = wave + 4 * half;
            f32x16 acc = {};
            gemm_tn((half ? B2 : B1) + wave * 32, LDA, sO, SOW, acc, lane);
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = (reg & 3) + 8 * (reg >> 2) + 4 * hh;
                if (half == 0) { if (li < 8) mslab[FA_MSLAB_W9C + (kb * 32 + row) * 8 + li] = acc[reg]; }
                else if (li == 8) mslab[FA_MSLAB_W9C + 1024 + wave * 32 + row] = acc[reg];
            }
        }
        {   // db9 = column sums of dOUT: eight 4-row partial sums per column, folded behind the next barrier
            const int col = tid & 31, part = tid >> 5;
            float sum = 0.0f;
#pragma unroll
            for (int r = 0; r < TR / 8; ++r) sum += sO[(part * (TR / 8) + r) * SOW + col];
            sO2[part * 32 + col] = sum;
        }
    }
    {   // d[P | V] = dOUT W9^T (K = 32 -> 256 columns: blocks wave, wave + 4), through the relus in place
        BHead<32> h0, h1;
        const float4 *wp0 = Tq + FA_TOFF_W9T / 4 + cbw * 4 * 64, *wp1 = Tq + FA_TOFF_W9T / 4 + (4 + cbw) * 4 * 64;
        prefetch_b<32>(wp0, lane, h0);
        prefetch_b<32>(wp1, lane, h1);
        f32x16 ap = {}, av = {};
#pragma unroll
        for (int c = 0; c < 4; ++c) {   // A = sO (row stride SOW): one 16-byte read per four MFMAs
            const float4 x = *reinterpret_cast<const float4 *>(sO + li * SOW + hh * 16 + c * 4);
            ap = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, h0.v[c].x, ap, 0, 0, 0);
            ap = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, h0.v[c].y, ap, 0, 0, 0);
            ap = __builtin_amdgcn_mfma_f32_32x32x2f32(x.z, h0.v[c].z, ap, 0, 0, 0);
            ap = __builtin_amdgcn_mfma_f32_32x32x2f32(x.w, h0.v[c].w, ap, 0, 0, 0);
            av = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, h1.v[c].x, av, 0, 0, 0);
            av = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, h1.v[c].y, av, 0, 0, 0);
            av = __builtin_amdgcn_mfma_f32_32x32x2f32(x.z, h1.v[c].z, av, 0, 0, 0);
            av = __builtin_amdgcn_mfma_f32_32x32x2f32(x.w, h1.v[c].w, av, 0, 0, 0);
        }
        __syncthreads(); // dW9 has read P and V
        if (tid < 32) {
            float sum = 0.0f;
#pragma unroll
            for (int part = 0; part < 8; ++part) sum += sO2[part * 32 + tid];
            mslab[FA_MSLAB_B9 + tid] = sum;
        }
        store_acc_relu_mask(B1 + cbw * 32, ap, lane);
        store_acc_relu_mask(B2 + cbw * 32, av, lane);
    }
    __syncthreads();
    {   // [dP | dV] -> the record of dW8 = h3^T [dP | dV] (fa_train_dw_kernel);  db8 = its column sums
        for (int k = tid; k < TR * 64; k += NTH) {
            const int r = k >> 6, c4 = k & 63;
            reinterpret_cast<float4 *>(recB + FA_RECB_DPV)[k] =
                *reinterpret_cast<const float4 *>((c4 < 32 ? B1 : B2) + r * LDA + (c4 & 31) * 4);
        }
        const float *src = (tid < 128 ? B1 : B2) + (tid & 127);
        float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
        for (int r = 0; r < TR; r += 2) { s0 += src[r * LDA]; s1 += src[(r + 1) * LDA]; }
        mslab[FA_MSLAB_B8 + tid] = s0 + s1;
    }
    {   // dh3 = [dP | dV] W8^T (K = 256: half 0 walks dP, half 1 dV) -> B3
        BHead<256> hd;
        const float4 *wp = Tq + FA_TOFF_W8T / 4 + cbw * 32 * 64;
        prefetch_b<256>(wp, lane, hd);
        f32x16 acc[1] = {};
        gemm_cb<256, 1>((hh ? B2 : B1) + li * LDA, wp, acc, lane, hd);
        store_acc_gate(B3 + cbw * 32, B0 + cbw * 32, acc[0], lane); // = dZ of round 2: through the relu of h3 (B0)
    }
    __syncthreads();

    FA_TR_TICK(5)
    // ---- the three rounds, last first: B0 = the round's output h, B3 = dL/d(output) ---------------------
    float dbu = 0.0f; // threads 0..127: their column of the update bias gradient
    for (int round = 2; round >= 0; --round) {
        float *rec = recA + round * FA_RECA_FLOATS;
        // B3 = dZ of this round (the relu gate was applied by the store that completed it).  h_in is requested now and
        // lands in B0 behind the first GEMM, which needs dZ only.
        // (order of the requests: the vector-memory counter retires in order, so the first GEMM's weights go first and
        //  h_in -- an HBM round trip that the GEMM covers -- last)
        BHead<128> ha, hm;
        const float4 *wpa = Tq + FA_TOFF_W7T / 4 + cbw * 16 * 64, *wpm = Tq + FA_TOFF_W7T / 4 + (4 + cbw) * 16 * 64;
        prefetch_b<128>(wpa, lane, ha);
        prefetch_b<128>(wpm, lane, hm);
        {   // the update bias gradient: two 16-row partial sums per column of dZ, folded behind the next barrier
            const int col = tid & 127, q = tid >> 7;
            float sum = 0.0f;
#pragma unroll
            for (int r = 0; r < TR / 2; ++r) sum += B3[(q * (TR / 2) + r) * LDA + col];
            sO2[q * 128 + col] = sum;
        }
        save_tile(B3, rec + FA_RECA_DZ);
        float4 hin[TR * 32 / NTH];
#pragma unroll
        for (int j = 0; j < TR * 32 / NTH; ++j) hin[j] = reinterpret_cast<const float4 *>(rec + FA_RECA_HIN)[tid + j * NTH];
        float4 gin[TR * 32 / NTH];
#pragma unroll
        for (int j = 0; j < TR * 32 / NTH; ++j) gin[j] = reinterpret_cast<const float4 *>(recG + round * FA_REC_PLANE)[tid + j * NTH];
        __builtin_amdgcn_sched_barrier(0);
        FA_TR_TICK(30 + 8 * (2 - round))
        {   // [dh_a | dhmix] = dZ W7^T (K = 128 -> 256 columns)
            f32x16 aa[1] = {}, am[1] = {};
            gemm_cb<128, 1>(B3 + li * LDA + hh * 64, wpa, aa, lane, ha);
            gemm_cb<128, 1>(B3 + li * LDA + hh * 64, wpm, am, lane, hm);
            FA_TR_TICK(31 + 8 * (2 - round))
            __syncthreads(); // every wave has read dZ (and the gate in B0, when the store above was the previous round's)
            store_acc<false>(B3 + cbw * 32, 0, aa[0], 0.0f, lane);
            store_acc<false>(B1 + cbw * 32, 0, am[0], 0.0f, lane);
        }
#pragma unroll
        for (int j = 0; j < TR * 32 / NTH; ++j) {
            const int k = tid + j * NTH;
            *reinterpret_cast<float4 *>(B0 + (k >> 5) * LDA + (k & 31) * 4) = hin[j];
        }
        // g -> B2 (rows beyond the tile's envs hold the g of padding rows: their dg is zero)
#pragma unroll
        for (int j = 0; j < TR * 32 / NTH; ++j) {
            const int k = tid + j * NTH;
            *reinterpret_cast<float4 *>(B2 + (k >> 5) * LDA + (k & 31) * 4) = (k >> 5) < RU ? gin[j] : float4{0, 0, 0, 0};
        }
        BHead<128> hd;
        const float4 *wp = Tq + FA_TOFF_AMT / 4 + cbw * 16 * 64;
        __syncthreads();
        FA_TR_TICK(33 + 8 * (2 - round))
        if (tid < 128) dbu += sO2[tid] + sO2[128 + tid];
        // attention backward per env: dhmix (B1), g (B2) -> dg (B2 in place), dkeys added into B3
        for (int el = wave * 4 + (lane >> 4); el < ET; el += NWV * 4)
            attend_env_bwd<128, true, MT>(B1 + (el * n) * LDA, B2 + (el * n) * LDA, B0 + (el * n) * LDA, B3 + (el * n) * LDA,
                                          sAttn[1 + round] + (el * n) * 8, n, n, q16);
        __syncthreads();
        FA_TR_TICK(34 + 8 * (2 - round))
        prefetch_b<128>(wp, lane, hd); // (requested here, not ahead of the attention: 16 registers less across it, 1-2 % faster)
        save_tile(B2, rec + FA_RECA_DG);
        {   // dh += dg A_m^T, and -- rounds 2, 1 -- through the relu of the round below (its output is this round's h_in)
            f32x16 acc[1] = {};
            gemm_cb<128, 1>(B2 + li * LDA + hh * 64, wp, acc, lane, hd);
            if (round > 0) store_acc_add_gate(B3 + cbw * 32, B0 + cbw * 32, acc[0], lane);
            else store_acc_add(B3 + cbw * 32, acc[0], lane);
        }
        __syncthreads();
        FA_TR_TICK(35 + 8 * (2 - round))
    }
    if (tid < 128) mslab[FA_MSLAB_BU + tid] = dbu;

    FA_TR_TICK(60)
    // ---- opponent stage: B0 = [h1 | e_opp], B3 = [dh1 (so far) | de_opp] --------------------------------
    encoders(nullptr, B2); // ho -> B2[:, 0:64] (opponent rows), recomputed
    __syncthreads();
    project_opp();         // g_o -> B1[:, 64:128], recomputed
    save_tile64(B3 + 64, recB + FA_RECB_DE);   // dB_o = mix_o^T de_opp (mix_o: saved by the forward)
    save_tile64(B0, recB + FA_RECB_H1);        // dA_o = h1^T dg_o
    if (wave < 2) {   // dmix_o = de_opp B_o^T -> B2[:, 64:128]
        BHead<64> hd;
        const float4 *wp = Tq + FA_TOFF_BOT / 4 + wave * 8 * 64;
        prefetch_b<64>(wp, lane, hd);
        f32x16 acc[1] = {};
        gemm_cb<64, 1>(B3 + li * LDA + 64 + hh * 32, wp, acc, lane, hd);
        store_acc<false>(B2 + 64 + wave * 32, 0, acc[0], 0.0f, lane);
    }
    __syncthreads();
    // opponent attention backward per env: dmix_o (B2[:, 64:]), g_o (B1[:, 64:]) -> dg_o in place; dho -> B1[:, 0:64]
    for (int el = wave * 4 + (lane >> 4); el < ET; el += NWV * 4)
        attend_env_bwd<64, false, MT>(B2 + (el * n) * LDA + 64, B1 + (el * n) * LDA + 64, B2 + (el * m) * LDA, B1 + (el * m) * LDA,
                                      sAttn[0] + (el * n) * 8, n, m, q16);
    __syncthreads();
    save_tile64(B1 + 64, recB + FA_RECB_DGO);
    if (wave < 2) {   // dh1 += dg_o A_o^T
        BHead<64> hd;
        const float4 *wp = Tq + FA_TOFF_AOT / 4 + wave * 8 * 64;
        prefetch_b<64>(wp, lane, hd);
        f32x16 acc2[1] = {};
        gemm_cb<64, 1>(B1 + li * LDA + 64 + hh * 32, wp, acc2, lane, hd);
        store_acc_add(B3 + wave * 32, acc2[0], lane);
    }
    __syncthreads();
    FA_TR_TICK(61)
    // ---- encoders: through the relus, then [dW ; db] = [x | 1]^T dpre as one 32 x 32 MFMA tile per 32 columns ----
    // (own side: waves 0, 1; opponent side: waves 2, 3).  [x | 1 | 0...] rows go to sO / sO2.
    for (int k = tid; k < 2 * TR * 64; k += NTH) {
        const int side = k / (TR * 64), kk = k - side * TR * 64, r = kk >> 6, c = kk & 63;
        if (side == 0) { if (!(B0[r * LDA + c] > 0.0f)) B3[r * LDA + c] = 0.0f; }
        else if (!(B2[r * LDA + c] > 0.0f)) B1[r * LDA + c] = 0.0f;
    }
    for (int k = tid; k < 2 * TR * SOW; k += NTH) {
        const int side = k / (TR * SOW), kk = k - side * TR * SOW, r = kk / SOW, c = kk - r * SOW;
        const int per = side == 0 ? n : m, first = side == 0 ? own0 : opp0, used = side == 0 ? RU : RO;
        float v = 0.0f;
        if (r < used) {
            const int el = r / per, i = r - el * per;
            v = c < FA_OBS_DIM ? sX[(el * N + first + i) * FA_OBS_DIM + c] : (c == FA_OBS_DIM ? 1.0f : 0.0f);
        }
        (side == 0 ? sO : sO2)[kk] = v;
    }
    __syncthreads();
    {
        const int side = wave >> 1, cb = wave & 1;
        f32x16 acc = {};
        gemm_tn(side == 0 ? sO : sO2, SOW, (side == 0 ? B3 : B1) + cb * 32, LDA, acc, lane);
        float *dw = mslab + (side == 0 ? FA_POFF_WE : FA_POFF_WOE), *db = mslab + (side == 0 ? FA_POFF_BE : FA_POFF_BOE);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = (reg & 3) + 8 * (reg >> 2) + 4 * hh; // = k of [x | 1]
            if (row < FA_OBS_DIM) dw[row * 64 + cb * 32 + li] = acc[reg];
            else if (row == FA_OBS_DIM) db[cb * 32 + li] = acc[reg];
        }
    }
    FA_TR_TICK(62)
}

template <bool GATHER, int MT>
__global__ __launch_bounds__(NTH, 2) void fa_train_kernel(FaTrainArgs a) { fa_train_body<GATHER, MT>(a); }
// (Rounds 2-3 had a second, register-capped build for the two teams' concurrent update chains -- it left room on a CU for
//  the other chain's small launches.  With two 4-wave workgroups per CU the capped build spills more than the sharing
//  gains: 0.229 s per update capped, 0.207 s uncapped.  FaTrainArgs::share_cu is accepted and ignored.)

// The alive-mask sum of the minibatch's own-team rows as FA_MASK_PARTS partial sums (one workgroup each; the
// train kernel's workgroups fold them in a fixed order: reproducible, and no single-workgroup latency chain)
__global__ __launch_bounds__(64) void fa_mask_part_kernel(FaTrainArgs a, float *__restrict__ part) {
    // (one wave, no LDS: a CU busy with fa_train_kernel has 1 KB of LDS left, below the allocation granule)
    const int N = a.G + a.A, n = a.team == 0 ? a.G : a.A, own0 = a.team == 0 ? 0 : a.G;
    float s = 0.0f;
    for (int b = blockIdx.x * 64 + threadIdx.x; b < a.B; b += FA_MASK_PARTS * 64) {
        const float *row = a.obs + (size_t)(a.idx ? a.idx[b] : b) * N * FA_OBS_DIM;
        for (int i = 0; i < n; ++i) s += row[(own0 + i) * FA_OBS_DIM];
    }
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// ---- clip_grad_norm_ + Adam on the flat parameter buffer -------------------------------------------------------
// ||g||^2 as FA_NORM_PARTS fp64 partial sums (scratch + 4, 16-byte aligned); the step counters advance here.
__global__ __launch_bounds__(64) void fa_sqnorm_part_kernel(const float *__restrict__ g, int n, double *__restrict__ part,
                                                            float *__restrict__ steps, int nseg) {
    double s = 0.0; // (one wave, no LDS: see fa_mask_part_kernel)
    for (int k = blockIdx.x * 64 + threadIdx.x; k < n; k += FA_NORM_PARTS * 64) s += (double)g[k] * (double)g[k];
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
    if (blockIdx.x == 0)
        for (int k = threadIdx.x; k < nseg; k += 64) steps[k] += 1.0f;
}

// coef = min(1, max_norm / (||g|| + 1e-6)) (torch.nn.utils.clip_grad_norm_), then torch.optim.Adam's update
__global__ __launch_bounds__(256) void fa_adam_kernel(float *__restrict__ p, float *__restrict__ g, float *__restrict__ m,
                                                      float *__restrict__ v, const float *__restrict__ steps,
                                                      const int32_t *__restrict__ seg, int nseg, float lr, float beta1,
                                                      float beta2, float eps, float max_norm, float *__restrict__ scratch,
                                                      const float *__restrict__ hyper) {
    if (hyper) { // (lr, beta1, beta2, eps, max_norm) read at run time: a captured graph follows an LR schedule
        lr = hyper[0]; beta1 = hyper[1]; beta2 = hyper[2]; eps = hyper[3]; max_norm = hyper[4];
    }
    double t = reinterpret_cast<const double *>(scratch + 4)[threadIdx.x & 63];
    for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d);
    const float coef = fminf(1.0f, max_norm / ((float)sqrt(t) + 1e-6f));
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k == 0) scratch[0] = coef;
    if (k >= seg[nseg]) return;
    int sg = 0;
    while (sg + 1 < nseg && k >= seg[sg + 1]) ++sg; // the parameter tensor this element belongs to: its step count
    const float st = steps[sg];
    const float bc1 = 1.0f - powf(beta1, st), bc2 = 1.0f - powf(beta2, st);
    const float gr = g[k] * coef;
    const float mk = m[k] + (gr - m[k]) * (1.0f - beta1); // exp_avg.lerp_(grad, 1 - beta1)
    const float vk = v[k] * beta2 + (1.0f - beta2) * gr * gr;
    g[k] = gr;
    m[k] = mk;
    v[k] = vk;
    p[k] -= (lr / bc1) * mk / (sqrtf(vk) / sqrtf(bc2) + eps);
}
} // namespace

int fa_train_tile_envs(int G, int A) { return TR / (G > A ? G : A); }

hipError_t fa_launch_train(const FaTrainArgs &a, hipStream_t st) {
    const int ET = fa_train_tile_envs(a.G, a.A);
    const dim3 grid(a.ntiles > 0 ? a.ntiles : (a.B + ET - 1) / ET), block(NTH);
    const int big = a.G > a.A ? a.G : a.A;
#define FA_TRAIN_LAUNCH(MT)                                                                                   \
    do {                                                                                                      \
        if (a.idx) hipLaunchKernelGGL((fa_train_kernel<true, MT>), grid, block, 0, st, a);                    \
        else hipLaunchKernelGGL((fa_train_kernel<false, MT>), grid, block, 0, st, a);                         \
    } while (0)
    if (big <= 4) FA_TRAIN_LAUNCH(4);
    else if (big <= 6) FA_TRAIN_LAUNCH(6);
    else FA_TRAIN_LAUNCH(FA_POLICY_MAX_TEAM);
#undef FA_TRAIN_LAUNCH
    return hipGetLastError();
}

hipError_t fa_launch_mask_parts(const FaTrainArgs &a, float *part, hipStream_t st) {
    hipLaunchKernelGGL(fa_mask_part_kernel, dim3(FA_MASK_PARTS), dim3(64), 0, st, a, part);
    return hipGetLastError();
}

hipError_t fa_launch_adam(float *p, float *g, float *m, float *v, float *steps, const int32_t *seg, int nseg, int n,
                          float lr, float beta1, float beta2, float eps, float max_norm, float *scratch, const float *hyper,
                          hipStream_t st) {
    hipLaunchKernelGGL(fa_sqnorm_part_kernel, dim3(FA_NORM_PARTS), dim3(64), 0, st, g, n, reinterpret_cast<double *>(scratch + 4),
                       steps, nseg);
    hipLaunchKernelGGL(fa_adam_kernel, dim3((n + 255) / 256), dim3(256), 0, st, p, g, m, v, steps, seg, nseg, lr, beta1, beta2,
                       eps, max_norm, scratch, hyper);
    return hipGetLastError();
}
