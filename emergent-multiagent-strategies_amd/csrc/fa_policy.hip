// fa_policy.hip -- the MPNN actor-critic forward (reference mpnn.py:117-192: _fwd, act, get_value) of a
// whole batch of envs as ONE fused kernel for CDNA4: encoders, opponent attention, the K = 3 message
// passing rounds, policy / value heads, log-softmax and categorical sampling, written straight into
// the rollout rows value_preds[s] / actions[s] / action_log_probs[s].
//
// Why a kernel of its own: as PyTorch ops one forward of both teams at 4096 envs is ~80 launches and
// 0.9 ms (18 small fp32 GEMMs at ~50 TFLOP/s plus 60 elementwise / reduction kernels on (24576 x 128)
// activations); the step kernel next to it takes 7 us.  Here the activations of a tile of envs never
// leave the CU: 96 rows (= 32 envs x 3 agents at 3v3) live in two LDS buffers, every dense layer is a
// chain of v_mfma_f32_32x32x2_f32 (exact fp32, 157 TFLOP/s peak) whose B operand -- the weights,
// pre-packed in lane order by the host -- streams from L2 straight into registers, and the attention
// between the agents of an env is done on the LDS-resident rows.
//
// Algebra.  The network is the reference's (same parameters); three products of consecutive linear maps
// are folded on the host (mpnn_pack.py), which is exact in real arithmetic and changes fp32 rounding
// only (~1e-6 relative, the size of a GEMM's own summation-order noise):
//   scores_ij = norm * (h_i Wq)(h_j Wk)^T           = (h_i A) . h_j          A  = norm * Wq Wk^T
//   msg_i     = (sum_j a_ij (h_j Wv)) Wout           = (sum_j a_ij h_j) Wv Wout
//   update    = relu([h | msg] Wu^T + b)             = relu([h | hmix] [Wu1 ; Wv Wout Wu2] + b)
// so a round is: g = h A (128x128), attention mix on LDS rows, h' = relu([h | hmix] W7 + b) (256x128)
// -- 48k MACs per row instead of 98k -- and needs two LDS buffers instead of four.  The opponent
// attention (mpnn.py:372-443) folds the same way: A_o = norm * Wkey Wquery^T, B_o = Wval Wout.
//
// Tiling.  Workgroup = one tile of ET = 96 / max(n, m) envs of ONE team (blockIdx.y), eight waves (two per SIMD; four
// for the 64-row tile of small batches).  Rows = (env, own agent).  A 32x32x2 MFMA takes A[i = lane & 31][k = lane >> 5]: the K axis is permuted so that
// lane half hh covers k in [hh*K/2, (hh+1)*K/2) -- then a lane's A values of four consecutive MFMAs are
// one contiguous 16-byte LDS read, and for the K = 256 update layer half 0 reads h and half 1 reads hmix
// from their own buffers with no concatenation.  The packed B operand uses the same permutation:
// float4 index (cb * K/8 + t4) * 64 + lane holds W[k = hh*K/2 + 4*t4 + q][col = 32*cb + (lane & 31)],
// q = 0..3.  A wave owns one 32-column block of the layer's output -- waves 0..3 for row blocks 0 and 1, waves 4..7
// for row block 2 (see FA_POLICY_SPLIT) -- and its B registers are reused for each of its row blocks; LDS rows are
// padded to 132 floats (bank-conflict-free 16-byte column reads).
//
// Sampling: Gumbel-max over the 8 logits with Philox4x32-10 uniforms keyed by (seed; rollout counter,
// rollout step, global env index, agent) -- a draw from softmax(logits), i.e. the distribution of
// FixedCategorical.sample (rlcore/distributions.py:12-13); the stream is the engine's own (torch's
// generator cannot be reproduced from inside a kernel).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "fa_policy.h"
#include "fa_mfma.h"
#define FA_PROBE_POLICY_TU
#include "fa_probe.h"

#ifndef FA_POLICY_WAVES
#define FA_POLICY_WAVES 8 // waves of the 96-row tile's workgroup (4: one per SIMD, the round-2 shape)
#endif
// The dense layers' GEMM: 1 = the bf16 matrix cores with the three-way operand split (fa_mfma.h gemm_cb3: fp32-class accuracy at
// 6/16 of the fp32 MFMA time; weights pre-split in the pack, FA_POFF3_*), 0 = v_mfma_f32_32x32x2_f32 (the exact fp32 fmaf chain:
// rounds 2-5, kept as the definition the split form is tested against -- tools/build_variant.py ... -DFA_POLICY_X3=0)
#ifndef FA_POLICY_X3
#define FA_POLICY_X3 1
#endif
namespace {
#if FA_POLICY_X3
typedef fa_u32x4 pw_t;                          // one 16-byte step of packed weights
template <int K> using PBHead = BHead3<K>;
template <int K> __device__ __forceinline__ void p_prefetch(const pw_t *__restrict__ wp, int lane, PBHead<K> &h) { prefetch_b3<K>(wp, lane, h); }
template <int K, int NRB>
__device__ __forceinline__ void p_gemm(const float *arow, const pw_t *__restrict__ wp, f32x16 (&acc)[NRB], int lane, const PBHead<K> &h) {
    gemm_cb3<K, NRB>(arow, wp, acc, lane, h);
}
// the column block cb of the packed (K x C) matrix at float offset OFF3
#define FA_PW(OFF, OFF3, K, cb) (reinterpret_cast<const pw_t *>(W + (OFF3)) + (size_t)(cb) * ((K) / 16) * 3 * 64)
#else
typedef float4 pw_t;
template <int K> using PBHead = BHead<K>;
template <int K> __device__ __forceinline__ void p_prefetch(const pw_t *__restrict__ wp, int lane, PBHead<K> &h) { prefetch_b<K>(wp, lane, h); }
template <int K, int NRB>
__device__ __forceinline__ void p_gemm(const float *arow, const pw_t *__restrict__ wp, f32x16 (&acc)[NRB], int lane, const PBHead<K> &h) {
    gemm_cb<K, NRB>(arow, wp, acc, lane, h);
}
#define FA_PW(OFF, OFF3, K, cb) (reinterpret_cast<const pw_t *>(W + (OFF)) + (size_t)(cb) * ((K) / 8) * 64)
#endif
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// NRB = 32-row blocks per tile: 3 (96 rows, one workgroup per CU) or 2 (64 rows, 75 KB of LDS: two workgroups
// per CU; small batches).  NW = waves per workgroup.  The 96-row tile runs EIGHT waves, two per SIMD: waves 0..3
// own column block `wave` of row blocks 0 and 1, waves 4..7 the same column block of row block 2 -- a SIMD's MFMA
// work is what one wave did before (three 32 x 32 tiles per layer), but while one of its waves stores accumulators,
// waits at a barrier or for an LDS read, the other's MFMA chain runs, and the phases no MFMA runs in (encoders,
// attention, sampling: latency-bound LDS / DPP chains) have twice the waves to hide their latencies with.
// A barrier INSIDE the wave split below: the two branches are taken by whole waves (`wave` is wave-uniform), so each wave
// executes exactly one of the two barrier instructions.  The hardware barrier counts waves, whichever instruction they
// arrive at; the HIP-level __syncthreads() does not promise that, so these are spelled as what they are -- the LDS
// writes / reads of the wave drained, then s_barrier -- opaque to the compiler (it cannot duplicate, merge or move
// memory operations across them).  Vector-memory requests (the weight prefetch) stay in flight across it.
#define FA_WAVES_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define FA_POLICY_SPLIT(...)                                                          \
    if constexpr (NW == 8) {                                                          \
        if (wave < 4) { constexpr int R0 = 0, NR = 2; __VA_ARGS__ }                   \
        else { constexpr int R0 = 2, NR = 1; __VA_ARGS__ }                            \
    } else if constexpr (NW == 12) {                                                  \
        const int R0 = wave >> 2; constexpr int NR = 1; __VA_ARGS__                   \
    } else { constexpr int R0 = 0, NR = NRB; __VA_ARGS__ }
template <int NRB, int NW>
__global__ __launch_bounds__(NW * 64, NRB == 2 ? 2 : 1) void fa_policy_kernel(FaPolicyArgs a) {
    static_assert((NRB == 3 && (NW == 8 || NW == 12)) || NW == 4, "eight / twelve waves: the 96-row tile");
    constexpr int PR = 32 * NRB; // rows (env, agent) per tile
    constexpr int NT = NW * 64;
    __shared__ __attribute__((aligned(16))) float sH[PR * LDA]; // own hidden state h (128 wide)
    __shared__ __attribute__((aligned(16))) float sG[PR * LDA]; // g / hmix; opponent stage scratch
    __shared__ float sX[PR * 2 * FA_OBS_DIM];                    // observations of the tile's envs (all agents)
    __shared__ float sO[PR * 16];                                // logits (8) + value per row
    __shared__ float sNz[NW == 8 ? PR * FA_NUM_ACTIONS : 1];     // eight waves: log(-log u) per (row, action), see below

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int team = blockIdx.y;
    const int N = a.G + a.A;
    const int n = team == 0 ? a.G : a.A, m = N - n;   // own / opponent team size
    const int own0 = team == 0 ? 0 : a.G, opp0 = team == 0 ? a.G : 0;
    const int ET = PR / (n > m ? n : m);               // envs per tile
    // r / n, r / m, k / (6 N) for row indices below 2048 as a multiply and a shift (exact for divisors up to 16 -- team
    // sizes -- and checked for 6 N <= 96 on the host side of this file); a run-time integer division is ~35 instructions
    const unsigned inv_n = 65536u / (unsigned)n + 1u, inv_m = 65536u / (unsigned)m + 1u;
    auto div_n = [&](int r) { return (int)(((unsigned)r * inv_n) >> 16); };
    auto div_m = [&](int r) { return (int)(((unsigned)r * inv_m) >> 16); };
    // which envs: a contiguous range, or -- ensemble of attacker strategies -- the tile's slots of the
    // strategy-sorted env list (every env of a tile then shares one set of attacker weights)
    __shared__ int sE[PR];
    const float *W = a.w[team];
    if (a.env_list) {
        const int strat = a.tile_strategy[blockIdx.x];
        if (strat < 0) return;
        if (team == 1) W = a.pool + (size_t)strat * FA_POLICY_WEIGHT_FLOATS;
        if (tid < ET) sE[tid] = a.env_list[blockIdx.x * ET + tid];
    } else {
        const int e0 = blockIdx.x * ET;
        if (e0 >= a.E) return;
        if (tid < ET) sE[tid] = e0 + tid < a.E ? e0 + tid : -1;
    }
    // (the encoder weights of this thread's column: requested ahead of the observation round trip)
    float we[FA_OBS_DIM], wo[FA_OBS_DIM];
#pragma unroll
    for (int k = 0; k < FA_OBS_DIM; ++k) {
        we[k] = W[FA_POFF_WE + k * 64 + (tid & 63)];
        wo[k] = W[FA_POFF_WOE + k * 64 + (tid & 63)];
    }
    const float be = W[FA_POFF_BE + (tid & 63)], bo = W[FA_POFF_BOE + (tid & 63)];
    __syncthreads();

    // ---- observations of the tile's envs (N * 6 contiguous floats per env) -------------------------------
    for (int k = tid; k < ET * N * FA_OBS_DIM; k += NT) {
        const int el = k / (N * FA_OBS_DIM), e = sE[el];
        sX[k] = e >= 0 ? a.obs[(size_t)e * N * FA_OBS_DIM + (k - el * N * FA_OBS_DIM)] : 0.0f;
    }
    __syncthreads();

    FA_PL_TICK(0)
    const int li = lane & 31, hh = lane >> 5;
    const int cbw = wave & 3;                 // this wave's column block of the 128-wide layers
    // the 64-wide layers of the opponent stage have 2 x NRB output tiles: column block ocb, row block(s) from orb
    const bool opp_two = NW == 4 && NRB == 3 && wave < 2;                 // four waves: waves 0, 1 take two row blocks
    const bool opp_on = NW == 4 || wave < 2 * NRB;
    const int ocb = wave & 1, orb = NW >= 8 ? (wave >> 1) : (NRB == 3 ? 2 : (wave >> 1));
    const pw_t *wp_ao = FA_PW(FA_POFF_AO, FA_POFF3_AO, 64, ocb);
    const pw_t *wp_bo = FA_PW(FA_POFF_BO, FA_POFF3_BO, 64, ocb);
    const pw_t *wp_am = FA_PW(FA_POFF_AM, FA_POFF3_AM, 128, cbw);
    const pw_t *wp_w7 = FA_PW(FA_POFF_W7, FA_POFF3_W7, 256, cbw);
    const pw_t *wp_w8p = FA_PW(FA_POFF_W8, FA_POFF3_W8, 128, cbw);
    const pw_t *wp_w8v = FA_PW(FA_POFF_W8, FA_POFF3_W8, 128, 4 + cbw);
    const pw_t *wp_w9 = FA_PW(FA_POFF_W9, FA_POFF3_W9, 256, 0);
    PBHead<64> hd_o;
    if (opp_on) p_prefetch<64>(wp_ao, lane, hd_o);
    // ---- encoders (mpnn.py:37-38): h1 = relu(x We + be) -> sH[:, 0:64] (own rows), ho -> sG[:, 0:64] (opp rows)
    {
        const int col = tid & 63, grp = tid >> 6;
        for (int r = grp; r < PR; r += NW) {
            float vo = 0.0f, vp = 0.0f;
            if (r < ET * n) {
                const int el = div_n(r), i = r - el * n;
                const float *x = sX + (el * N + own0 + i) * FA_OBS_DIM;
                vo = be;
#pragma unroll
                for (int k = 0; k < FA_OBS_DIM; ++k) vo = fmaf(x[k], we[k], vo);
                vo = fmaxf(vo, 0.0f);
            }
            if (r < ET * m) {
                const int el = div_m(r), j = r - el * m;
                const float *x = sX + (el * N + opp0 + j) * FA_OBS_DIM;
                vp = bo;
#pragma unroll
                for (int k = 0; k < FA_OBS_DIM; ++k) vp = fmaf(x[k], wo[k], vp);
                vp = fmaxf(vp, 0.0f);
            }
            sH[r * LDA + col] = vo;
            sG[r * LDA + col] = vp;
        }
    }
    __syncthreads();
    FA_PL_TICK(1)

    // ---- opponent attention (mpnn.py:372-443): g_o = h1 A_o -> sG[:, 64:128] --------------------------------
    if (opp_two) {
        f32x16 acc[2] = {};
        p_gemm<64, 2>(sH + li * LDA + hh * 32, wp_ao, acc, lane, hd_o);
        p_prefetch<64>(wp_bo, lane, hd_o);
        store_acc<false>(sG + 64 + ocb * 32, 0, acc[0], 0.0f, lane);
        store_acc<false>(sG + 64 + ocb * 32, 1, acc[1], 0.0f, lane);
    } else if (opp_on) {
        f32x16 acc[1] = {};
        p_gemm<64, 1>(sH + (orb * 32 + li) * LDA + hh * 32, wp_ao, acc, lane, hd_o);
        p_prefetch<64>(wp_bo, lane, hd_o);
        store_acc<false>(sG + 64 + ocb * 32, orb, acc[0], 0.0f, lane);
    }
    __syncthreads();
    FA_PL_TICK(2)
    // scores against the env's opponents, softmax, mix of the opponents' encodings -> sG[r][64:128]
    // (a wave's rows are all computed before any is stored: the loads of one row overlap the DPP / exp chains of another)
    constexpr int AIT = PR / (NW * 4); // attention rows per 16-lane sub-group
    const int q16 = lane & 15, arow0 = wave * 4 + (lane >> 4), RU = ET * n;
    const bool small_teams = (n > m ? n : m) <= 4, mid_teams = (n > m ? n : m) <= 6; // keys held in registers: 4, 6 or 8
    auto opp_attention = [&](auto MTc) __attribute__((always_inline)) {
        constexpr int MT = decltype(MTc)::value;
        float ov[AIT][4];
#pragma unroll
        for (int k = 0; k < AIT; ++k) {
            const int r = arow0 + k * NW * 4, rr = r < RU ? r : RU - 1;
            attend_row_regs<64, MT>(sG + rr * LDA + 64, sG + (div_n(rr) * m) * LDA, m, -1, q16, ov[k]);
        }
#pragma unroll
        for (int k = 0; k < AIT; ++k) {
            const int r = arow0 + k * NW * 4;
            if (r < RU) store_row_regs<64>(sG + r * LDA + 64, q16, ov[k]);
        }
    };
    if (small_teams) opp_attention(std::integral_constant<int, 4>{});
    else if (mid_teams) opp_attention(std::integral_constant<int, 6>{});
    else opp_attention(std::integral_constant<int, FA_POLICY_MAX_TEAM>{});
    __syncthreads();
    FA_PL_TICK(3)
    // e_opp = hmix_o B_o -> sH[:, 64:128]   (h = [h1 | e_opp], mpnn.py:143)
    PBHead<128> hd_m;
    if (opp_two) {
        f32x16 acc[2] = {};
        p_gemm<64, 2>(sG + li * LDA + 64 + hh * 32, wp_bo, acc, lane, hd_o);
        p_prefetch<128>(wp_am, lane, hd_m);
        store_acc<false>(sH + 64 + ocb * 32, 0, acc[0], 0.0f, lane);
        store_acc<false>(sH + 64 + ocb * 32, 1, acc[1], 0.0f, lane);
    } else if (opp_on) {
        f32x16 acc[1] = {};
        p_gemm<64, 1>(sG + (orb * 32 + li) * LDA + 64 + hh * 32, wp_bo, acc, lane, hd_o);
        p_prefetch<128>(wp_am, lane, hd_m);
        store_acc<false>(sH + 64 + ocb * 32, orb, acc[0], 0.0f, lane);
    } else {
        p_prefetch<128>(wp_am, lane, hd_m);
    }
    __syncthreads();

    // ---- K = 3 rounds of message passing with shared weights (mpnn.py:155-157) ------------------------------
    auto team_attention = [&](auto MTc) __attribute__((always_inline)) {
        constexpr int MT = decltype(MTc)::value;
        float ov[AIT][8];
#pragma unroll
        for (int k = 0; k < AIT; ++k) {
            const int r = arow0 + k * NW * 4, rr = r < RU ? r : RU - 1, el = div_n(rr);
            attend_row_regs<128, MT>(sG + rr * LDA, sH + (el * n) * LDA, n, rr - el * n, q16, ov[k]);
        }
#pragma unroll
        for (int k = 0; k < AIT; ++k) {
            const int r = arow0 + k * NW * 4;
            if (r < RU) store_row_regs<128>(sG + r * LDA, q16, ov[k]);
        }
    };
    PBHead<256> hd_u;
    for (int round = 0; round < 3; ++round) {
        FA_PL_TICK(4 + round * 8)
        FA_POLICY_SPLIT({   // g = h A -> sG
            f32x16 acc[NR] = {};
            p_gemm<128, NR>(sH + (R0 * 32 + li) * LDA + hh * 64, wp_am, acc, lane, hd_m);
            p_prefetch<256>(wp_w7, lane, hd_u);
FA_PL_TICK(5 + round * 8)
_Pragma("unroll")
            for (int rb = 0; rb < NR; ++rb) store_acc<false>(sG + cbw * 32, R0 + rb, acc[rb], 0.0f, lane);
        })
        FA_PL_TICK(6 + round * 8)
        __syncthreads();
        FA_PL_TICK(7 + round * 8)
        // team attention, self excluded (mpnn.py:297-298): hmix -> sG rows
        if (small_teams) team_attention(std::integral_constant<int, 4>{});
        else if (mid_teams) team_attention(std::integral_constant<int, 6>{});
        else team_attention(std::integral_constant<int, FA_POLICY_MAX_TEAM>{});
        FA_PL_TICK(8 + round * 8)
        __syncthreads();
        FA_PL_TICK(9 + round * 8)
        FA_POLICY_SPLIT({   // h' = relu([h | hmix] W7 + bu): lane half 0 walks h, half 1 walks hmix
            f32x16 acc[NR] = {};
            p_gemm<256, NR>((hh ? sG : sH) + (R0 * 32 + li) * LDA, wp_w7, acc, lane, hd_u);
            p_prefetch<128>(round < 2 ? wp_am : wp_w8p, lane, hd_m); // next: the following round's g, or the policy head
            const float bias = W[FA_POFF_BU + cbw * 32 + li];
            FA_PL_TICK(10 + round * 8)
            FA_WAVES_BARRIER(); // every wave has read the old h
            FA_PL_TICK(11 + round * 8)
_Pragma("unroll")
            for (int rb = 0; rb < NR; ++rb) store_acc<true>(sH + cbw * 32, R0 + rb, acc[rb], bias, lane);
        })
        __syncthreads();
    }

    // ---- heads: [p | v] = relu(h [Wp0 | Wv0] + b) (mpnn.py:66-72), p -> sG, v -> sH ------------------------
    FA_PL_TICK(28)
    // (p goes to sG straight after its GEMM: hmix is dead since the barrier behind the last W7 layer; only v -> sH
    // has to wait until every wave has read h.  One accumulator set live at a time.)
    FA_POLICY_SPLIT({
        PBHead<128> hd_v;
        p_prefetch<128>(wp_w8v, lane, hd_v);
        {
            f32x16 accp[NR] = {};
            p_gemm<128, NR>(sH + (R0 * 32 + li) * LDA + hh * 64, wp_w8p, accp, lane, hd_m);
            const float bp = W[FA_POFF_B8 + cbw * 32 + li];
_Pragma("unroll")
            for (int rb = 0; rb < NR; ++rb) store_acc<true>(sG + cbw * 32, R0 + rb, accp[rb], bp, lane);
        }
        f32x16 accv[NR] = {};
        p_gemm<128, NR>(sH + (R0 * 32 + li) * LDA + hh * 64, wp_w8v, accv, lane, hd_v);
        if (wave < NRB) p_prefetch<256>(wp_w9, lane, hd_u);
        const float bv = W[FA_POFF_B8 + 128 + cbw * 32 + li];
        FA_WAVES_BARRIER(); // every wave has read h
_Pragma("unroll")
        for (int rb = 0; rb < NR; ++rb) store_acc<true>(sH + cbw * 32, R0 + rb, accv[rb], bv, lane);
    })
    __syncthreads();
    FA_PL_TICK(29)
    // log(-log u) of a row's eight Gumbel-max draws: Philox4x32-10 keyed by (seed; counter, step, global env, agent)
    auto gumbel_row = [&](int row, float (&g)[FA_NUM_ACTIONS]) __attribute__((always_inline)) {
        const int el = div_n(row), i = row - el * n;
        const uint64_t ge = (uint64_t)(a.env_offset + sE[el]);
        const uint32_t ctr = a.counter ? (uint32_t)a.counter[0] : 0u;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            uint32_t c[4] = {(uint32_t)ge, (uint32_t)(ge >> 32) ^ ((uint32_t)(own0 + i) << 16) ^ ((uint32_t)half << 31),
                             (uint32_t)a.step, ctr};
            philox4x32_10(c, (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float u = ((float)(c[k] >> 8) + 0.5f) * (1.0f / 16777216.0f); // (0, 1)
                g[half * 4 + k] = logf(-logf(u));
            }
        }
    };
    // logits (8) | value (1) = [p | v] W9 + b9, W9 block diagonal in a 32-column block: wave = row block.  Eight waves:
    // the draws do not depend on the logits -- two of the waves without a row block make them meanwhile
    if (NW == 8 && wave >= 4 && wave < 6 && !a.deterministic && !a.value_only) {
        const int row = tid - 256;
        if (row < RU && sE[div_n(row)] >= 0) {
            float g[FA_NUM_ACTIONS];
            gumbel_row(row, g);
#pragma unroll
            for (int k = 0; k < FA_NUM_ACTIONS; ++k) sNz[row * FA_NUM_ACTIONS + k] = g[k];
        }
    }
    if (wave < NRB) {
        f32x16 acc[1] = {};
        p_gemm<256, 1>((hh ? sH : sG) + (wave * 32 + li) * LDA, wp_w9, acc, lane, hd_u);
        if (li < 16) {
            const float bias = W[FA_POFF_B9 + li];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = wave * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
                sO[row * 16 + li] = acc[0][reg] + bias;
            }
        }
    }
    __syncthreads();
    FA_PL_TICK(30)

    // ---- value, log-softmax, sample, log-prob of the sample -> rollout rows ----------------------------------
    const int el_out = tid < PR ? div_n(tid) : 0;
    if (tid < ET * n && sE[el_out] >= 0) {
        const int el = el_out, i = tid - el * n;
        const int e = sE[el];
        const size_t o = (size_t)e * N + own0 + i;
        const float *lo = sO + tid * 16;
        if (a.value) a.value[o] = lo[8];
        if (!a.value_only) {
            float lg[FA_NUM_ACTIONS], mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < FA_NUM_ACTIONS; ++k) { lg[k] = lo[k]; mx = fmaxf(mx, lg[k]); }
            float se = 0.0f;
#pragma unroll
            for (int k = 0; k < FA_NUM_ACTIONS; ++k) se += expf(lg[k] - mx);
            const float lse = mx + logf(se);
            int act = 0;
            if (a.deterministic) { // FixedCategorical.mode (distributions.py:16-17)
                float best = lg[0];
#pragma unroll
                for (int k = 1; k < FA_NUM_ACTIONS; ++k)
                    if (lg[k] > best) { best = lg[k]; act = k; }
            } else {
                float g[FA_NUM_ACTIONS];
                if constexpr (NW == 8) {
#pragma unroll
                    for (int k = 0; k < FA_NUM_ACTIONS; ++k) g[k] = sNz[tid * FA_NUM_ACTIONS + k];
                } else {
                    gumbel_row(tid, g);
                }
                float best = -INFINITY;
#pragma unroll
                for (int k = 0; k < FA_NUM_ACTIONS; ++k) {
                    const float z = lg[k] - g[k]; // Gumbel-max
                    if (z > best) { best = z; act = k; }
                }
            }
            float la = lg[0];
#pragma unroll
            for (int k = 1; k < FA_NUM_ACTIONS; ++k) la = (k == act) ? lg[k] : la;
            a.action[o] = (int64_t)act;
            a.logp[o] = la - lse;
        }
    }
    FA_PL_TICK(31)
}
#undef FA_POLICY_SPLIT
// Ensemble: sort the envs into tiles of equal strategy.  One workgroup: histogram, tile ranges per
// strategy, scatter (the order inside a strategy's tiles is arbitrary -- every output is per env and the
// sampling key holds the env index, so results do not depend on it).
__global__ __launch_bounds__(1024) void fa_group_envs_kernel(const int32_t *__restrict__ strat, int E, int K, int ET, int tiles_max,
                                                            int32_t *__restrict__ env_list, int32_t *__restrict__ tile_strategy) {
    __shared__ int cnt[FA_POLICY_MAX_POOL], start[FA_POLICY_MAX_POOL], cur[FA_POLICY_MAX_POOL];
    const int tid = threadIdx.x;
    if (tid < FA_POLICY_MAX_POOL) { cnt[tid] = 0; cur[tid] = 0; }
    __syncthreads();
    auto clampk = [&](int k) { return k < 0 ? 0 : (k >= K ? K - 1 : k); };
    for (int e = tid; e < E; e += blockDim.x) atomicAdd(&cnt[clampk(strat[e])], 1);
    for (int s = tid; s < tiles_max * ET; s += blockDim.x) env_list[s] = -1;
    __syncthreads();
    if (tid == 0) {
        int t = 0;
        for (int k = 0; k < K; ++k) {
            start[k] = t * ET;
            const int nt = (cnt[k] + ET - 1) / ET;
            for (int j = 0; j < nt && t + j < tiles_max; ++j) tile_strategy[t + j] = k;
            t += nt;
        }
        for (; t < tiles_max; ++t) tile_strategy[t] = -1;
    }
    __syncthreads();
    for (int e = tid; e < E; e += blockDim.x) {
        const int k = clampk(strat[e]);
        env_list[start[k] + atomicAdd(&cur[k], 1)] = e;
    }
}
} // namespace

// 96-row tiles unless they would leave a quarter of the CUs without a workgroup (small batches): then 64-row
// tiles spread the same rows over more CUs.  (At 4096 envs the 96-row tile -- exactly one workgroup per CU at
// 3v3 -- is the faster one: 0.127 vs 0.140 ms per env-step; two 64-row workgroups per CU did not overlap
// their phases enough to pay for reusing the weights twice instead of three times.)
static int policy_rows(int E, int G, int A) {
    const int n_max = G > A ? G : A, et96 = FA_POLICY_ROWS / n_max;
    const int wgs96 = 2 * ((E + et96 - 1) / et96);
    return wgs96 < 192 ? 64 : FA_POLICY_ROWS;
}
int fa_policy_tile_envs(int E, int G, int A) { return policy_rows(E, G, A) / (G > A ? G : A); }

hipError_t fa_launch_group_envs(const int32_t *env_strategy, int E, int pool_size, int G, int A, int32_t *env_list,
                                int32_t *tile_strategy, int tiles_max, hipStream_t st) {
    hipLaunchKernelGGL(fa_group_envs_kernel, dim3(1), dim3(1024), 0, st, env_strategy, E, pool_size, fa_policy_tile_envs(E, G, A),
                       tiles_max, env_list, tile_strategy);
    return hipGetLastError();
}

hipError_t fa_launch_policy(const FaPolicyArgs &a, hipStream_t st) {
    const int ET = fa_policy_tile_envs(a.E, a.G, a.A);
    const int tiles = a.env_list ? a.tiles : (a.E + ET - 1) / ET;
    if (policy_rows(a.E, a.G, a.A) == 64) hipLaunchKernelGGL((fa_policy_kernel<2, 4>), dim3(tiles, 2), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((fa_policy_kernel<3, FA_POLICY_WAVES>), dim3(tiles, 2), dim3(FA_POLICY_WAVES * 64), 0, st, a);
    return hipGetLastError();
}
