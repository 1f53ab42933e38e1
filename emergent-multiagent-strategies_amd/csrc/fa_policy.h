// fa_policy.h -- internal declarations of the fused MPNN policy kernel (fa_policy.hip) and the layout of
// its packed weight buffer (filled by emergent-multiagent-strategies_amd/mpnn_pack.py; documented for
// other hosts in include/fortattack.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fortattack.h"

#define FA_POLICY_ROWS 96     // (env, agent) rows per workgroup tile = three 32-row MFMA blocks
#define FA_POLICY_MAX_TEAM 8  // agents per team the attention loops are unrolled for

// Offsets (in floats) into one team's packed weight buffer, hidden_dim = 128 (mpnn.py:27-74).
// "packed (K x C)" = MFMA B-operand order: float4 index (cb * K/8 + t4) * 64 + lane holds
// W[k = (lane >> 5) * K/2 + 4 * t4 + q][col = 32 * cb + (lane & 31)], q = 0..3.
#define FA_POFF_WE 0        // encoder weight^T, plain [6][64]
#define FA_POFF_BE 384      // encoder bias [64]
#define FA_POFF_WOE 448     // oppEncoder weight^T, plain [6][64]
#define FA_POFF_BOE 832     // oppEncoder bias [64]
#define FA_POFF_AO 896      // packed (64 x 64):  norm_o * oppAttn.W_key W_query^T
#define FA_POFF_BO 4992     // packed (64 x 64):  oppAttn.W_val W_out
#define FA_POFF_AM 9088     // packed (128 x 128): norm * messages.W_query W_key^T
#define FA_POFF_W7 25472    // packed (256 x 128): [update.weight[:, :128]^T ; messages.W_val W_out update.weight[:, 128:]^T]
#define FA_POFF_BU 58240    // update bias [128]
#define FA_POFF_W8 58368    // packed (128 x 256): [policy_head.0.weight^T | value_head.0.weight^T]
#define FA_POFF_B8 91136    // [policy_head.0.bias | value_head.0.bias] [256]
#define FA_POFF_W9 91392    // packed (256 x 32): rows 0..127 x cols 0..7 = dist.linear.weight^T, rows 128..255 x col 8 = value_head.2.weight^T
#define FA_POFF_B9 99584    // [dist.linear.bias (8) | value_head.2.bias (1) | 0 ...] [32]
#define FA_POLICY_PLAIN_FLOATS 99616  // end of the float32 sections above (= the layout of the plain / gradient buffers)
// The same six dense matrices once more, for the bf16 matrix cores: every float32 weight split EXACTLY into three bf16 terms
// (hi = round-to-nearest at 8 significant bits, mid = the remainder rounded likewise, lo = the rest; fa_mfma.h gemm_cb3),
// in the B-operand order of v_mfma_f32_32x32x16_bf16: 16-byte index ((cb * K/16 + s) * 3 + term) * 64 + lane holds the eight
// bf16 W[k = (lane >> 5) * K/2 + 8 s + j][col = 32 cb + (lane & 31)], j = 0..7, of term 0 / 1 / 2 = hi / mid / lo.
// 1.5 floats of buffer per weight.  Written by fa_pack_weights (device) / mpnn_pack.pack_policy (host) next to the float32 form.
#define FA_POFF3_AO 99616   // (64 x 64)
#define FA_POFF3_BO 105760  // (64 x 64)
#define FA_POFF3_AM 111904  // (128 x 128)
#define FA_POFF3_W7 136480  // (256 x 128)
#define FA_POFF3_W8 185632  // (128 x 256)
#define FA_POFF3_W9 234784  // (256 x 32)
#define FA_POLICY_WEIGHT_FLOATS 247072

struct FaPolicyArgs {
    const float *obs;      // (E, N, 6) observation row
    const float *w[2];     // packed weights: guards' policy, attackers' policy
    float *value;          // (E, N) or null
    int64_t *action;       // (E, N)
    float *logp;           // (E, N)
    const int64_t *counter; // device scalar mixed into the sampling stream (bumped once per rollout) or null
    uint64_t seed;
    int64_t env_offset;
    int32_t E, G, A, step, deterministic, value_only;
    // ensemble of frozen attacker strategies (learner.py:119-140): envs grouped by strategy
    const float *pool;             // pool_size packed weight buffers back to back, or null
    const int32_t *env_list;       // [tiles][ET] env index of every tile slot (-1 = empty), from fa_group_envs_kernel
    const int32_t *tile_strategy;  // [tiles] strategy of the tile's envs (-1 = unused tile)
    int32_t tiles;                 // grid size in tiles when env_list is set
};
#define FA_POLICY_MAX_POOL 64 // strategies in an ensemble

hipError_t fa_launch_policy(const FaPolicyArgs &a, hipStream_t st);
// envs -> tiles of equal strategy: env_list / tile_strategy for a launch with `pool`
hipError_t fa_launch_group_envs(const int32_t *env_strategy, int E, int pool_size, int G, int A, int32_t *env_list,
                                int32_t *tile_strategy, int tiles_max, hipStream_t st);
int fa_policy_tile_envs(int E, int G, int A);
