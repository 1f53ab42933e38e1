// fa_train.h -- internal declarations of the fused PPO minibatch kernel (fa_train.hip): the MPNN forward, the
// alive-masked PPO losses and the complete backward pass of one team's minibatch in one launch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fa_policy.h"
#include "fortattack.h"

// the records' stores (tile kernel) and loads (weight-gradient GEMM) carry the non-temporal hint: each is written once and
// read once by another kernel (0: plain accesses -- the A/B of the chunked hand-off, profiles/r06_experiments/)
#ifndef FA_REC_NT
#define FA_REC_NT 1
#endif
#define FA_TR_ROWS 32 // (env, agent) rows per workgroup tile: four 32 x 132-float LDS buffers, two workgroups per CU

// Transposed weights for the backward's dX = dY W^T GEMMs, each in the packed B-operand order of
// fa_policy.h ("packed (K x C)" of the TRANSPOSE).  Offsets in floats.
#define FA_TOFF_AOT 0      // packed (64 x 64):   A_o^T
#define FA_TOFF_BOT 4096   // packed (64 x 64):   B_o^T
#define FA_TOFF_AMT 8192   // packed (128 x 128): A_m^T
#define FA_TOFF_W7T 24576  // packed (128 x 256): W7^T
#define FA_TOFF_W8T 57344  // packed (256 x 128): W8^T
#define FA_TOFF_W9T 90112  // packed (32 x 256):  W9^T
#define FA_TRANS_FLOATS 98304

// The result of fa_ppo_grad (`out`): the gradient of every kernel-facing matrix in PLAIN row-major layout at the offsets
// of the forward pack (FA_POFF_*: We (6x64) | be | Woe | boe | A_o (64x64) | B_o | A_m (128x128) | W7 (256x128)
// | bu | W8 (128x256) | b8 | W9 (256x32) | b9), then the minibatch's loss sums.
#define FA_SLAB_LOSS FA_POLICY_PLAIN_FLOATS // [value_loss sum, action_loss sum, entropy*mask sum, mask sum]
#define FA_SLAB_FLOATS (FA_POLICY_PLAIN_FLOATS + 16)
#define FA_MASK_PARTS 64
#define FA_NORM_PARTS 64
#define FA_ADAM_SCRATCH (4 + 2 * FA_NORM_PARTS)

// What a tile of fa_train_kernel leaves in global memory.
// (1) The operands of the weight-gradient GEMMs dW = X^T dY, which fa_train_dw_kernel sums over ALL rows of the minibatch
//     (so the tile kernel carries no accumulator across its three rounds and writes no per-tile weight-gradient slab):
//     one record per (tile, round) of four dense 32 x 128 planes, and one record per tile for the heads / opponent stage.
#define FA_REC_PLANE (FA_TR_ROWS * 128)
#define FA_RECA_HIN 0                      // h entering the round (also what the tile's own backward reloads)
#define FA_RECA_HMIX (1 * FA_REC_PLANE)    // the attention mix of the round
#define FA_RECA_DZ (2 * FA_REC_PLANE)      // dL/d(pre-activation of the update layer)
#define FA_RECA_DG (3 * FA_REC_PLANE)      // dL/dg, g = h A_m
#define FA_RECA_FLOATS (4 * FA_REC_PLANE)  // dW7 += [HIN | HMIX]^T DZ ;  dA_m += HIN^T DG
#define FA_RECB_H3 0                                   // 32 x 128: h after the last round
#define FA_RECB_DPV (FA_REC_PLANE)                     // 32 x 256: [dP | dV] through the head relus
#define FA_RECB_MO (3 * FA_REC_PLANE)                  // 32 x 64: the opponent-attention mix
#define FA_RECB_DE (3 * FA_REC_PLANE + FA_TR_ROWS * 64)  // 32 x 64: dL/de_opp
#define FA_RECB_H1 (4 * FA_REC_PLANE)                  // 32 x 64: the own encodings
#define FA_RECB_DGO (4 * FA_REC_PLANE + FA_TR_ROWS * 64) // 32 x 64: dL/dg_o, g_o = h1 A_o
#define FA_RECB_FLOATS (5 * FA_REC_PLANE)  // dW8 = H3^T DPV ;  dB_o = MO^T DE ;  dA_o = H1^T DGO
// (3) g = h_in A_m of every round (32 x 128 each): what the tile's own backward reloads instead of recomputing
#define FA_TR_SAVE_FLOATS (3 * FA_RECA_FLOATS + FA_RECB_FLOATS + 3 * FA_REC_PLANE) // per tile
// (2) The tile's SMALL gradients: encoders and biases at their FA_POFF_* offsets (0 .. 895), then the update / head
//     biases, the 1 152 real entries of dW9 (W9 is block diagonal: 128 x 8 logits weights + 128 value weights) and the
//     tile's loss sums.  Summed over the tiles in two stages (fa_train_mred_kernel, then the final reduction).
#define FA_MSLAB_BU 896
#define FA_MSLAB_B8 1024
#define FA_MSLAB_B9 1280
#define FA_MSLAB_W9C 1312
#define FA_MSLAB_LOSS 2464
#define FA_MSLAB_FLOATS 2480
#define FA_MRED_PARTS 64
// fa_train_dw_kernel: FA_DW_WGS_A workgroups share the (tile, round) records, FA_DW_WGS_B the per-tile records (split by
// their MFMA work, 3.6 : 1); each leaves one partial slab: A = [dW7 (256 x 128) | dA_m (128 x 128)], B = [dW8 (128 x 256) |
// dB_o (64 x 64) | dA_o (64 x 64)], plain row-major.
#define FA_DW_WGS_A 200
#define FA_DW_WGS_B 56
#define FA_DWA_FLOATS (256 * 128 + 128 * 128)
#define FA_DWB_FLOATS (128 * 256 + 2 * 64 * 64)

struct FaTrainArgs {
    const float *obs;          // (B, N, 6) minibatch observations
    const int64_t *action;     // (B, N) the actions taken (own agents' columns are read)
    const float *value_pred;   // (B, N) value_preds of the rollout
    const float *ret;          // (B, N) returns
    const float *old_logp;     // (B, N) action_log_probs of the rollout
    const float *adv;          // (B, N) normalised advantages, or null with adv_mean / adv_std
    const double *adv_mean, *adv_std; // per agent (N): the kernel normalises ret - value_pred itself (ppo.py:121-124)
    const int64_t *idx;        // the minibatch: sample b is row idx[b] of the six arrays above (null: row b)
    const float *w;            // forward pack (FA_POFF_*)
    const float *wt;           // transposed pack (FA_TOFF_*)
    float *mslab;              // [tiles][FA_MSLAB_FLOATS]
    float *rec_a;              // [tiles][3][FA_RECA_FLOATS]
    float *rec_b;              // [tiles][FA_RECB_FLOATS]
    float *rec_g;              // [tiles][3][FA_REC_PLANE]
    int32_t B, G, A, team;     // team 0: the guards' policy on the guards' rows; 1: the attackers'
    int32_t tile0, ntiles;     // this launch covers tiles [tile0, tile0 + ntiles) of the minibatch (ntiles 0: all of them);
    int32_t rec_tile0;         // its records go to slots [rec_tile0, rec_tile0 + ntiles) of rec_a / rec_b / rec_g (chunked
                               // hand-off to fa_train_dw_kernel: fa_api.hip fa_ppo_grad); mslab stays indexed by the tile
    const float *mask_part;    // with scale == null: FA_MASK_PARTS partial alive-mask sums (fa_launch_mask_parts); the
    float *scale_out;          // kernel derives the scale pair itself (`normalize`: divide by the mask mean) and
    int32_t normalize;         // workgroup 0 leaves it at scale_out[0..1]
    int32_t share_cu;          // (ignored since round 4: see fa_train.hip)
    const float *scale;        // device float[2]: {1 / (B n mask_mean'), mask_mean'} with mask_mean' = the alive-mask
                               // mean of the minibatch (1 where that is 0, or when the caller normalises later:
                               // several ranks).  [0] multiplies every loss gradient; [1] undoes it for the
                               // unclipped value loss, which the reference does not mask (ppo.py:178-182)
    float clip;                // PPO clip parameter
    float c_value, c_entropy;  // loss coefficients
    int32_t clipped_value_loss;
};

hipError_t fa_launch_tasks(const fa_task *tasks, int n, hipStream_t st);
hipError_t fa_launch_pack(const float *plain, float *w, float *wt, hipStream_t st);
int fa_train_tile_envs(int G, int A);
hipError_t fa_launch_train(const FaTrainArgs &a, hipStream_t st);
// FA_MASK_PARTS partial sums of the alive mask over the minibatch's own-team rows (FaTrainArgs::mask_part)
hipError_t fa_launch_mask_parts(const FaTrainArgs &a, float *part, hipStream_t st);
// Adam (torch.optim.Adam, no amsgrad / weight decay) on flat buffers after the global-norm clip of
// nn.utils.clip_grad_norm_: two launches.  seg: nseg + 1 offsets; steps: nseg step counters (float);
// scratch: FA_ADAM_SCRATCH floats, 16-byte aligned ([0] receives the clip coefficient); hyper: null, or 5 device
// floats (lr, beta1, beta2, eps, max_norm) that replace the by-value arguments at run time
hipError_t fa_launch_adam(float *p, float *g, float *m, float *v, float *steps, const int32_t *seg, int nseg, int n, float lr,
                          float beta1, float beta2, float eps, float max_norm, float *scratch, const float *hyper,
                          hipStream_t st);
// The weight-gradient GEMMs over the records of `tiles` tiles -> FA_DW_WGS_A + FA_DW_WGS_B partial slabs at dw_slabs
// (accumulate: the partial slabs already hold the sums of earlier chunks of the minibatch -- start from them)
hipError_t fa_launch_train_dw(const float *rec_a, const float *rec_b, int tiles, float *dw_slabs, bool accumulate, hipStream_t st);
// mslab of `tiles` tiles -> mpart[FA_MRED_PARTS][FA_MSLAB_FLOATS]; then out[k], k < FA_SLAB_LOSS + 8, from the partial
// slabs and mpart (fixed orders: reproducible); out[FA_SLAB_LOSS + 8..9] is where fa_ppo_grad keeps the scale pair when
// the caller passes none
hipError_t fa_launch_train_reduce(const float *mslab, int tiles, float *mpart, const float *dw_slabs, float *out, hipStream_t st);
