// fa_train.h -- internal declarations of the fused PPO minibatch kernel (fa_train.hip): the MPNN forward, the
// alive-masked PPO losses and the complete backward pass of one team's minibatch in one launch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fa_policy.h"
#include "fortattack.h"

#define FA_TR_ROWS 64 // (env, agent) rows per workgroup tile: four 64 x 132-float LDS buffers

// Transposed weights for the backward's dX = dY W^T GEMMs, each in the packed B-operand order of
// fa_policy.h ("packed (K x C)" of the TRANSPOSE).  Offsets in floats.
#define FA_TOFF_AOT 0      // packed (64 x 64):   A_o^T
#define FA_TOFF_BOT 4096   // packed (64 x 64):   B_o^T
#define FA_TOFF_AMT 8192   // packed (128 x 128): A_m^T
#define FA_TOFF_W7T 24576  // packed (128 x 256): W7^T
#define FA_TOFF_W8T 57344  // packed (256 x 128): W8^T
#define FA_TOFF_W9T 90112  // packed (32 x 256):  W9^T
#define FA_TRANS_FLOATS 98304

// A tile's output slab: the gradient of every kernel-facing matrix in PLAIN row-major layout at the offsets
// of the forward pack (FA_POFF_*: We (6x64) | be | Woe | boe | A_o (64x64) | B_o | A_m (128x128) | W7 (256x128)
// | bu | W8 (128x256) | b8 | W9 (256x32) | b9), then the tile's loss sums.
#define FA_SLAB_LOSS FA_POLICY_WEIGHT_FLOATS // [value_loss sum, action_loss sum, entropy*mask sum, mask sum]
#define FA_SLAB_FLOATS (FA_POLICY_WEIGHT_FLOATS + 16)
#define FA_MASK_PARTS 64
#define FA_NORM_PARTS 64
#define FA_ADAM_SCRATCH (4 + 2 * FA_NORM_PARTS)
#define FA_TR_SAVE_FLOATS (3 * FA_TR_ROWS * 128) // per tile: h after the opponent stage and after rounds 1, 2

struct FaTrainArgs {
    const float *obs;          // (B, N, 6) minibatch observations
    const int64_t *action;     // (B, N) the actions taken (own agents' columns are read)
    const float *value_pred;   // (B, N) value_preds of the rollout
    const float *ret;          // (B, N) returns
    const float *old_logp;     // (B, N) action_log_probs of the rollout
    const float *adv;          // (B, N) normalised advantages
    const int64_t *idx;        // the minibatch: sample b is row idx[b] of the six arrays above (null: row b)
    const float *w;            // forward pack (FA_POFF_*)
    const float *wt;           // transposed pack (FA_TOFF_*)
    float *slabs;              // [tiles][FA_SLAB_FLOATS]
    float *hsave;              // [tiles][FA_TR_SAVE_FLOATS]
    int32_t B, G, A, team;     // team 0: the guards' policy on the guards' rows; 1: the attackers'
    const float *mask_part;    // with scale == null: FA_MASK_PARTS partial alive-mask sums (fa_launch_mask_parts); the
    float *scale_out;          // kernel derives the scale pair itself (`normalize`: divide by the mask mean) and
    int32_t normalize;         // workgroup 0 leaves it at scale_out[0..1]
    int32_t share_cu;          // leave 48 registers per lane of every CU to concurrent small launches (fa_train.hip)
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
// out[k] = sum over tiles of slabs[t][k], k < FA_SLAB_LOSS + 8 (fixed order: reproducible); out[FA_SLAB_LOSS + 8..9]
// is where fa_ppo_grad keeps the scale pair when the caller passes none
hipError_t fa_launch_train_reduce(const float *slabs, int tiles, float *out, hipStream_t st);
