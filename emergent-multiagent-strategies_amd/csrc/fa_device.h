// fa_device.h -- shared device-side declarations of the FortAttack HIP engine (gfx950).
//
// Data layout in HBM (E envs, N agents/env, guards first):
//   world state   fp64 SoA, env-major: field[e*N + i].  A wave64 carries EPW = 64/N whole
//                 envs (10 at N=6, 6 at N=10), lane = agent, so every field access of a
//                 wave is one contiguous span of EPW*N elements.
//   RNG state     MT19937 words, env-major mt[e*624 + k] + a cursor per env.
//   rollout rows  (E, N, ...) float32 rows of the caller's joint RolloutStorage tensors.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FA_WAVE 64
#define FA_MT_N 624
#define FA_MT_M 397
#define FA_MAX_AGENTS_DEV 16
#define FA_ACT_BATCH 16 // env-steps of actions staged in LDS per batch (power of two)
#define FA_TWO_WAVE_MAX_GRID 1024  // workgroups up to which the two-wave step kernel is used
#define FA_THREE_WAVE_MAX_GRID 680 // ... and the three-wave one (3 x 680 waves = 2 per SIMD)
#define FA_PIPE_MAX_GRID 768       // pipelined kernel: 4 waves x 3 workgroups per CU x 256 CUs in one round
#define FA_PIPE_MIN_STEPS 8        // ... and only for rollout launches

// Host-derived constants (evaluated once in double, in the reference's expression order).
struct FaDerived {
    double agent_size, accel, max_speed, fort_dim, door_x, door_y, dt, one_minus_damping;
    double contact_force, contact_margin, dist_min;
    double wall_xmin, wall_xmax, wall_ymin, wall_ymax;
    double shoot_rad, half_win;
    double cos_hw, sin_hw;          // cos/sin(shootWin/2), host libm
    double shoot_far;               // shoot_rad * cos(shootWin/2): distance of the wedge's far edge
    double speed2_max;              // max{x : sqrt_rn(x) <= max_speed}
    double fort2_max;               // max{x : sqrt_rn(x) <  fort_dim}
    double rot_pos, rot_neg;        // (+max_rot) % 2pi, (-max_rot) % 2pi  (Python modulo, core.py:336)
    double ang_guard, ang_attacker; // 3pi/2, pi/2                         (fortattack_env_v1.py:59)
    double att_x_lo, att_x_rng, att_y_lo, att_y_rng; // fortattack_env_v1.py:66
    double grd_x_lo, grd_x_rng, grd_y_lo, grd_y_rng; // fortattack_env_v1.py:70
    double contact_skip_d2;         // squared distance beyond which the soft contact is exactly 0
    double wall_skip;               // wall clearance beyond which the soft contact is exactly 0
};

struct FaState {
    double *px, *py, *vx, *vy, *ang, *prev; // (E,N)
    uint8_t *alive;                          // (E,N)
    int32_t *tstep;                          // (E)
    int32_t *num_hit, *num_was_hit;          // (E,N) or null when counters are off
    uint8_t *game_result;                    // (E,3)
    uint32_t *result_count;                  // (E,3)
    uint32_t *mt;                            // (E,624)
    int32_t *mt_pos;                         // (E)
    uint32_t *reset_count;                   // (E)   Philox counter
    double *ep_rew, *ep_rew_sum;             // (E,N) running / finished-episode sum of reward*mask
    uint32_t *alive_end;                     // (E,N) episodes this agent was alive at the end of
};

struct FaStepArgs {
    FaState s;
    const int64_t *actions;
    int64_t as_t, as_e, as_i; // element strides: rollout step, env, agent
    float *obs32, *rew32, *mask32;
    uint8_t *done;
    double *obs64, *rew64;
    uint8_t *hit, *was_hit;
    const uint8_t *reset_mask; // fa_reset only
    int32_t E, G, A, max_t;
    int32_t auto_reset, rng_mode, track_counters, nsteps; // nsteps: env-steps per launch
    int32_t step_kernel;                                  // FA_KERNEL_* of the handle (host side only)
    int32_t choice_k;                                     // > 0: np.random.choice(choice_k) follows every reset
    int32_t *choice_out;                                  // (E) the drawn index, see fa_set_reset_choice
    uint64_t seed;
    int64_t env_offset;
    FaDerived c;
};
