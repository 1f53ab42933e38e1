/*
 * fortattack.h -- C ABI of the MI355X-native FortAttack rollout engine.
 *
 * The reference (Ankur-Deka/Emergent-Multiagent-Strategies) has no FFI / plugin
 * interface: its boundary is a duck-typed Python API.  This header is
 * the boundary a binding for that API sits on: every entry point names the reference
 * interface it replaces (file:line relative to the reference root).  INTEGRATION.md
 * shows the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *   - return 0 (FA_OK) on success, negative on error; text via fa_last_error()
 *     (thread local).
 *   - the CALLER owns every buffer passed in (torch tensors, hipMalloc'd memory); the
 *     library owns only the per-handle fp64 SoA world state and RNG state.
 *   - all pointers in fa_step_io / fa_storage are DEVICE pointers on the handle's
 *     device; fa_state_host pointers are HOST pointers.
 *   - every call that takes `stream` (a hipStream_t, passed as void*) is asynchronous
 *     and stream ordered: no hidden synchronisation, no allocation after fa_create.
 *   - one handle per device; a handle is not thread safe (the reference is single
 *     threaded, train_fortattack.py:199); distinct handles are independent.
 *   - nothing is printed (the reference prints on every episode end,
 *     fortattack.py:208,214,220).
 *
 * Layouts (E = num_envs on this handle, N = num_guards + num_attackers, guards first
 * as in fortattack_env_v1.py:26-31):
 *   state        fp64 SoA, element (e, i) at [e*N + i]          (env-major: a wave
 *                touches one contiguous span per field)
 *   obs          (E, N, 6) = [alive, px, py, ang, vx, vy]        fortattack_env_v1.py:238
 *   storage      joint tensors (T[+1], E, N, ...).  Agent i's reference-shaped
 *                RolloutStorage tensor (T[+1], P=E, ...) is the strided view [:, :, i]
 *                (rlcore/storage.py:10-21).
 */
#ifndef FORTATTACK_H
#define FORTATTACK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FA_MAX_AGENTS 16
#define FA_OBS_DIM 6
#define FA_NUM_ACTIONS 8

enum { FA_OK = 0, FA_ERR_INVALID = -1, FA_ERR_HIP = -2, FA_ERR_STATE = -3 };
enum { FA_KERNEL_AUTO = 0,
       FA_KERNEL_PIPE = 1,    /* fa_step_pipe_kernel, two workgroups per CU build (3v3 / 5v5, num_steps >= 2) */
       FA_KERNEL_PIPE3 = 2,   /* fa_step_pipe_kernel, three workgroups per CU build (<= 168 VGPRs) */
       FA_KERNEL_WAVES1 = 3,  /* fa_step_kernel, one wave per workgroup */
       FA_KERNEL_WAVES2 = 4,  /* ... with the force wave (3v3 / 5v5) */
       FA_KERNEL_WAVES3 = 5 }; /* ... with force and wall waves (3v3 / 5v5) */
/* (values >= 64 name experiment kernels that exist in variant libraries only -- csrc/experiments/fa_step_experiments.h,
 * tools/build_variant.py; this library's fa_create refuses them) */
enum { FA_RNG_MT19937 = 0, /* numpy legacy RandomState stream: parity with the reference */
       FA_RNG_PHILOX = 1 }; /* counter based, stateless: perf mode */

typedef struct fa_env fa_env; /* opaque */

/* World constants: literals of gym_fortattack/core.py:32,100-101,117-128 and
 * gym_fortattack/envs/fortattack_env_v1.py:16-17,33-35.  fa_config_default fills them. */
typedef struct fa_world_consts {
    double agent_size;     /* 0.05   core.py:32 */
    double accel;          /* 3      fortattack_env_v1.py:33 */
    double max_speed;      /* 3      fortattack_env_v1.py:34 */
    double max_rot;        /* 0.17   fortattack_env_v1.py:35 */
    double fort_dim;       /* 0.15   fortattack_env_v1.py:16 */
    double door_x, door_y; /* 0, 0.8 fortattack_env_v1.py:17 */
    double dt;             /* 0.1    core.py:121 */
    double damping;        /* 0.25   core.py:123 */
    double contact_force;  /* 100    core.py:125 */
    double contact_margin; /* 1e-10  core.py:126 */
    double wall_xmin, wall_xmax, wall_ymin, wall_ymax; /* -1 1 -0.8 0.8  core.py:128 */
    double shoot_rad;      /* 0.8    core.py:100 */
    double shoot_win;      /* pi/4   core.py:101 */
} fa_world_consts;

typedef struct fa_config {
    int32_t num_envs;       /* E: env instances on this handle */
    int32_t num_guards;     /* reference hard-codes 5 (fortattack_env_v1.py:18) */
    int32_t num_attackers;  /* reference hard-codes 5 (fortattack_env_v1.py:19) */
    int32_t max_time_steps; /* world.max_time_steps, fortattack.py:21 */
    int32_t device_id;
    int32_t rng_mode;       /* FA_RNG_* */
    uint64_t base_seed;     /* env e == reference under np.random.seed(base_seed + env_offset + e) */
    int64_t env_offset;     /* global index of this handle's env 0 (multi-GPU sharding) */
    int32_t rng_skip_doubles; /* random_sample() draws consumed before the first reset
                                 (scenario __init__ -> reset_world, fortattack_env_v1.py:45);
                                 negative = 2*N */
    int32_t track_counters; /* maintain Agent.numHit / numWasHit (core.py:93-94) */
    int32_t step_kernel;    /* FA_KERNEL_*: which step kernel this handle's launches use.  AUTO picks by
                               team size, grid size and launch length (fa_step_variant reports it); the
                               others pin one build so that tests can drive every instantiation at any
                               size.  All variants produce identical bits. */
    fa_world_consts world;
} fa_config;

/* One env.step over all E envs.  Any output pointer may be NULL. */
typedef struct fa_step_io {
    const int64_t *actions;    /* element (e,i) at actions[e*act_stride_env + i*act_stride_agent];
                                  values 0..7 (fortattack.py:253-263) */
    int64_t act_stride_env, act_stride_agent;
    float *obs_f32;            /* (E,N,6)  as stored by learner.py:240 (.float())          */
    float *reward_f32;         /* (E,N)    train_fortattack.py:73                            */
    float *mask_f32;           /* (E,N)    obs[:,0] BEFORE the step, train_fortattack.py:53;
                                  1 for every agent of an env auto-reset by this step      */
    uint8_t *done;             /* (E)      fortattack.py:166                                 */
    double *obs_f64;           /* (E,N,6)  what env.step returns, fortattack.py:172          */
    double *reward_f64;        /* (E,N)    fortattack.py:173                                 */
    uint8_t *hit, *was_hit;    /* (E,N)    Agent.hit / wasHit of this step (core.py:95-96)   */
    int32_t auto_reset;        /* !=0: envs that finish are reset in the same launch and obs_*
                                  hold the post-reset observation -- the trainer's
                                  `if done: obs = env.reset()` (train_fortattack.py:97-104,
                                  rlagent.py:28-31); mask for the next step is then 1 */
    int32_t num_steps;         /* env-steps advanced by this ONE launch (0 or 1 = one step).
                                  With K > 1 the actions of step k are read at
                                  actions + k*act_stride_step and every output is a stack of K
                                  rows ((K,E,N,6), (K,E,N), (K,E)): an open-loop rollout whose
                                  world state never leaves the registers between steps.
                                  Requires auto_reset != 0 when episodes may end. */
    int64_t act_stride_step;
} fa_step_io;

/* Joint rollout buffers (device).  Shapes use T = num_steps. */
typedef struct fa_storage {
    int32_t num_steps;
    float *obs;                     /* (T+1, E, N, 6)  storage.py:11 */
    float *recurrent_hidden_states; /* (T+1, E, N)     storage.py:12 (size 1, unused by MPNN) */
    float *rewards;                 /* (T,   E, N)     storage.py:13 */
    float *value_preds;             /* (T+1, E, N)     storage.py:14 */
    float *returns;                 /* (T+1, E, N)     storage.py:15 */
    float *action_log_probs;        /* (T,   E, N)     storage.py:16 */
    int64_t *actions;               /* (T,   E, N)     storage.py:17-18 */
    float *masks;                   /* (T+1, E, N)     storage.py:19 (init 1) */
    uint8_t *done;                  /* (T,   E)        episode ended at step t: the per-env
                                       end_pts list of train_fortattack.py:98 as flags */
} fa_storage;

/* Host-side fp64 snapshot of the world state (tests, checkpoint).  Any pointer may be
 * NULL.  prev_dist: NaN encodes prevDist == None (core.py:104). */
typedef struct fa_state_host {
    double *pos_x, *pos_y, *vel_x, *vel_y, *ang, *prev_dist; /* (E,N) */
    uint8_t *alive;                                           /* (E,N) */
    int32_t *time_step;                                       /* (E)   world.time_step */
    int32_t *num_hit, *num_was_hit;                           /* (E,N) */
    uint8_t *game_result;                                     /* (E,3) world.gameResult of the
                                                                 last finished episode */
    int64_t *result_count;                                    /* (E,3) finished episodes by
                                                                 ending since fa_create */
    /* evaluation statistics of test_fortattack_v2.py:88-101, summed over the finished episodes
     * of each env (track_counters only): sum of per-episode returns (reward * alive-before
     * mask, train_fortattack.py:88) and number of episodes the agent was alive at the end of */
    double *episode_reward_sum;                               /* (E,N) */
    int64_t *alive_at_end;                                    /* (E,N) */
} fa_state_host;

/* ---- lifecycle ---------------------------------------------------------------- */
int fa_config_default(fa_config *cfg);            /* reference literals; 3v3, E=1 */
/* replaces make_fortattack_env (fortattack.py:17-27) + gym.make('fortattack-v1')
 * (fortattack_env_v1.py:10-45) for E independent worlds; seeds the per-env RNG and
 * consumes the construction draws. */
int fa_create(const fa_config *cfg, fa_env **out);
void fa_destroy(fa_env *env);
const char *fa_last_error(void);
int fa_num_agents(const fa_env *env);
int fa_num_envs(const fa_env *env);

/* ---- env API ------------------------------------------------------------------- */
/* replaces FortAttackGlobalEnv.reset (fortattack.py:175-186) -> reset_world
 * (fortattack_env_v1.py:47-75).  env_mask: device (E) bytes, NULL = every env. */
int fa_reset(fa_env *env, const uint8_t *env_mask, float *obs_f32, double *obs_f64, void *stream);
/* replaces FortAttackGlobalEnv.step (fortattack.py:127-173): _set_action (:235-302),
 * World.step (core.py:191-218), observation / reward callbacks
 * (fortattack_env_v1.py:87-238), _get_done (fortattack.py:202-225). */
int fa_step(fa_env *env, const fa_step_io *io, void *stream);

/* Ensemble-attacker path (train_fortattack_v2.py): every env.reset() -- the first one (:29-35) and each
 * episode-end one (:104-111) -- is followed by Learner.sample_attacker (learner.py:119-121), whose
 * np.random.choice(attacker_ckpts) draws from the SAME global numpy stream as the reset positions.  With
 * k > 0 every reset of an env (fa_reset, fa_collect_reset, auto-reset inside fa_step / fa_collect_*) is
 * followed by that draw on the env's stream and the chosen index (0..k-1) is stored in choice_out[e]
 * (device, E int32, caller owned, must stay valid).  k = 0 switches it off.  Launches then use
 * fa_step_kernel (the pipelined kernel draws its resets ahead and is not used with a choice). */
int fa_set_reset_choice(fa_env *env, int32_t k, int32_t *choice_out);

/* ---- collector ------------------------------------------------------------------ */
/* Attach caller-owned rollout buffers (RolloutStorage.__init__/to, storage.py:10-31). */
int fa_bind_storage(fa_env *env, const fa_storage *st);
/* env.step + RolloutStorage.insert of the env-produced fields for rollout index `step`
 * (storage.py:33-43 via rlagent.py:33-34, learner.py:239-243): reads
 * actions[step], writes obs[step+1], rewards[step], masks[step+1], done[step]; with
 * auto_reset also Neo.initialize_new_episode (rlagent.py:28-31).  The policy-produced
 * fields (actions, action_log_probs, value_preds) are written by the caller. */
int fa_collect_step(fa_env *env, int32_t step, int32_t auto_reset, void *stream);
/* The same for rollout indices [step_begin, step_begin + num_steps) in ONE launch, for
 * open-loop policies (scripted / random actions already in storage.actions): the host
 * loop of train_fortattack.py:51-105 without the per-step policy call. */
int fa_collect_rollout(fa_env *env, int32_t step_begin, int32_t num_steps, int32_t auto_reset, void *stream);
/* FortAttackGlobalEnv.reset + Neo.initialize_obs (rlagent.py:23-26): reset every env and
 * write the observation to obs[0], masks[0] = 1. */
int fa_collect_reset(fa_env *env, void *stream);
/* Learner.wrap_horizon (learner.py:191-211) + RolloutStorage.compute_returns
 * (storage.py:59-66, GAE branch) for all (env, agent) with per-env episode boundaries.
 * value_preds[T] must hold V(obs[T]).  Reproduces quirk Q7 (returns[end_pt] not
 * recomputed) exactly. */
int fa_gae(fa_env *env, double gamma, double tau, void *stream);
/* fa_gae + the advantage statistics of ppo.py:121-123 (instead of the two passes + two folds of fa_adv_stats): per
 * agent, fp64 sums of (A - P) and (A - P)^2 around a pivot P that is an actual sample (no cancellation), folded over the
 * workgroups in a fixed order -- bitwise reproducible.  Up to 32 768 (env, agent) columns the scan itself leaves the
 * sums (A = returns[t] - value_preds[t] in float32 from the value it is about to store; the stale entries at the
 * episode ends from the old returns) and one workgroup folds them: two launches; beyond that fa_gae, a one-pass sweep
 * and its fold.  Writes moments[i] = {n, mean, M2} (the fa_adv_moments / fa_adv_merge format), mean[i] and the unbiased
 * std[i] of THIS handle's samples; any of the three may be null.  Device pointers. */
int fa_gae_moments(fa_env *env, double gamma, double tau, double *moments, double *mean, double *std_, void *stream);
/* The whole collector tail of one rank -- Learner.wrap_horizon + compute_returns (learner.py:191-211, storage.py:59-66),
 * the advantage statistics and the normalisation (A - mean) / (std + 1e-5) of rlcore/algo/ppo.py:121-124 -- in two
 * launches: the scan of fa_gae_moments, then ONE kernel in which every workgroup folds the moment partials (same order,
 * same bits everywhere) and normalises its share into adv_out, (T, E, N) float32; returns / value_preds are read once
 * more, nothing else (beyond 32 768 columns: fa_gae, the one-pass sweep and the same fold + normalisation launch on its
 * partials -- three launches).  Equals fa_gae_moments + fa_adv_normalize bit for bit; moments / mean / std_ (may be null) as
 * there.  With several ranks the statistics are exchanged between the two halves: fa_gae_moments, the all-gather,
 * fa_adv_merge, fa_adv_normalize. */
int fa_gae_normalize(fa_env *env, double gamma, double tau, float *adv_out, double *moments, double *mean, double *std_,
                     void *stream);
/* The second half of fa_gae_moments alone -- the one-pass per-agent advantage moments (n, mean, M2), mean and unbiased
 * std of ppo.py:121-123 from the bound storage's returns / value_preds -- for a caller that puts the statistics on
 * another stream than the GAE scan (they read returns / value_preds only; the next rollout does not touch those). */
int fa_adv_moments_onepass(fa_env *env, double *moments, double *mean, double *std_, void *stream);
/* Advantage statistics of JointPPO.update (rlcore/algo/ppo.py:121-123), per agent, in
 * fp64: pass 0 writes stats[i] = {n, sum(A), 0}; pass 1 reads mean[i] and writes
 * only stats[i][2] = sum((A-mean)^2) (stats[i][0..1] are left as they are).  A = returns[:-1] - value_preds[:-1].
 * stats: device (N,3) doubles; mean: device (N) doubles.  The only quantities that
 * cross GPUs (all-reduce sum of `stats`). */
int fa_adv_stats(fa_env *env, int32_t pass, const double *mean, double *stats, void *stream);
/* Single-GPU form of ppo.py:122-123: both passes on this handle's samples only; writes
 * mean[i] and the unbiased std[i] (device, N doubles each).  With several GPUs use
 * fa_adv_stats and all-reduce between the passes instead. */
int fa_adv_mean_std(fa_env *env, double *mean_out, double *std_out, void *stream);
/* Multi-GPU form with ONE collective: moments_out (N,3) = {n, mean, M2} of this handle's
 * samples (two passes, M2 = sum of squared deviations from the local mean); all-gather the
 * triples of every rank and give them to fa_adv_merge, which combines them exactly
 * (Chan-Golub-LeVeque, rank order => identical on all ranks) into the global mean / unbiased
 * std.  gathered: device (world, N, 3) doubles. */
int fa_adv_moments(fa_env *env, double *moments_out, void *stream);
int fa_adv_merge(fa_env *env, const double *gathered, int32_t world, double *mean_out, double *std_out,
                 void *stream);
/* ppo.py:123: adv_out (T,E,N) = (A - mean[i]) / (std[i] + 1e-5), float32. */
int fa_adv_normalize(fa_env *env, const double *mean, const double *std, float *adv_out, void *stream);
/* fa_adv_merge + fa_adv_normalize as ONE launch (the several-rank tail behind the all-gather): every workgroup merges the
 * `world` ranks' gathered (n, mean, M2) triples itself -- fa_adv_merge's loop, rank order, the same bits on every workgroup and
 * every rank -- and normalises its share of this handle's advantages into adv_out (ppo.py:121-124 over ALL ranks' samples).
 * mean_out / std_out (device, N doubles, may be null) receive the merged statistics.  Bit for bit the two separate calls. */
int fa_adv_merge_normalize(fa_env *env, const double *gathered, int32_t world, float *adv_out, double *mean_out,
                           double *std_out, void *stream);
/* RolloutStorage.after_update (storage.py:51-56). */
int fa_after_update(fa_env *env, void *stream);

/* ---- policy in the loop ------------------------------------ */
/* The MPNN actor-critic forward of both teams (reference mpnn.py:117-192: _fwd + act / get_value;
 * called per env-step from learner.py:143-172) as ONE fused launch: encoders, opponent attention,
 * the K = 3 message-passing rounds, policy and value heads, log-softmax and categorical sampling
 * (FixedCategorical.sample / .mode, rlcore/distributions.py:12-17), hidden_dim = 128, float32 results: the dense
 * layers run on the bf16 matrix cores with every float32 operand split EXACTLY into three bf16 terms (fp32 accumulate;
 * fp32-class accuracy, csrc/fa_mfma.h gemm_cb3).
 * weights[t] is team t's policy (0 = guards, 1 = attackers) in the packed layout of
 * csrc/fa_policy.h: fa_policy_weight_floats() floats = the float32 sections (FA_POFF_*, fa_policy_plain_floats() floats:
 * also the layout of the plain / gradient buffers of the update) followed by the six dense matrices once more as bf16
 * triples in MFMA operand order (FA_POFF3_*).  Filled from a reference state_dict by mpnn_pack.pack_policy, or on the
 * device by fa_pack_weights -- three pairs of consecutive linear maps are pre-multiplied there, which is exact in real
 * arithmetic.  Outputs are (E, N) rows, guards first: value, action (0..7), log-prob of the
 * action.  Sampling stream: Philox4x32-10 keyed by (seed; *counter, step, global env index, agent).
 * Teams of up to 8 agents. */
typedef struct fa_policy_io {
    const float *obs;            /* (E, N, 6) device: the observation row the policies act on */
    const float *weights[2];     /* device, packed (see above) */
    float *value;                /* (E, N) or NULL */
    int64_t *action;             /* (E, N); ignored when value_only */
    float *log_prob;             /* (E, N); ignored when value_only */
    const int64_t *counter;      /* device scalar or NULL: bump it once per rollout so that equal (seed, step)
                                    of different rollouts draw differently */
    uint64_t seed;
    int32_t step;                /* rollout index of this call (part of the sampling key) */
    int32_t deterministic;       /* != 0: argmax instead of a sample */
    int32_t value_only;          /* != 0: get_value (mpnn.py:202-205): only `value` is written */
    /* Ensemble of frozen attacker strategies (train_fortattack_v2.py; Learner.select_attacker loads one
     * checkpoint into the attacker agents, learner.py:132-140): with pool_size > 0 env e's attackers run
     * strategy env_strategy[e] of the pool instead of weights[1].  The envs are sorted into tiles of equal
     * strategy first (one extra small launch), so each strategy's network runs once over ITS envs. */
    const float *attacker_pool;  /* device: pool_size packed buffers back to back, or NULL */
    int32_t pool_size;           /* 0 = no ensemble; at most 64 */
    const int32_t *env_strategy; /* device (E): 0..pool_size-1, e.g. the choice_out of fa_set_reset_choice */
} fa_policy_io;
int fa_policy_act(fa_env *env, const fa_policy_io *io, void *stream);
/* The same on the bound storage (io->obs / value / action / log_prob are ignored): reads obs[step], writes
 * value_preds[step], actions[step], action_log_probs[step] (Learner.act + the policy half of
 * RolloutStorage.insert, learner.py:143-172, storage.py:33-43); with value_only only value_preds[step]
 * (wrap_horizon's V(obs[T]), learner.py:196-202). */
int fa_collect_act(fa_env *env, int32_t step, const fa_policy_io *io, void *stream);
/* number of floats of one team's packed weight buffer; of its leading float32 sections (= a plain-layout buffer) */
int64_t fa_policy_weight_floats(void);
int64_t fa_policy_plain_floats(void);

/* ---- PPO update: the attention between the agents of one env, forward and backward ---------------
 * (reference mpnn.py:250-332 MultiHeadAttention / :372-443 MultiHeadOppAttention, one head, inside
 * evaluate_actions as JointPPO.update calls it, ppo.py:146-147).  With g = h (norm W_query W_key^T) already
 * projected by a GEMM:  s_ij = g_i . k_j over the env's nk key rows (j == i excluded when skip_self),
 * a_i = softmax_j s_ij, out_i = sum_j a_ij k_j  (W_val W_out are folded into the next GEMM).
 * g, out, dout, dg: (B*n, width) rows env-major; keys, dkeys: (B, nk, width); attn: (B*n, nk), written by
 * the forward and read by the backward; width 64 or 128; n, nk <= 8.  Device pointers on the calling
 * thread's current device; stream ordered.  The caller adds dkeys to the key tensor's gradient. */
int fa_attend_forward(const float *g, const float *keys, float *out, float *attn, int32_t B, int32_t n, int32_t nk,
                      int32_t width, int32_t skip_self, void *stream);
int fa_attend_backward(const float *g, const float *keys, const float *attn, const float *dout, float *dg,
                       float *dkeys, int32_t B, int32_t n, int32_t nk, int32_t width, int32_t skip_self, void *stream);

/* One team's PPO minibatch as one fused launch (+ a small reduction): the MPNN forward of evaluate_actions
 * (mpnn.py:194-200), the alive-masked clipped losses of JointPPO.update (rlcore/algo/ppo.py:146-187) and the
 * complete backward pass, hidden_dim = 128, teams of up to 8.  `weights` is the team's policy in the packed
 * layout of fa_policy_act, `weights_t` the transposes the backward needs (csrc/fa_train.h FA_TOFF_*; both
 * written by mpnn_pack).  `out` receives FA_SLAB floats: the gradient of the loss
 *     value_loss * c_value + action_loss - entropy * c_entropy
 * with respect to every kernel-facing matrix, in PLAIN row-major layout at the offsets of the packed buffer
 * (csrc/fa_policy.h FA_POFF_*: encoders, A_o, B_o, A_m, W7, W8, W9 and the biases -- the host applies the
 * chain rule to the module's own parameters, mpnn_pack.KernelParams), followed at fa_ppo_grad_floats() - 16
 * by the sums over the minibatch of {value loss, action loss, entropy * mask, mask}.
 * The minibatch is rows idx[0..B) of the arrays (magent_feed_forward_generator's index set, ppo.py:213-246; idx
 * NULL: rows 0..B): obs (rows, N, 6); action / value_pred / ret / old_log_prob / adv (rows, N), of which the team's
 * columns are read.
 * scale: NULL -> the library takes the minibatch's alive-mask mean itself (one small launch) and divides every loss
 * by it as ppo.py:150-187 does (`normalize` = 1), or leaves the division to the caller (`normalize` = 0: several
 * ranks divide by the all-rank mean after the gradient exchange; the mask sum is the 4th loss float).  Otherwise
 * a device float[2] = {1 / (B n mask_mean'), mask_mean'} supplied by the caller (mask_mean' = 1 where the mean is 0).
 * slabs / hsave: scratch of fa_ppo_grad_scratch() floats.  Bitwise reproducible (no atomics). */
typedef struct fa_ppo_grad_io {
    const float *obs;
    const int64_t *action;
    const float *value_pred, *ret, *old_log_prob, *adv;
    const int64_t *idx;        /* B row indices, or NULL */
    const float *weights, *weights_t;
    const float *scale;        /* or NULL */
    float *slabs, *hsave;      /* scratch */
    float *out;                /* fa_ppo_grad_floats() floats */
    int32_t B;                 /* envs in the minibatch */
    int32_t num_guards, num_attackers;
    int32_t team;              /* 0: the guards' policy on the guards' rows, 1: the attackers' */
    float clip_param, value_loss_coef, entropy_coef;
    int32_t clipped_value_loss;
    int32_t normalize;         /* with scale == NULL: 1 = divide by the alive-mask mean here, 0 = the caller will */
    int32_t share_cu;          /* accepted and ignored (rounds 2-3: a register-capped build of the tile kernel that left
                                  room on its CUs for another stream's launches; the round-4 kernel does better uncapped) */
    /* Advantage normalisation inside the kernel (rlcore/algo/ppo.py:121-124): when adv_mean / adv_std are given
     * (device, one double per agent: what fa_gae_moments / fa_adv_mean_std / fa_adv_allreduce leave), `adv` may be NULL
     * and a row's advantage is (ret - value_pred - (float)mean[agent]) / ((float)std[agent] + 1e-5f) -- bit for bit
     * what fa_adv_normalize would have written -- so a trainer never materialises the (T, E, N) advantage tensor. */
    const double *adv_mean, *adv_std;
} fa_ppo_grad_io;
int fa_ppo_grad(const fa_ppo_grad_io *io, void *stream);
int64_t fa_ppo_grad_floats(void);
/* scratch sizes in floats for a minibatch of B envs */
int fa_ppo_grad_scratch(int32_t B, int32_t num_guards, int32_t num_attackers, int64_t *slab_floats, int64_t *hsave_floats);
int64_t fa_policy_weight_t_floats(void);

/* The small dense algebra around fa_ppo_grad -- building the kernel-facing matrices from the module's
 * parameters and carrying their gradients back (products of 64..128-wide matrices, transposes, slices) -- as a
 * list of strided tasks run by ONE launch (one workgroup per task), and the packing of the plain matrices
 * into fa_policy_act's layout (forward and transposed) by another.  type 0: C[i*ldc + j] = alpha * sum_k
 * A[i*a_rs + k*a_cs] * B[k*b_rs + j*b_cs]; type 1: C[i*ldc + j] = alpha * A[i*a_rs + j*a_cs] (i < M, j < N).
 * `tasks` is a DEVICE array; the pointers inside are device pointers. */
typedef struct fa_task {
    float *C;
    const float *A, *B;
    int32_t ldc, M, N, K, a_rs, a_cs, b_rs, b_cs;
    float alpha;
    int32_t type;
} fa_task;
int fa_run_tasks(const fa_task *tasks, int32_t n, void *stream);
/* plain: fa_policy_plain_floats() floats, row-major matrices at the FA_POFF_* offsets (csrc/fa_policy.h) ->
 * weights (fa_policy_weight_floats(): both halves, see fa_policy_io) and weights_t (fa_policy_weight_t_floats()) */
int fa_pack_weights(const float *plain, float *weights, float *weights_t, void *stream);

/* One optimizer step on a flat parameter buffer: nn.utils.clip_grad_norm_(max_grad_norm) over all n gradients,
 * then torch.optim.Adam's update (rlcore/algo/ppo.py:35 optim.Adam(lr, eps); no amsgrad, no weight decay) -- two
 * launches instead of ~70 (the multi-tensor kernels plus one pow kernel per parameter tensor and moment).
 * params / grads / exp_avg / exp_avg_sq: n floats; grads are left clipped.  seg: nseg + 1 int32 offsets of the
 * parameter tensors inside the buffer (seg[nseg] = n), steps: their nseg step counters (float, advanced here);
 * scratch: fa_adam_scratch_floats() floats, 16-byte aligned ([0] receives the clip coefficient).  All device
 * pointers.  The squared norm is accumulated in fp64 in a fixed order: reproducible. */
int fa_adam_step(float *params, float *grads, float *exp_avg, float *exp_avg_sq, float *steps, const int32_t *seg,
                 int32_t nseg, int32_t n, float lr, float beta1, float beta2, float eps, float max_grad_norm, float *scratch,
                 void *stream);
/* The same step with the hyper-parameters read from device memory at run time: hyper = 5 floats (lr, beta1, beta2,
 * eps, max_grad_norm).  A launch captured in a hipGraph then follows optimizer.param_groups[0] (a learning-rate
 * schedule, a resumed run with another --lr; rlcore/algo/ppo.py:35 reads lr once) without a re-capture: rewrite the
 * five floats, replay. */
int fa_adam_step_dev(float *params, float *grads, float *exp_avg, float *exp_avg_sq, float *steps, const int32_t *seg,
                     int32_t nseg, int32_t n, const float *hyper, float *scratch, void *stream);
int64_t fa_adam_scratch_floats(void);

/* ---- the cross-GPU exchange inside the library (RCCL, opened with dlopen at first use) -------------
 * The reference is ONE process (train_fortattack.py:199).  With the env batch sharded over GPUs, two quantities
 * must cover every rank's samples: the per-agent advantage mean / unbiased std of JointPPO.update
 * (rlcore/algo/ppo.py:121-123) and -- for a consistent data-parallel learner -- the gradients of every optimizer
 * step (ppo.py:189-193).  The Python package does both through torch.distributed (dist.py); these entry points are
 * the same exchange for a consumer without torch, stream-ordered on the caller's stream, no host round trip.
 * RCCL is not linked: librccl.so.1 (FA_RCCL_LIB overrides the name) is opened on first use; without it these calls
 * return FA_ERR_STATE and everything else works.  `nccl_comm` is an ncclComm_t -- the caller's own, or one made by
 * fa_rccl_comm_create (the 128-byte id from fa_rccl_unique_id on rank 0 reaches the other ranks by the caller's
 * means: MPI, a file, torch.distributed.broadcast). */
#define FA_RCCL_UNIQUE_ID_BYTES 128
int fa_rccl_available(void);                 /* 1 when RCCL could be opened */
const char *fa_rccl_library(void);           /* the name it was opened under ("" when unavailable) */
int fa_rccl_unique_id(void *id_out);         /* ncclGetUniqueId: FA_RCCL_UNIQUE_ID_BYTES bytes (host) */
int fa_rccl_comm_create(void **comm_out, int32_t nranks, const void *id, int32_t rank, int32_t device_id);
int fa_rccl_comm_destroy(void *comm);
int fa_rccl_comm_ranks(void *comm);          /* ncclCommCount (>= 1), negative on error */
/* ppo.py:121-123 over ALL ranks: ncclAllGather of this rank's moments (N,3) = {n, mean, M2} (fa_gae_moments /
 * fa_adv_moments) into gathered (world, N, 3) -- caller-owned device scratch -- then the exact merge of
 * fa_adv_merge in rank order: every rank ends with the same bits in mean_out / std_out (N doubles each). */
int fa_adv_allreduce(fa_env *env, const double *moments, double *gathered, void *nccl_comm, double *mean_out,
                     double *std_out, void *stream);
/* The WHOLE several-rank collector tail of one rollout as ONE call (what bench.py --gpus N and a torch-free consumer
 * enqueue per rollout; learner.py:191-211 + ppo.py:121-124 over all ranks): fa_gae_moments (GAE scan + this rank's
 * moments -> `moments`, N x 3), ncclAllGather into `gathered` (world, N, 3), then fa_adv_merge_normalize (merge in rank
 * order + normalisation -> adv_out (T, E, N), mean_out / std_out).  Four device operations enqueued back to back on
 * `stream`, nothing on the host in between; capturable in a hipGraph together with fa_collect_rollout (the collective
 * becomes a graph node: tests/test_gpu_rccl.py).  == fa_gae_moments + fa_adv_allreduce + fa_adv_normalize bit for bit. */
int fa_gae_allreduce_normalize(fa_env *env, double gamma, double tau, void *nccl_comm, double *moments, double *gathered,
                               float *adv_out, double *mean_out, double *std_out, void *stream);
/* ncclAllReduce(sum, float32) of `n` floats in place: the flat gradient buffer of one optimizer step (the
 * un-normalised gradients + loss sums + this rank's alive-mask mean, learner.py GraphedPPOStep). */
int fa_grad_allreduce(float *flat, int64_t n, void *nccl_comm, void *stream);

/* ---- state access (synchronous; tests / checkpoint) ------------------------------ */
int fa_get_state(fa_env *env, const fa_state_host *out);
int fa_set_state(fa_env *env, const fa_state_host *in); /* pos/vel/ang/prev_dist/alive/time_step/num_hit/
                                                           num_was_hit/game_result; NULL fields are left alone */
/* Device self-test: the step kernel's hand-sequenced fp64 divide / sqrt (no range-scaling
 * wrappers) against the compiler's `/` and sqrt() on >= `samples` random operands of the
 * magnitudes the step uses; mismatch_host[0] = differing quotients, [1] = differing roots,
 * [2] = largest deviation of the heading sin/cos from the device libm, in 1/1000 ulp
 * (mismatch_host holds 3 values). */
int fa_selftest_math(fa_env *env, uint64_t samples, uint64_t seed, uint64_t *mismatch_host);
/* Name of the step kernel variant a fa_step / fa_collect_rollout launch of `num_steps` env-steps
 * uses on this env (reporting only: bench.py labels its roofline line with it). */
const char *fa_step_variant(fa_env *env, int32_t num_steps);
/* which fa_policy_kernel shape fa_policy_act / fa_collect_act launch for this handle:
 * "fa_policy_kernel<3, 8>" (96-row tiles, eight waves) or "fa_policy_kernel<2, 4>" (64-row tiles, four waves: small batches) */
const char *fa_policy_variant(fa_env *env);
/* next `count` random_sample() doubles env e would draw (does not advance the stream) */
int fa_rng_peek(fa_env *env, int32_t e, int32_t count, double *out_host);

#ifdef __cplusplus
}
#endif
#endif /* FORTATTACK_H */
