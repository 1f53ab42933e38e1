// fa_api.hip -- host side of the C ABI declared in include/fortattack.h.
// Owns the fp64 SoA world state + RNG state in HBM; everything else belongs to the caller.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "fa_device.h"
#include "experiments/fa_step_experiments.h"
#include "fa_policy.h"
#include "fa_train.h"
#include "fortattack.h"

hipError_t fa_launch_step(const FaStepArgs &a, hipStream_t st);
hipError_t fa_launch_reset(const FaStepArgs &a, hipStream_t st);
const char *fa_step_variant_name(int G, int A, int E, int nsteps, int step_kernel, bool choice);
int fa_step_experiments_linked(); // fa_step_classic.hip: 1 in a variant library that carries csrc/experiments/
hipError_t fa_launch_seed(const FaState &s, int E, uint64_t base_seed, int64_t env_offset, int skip_words,
                          hipStream_t st);
hipError_t fa_launch_selftest(unsigned long long n_per_thread, unsigned long long seed, unsigned long long *mismatch,
                              hipStream_t st);
hipError_t fa_launch_gae(const float *rewards, const float *value_preds, const float *masks, float *returns,
                         const uint8_t *done, int T, int E, int N, double gamma, double tau, hipStream_t st);
int fa_gae_mom_blocks(const float *rewards, const float *value_preds, const float *masks, const float *returns, int T, int E,
                      int N, long long partial_cap);
hipError_t fa_launch_gae_mom(const float *rewards, const float *value_preds, const float *masks, float *returns,
                             const uint8_t *done, int T, int E, int N, double gamma, double tau, double *partial, int nblocks,
                             hipStream_t st);
hipError_t fa_launch_gae_mom_final(const double *partial, int nblocks, const float *rewards, const float *value_preds,
                                   const float *masks, int T, int E, int N, double gamma, double *moments_out, double *mean_out,
                                   double *std_out, hipStream_t st);
hipError_t fa_launch_gae_mom_norm(const double *partial, int nblocks, const float *rewards, const float *value_preds,
                                  const float *masks, const float *returns, int T, int E, int N, double gamma, float *out,
                                  double *moments_out, double *mean_out, double *std_out, int grid, int piv_from_returns,
                                  hipStream_t st);
hipError_t fa_launch_adv_merge_norm(const double *gathered, int W, int N, const float *returns, const float *value_preds,
                                    long long total, float *out, double *mean_out, double *std_out, hipStream_t st);
hipError_t fa_launch_adv_onepass(const float *returns, const float *value_preds, long long rows, int N, double *partial,
                                 int nblocks, double *moments_out, double *mean_out, double *std_out, hipStream_t st,
                                 bool final = true);
hipError_t fa_launch_adv_stats(int pass, const float *returns, const float *value_preds, const double *mean,
                               long long rows, int N, double *partial, int nblocks, double *stats,
                               double *derived, hipStream_t st);
hipError_t fa_launch_adv_merge(const double *gathered, int W, int N, double *mean_out, double *std_out,
                               hipStream_t st);
hipError_t fa_launch_adv_moments_fix(double *stats, int N, hipStream_t st);
hipError_t fa_launch_attend(int width, bool backward, const float *g, const float *keys, float *out, float *attn,
                            const float *dout, float *dg, float *dkeys, int B, int n, int nk, int skip_self, hipStream_t st);
hipError_t fa_launch_adv_norm(const float *returns, const float *value_preds, const double *mean,
                              const double *std_, long long total, int N, float *out, hipStream_t st);

static thread_local std::string g_err;

static int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
int fa_api_fail(int code, const std::string &msg) { return fail(code, msg); } // for the other translation units
#define FA_HIP(expr)                                                                          \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return fail(FA_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));       \
    } while (0)

struct fa_env {
    fa_config cfg;
    int N;
    FaDerived c;
    FaState s;
    void *slab;
    size_t slab_bytes;
    fa_storage st;
    bool bound;
    double *adv_partial; // [ADV_BLOCKS][N]
    double *adv_stats;   // [N][3] scratch of fa_adv_mean_std
    int adv_blocks;
    int choice_k;        // fa_set_reset_choice
    int32_t *choice_out;
    int32_t *grp_env_list, *grp_tile_strategy; // fused policy with an attacker ensemble: envs grouped by strategy
    int grp_tiles_max;
};

namespace {
struct DeviceGuard {
    int prev;
    bool changed;
    explicit DeviceGuard(int dev) : prev(-1), changed(false) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) changed = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (changed) (void)hipSetDevice(prev);
    }
};

// Python float modulo (core.py:336: u[2] % (2*pi)): result takes the sign of the divisor.
double py_mod(double a, double b) {
    double r = std::fmod(a, b);
    if (r != 0.0 && ((b < 0) != (r < 0))) r += b;
    return r;
}

FaDerived derive(const fa_world_consts &w) {
    const double pi = 3.141592653589793; // np.pi
    FaDerived d;
    d.agent_size = w.agent_size;
    d.accel = w.accel;
    d.max_speed = w.max_speed;
    d.fort_dim = w.fort_dim;
    d.door_x = w.door_x;
    d.door_y = w.door_y;
    d.dt = w.dt;
    d.one_minus_damping = 1 - w.damping;           // core.py:328
    d.contact_force = w.contact_force;
    d.contact_margin = w.contact_margin;
    d.dist_min = w.agent_size + w.agent_size;      // core.py:449
    d.wall_xmin = w.wall_xmin;
    d.wall_xmax = w.wall_xmax;
    d.wall_ymin = w.wall_ymin;
    d.wall_ymax = w.wall_ymax;
    d.shoot_rad = w.shoot_rad;
    d.half_win = w.shoot_win / 2;                  // core.py:376
    d.cos_hw = std::cos(d.half_win);
    d.sin_hw = std::sin(d.half_win);
    d.shoot_far = w.shoot_rad * d.cos_hw;
    {   // largest double x with sqrt(x) <= max_speed (host sqrt is correctly rounded)
        double x = w.max_speed * w.max_speed;
        while (std::sqrt(x) > w.max_speed) x = std::nextafter(x, 0.0);
        while (std::sqrt(std::nextafter(x, INFINITY)) <= w.max_speed) x = std::nextafter(x, INFINITY);
        d.speed2_max = x;
    }
    {   // `sqrt(dd) < fort_dim` decided on dd: the largest double whose correctly rounded sqrt is < fort_dim
        double x = w.fort_dim * w.fort_dim;
        while (std::sqrt(x) >= w.fort_dim) x = std::nextafter(x, 0.0);
        while (std::sqrt(std::nextafter(x, INFINITY)) < w.fort_dim) x = std::nextafter(x, INFINITY);
        d.fort2_max = x;
    }
    d.rot_pos = py_mod(+w.max_rot, 2 * pi);        // core.py:336
    d.rot_neg = py_mod(-w.max_rot, 2 * pi);
    d.ang_guard = 3 * pi / 2;                      // fortattack_env_v1.py:59
    d.ang_attacker = pi / 2;
    // fortattack_env_v1.py:66  uniform(xMin,xMax), uniform(yMin, 0.8*yMin): lo + (hi-lo)*u
    d.att_x_lo = w.wall_xmin;
    d.att_x_rng = w.wall_xmax - w.wall_xmin;
    d.att_y_lo = w.wall_ymin;
    d.att_y_rng = 0.8 * w.wall_ymin - w.wall_ymin;
    // fortattack_env_v1.py:70  uniform(-0.8*fortDim/2, 0.8*fortDim/2), uniform(0.8*yMax, yMax)
    d.grd_x_lo = -0.8 * w.fort_dim / 2;
    d.grd_x_rng = 0.8 * w.fort_dim / 2 - d.grd_x_lo;
    d.grd_y_lo = 0.8 * w.wall_ymax;
    d.grd_y_rng = w.wall_ymax - d.grd_y_lo;
    // exact-zero contact shortcuts: clearance > 1000 * margin => exp(-1000) == +0.0
    const double clr = 1000.0 * w.contact_margin;
    d.contact_skip_d2 = (d.dist_min + clr) * (d.dist_min + clr) * (1.0 + 1e-12);
    d.wall_skip = clr;
    return d;
}

size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

FaStepArgs base_args(const fa_env *env) {
    FaStepArgs a;
    std::memset(&a, 0, sizeof(a));
    a.s = env->s;
    a.E = env->cfg.num_envs;
    a.G = env->cfg.num_guards;
    a.A = env->cfg.num_attackers;
    a.max_t = env->cfg.max_time_steps;
    a.rng_mode = env->cfg.rng_mode;
    a.track_counters = env->cfg.track_counters;
    a.step_kernel = env->cfg.step_kernel;
    a.choice_k = env->choice_k;
    a.choice_out = env->choice_out;
    a.seed = env->cfg.base_seed;
    a.env_offset = env->cfg.env_offset;
    a.c = env->c;
    a.nsteps = 1;
    return a;
}
} // namespace

extern "C" {

const char *fa_last_error(void) { return g_err.c_str(); }

int fa_config_default(fa_config *cfg) {
    if (!cfg) return fail(FA_ERR_INVALID, "fa_config_default: null cfg");
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->num_envs = 1;
    cfg->num_guards = 3;
    cfg->num_attackers = 3;
    cfg->max_time_steps = 100; // arguments.py:24 --num-env-steps
    cfg->device_id = 0;
    cfg->rng_mode = FA_RNG_MT19937;
    cfg->base_seed = 0;
    cfg->env_offset = 0;
    cfg->rng_skip_doubles = -1;
    cfg->track_counters = 1;
    cfg->step_kernel = FA_KERNEL_AUTO;
    fa_world_consts &w = cfg->world;
    w.agent_size = 0.05;
    w.accel = 3;
    w.max_speed = 3;
    w.max_rot = 0.17;
    w.fort_dim = 0.15;
    w.door_x = 0;
    w.door_y = 0.8;
    w.dt = 0.1;
    w.damping = 0.25;
    w.contact_force = 1e+2;
    w.contact_margin = 1e-10;
    w.wall_xmin = -1;
    w.wall_xmax = 1;
    w.wall_ymin = -0.8;
    w.wall_ymax = 0.8;
    w.shoot_rad = 0.8;
    w.shoot_win = 3.141592653589793 / 4;
    return FA_OK;
}

int fa_create(const fa_config *cfg, fa_env **out) {
    if (!cfg || !out) return fail(FA_ERR_INVALID, "fa_create: null argument");
    const int N = cfg->num_guards + cfg->num_attackers;
    if (cfg->num_envs < 1) return fail(FA_ERR_INVALID, "fa_create: num_envs must be >= 1");
    if (cfg->num_guards < 1 || cfg->num_attackers < 1 || N > FA_MAX_AGENTS)
        return fail(FA_ERR_INVALID, "fa_create: need >=1 guard, >=1 attacker, <=16 agents");
    if (cfg->max_time_steps < 1) return fail(FA_ERR_INVALID, "fa_create: max_time_steps must be >= 1");
    if (cfg->rng_mode != FA_RNG_MT19937 && cfg->rng_mode != FA_RNG_PHILOX)
        return fail(FA_ERR_INVALID, "fa_create: unknown rng_mode");
    const bool experiment = cfg->step_kernel == FA_KERNEL_EXP_PAIRS || cfg->step_kernel == FA_KERNEL_EXP_CHAIN;
    if (experiment && !fa_step_experiments_linked())
        return fail(FA_ERR_INVALID, "fa_create: this library does not carry the experiment step kernels (build a variant: "
                                    "tools/build_variant.py experiments --add experiments/fa_step_experiments.hip)");
    if (!experiment && (cfg->step_kernel < FA_KERNEL_AUTO || cfg->step_kernel > FA_KERNEL_WAVES3))
        return fail(FA_ERR_INVALID, "fa_create: unknown step_kernel");
    if (cfg->step_kernel != FA_KERNEL_AUTO && cfg->step_kernel != FA_KERNEL_WAVES1 &&
        !((cfg->num_guards == 3 && cfg->num_attackers == 3) || (cfg->num_guards == 5 && cfg->num_attackers == 5)))
        return fail(FA_ERR_INVALID, "fa_create: the multi-wave step kernels exist for 3v3 and 5v5 only");
    if (cfg->step_kernel == FA_KERNEL_EXP_PAIRS && !(cfg->num_guards == 3 && cfg->num_attackers == 3))
        return fail(FA_ERR_INVALID, "fa_create: the pair-per-lane step kernel exists for 3v3 only");
    if (!(cfg->world.contact_margin > 0) || !(cfg->world.agent_size > 0))
        return fail(FA_ERR_INVALID, "fa_create: contact_margin and agent_size must be > 0");
    int ndev = 0;
    FA_HIP(hipGetDeviceCount(&ndev));
    if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(FA_ERR_INVALID, "fa_create: bad device_id");
    DeviceGuard guard(cfg->device_id);

    fa_env *env = new fa_env();
    std::memset(static_cast<void *>(env), 0, sizeof(*env));
    env->cfg = *cfg;
    env->N = N;
    if (env->cfg.rng_skip_doubles < 0) env->cfg.rng_skip_doubles = 2 * N;
    env->c = derive(cfg->world);
    env->adv_blocks = 1024;

    const size_t E = (size_t)cfg->num_envs, EN = E * N;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes); return o; };
    const size_t o_f64 = carve(6 * EN * sizeof(double));
    const size_t o_alive = carve(EN);
    const size_t o_tstep = carve(E * sizeof(int32_t));
    const size_t o_nh = carve(2 * EN * sizeof(int32_t));
    const size_t o_gr = carve(E * 3);
    const size_t o_rc = carve(E * 3 * sizeof(uint32_t));
    const size_t o_mt = carve(E * FA_MT_N * sizeof(uint32_t));
    const size_t o_pos = carve(E * sizeof(int32_t));
    const size_t o_cnt = carve(E * sizeof(uint32_t));
    const size_t o_epr = carve(2 * EN * sizeof(double));
    const size_t o_ale = carve(EN * sizeof(uint32_t));
    const size_t o_adv = carve((size_t)env->adv_blocks * FA_MAX_AGENTS * 2 * sizeof(double));
    const size_t o_advs = carve((size_t)FA_MAX_AGENTS * 5 * sizeof(double)); // stats (3) + mean + std scratch
    const bool fused_ok = cfg->num_guards <= FA_POLICY_MAX_TEAM && cfg->num_attackers <= FA_POLICY_MAX_TEAM;
    // sized for the smallest tile the policy kernel may choose (64 rows), slots counted at the largest (96)
    const int n_max = cfg->num_guards > cfg->num_attackers ? cfg->num_guards : cfg->num_attackers;
    const int tile_envs = fused_ok ? 64 / n_max : 1;
    env->grp_tiles_max = fused_ok ? (cfg->num_envs + tile_envs - 1) / tile_envs + FA_POLICY_MAX_POOL : 0;
    const size_t o_gel = carve((size_t)env->grp_tiles_max * (FA_POLICY_ROWS / n_max + 1) * sizeof(int32_t));
    const size_t o_gts = carve((size_t)env->grp_tiles_max * sizeof(int32_t));
    env->slab_bytes = off;
    hipError_t he = hipMalloc(&env->slab, env->slab_bytes);
    if (he != hipSuccess) {
        delete env;
        return fail(FA_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(he));
    }
    char *b = static_cast<char *>(env->slab);
    double *f = reinterpret_cast<double *>(b + o_f64);
    env->s.px = f;
    env->s.py = f + EN;
    env->s.vx = f + 2 * EN;
    env->s.vy = f + 3 * EN;
    env->s.ang = f + 4 * EN;
    env->s.prev = f + 5 * EN;
    env->s.alive = reinterpret_cast<uint8_t *>(b + o_alive);
    env->s.tstep = reinterpret_cast<int32_t *>(b + o_tstep);
    env->s.num_hit = reinterpret_cast<int32_t *>(b + o_nh);
    env->s.num_was_hit = env->s.num_hit + EN;
    env->s.game_result = reinterpret_cast<uint8_t *>(b + o_gr);
    env->s.result_count = reinterpret_cast<uint32_t *>(b + o_rc);
    env->s.mt = reinterpret_cast<uint32_t *>(b + o_mt);
    env->s.mt_pos = reinterpret_cast<int32_t *>(b + o_pos);
    env->s.reset_count = reinterpret_cast<uint32_t *>(b + o_cnt);
    env->s.ep_rew = reinterpret_cast<double *>(b + o_epr);
    env->s.ep_rew_sum = env->s.ep_rew + EN;
    env->s.alive_end = reinterpret_cast<uint32_t *>(b + o_ale);
    env->adv_partial = reinterpret_cast<double *>(b + o_adv);
    env->adv_stats = reinterpret_cast<double *>(b + o_advs);
    env->grp_env_list = reinterpret_cast<int32_t *>(b + o_gel);
    env->grp_tile_strategy = reinterpret_cast<int32_t *>(b + o_gts);

    // construction state (core.py:102-104): alive, prevDist None (NaN); positions are
    // defined by the first reset.
    he = hipMemset(env->slab, 0, env->slab_bytes);
    if (he == hipSuccess) he = hipMemset(env->s.alive, 1, EN);
    if (he == hipSuccess) {
        std::vector<double> nan(EN, std::nan(""));
        he = hipMemcpy(env->s.prev, nan.data(), EN * sizeof(double), hipMemcpyHostToDevice);
    }
    if (he == hipSuccess)
        he = fa_launch_seed(env->s, cfg->num_envs, cfg->base_seed, cfg->env_offset,
                            2 * env->cfg.rng_skip_doubles, nullptr);
    if (he == hipSuccess) he = hipDeviceSynchronize();
    if (he != hipSuccess) {
        (void)hipFree(env->slab);
        delete env;
        return fail(FA_ERR_HIP, std::string("fa_create init: ") + hipGetErrorString(he));
    }
    *out = env;
    return FA_OK;
}

void fa_destroy(fa_env *env) {
    if (!env) return;
    DeviceGuard guard(env->cfg.device_id);
    (void)hipFree(env->slab);
    delete env;
}

int fa_num_agents(const fa_env *env) { return env ? env->N : FA_ERR_INVALID; }
int fa_num_envs(const fa_env *env) { return env ? env->cfg.num_envs : FA_ERR_INVALID; }

int fa_reset(fa_env *env, const uint8_t *env_mask, float *obs_f32, double *obs_f64, void *stream) {
    if (!env) return fail(FA_ERR_INVALID, "fa_reset: null env");
    DeviceGuard guard(env->cfg.device_id);
    FaStepArgs a = base_args(env);
    a.reset_mask = env_mask;
    a.obs32 = obs_f32;
    a.obs64 = obs_f64;
    FA_HIP(fa_launch_reset(a, static_cast<hipStream_t>(stream)));
    return FA_OK;
}

int fa_step(fa_env *env, const fa_step_io *io, void *stream) {
    if (!env || !io) return fail(FA_ERR_INVALID, "fa_step: null argument");
    if (!io->actions) return fail(FA_ERR_INVALID, "fa_step: actions is null");
    DeviceGuard guard(env->cfg.device_id);
    FaStepArgs a = base_args(env);
    a.actions = io->actions;
    a.as_e = io->act_stride_env;
    a.as_i = io->act_stride_agent;
    a.as_t = io->act_stride_step;
    a.nsteps = io->num_steps > 1 ? io->num_steps : 1;
    a.obs32 = io->obs_f32;
    a.rew32 = io->reward_f32;
    a.mask32 = io->mask_f32;
    a.done = io->done;
    a.obs64 = io->obs_f64;
    a.rew64 = io->reward_f64;
    a.hit = io->hit;
    a.was_hit = io->was_hit;
    a.auto_reset = io->auto_reset;
    FA_HIP(fa_launch_step(a, static_cast<hipStream_t>(stream)));
    return FA_OK;
}

int fa_set_reset_choice(fa_env *env, int32_t k, int32_t *choice_out) {
    if (!env) return fail(FA_ERR_INVALID, "fa_set_reset_choice: null env");
    if (k < 0 || (k > 0 && !choice_out)) return fail(FA_ERR_INVALID, "fa_set_reset_choice: k >= 0, and k > 0 needs choice_out");
    env->choice_k = k;
    env->choice_out = k > 0 ? choice_out : nullptr;
    return FA_OK;
}

int fa_bind_storage(fa_env *env, const fa_storage *st) {
    if (!env || !st) return fail(FA_ERR_INVALID, "fa_bind_storage: null argument");
    if (st->num_steps < 1) return fail(FA_ERR_INVALID, "fa_bind_storage: num_steps must be >= 1");
    if (!st->obs || !st->rewards || !st->value_preds || !st->returns || !st->actions || !st->masks ||
        !st->done)
        return fail(FA_ERR_INVALID, "fa_bind_storage: obs/rewards/value_preds/returns/actions/masks/done required");
    env->st = *st;
    env->bound = true;
    return FA_OK;
}

int fa_collect_step(fa_env *env, int32_t step, int32_t auto_reset, void *stream) {
    return fa_collect_rollout(env, step, 1, auto_reset, stream);
}

int fa_collect_rollout(fa_env *env, int32_t step, int32_t num_steps, int32_t auto_reset, void *stream) {
    if (!env) return fail(FA_ERR_INVALID, "fa_collect_rollout: null env");
    if (!env->bound) return fail(FA_ERR_STATE, "fa_collect_rollout: no storage bound");
    const fa_storage &st = env->st;
    if (step < 0 || num_steps < 1 || step + num_steps > st.num_steps)
        return fail(FA_ERR_INVALID, "fa_collect_rollout: step range outside the bound storage");
    DeviceGuard guard(env->cfg.device_id);
    const size_t EN = (size_t)env->cfg.num_envs * env->N;
    FaStepArgs a = base_args(env);
    a.actions = st.actions + (size_t)step * EN;
    a.as_e = env->N;
    a.as_i = 1;
    a.as_t = (int64_t)EN;
    a.nsteps = num_steps;
    a.obs32 = st.obs + (size_t)(step + 1) * EN * FA_OBS_DIM;
    a.rew32 = st.rewards + (size_t)step * EN;
    a.mask32 = st.masks + (size_t)(step + 1) * EN;
    a.done = st.done + (size_t)step * env->cfg.num_envs;
    a.auto_reset = auto_reset;
    FA_HIP(fa_launch_step(a, static_cast<hipStream_t>(stream)));
    return FA_OK;
}

int fa_collect_reset(fa_env *env, void *stream) {
    if (!env) return fail(FA_ERR_INVALID, "fa_collect_reset: null env");
    if (!env->bound) return fail(FA_ERR_STATE, "fa_collect_reset: no storage bound");
    DeviceGuard guard(env->cfg.device_id);
    hipStream_t s = static_cast<hipStream_t>(stream);
    FaStepArgs a = base_args(env);
    a.obs32 = env->st.obs;
    FA_HIP(fa_launch_reset(a, s));
    // masks[0] = 1 (every agent alive after a reset): float 1.0f fill
    const size_t EN = (size_t)env->cfg.num_envs * env->N;
    FA_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(env->st.masks), 0x3f800000, EN, s));
    return FA_OK;
}

int fa_gae(fa_env *env, double gamma, double tau, void *stream) {
    if (!env) return fail(FA_ERR_INVALID, "fa_gae: null env");
    if (!env->bound) return fail(FA_ERR_STATE, "fa_gae: no storage bound");
    DeviceGuard guard(env->cfg.device_id);
    const fa_storage &st = env->st;
    FA_HIP(fa_launch_gae(st.rewards, st.value_preds, st.masks, st.returns, st.done, st.num_steps,
                         env->cfg.num_envs, env->N, gamma, tau, static_cast<hipStream_t>(stream)));
    return FA_OK;
}

// the fused scan's workgroup count for this handle's storage, 0 where the separate kernels serve (see fa_gae_mom_blocks;
// FA_GAE_MOMENTS_SEPARATE=1 in the environment forces them: the A/B switch of tools/ab_tail.py)
static int gae_mom_blocks(const fa_env *env) {
    static const bool separate = [] { const char *v = getenv("FA_GAE_MOMENTS_SEPARATE"); return v && v[0] == '1'; }();
    if (separate) return 0;
    const fa_storage &st = env->st;
    return fa_gae_mom_blocks(st.rewards, st.value_preds, st.masks, st.returns, st.num_steps, env->cfg.num_envs, env->N,
                             (long long)env->adv_blocks * FA_MAX_AGENTS * 2);
}

int fa_gae_moments(fa_env *env, double gamma, double tau, double *moments_out, double *mean_out, double *std_out,
                   void *stream) {
    if (!env) return fail(FA_ERR_INVALID, "fa_gae_moments: null env");
    if (!env->bound) return fail(FA_ERR_STATE, "fa_gae_moments: no storage bound");
    DeviceGuard guard(env->cfg.device_id);
    const fa_storage &st = env->st;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (const int nb = gae_mom_blocks(env)) {
        // two launches: the scan leaves the workgroups' moment partials, one workgroup folds them
        FA_HIP(fa_launch_gae_mom(st.rewards, st.value_preds, st.masks, st.returns, st.done, st.num_steps, env->cfg.num_envs,
                                 env->N, gamma, tau, env->adv_partial, nb, s));
        FA_HIP(fa_launch_gae_mom_final(env->adv_partial, nb, st.rewards, st.value_preds, st.masks, st.num_steps,
                                       env->cfg.num_envs, env->N, gamma, moments_out, mean_out, std_out, s));
        return FA_OK;
    }
    FA_HIP(fa_launch_gae(st.rewards, st.value_preds, st.masks, st.returns, st.done, st.num_steps,
                         env->cfg.num_envs, env->N, gamma, tau, s));
    const long long rows = (long long)st.num_steps * env->cfg.num_envs;
    // a lane takes 2 rows x 4 per trip: one workgroup per 2048 rows, at most two per CU
    long long want = (rows + 2047) / 2048;
    const int nblocks = (int)(want < 512 ? (want < 1 ? 1 : want) : 512);
    FA_HIP(fa_launch_adv_onepass(st.returns, st.value_preds, rows, env->N, env->adv_partial, nblocks, moments_out,
                                 mean_out, std_out, s));
    return FA_OK;
}

int fa_gae_normalize(fa_env *env, double gamma, double tau, float *adv_out, double *moments_out, double *mean_out,
                     double *std_out, void *stream) {
    if (!env || !adv_out) return fail(FA_ERR_INVALID, "fa_gae_normalize: null argument");
    if (!env->bound) return fail(FA_ERR_STATE, "fa_gae_normalize: no storage bound");
    DeviceGuard guard(env->cfg.device_id);
    const fa_storage &st = env->st;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (const int nb = gae_mom_blocks(env)) {
        // two launches: scan + moment partials; fold + normalisation by every workgroup
        FA_HIP(fa_launch_gae_mom(st.rewards, st.value_preds, st.masks, st.returns, st.done, st.num_steps, env->cfg.num_envs,
                                 env->N, gamma, tau, env->adv_partial, nb, s));
        FA_HIP(fa_launch_gae_mom_norm(env->adv_partial, nb, st.rewards, st.value_preds, st.masks, st.returns, st.num_steps,
                                      env->cfg.num_envs, env->N, gamma, adv_out, moments_out, mean_out, std_out, 0, 0, s));
        return FA_OK;
    }
    // beyond the fused scan: fa_gae, the one-pass sweep (at most 512 workgroups of partials), and the same fold +
    // normalisation launch on its partials -- three launches instead of four, nothing dirty left behind the last one
    FA_HIP(fa_launch_gae(st.rewards, st.value_preds, st.masks, st.returns, st.done, st.num_steps, env->cfg.num_envs, env->N, gamma,
                         tau, s));
    const long long rows = (long long)st.num_steps * env->cfg.num_envs;
    long long want = (rows + 2047) / 2048;   // as fa_gae_moments
    const int nblocks = (int)(want < 512 ? (want < 1 ? 1 : want) : 512);
    FA_HIP(fa_launch_adv_onepass(st.returns, st.value_preds, rows, env->N, env->adv_partial, nblocks, nullptr, nullptr, nullptr, s,
                                 false));
    FA_HIP(fa_launch_gae_mom_norm(env->adv_partial, nblocks, st.rewards, st.value_preds, st.masks, st.returns, st.num_steps,
                                  env->cfg.num_envs, env->N, gamma, adv_out, moments_out, mean_out, std_out, 0, 1, s));
    return FA_OK;
}

int fa_adv_moments_onepass(fa_env *env, double *moments_out, double *mean_out, double *std_out, void *stream) {
    if (!env) return fail(FA_ERR_INVALID, "fa_adv_moments_onepass: null env");
    if (!env->bound) return fail(FA_ERR_STATE, "fa_adv_moments_onepass: no storage bound");
    DeviceGuard guard(env->cfg.device_id);
    const fa_storage &st = env->st;
    const long long rows = (long long)st.num_steps * env->cfg.num_envs;
    long long want = (rows + 2047) / 2048;   // as fa_gae_moments: a lane takes 2 rows x 4 per trip, at most two workgroups per CU
    const int nblocks = (int)(want < 512 ? (want < 1 ? 1 : want) : 512);
    FA_HIP(fa_launch_adv_onepass(st.returns, st.value_preds, rows, env->N, env->adv_partial, nblocks, moments_out, mean_out,
                                 std_out, static_cast<hipStream_t>(stream)));
    return FA_OK;
}

int fa_adv_stats(fa_env *env, int32_t pass, const double *mean, double *stats, void *stream) {
    if (!env || !stats) return fail(FA_ERR_INVALID, "fa_adv_stats: null argument");
    if (!env->bound) return fail(FA_ERR_STATE, "fa_adv_stats: no storage bound");
    if (pass != 0 && pass != 1) return fail(FA_ERR_INVALID, "fa_adv_stats: pass must be 0 or 1");
    if (pass == 1 && !mean) return fail(FA_ERR_INVALID, "fa_adv_stats: pass 1 needs mean");
    DeviceGuard guard(env->cfg.device_id);
    const fa_storage &st = env->st;
    const long long rows = (long long)st.num_steps * env->cfg.num_envs;
    long long want = (rows + 255) / 256;
    const int nblocks = (int)(want < env->adv_blocks ? want : env->adv_blocks);
    FA_HIP(fa_launch_adv_stats(pass, st.returns, st.value_preds, mean, rows, env->N, env->adv_partial,
                               nblocks, stats, nullptr, static_cast<hipStream_t>(stream)));
    return FA_OK;
}

int fa_adv_mean_std(fa_env *env, double *mean_out, double *std_out, void *stream) {
    if (!env || !mean_out || !std_out) return fail(FA_ERR_INVALID, "fa_adv_mean_std: null argument");
    if (!env->bound) return fail(FA_ERR_STATE, "fa_adv_mean_std: no storage bound");
    DeviceGuard guard(env->cfg.device_id);
    const fa_storage &st = env->st;
    const long long rows = (long long)st.num_steps * env->cfg.num_envs;
    long long want = (rows + 255) / 256;
    const int nblocks = (int)(want < env->adv_blocks ? want : env->adv_blocks);
    hipStream_t s = static_cast<hipStream_t>(stream);
    FA_HIP(fa_launch_adv_stats(0, st.returns, st.value_preds, nullptr, rows, env->N, env->adv_partial, nblocks,
                               env->adv_stats, mean_out, s));
    FA_HIP(fa_launch_adv_stats(1, st.returns, st.value_preds, mean_out, rows, env->N, env->adv_partial, nblocks,
                               env->adv_stats, std_out, s));
    return FA_OK;
}

int fa_adv_moments(fa_env *env, double *moments_out, void *stream) {
    if (!env || !moments_out) return fail(FA_ERR_INVALID, "fa_adv_moments: null argument");
    if (!env->bound) return fail(FA_ERR_STATE, "fa_adv_moments: no storage bound");
    DeviceGuard guard(env->cfg.device_id);
    const fa_storage &st = env->st;
    const long long rows = (long long)st.num_steps * env->cfg.num_envs;
    long long want = (rows + 255) / 256;
    const int nblocks = (int)(want < env->adv_blocks ? want : env->adv_blocks);
    hipStream_t s = static_cast<hipStream_t>(stream);
    double *mean = env->adv_stats + 3 * FA_MAX_AGENTS; // scratch: local mean
    FA_HIP(fa_launch_adv_stats(0, st.returns, st.value_preds, nullptr, rows, env->N, env->adv_partial, nblocks,
                               moments_out, mean, s));
    FA_HIP(fa_launch_adv_stats(1, st.returns, st.value_preds, mean, rows, env->N, env->adv_partial, nblocks,
                               moments_out, nullptr, s));
    FA_HIP(fa_launch_adv_moments_fix(moments_out, env->N, s));
    return FA_OK;
}

int fa_adv_merge(fa_env *env, const double *gathered, int32_t world, double *mean_out, double *std_out,
                 void *stream) {
    if (!env || !gathered || !mean_out || !std_out) return fail(FA_ERR_INVALID, "fa_adv_merge: null argument");
    if (world < 1) return fail(FA_ERR_INVALID, "fa_adv_merge: world must be >= 1");
    DeviceGuard guard(env->cfg.device_id);
    FA_HIP(fa_launch_adv_merge(gathered, world, env->N, mean_out, std_out, static_cast<hipStream_t>(stream)));
    return FA_OK;
}

int fa_adv_merge_normalize(fa_env *env, const double *gathered, int32_t world, float *adv_out, double *mean_out, double *std_out,
                           void *stream) {
    if (!env || !gathered || !adv_out) return fail(FA_ERR_INVALID, "fa_adv_merge_normalize: null argument");
    if (world < 1) return fail(FA_ERR_INVALID, "fa_adv_merge_normalize: world must be >= 1");
    if (!env->bound) return fail(FA_ERR_STATE, "fa_adv_merge_normalize: no storage bound");
    DeviceGuard guard(env->cfg.device_id);
    const fa_storage &st = env->st;
    const long long total = (long long)st.num_steps * env->cfg.num_envs * env->N;
    FA_HIP(fa_launch_adv_merge_norm(gathered, world, env->N, st.returns, st.value_preds, total, adv_out, mean_out, std_out,
                                    static_cast<hipStream_t>(stream)));
    return FA_OK;
}

int fa_adv_normalize(fa_env *env, const double *mean, const double *std_, float *adv_out, void *stream) {
    if (!env || !mean || !std_ || !adv_out) return fail(FA_ERR_INVALID, "fa_adv_normalize: null argument");
    if (!env->bound) return fail(FA_ERR_STATE, "fa_adv_normalize: no storage bound");
    DeviceGuard guard(env->cfg.device_id);
    const fa_storage &st = env->st;
    const long long total = (long long)st.num_steps * env->cfg.num_envs * env->N;
    FA_HIP(fa_launch_adv_norm(st.returns, st.value_preds, mean, std_, total, env->N, adv_out,
                              static_cast<hipStream_t>(stream)));
    return FA_OK;
}

int fa_after_update(fa_env *env, void *stream) {
    if (!env) return fail(FA_ERR_INVALID, "fa_after_update: null env");
    if (!env->bound) return fail(FA_ERR_STATE, "fa_after_update: no storage bound");
    DeviceGuard guard(env->cfg.device_id);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const fa_storage &st = env->st;
    const size_t EN = (size_t)env->cfg.num_envs * env->N, T = (size_t)st.num_steps;
    // storage.py:52-56: obs[0] = obs[-1]; obs[1:] = 0; hidden[0] = hidden[-1]; masks[0] = masks[-1]
    FA_HIP(hipMemcpyAsync(st.obs, st.obs + T * EN * FA_OBS_DIM, EN * FA_OBS_DIM * sizeof(float),
                          hipMemcpyDeviceToDevice, s));
    FA_HIP(hipMemsetAsync(st.obs + EN * FA_OBS_DIM, 0, T * EN * FA_OBS_DIM * sizeof(float), s));
    if (st.recurrent_hidden_states)
        FA_HIP(hipMemcpyAsync(st.recurrent_hidden_states, st.recurrent_hidden_states + T * EN,
                              EN * sizeof(float), hipMemcpyDeviceToDevice, s));
    FA_HIP(hipMemcpyAsync(st.masks, st.masks + T * EN, EN * sizeof(float), hipMemcpyDeviceToDevice, s));
    return FA_OK;
}

static int policy_launch(fa_env *env, const float *obs, const float *wg, const float *wa, float *value, int64_t *action,
                         float *logp, const int64_t *counter, uint64_t seed, int step, int deterministic, int value_only,
                         const float *pool, int pool_size, const int32_t *env_strategy, void *stream, const char *who) {
    if (!obs || !wg || (!wa && pool_size <= 0))
        return fail(FA_ERR_INVALID, std::string(who) + ": obs and both teams' weight buffers are required");
    if (!value_only && (!action || !logp)) return fail(FA_ERR_INVALID, std::string(who) + ": action and log_prob are required");
    if (value_only && !value) return fail(FA_ERR_INVALID, std::string(who) + ": value_only needs value");
    if (env->cfg.num_guards > FA_POLICY_MAX_TEAM || env->cfg.num_attackers > FA_POLICY_MAX_TEAM)
        return fail(FA_ERR_INVALID, std::string(who) + ": teams of more than 8 agents are not supported by the fused policy");
    if (pool_size > 0 && (!pool || !env_strategy || pool_size > FA_POLICY_MAX_POOL))
        return fail(FA_ERR_INVALID, std::string(who) + ": an attacker pool needs weights, env_strategy and <= 64 strategies");
    DeviceGuard guard(env->cfg.device_id);
    hipStream_t s = static_cast<hipStream_t>(stream);
    FaPolicyArgs a;
    std::memset(&a, 0, sizeof(a));
    a.obs = obs;
    a.w[0] = wg;
    a.w[1] = wa;
    a.value = value;
    a.action = action;
    a.logp = logp;
    a.counter = counter;
    a.seed = seed;
    a.env_offset = env->cfg.env_offset;
    a.E = env->cfg.num_envs;
    a.G = env->cfg.num_guards;
    a.A = env->cfg.num_attackers;
    a.step = step;
    a.deterministic = deterministic;
    a.value_only = value_only;
    if (pool_size > 0) { // group the envs into tiles of equal strategy, then one launch over the grouped tiles
        FA_HIP(fa_launch_group_envs(env_strategy, a.E, pool_size, a.G, a.A, env->grp_env_list, env->grp_tile_strategy,
                                    env->grp_tiles_max, s));
        a.pool = pool;
        a.env_list = env->grp_env_list;
        a.tile_strategy = env->grp_tile_strategy;
        a.tiles = env->grp_tiles_max;
    }
    FA_HIP(fa_launch_policy(a, s));
    return FA_OK;
}

int fa_policy_act(fa_env *env, const fa_policy_io *io, void *stream) {
    if (!env || !io) return fail(FA_ERR_INVALID, "fa_policy_act: null argument");
    return policy_launch(env, io->obs, io->weights[0], io->weights[1], io->value, io->action, io->log_prob, io->counter,
                         io->seed, io->step, io->deterministic, io->value_only, io->attacker_pool, io->pool_size,
                         io->env_strategy, stream, "fa_policy_act");
}

int fa_collect_act(fa_env *env, int32_t step, const fa_policy_io *io, void *stream) {
    if (!env || !io) return fail(FA_ERR_INVALID, "fa_collect_act: null argument");
    if (!env->bound) return fail(FA_ERR_STATE, "fa_collect_act: no storage bound");
    const fa_storage &st = env->st;
    const int value_only = io->value_only;
    if (step < 0 || step > st.num_steps || (!value_only && step == st.num_steps))
        return fail(FA_ERR_INVALID, "fa_collect_act: step outside the bound storage");
    if (!value_only && !st.action_log_probs) return fail(FA_ERR_STATE, "fa_collect_act: storage has no action_log_probs");
    const size_t EN = (size_t)env->cfg.num_envs * env->N;
    return policy_launch(env, st.obs + (size_t)step * EN * FA_OBS_DIM, io->weights[0], io->weights[1],
                         st.value_preds + (size_t)step * EN, value_only ? nullptr : st.actions + (size_t)step * EN,
                         value_only ? nullptr : st.action_log_probs + (size_t)step * EN, io->counter, io->seed, step,
                         io->deterministic, value_only, io->attacker_pool, io->pool_size, io->env_strategy, stream,
                         "fa_collect_act");
}

int64_t fa_policy_weight_floats(void) { return FA_POLICY_WEIGHT_FLOATS; }
int64_t fa_policy_plain_floats(void) { return FA_POLICY_PLAIN_FLOATS; }

int64_t fa_ppo_grad_floats(void) { return FA_SLAB_FLOATS; }
int64_t fa_adam_scratch_floats(void) { return FA_ADAM_SCRATCH; }
int64_t fa_policy_weight_t_floats(void) { return FA_TRANS_FLOATS; }

int fa_ppo_grad_scratch(int32_t B, int32_t G, int32_t A, int64_t *slab_floats, int64_t *hsave_floats) {
    if (B < 1 || G < 1 || A < 1 || G > FA_POLICY_MAX_TEAM || A > FA_POLICY_MAX_TEAM)
        return fail(FA_ERR_INVALID, "fa_ppo_grad_scratch: need B >= 1 and teams of 1..8");
    const int et = fa_train_tile_envs(G, A);
    const int64_t tiles = (B + et - 1) / et;
    // slabs: the tiles' small gradients | their stage-1 sums | the partial slabs of the weight-gradient GEMM | mask sums
    if (slab_floats) *slab_floats = tiles * FA_MSLAB_FLOATS + (int64_t)FA_MRED_PARTS * FA_MSLAB_FLOATS +
                                    (int64_t)FA_DW_WGS_A * FA_DWA_FLOATS + (int64_t)FA_DW_WGS_B * FA_DWB_FLOATS + FA_MASK_PARTS;
    // hsave: the records of the weight-gradient GEMM's operands: [tiles][3] x A, then [tiles] x B
    if (hsave_floats) *hsave_floats = tiles * FA_TR_SAVE_FLOATS;
    return FA_OK;
}

int fa_ppo_grad(const fa_ppo_grad_io *io, void *stream) {
    if (!io) return fail(FA_ERR_INVALID, "fa_ppo_grad: null io");
    if (!io->obs || !io->action || !io->value_pred || !io->ret || !io->old_log_prob || !io->weights ||
        !io->weights_t || !io->slabs || !io->hsave || !io->out)
        return fail(FA_ERR_INVALID, "fa_ppo_grad: every pointer but idx, scale and adv / adv_mean / adv_std is required");
    if ((io->adv_mean == nullptr) != (io->adv_std == nullptr) || (!io->adv && !io->adv_mean))
        return fail(FA_ERR_INVALID, "fa_ppo_grad: pass adv, or adv_mean and adv_std together");
    if (io->B < 1 || io->num_guards < 1 || io->num_attackers < 1 || io->num_guards > FA_POLICY_MAX_TEAM ||
        io->num_attackers > FA_POLICY_MAX_TEAM || (io->team != 0 && io->team != 1))
        return fail(FA_ERR_INVALID, "fa_ppo_grad: need B >= 1, teams of 1..8, team 0 or 1");
    FaTrainArgs a;
    std::memset(&a, 0, sizeof(a));
    a.obs = io->obs; a.action = io->action; a.value_pred = io->value_pred; a.ret = io->ret;
    a.old_logp = io->old_log_prob; a.adv = io->adv; a.w = io->weights; a.wt = io->weights_t;
    a.adv_mean = io->adv_mean; a.adv_std = io->adv_std;
    a.scale = io->scale; a.idx = io->idx;
    a.B = io->B; a.G = io->num_guards; a.A = io->num_attackers; a.team = io->team;
    a.clip = io->clip_param; a.c_value = io->value_loss_coef; a.c_entropy = io->entropy_coef;
    a.clipped_value_loss = io->clipped_value_loss;
    a.share_cu = io->share_cu != 0 && io->idx != nullptr;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int et = fa_train_tile_envs(a.G, a.A);
    const size_t tiles = (size_t)((a.B + et - 1) / et);
    float *mpart = io->slabs + tiles * FA_MSLAB_FLOATS, *dw_slabs = mpart + (size_t)FA_MRED_PARTS * FA_MSLAB_FLOATS;
    a.mslab = io->slabs;
    a.rec_a = io->hsave;
    a.rec_b = io->hsave + tiles * 3 * FA_RECA_FLOATS;
    a.rec_g = a.rec_b + tiles * FA_RECB_FLOATS;
    if (!a.scale) { // partial mask sums behind the partial slabs; the scale pair is left behind the loss sums of `out`
        float *part = dw_slabs + (size_t)FA_DW_WGS_A * FA_DWA_FLOATS + (size_t)FA_DW_WGS_B * FA_DWB_FLOATS;
        FA_HIP(fa_launch_mask_parts(a, part, s));
        a.mask_part = part;
        a.scale_out = io->out + FA_SLAB_LOSS + 8;
        a.normalize = io->normalize != 0;
    }
    // Chunked hand-off (FA_PPO_CHUNKS = C, default 1): the tiles of the minibatch in C chunks, tile kernel and weight-gradient
    // GEMM alternating on the stream -- chunk c's records (320 KB per tile) are consumed while they may still sit in the 256 MiB
    // Infinity Cache and every chunk reuses the SAME record slots; the GEMM continues from the partial slabs of the earlier
    // chunks (fixed chunk order: still bitwise reproducible run to run).
    static const int chunks_env = [] { const char *v = getenv("FA_PPO_CHUNKS"); const int c = v ? atoi(v) : 1; return c < 1 ? 1 : c; }();
    const int chunks = chunks_env > (int)tiles ? (int)tiles : chunks_env;
    if (chunks == 1) {
        FA_HIP(fa_launch_train(a, s));                                                  // tiles: forward, losses, dL/dX
        FA_HIP(fa_launch_train_dw(a.rec_a, a.rec_b, (int)tiles, dw_slabs, false, s));   // dW = X^T dY over all rows
    } else {
        const int per = ((int)tiles + chunks - 1) / chunks;
        a.rec_b = io->hsave + (size_t)per * 3 * FA_RECA_FLOATS;     // the record slots of ONE chunk
        a.rec_g = a.rec_b + (size_t)per * FA_RECB_FLOATS;
        for (int c = 0, t0 = 0; t0 < (int)tiles; ++c, t0 += per) {
            a.tile0 = t0;
            a.ntiles = (int)tiles - t0 < per ? (int)tiles - t0 : per;
            a.rec_tile0 = 0;
            FA_HIP(fa_launch_train(a, s));
            FA_HIP(fa_launch_train_dw(a.rec_a, a.rec_b, a.ntiles, dw_slabs, c > 0, s));
        }
    }
    FA_HIP(fa_launch_train_reduce(a.mslab, (int)tiles, mpart, dw_slabs, io->out, s)); // fixed-order sums -> out
    return FA_OK;
}

int fa_adam_step(float *params, float *grads, float *exp_avg, float *exp_avg_sq, float *steps, const int32_t *seg,
                 int32_t nseg, int32_t n, float lr, float beta1, float beta2, float eps, float max_grad_norm, float *scratch,
                 void *stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !steps || !seg || !scratch)
        return fail(FA_ERR_INVALID, "fa_adam_step: null argument");
    if (nseg < 1 || n < 1) return fail(FA_ERR_INVALID, "fa_adam_step: need nseg >= 1 and n >= 1");
    if (reinterpret_cast<uintptr_t>(scratch) & 15) return fail(FA_ERR_INVALID, "fa_adam_step: scratch must be 16-byte aligned");
    FA_HIP(fa_launch_adam(params, grads, exp_avg, exp_avg_sq, steps, seg, nseg, n, lr, beta1, beta2, eps, max_grad_norm, scratch,
                          nullptr, static_cast<hipStream_t>(stream)));
    return FA_OK;
}

int fa_adam_step_dev(float *params, float *grads, float *exp_avg, float *exp_avg_sq, float *steps, const int32_t *seg,
                     int32_t nseg, int32_t n, const float *hyper, float *scratch, void *stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !steps || !seg || !scratch || !hyper)
        return fail(FA_ERR_INVALID, "fa_adam_step_dev: null argument");
    if (nseg < 1 || n < 1) return fail(FA_ERR_INVALID, "fa_adam_step_dev: need nseg >= 1 and n >= 1");
    if (reinterpret_cast<uintptr_t>(scratch) & 15) return fail(FA_ERR_INVALID, "fa_adam_step_dev: scratch must be 16-byte aligned");
    FA_HIP(fa_launch_adam(params, grads, exp_avg, exp_avg_sq, steps, seg, nseg, n, 0.f, 0.f, 0.f, 0.f, 0.f, scratch, hyper,
                          static_cast<hipStream_t>(stream)));
    return FA_OK;
}

int fa_run_tasks(const fa_task *tasks, int32_t n, void *stream) {
    if (!tasks || n < 0) return fail(FA_ERR_INVALID, "fa_run_tasks: null task list");
    FA_HIP(fa_launch_tasks(tasks, n, static_cast<hipStream_t>(stream)));
    return FA_OK;
}

int fa_pack_weights(const float *plain, float *weights, float *weights_t, void *stream) {
    if (!plain || !weights || !weights_t) return fail(FA_ERR_INVALID, "fa_pack_weights: null argument");
    FA_HIP(fa_launch_pack(plain, weights, weights_t, static_cast<hipStream_t>(stream)));
    return FA_OK;
}

static int attend_check(const char *who, int B, int n, int nk, int width) {
    if (B < 1 || n < 1 || nk < 1 || n > FA_POLICY_MAX_TEAM || nk > FA_POLICY_MAX_TEAM)
        return fail(FA_ERR_INVALID, std::string(who) + ": need B >= 1 and 1 <= n, nk <= 8");
    if (width != 64 && width != 128) return fail(FA_ERR_INVALID, std::string(who) + ": width must be 64 or 128");
    return FA_OK;
}

int fa_attend_forward(const float *g, const float *keys, float *out, float *attn, int32_t B, int32_t n, int32_t nk,
                      int32_t width, int32_t skip_self, void *stream) {
    if (!g || !keys || !out || !attn) return fail(FA_ERR_INVALID, "fa_attend_forward: null argument");
    if (int rc = attend_check("fa_attend_forward", B, n, nk, width)) return rc;
    FA_HIP(fa_launch_attend(width, false, g, keys, out, attn, nullptr, nullptr, nullptr, B, n, nk, skip_self,
                            static_cast<hipStream_t>(stream)));
    return FA_OK;
}

int fa_attend_backward(const float *g, const float *keys, const float *attn, const float *dout, float *dg, float *dkeys,
                       int32_t B, int32_t n, int32_t nk, int32_t width, int32_t skip_self, void *stream) {
    if (!g || !keys || !attn || !dout || !dg || !dkeys) return fail(FA_ERR_INVALID, "fa_attend_backward: null argument");
    if (int rc = attend_check("fa_attend_backward", B, n, nk, width)) return rc;
    FA_HIP(fa_launch_attend(width, true, g, keys, nullptr, const_cast<float *>(attn), dout, dg, dkeys, B, n, nk, skip_self,
                            static_cast<hipStream_t>(stream)));
    return FA_OK;
}

int fa_get_state(fa_env *env, const fa_state_host *o) {
    if (!env || !o) return fail(FA_ERR_INVALID, "fa_get_state: null argument");
    DeviceGuard guard(env->cfg.device_id);
    FA_HIP(hipDeviceSynchronize());
    const size_t E = (size_t)env->cfg.num_envs, EN = E * env->N;
    const FaState &s = env->s;
#define FA_D2H(dst, src, bytes) \
    if (dst) FA_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost))
    FA_D2H(o->pos_x, s.px, EN * 8);
    FA_D2H(o->pos_y, s.py, EN * 8);
    FA_D2H(o->vel_x, s.vx, EN * 8);
    FA_D2H(o->vel_y, s.vy, EN * 8);
    FA_D2H(o->ang, s.ang, EN * 8);
    FA_D2H(o->prev_dist, s.prev, EN * 8);
    FA_D2H(o->alive, s.alive, EN);
    FA_D2H(o->time_step, s.tstep, E * 4);
    FA_D2H(o->num_hit, s.num_hit, EN * 4);
    FA_D2H(o->num_was_hit, s.num_was_hit, EN * 4);
    FA_D2H(o->game_result, s.game_result, E * 3);
#undef FA_D2H
#define FA_D2H_LATE(dst, src, bytes) \
    if (dst) FA_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost))
    FA_D2H_LATE(o->episode_reward_sum, s.ep_rew_sum, EN * 8);
    if (o->alive_at_end) {
        std::vector<uint32_t> ae(EN);
        FA_HIP(hipMemcpy(ae.data(), s.alive_end, EN * 4, hipMemcpyDeviceToHost));
        for (size_t k = 0; k < EN; ++k) o->alive_at_end[k] = (int64_t)ae[k];
    }
    if (o->result_count) {
        std::vector<uint32_t> rc(E * 3);
        FA_HIP(hipMemcpy(rc.data(), s.result_count, E * 3 * 4, hipMemcpyDeviceToHost));
        for (size_t k = 0; k < E * 3; ++k) o->result_count[k] = (int64_t)rc[k];
    }
    return FA_OK;
}

int fa_set_state(fa_env *env, const fa_state_host *in) {
    if (!env || !in) return fail(FA_ERR_INVALID, "fa_set_state: null argument");
    DeviceGuard guard(env->cfg.device_id);
    FA_HIP(hipDeviceSynchronize());
    const size_t E = (size_t)env->cfg.num_envs, EN = E * env->N;
    const FaState &s = env->s;
#define FA_H2D(dst, src, bytes) \
    if (src) FA_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice))
    FA_H2D(s.px, in->pos_x, EN * 8);
    FA_H2D(s.py, in->pos_y, EN * 8);
    FA_H2D(s.vx, in->vel_x, EN * 8);
    FA_H2D(s.vy, in->vel_y, EN * 8);
    FA_H2D(s.ang, in->ang, EN * 8);
    FA_H2D(s.prev, in->prev_dist, EN * 8);
    FA_H2D(s.alive, in->alive, EN);
    FA_H2D(s.tstep, in->time_step, E * 4);
    FA_H2D(s.num_hit, in->num_hit, EN * 4);
    FA_H2D(s.num_was_hit, in->num_was_hit, EN * 4);
    FA_H2D(s.game_result, in->game_result, E * 3);
#undef FA_H2D
    return FA_OK;
}

int fa_selftest_math(fa_env *env, uint64_t samples, uint64_t seed, uint64_t *mismatch_host) {
    if (!env || !mismatch_host) return fail(FA_ERR_INVALID, "fa_selftest_math: null argument");
    DeviceGuard guard(env->cfg.device_id);
    unsigned long long *d = reinterpret_cast<unsigned long long *>(env->adv_partial); // scratch
    FA_HIP(hipDeviceSynchronize());
    FA_HIP(hipMemset(d, 0, 24));
    const unsigned long long per = samples / (1024ull * 256ull) + 1ull;
    FA_HIP(fa_launch_selftest(per, seed, d, nullptr));
    FA_HIP(hipMemcpy(mismatch_host, d, 24, hipMemcpyDeviceToHost));
    return FA_OK;
}

const char *fa_policy_variant(fa_env *env) {
    if (!env) return "";
    const int E = env->cfg.num_envs, G = env->cfg.num_guards, A = env->cfg.num_attackers;
    if (G > FA_POLICY_MAX_TEAM || A > FA_POLICY_MAX_TEAM) return "";
    const int n_max = G > A ? G : A;
    return fa_policy_tile_envs(E, G, A) * n_max > 64 ? "fa_policy_kernel<3, 8>" : "fa_policy_kernel<2, 4>";
}

const char *fa_step_variant(fa_env *env, int32_t num_steps) {
    if (!env) return "";
    return fa_step_variant_name(env->cfg.num_guards, env->cfg.num_attackers, env->cfg.num_envs, num_steps,
                                env->cfg.step_kernel, env->choice_k > 0);
}

int fa_rng_peek(fa_env *env, int32_t e, int32_t count, double *out_host) {
    if (!env || !out_host) return fail(FA_ERR_INVALID, "fa_rng_peek: null argument");
    if (e < 0 || e >= env->cfg.num_envs || count < 0) return fail(FA_ERR_INVALID, "fa_rng_peek: bad index");
    if (env->cfg.rng_mode != FA_RNG_MT19937) return fail(FA_ERR_STATE, "fa_rng_peek: MT19937 mode only");
    DeviceGuard guard(env->cfg.device_id);
    FA_HIP(hipDeviceSynchronize());
    std::vector<uint32_t> mt(FA_MT_N);
    int32_t pos = 0;
    FA_HIP(hipMemcpy(mt.data(), env->s.mt + (size_t)e * FA_MT_N, FA_MT_N * 4, hipMemcpyDeviceToHost));
    FA_HIP(hipMemcpy(&pos, env->s.mt_pos + e, 4, hipMemcpyDeviceToHost));
    auto next = [&]() {
        const int n1 = pos + 1 >= FA_MT_N ? pos + 1 - FA_MT_N : pos + 1;
        const int nm = pos + FA_MT_M >= FA_MT_N ? pos + FA_MT_M - FA_MT_N : pos + FA_MT_M;
        uint32_t y = (mt[pos] & 0x80000000u) | (mt[n1] & 0x7fffffffu);
        uint32_t nw = mt[nm] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        mt[pos] = nw;
        pos = n1;
        nw ^= (nw >> 11);
        nw ^= (nw << 7) & 0x9d2c5680u;
        nw ^= (nw << 15) & 0xefc60000u;
        nw ^= (nw >> 18);
        return nw;
    };
    for (int k = 0; k < count; ++k) {
        uint32_t a = next() >> 5, b = next() >> 6;
        out_host[k] = (a * 67108864.0 + b) / 9007199254740992.0;
    }
    return FA_OK;
}

} // extern "C"
