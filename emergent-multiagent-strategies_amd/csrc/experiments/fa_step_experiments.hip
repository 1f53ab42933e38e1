// fa_step_experiments.hip -- the two structural experiments of round 4 on the step kernel, both bit-exact and both
// slower than fa_step_pipe_kernel (profiles/r04_experiments/step_lane_pair_mapping_assessment.md, step_one_barrier_chain.md).
// NOT part of the product library: tools/build_variant.py links this translation unit into tools/_build/lib_experiments.so,
// whose fa_launch_step_experiment the dispatch in fa_step_classic.hip finds (a weak symbol, null in the product);
// tests/test_gpu_experiment_kernels.py runs their parity tests against that library.
#include "../fa_step_common.h"
#include "../fa_probe.h"
#include "fa_step_experiments.h"

// ---- experiment (round 4): lane = (agent, partner), one wave, no workgroup barrier ---------------------
// The judge's round-3 question: does the step get faster when a lane owns ONE ordered pair (agent i, partner j) --
// its soft-contact force and its shooter -> target test -- so that the five-partner loop becomes one evaluation and
// nothing crosses a workgroup barrier?  N (N - 1) lanes per env (30 at 3v3: two envs per wave, four lanes idle); an
// agent's state lives in all N - 1 of its lanes, which integrate it redundantly (same operations on the same values:
// same bits); positions and the shooters' heading go through LDS by agent slot; a lane's pair force goes through LDS
// once and every lane of the agent adds its N - 1 partner terms in ascending partner order -- the reference's order
// (core.py:231-243); flag reductions are ballots shifted to the env's lane group, as in fa_step_kernel.  Everything
// else -- decode, walls, integration, rewards, done, reset, rows -- is fa_step_kernel's code.  3v3 only.
template <int TG, int TA, bool COLLECT, bool CHOICE>
__global__ __launch_bounds__(FA_WAVE) void fa_step_pair_kernel(FaStepArgs a) {
    constexpr int N = TG + TA, NP = N - 1, LPE = N * NP, EPW = FA_WAVE / LPE;
    static_assert(LPE <= FA_WAVE, "one env must fit a wave");
    const int lane = threadIdx.x;
    const int slot = lane / LPE, l = lane - slot * LPE;
    const int i = l / NP, p = l - i * NP, j = p + (p >= i ? 1 : 0); // agent, partner slot, partner agent
    const int gbase = slot * LPE;
    const int e = blockIdx.x * EPW + slot;
    if (!((slot < EPW) && (e < a.E))) return; // padding lanes leave: ballots count live lanes only
    const bool primary = p == 0;              // the lane that stores the agent's rows and state
    const bool is_att = i >= TG, opponent = (j >= TG) != is_att;
    const int sa = slot * N + i, sj = slot * N + j; // LDS slots of the agent and of the partner
    const size_t idx = (size_t)e * N + i;
    const size_t EN = (size_t)a.E * N;
    constexpr unsigned long long LPE_MASK = (1ull << LPE) - 1ull;
    unsigned long long att_primary = 0ull, hit_lanes = 0ull; // primary lanes of the attackers; lanes whose partner is i
#pragma unroll
    for (int t = 0; t < N; ++t) {
        if (t >= TG) att_primary |= 1ull << (t * NP);
        if (t != i) hit_lanes |= 1ull << (t * NP + (i - (i > t ? 1 : 0)));
    }
    const FaDerived &c = a.c;

    __shared__ double s_px[EPW * N], s_py[EPW * N], s_cs[EPW * N], s_sn[EPW * N];
    __shared__ int s_act[FA_ACT_BATCH][FA_WAVE];
    __shared__ double2 s_f[FA_WAVE];

    double px = a.s.px[idx], py = a.s.py[idx], vx = a.s.vx[idx], vy = a.s.vy[idx], ang = a.s.ang[idx], prev = a.s.prev[idx];
    bool alive = a.s.alive[idx] != 0;
    int t = a.s.tstep[e], nh = 0, nwh = 0;
    double ep_rew = 0.0;
    if (a.track_counters) { nh = a.s.num_hit[idx]; nwh = a.s.num_was_hit[idx]; ep_rew = a.s.ep_rew[idx]; }
    bool dirty = false;
    int mt_base = (a.rng_mode == 0 ? a.s.mt_pos[e] : 0) + 4 * i;

    const int nsteps = a.nsteps;
    const int64_t *act_ptr = a.actions + (int64_t)e * a.as_e + (int64_t)i * a.as_i;
    int av[FA_ACT_BATCH];
#pragma unroll
    for (int k = 0; k < FA_ACT_BATCH; ++k) av[k] = k < nsteps ? (int)act_ptr[(int64_t)k * a.as_t] : 0;
    for (int s = 0; s < nsteps; ++s) {
        if ((s & (FA_ACT_BATCH - 1)) == 0) {
#pragma unroll
            for (int k = 0; k < FA_ACT_BATCH; ++k) s_act[k][lane] = av[k];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < FA_ACT_BATCH; ++k)
                av[k] = (s + FA_ACT_BATCH + k < nsteps) ? (int)act_ptr[(int64_t)(s + FA_ACT_BATCH + k) * a.as_t] : 0;
        }
        const bool alive0 = alive;
        // ---- fortattack.py:253-263,:289 _set_action (all agents, dead ones too) ----
        const int act = s_act[s & (FA_ACT_BATCH - 1)][lane];
        double u0 = 0.0, u1 = 0.0, rot = 0.0;
        if (act == 1) u0 = +1.0;
        if (act == 2) u0 = -1.0;
        if (act == 3) u1 = +1.0;
        if (act == 4) u1 = -1.0;
        if (act == 5) rot = c.rot_pos;
        if (act == 6) rot = c.rot_neg;
        const bool shoot = act == 7;
        u0 *= c.accel;
        u1 *= c.accel;

        // ---- positions and the shooters' heading by agent slot (every lane of an agent writes the same value) ----
        s_px[sa] = px;
        s_py[sa] = py;
        const bool shooter = alive0 && shoot;
        if (shooter) {
            double sn, cs;
            sincos_heading(ang, sn, cs);
            s_cs[sa] = cs;
            s_sn[sa] = sn;
        }
        const unsigned long long shooters_b = __ballot(shooter);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const double qx = s_px[sj], qy = s_py[sj];

        // ---- core.py:254-302 apply_laser_effect: this lane's test is "is agent i inside partner j's wedge" ----
        bool was_hit = false;
        int hit_cnt = 0, was_hit_cnt = 0;
        if (shooters_b != 0ull) {
            bool hk = false;
            if (opponent && alive0 && ((shooters_b >> (gbase + j * NP)) & 1ull)) {
                double u, lhs, rhs;
                fa_wedge(c.agent_size, c.cos_hw, c.sin_hw, px, py, qx, qy, s_cs[sj], s_sn[sj], u, lhs, rhs);
                hk = (u <= c.shoot_far) & (lhs <= rhs);
            }
            const unsigned long long grp = (__ballot(hk) >> gbase) & LPE_MASK;
            const unsigned mine = (unsigned)(grp >> (i * NP)) & ((1u << NP) - 1u);
            was_hit = mine != 0u;
            was_hit_cnt = __popc(mine);
            hit_cnt = __popcll(grp & hit_lanes);
        }
        const bool hit = shooter && hit_cnt > 0;
        const bool alive1 = alive0 && !was_hit;       // :293-302 one shot kills
        const bool just_died = alive0 && was_hit;
        const unsigned long long grp_alive1 = (__ballot(alive1) >> gbase) & LPE_MASK;
        const int n_alive_att = __popcll(grp_alive1 & att_primary);

        // ---- core.py:231-243 + :440-456: this lane's pair force (exactly +0.0 unless both live and in range) ----
        {
            const double dx = px - qx, dy = py - qy, d2 = dx * dx + dy * dy;
            double fx = 0.0, fy = 0.0;
            if (alive1 && ((grp_alive1 >> (j * NP)) & 1ull) && !(d2 > c.contact_skip_d2)) fa_contact_force(c, dx, dy, d2, fx, fy);
            s_f[lane] = make_double2(fx, fy);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (alive1) {
            double Fx = u0 + 0.0, Fy = u1 + 0.0;      // core.py:221-228
#pragma unroll
            for (int k = 0; k < NP; ++k) {            // ascending partner order == the reference's pair order; a term
                const double2 f = s_f[gbase + i * NP + k]; // of a dead or distant partner is +0.0: F (never -0.0) unchanged
                Fx = f.x + Fx;
                Fy = f.y + Fy;
            }
            {
                double wx, wy;
                fa_wall_force(c, px, py, wx, wy);     // core.py:246-252 + :459-472
                Fx = wx + Fx;
                Fy = wy + Fy;
            }
            // core.py:324-338 integrate_state (mass == 1.0: F/1.0 is exact)
            vx = vx * c.one_minus_damping;
            vy = vy * c.one_minus_damping;
            vx += Fx * c.dt;
            vy += Fy * c.dt;
            const double speed2 = vx * vx + vy * vy;
            if (speed2 > c.speed2_max) {
                const double speed = sqrt_rn(speed2);
                vx = div_rn(vx, speed) * c.max_speed;
                vy = div_rn(vy, speed) * c.max_speed;
            }
            ang += rot;
            px += vx * c.dt;
            py += vy * c.dt;
        }

        // ---- rewards (fortattack_env_v1.py:87-188), after World.step ---------------
        const double ddx = px - c.door_x, ddy = py - c.door_y;
        const double dist_door = sqrt_rn(ddx * ddx + ddy * ddy);
        const bool any_in_fort = ((__ballot(is_att && alive1 && dist_door < c.fort_dim) >> gbase) & LPE_MASK) != 0ull;
        const bool rewarded = (alive1 || just_died);
        const double rew = fa_reward(is_att, rewarded, prev, dist_door, shoot, hit, was_hit, n_alive_att, any_in_fort,
                                     c.fort_dim, 0.3, 10.0, 3.0, 0.1);
        prev = rewarded ? dist_door : prev;

        // ---- fortattack.py:202-225 _get_done, :171 time_step += 1 ------------------
        const bool timeout = t == a.max_t - 1;
        const bool done = any_in_fort || n_alive_att == 0 || timeout;
        if (l == 0) {
            if (done) {
                const int which = any_in_fort ? 2 : (n_alive_att == 0 ? 0 : 1);
                uint8_t *gr = a.s.game_result + (size_t)e * 3;
                gr[0] = which == 0; gr[1] = which == 1; gr[2] = which == 2;
                atomicAdd(a.s.result_count + (size_t)e * 3 + which, 1u);
            }
            if (COLLECT || a.done) a.done[(size_t)s * a.E + e] = done ? 1 : 0;
        }
        t += 1;
        if (a.track_counters) {
            ep_rew += alive0 ? rew : 0.0;
            if (done) {
                if (primary) {
                    a.s.ep_rew_sum[idx] += ep_rew;
                    if (alive1) a.s.alive_end[idx] += 1u;
                }
                ep_rew = 0.0;
                dirty = true;
            }
        }
        const bool do_reset = done && a.auto_reset != 0;
        dirty = dirty || alive0;
        alive = alive1;
        nh += hit_cnt;
        nwh += was_hit_cnt;
        if (primary) {
            const size_t o = (size_t)s * EN + idx;
            const float mk = (alive0 || do_reset) ? 1.0f : 0.0f;
            if (COLLECT) {
                a.rew32[o] = (float)rew;
                a.mask32[o] = mk;
            } else {
                if (a.rew32) a.rew32[o] = (float)rew;
                if (a.rew64) a.rew64[o] = rew;
                if (a.mask32) a.mask32[o] = mk;
                if (a.hit) a.hit[o] = hit ? 1 : 0;
                if (a.was_hit) a.was_hit[o] = was_hit ? 1 : 0;
            }
        }

        // ---- fortattack_env_v1.py:47-75 reset_world (every lane of an agent draws the same words and stores the same
        // twisted words: one wave, lock step) --------------------------------------------
        if (__ballot(do_reset) != 0ull) {
            double npx = px, npy = py;
            reset_agent(a, e, i, N, is_att, do_reset, mt_base, npx, npy);
            if (do_reset) {
                px = npx; py = npy; vx = 0.0; vy = 0.0;
                ang = is_att ? c.ang_attacker : c.ang_guard;
                alive = true;
                t = 0;
                nh = 0; nwh = 0;
                ep_rew = 0.0;
                dirty = true;
                if (l == 0) reset_advance(a, e, mt_base);
            }
            if constexpr (CHOICE) {
                int extra = 0;
                if (do_reset && l == 0) {
                    const uint32_t rng = (uint32_t)(a.choice_k - 1);
                    uint32_t mask = rng, v = 0;
                    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
                    if (rng != 0u) {
                        if (a.rng_mode == 0) {
                            uint32_t *mt = a.s.mt + (size_t)e * FA_MT_N;
                            int cur = mt_base;
                            do {
                                const uint32_t nw = mt_twist(mt[cur], mt[mt_wrap(cur + 1)], mt[mt_wrap(cur + FA_MT_M)]);
                                mt[cur] = nw;
                                v = mt_temper(nw) & mask;
                                cur = mt_wrap(cur + 1);
                                ++extra;
                            } while (v > rng);
                            a.s.mt_pos[e] = cur;
                        } else {
                            const uint64_t genv = (uint64_t)(a.env_offset + e);
                            uint32_t ctr = 0;
                            do {
                                uint32_t cc[4] = {(uint32_t)genv, (uint32_t)(genv >> 32), a.s.reset_count[e], 255u | (ctr << 8)};
                                philox4x32_10(cc, (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
                                v = cc[0] & mask;
                                ++ctr;
                            } while (v > rng);
                        }
                    }
                    a.choice_out[e] = (int)v;
                }
                if (a.rng_mode == 0) {
                    extra = __shfl(extra, gbase);
                    if (do_reset) mt_base = (mt_base - 4 * i + extra) % FA_MT_N + 4 * i;
                }
            }
        }

        // ---- observation row (fortattack_env_v1.py:238) ------------------------------------
        if (primary) {
            const size_t o6 = ((size_t)s * EN + idx) * 6;
            fa_store_obs((COLLECT || a.obs32) ? a.obs32 + o6 : nullptr, (!COLLECT && a.obs64) ? a.obs64 + o6 : nullptr,
                         alive, px, py, ang, vx, vy);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    if (dirty && primary) {
        a.s.px[idx] = px; a.s.py[idx] = py; a.s.vx[idx] = vx; a.s.vy[idx] = vy;
        a.s.ang[idx] = ang; a.s.prev[idx] = prev;
        a.s.alive[idx] = alive ? 1 : 0;
        if (a.track_counters) { a.s.num_hit[idx] = nh; a.s.num_was_hit[idx] = nwh; a.s.ep_rew[idx] = ep_rew; }
    }
    if (l == 0) a.s.tstep[e] = t;
}

// ---- experiment (round 4, second structural one): ONE workgroup barrier per step ---------------------------------
// The pipelined kernel's step is two phases closed by two barriers: [laser | pair forces | walls] -> B2 -> [force sum,
// integrate, publish | rows] -> P, each phase headed by an LDS round trip and ended by ~250 cycles of barrier.  Here the
// wave that owns the state ("chain wave") computes the first DF partner offsets of the soft contacts ITSELF, from
// partner positions it read back from its own publish before the barrier (no LDS wait at the top of a step), and the
// three helpers deliver what else the force sum needs -- wave 1 the laser masks, wave 2 decoded action + wall forces,
// wave 3 the remaining partner offset(s) -- through LDS hand-offs closed by a tag word instead of a barrier: a helper
// writes its data, then tag = step; the chain wave reads the tags FIRST and the data behind them in one burst (DS
// operations of a wave execute in order, so a tag that reads `step` proves the data behind it is the step's) and
// repeats the burst while a tag is stale.  The helpers then do what no next state waits for (next heading's sin/cos,
// rewards + reset stream, observation / done / mask rows) and everybody meets at the one barrier P.  Same arithmetic in
// the same order as the pipelined kernel: same bits.  Compile-time team sizes; the ensemble path's choice is not
// interleaved (as in the pipelined kernel).
// MEASURED (profiles/r04_experiments/step_one_barrier_chain.md): bit-exact on the first run, 151.5-153.0 us per 128-step
// launch at 3v3 x 4096 against the pipelined kernel's 147.5 in the same run -- no gain: the step is still the chain
// P -> [helper: LDS read + laser 1 000-1 160 cycles] -> hand-off -> [force sum, integrate, done, publish: 800] -> barrier
// (366 with the LDS drain), the chain wave's own pair offsets (628) sit in the shadow of the laser.  FA_KERNEL_CHAIN,
// never picked by AUTO.
#ifndef FA_CHAIN_DF
#define FA_CHAIN_DF 2        // partner offsets the chain wave computes itself (1: 157.9 us against 153.7)
#endif
#ifndef FA_CHAIN_TRIG_WAVE
#define FA_CHAIN_TRIG_WAVE 1 // which helper makes the next heading's sin/cos: 1 (laser wave) or 3 (pair / rows wave: 163.0 us)
#endif
#ifndef FA_CHAIN_NOBAR
#define FA_CHAIN_NOBAR 0     // 1: no barrier inside the step loop at all -- the helpers poll the chain wave's publish tag
                             // (bit-exact; 162.3 / 165.0 / 167.4 us with s_sleep 0 / 1 / 3 between polls against 153.8 with P)
#endif
#ifndef FA_CHAIN_HSLEEP
#define FA_CHAIN_HSLEEP 1    // s_sleep between two polls of the publish tag by a helper
#endif
#ifndef FA_CHAIN_SLEEP
#define FA_CHAIN_SLEEP 0     // s_sleep between two polls of the hand-off tags (0: none; 1 / 4: 153.7 / 157.0 us against 151.5)
#endif
template <int TG, int TA, bool COLLECT>
__global__ __launch_bounds__(4 * FA_WAVE, 2) void fa_step_chain_kernel(FaStepArgs a) {
    constexpr int G = TG, A = TA, N = TG + TA;
    constexpr int NOFF = N / 2;                 // partner offsets that cover every unordered pair once
    constexpr int DF = NOFF > FA_CHAIN_DF ? FA_CHAIN_DF : NOFF - 1; // offsets 1..DF: the chain wave; DF+1..NOFF: wave 3
    constexpr int EPW = FA_WAVE / N;
    const int lane = threadIdx.x & (FA_WAVE - 1);
    const int wave_id = threadIdx.x / FA_WAVE;
    const int slot = lane / N;
    const int i = lane - slot * N;
    const int gbase = slot * N;
    const int e = blockIdx.x * EPW + slot;
    if (!((slot < EPW) && (e < a.E))) return;
    const bool is_att = i >= G;
    const size_t idx = (size_t)e * N + i;
    const size_t EN = (size_t)a.E * N;
    constexpr unsigned long long grp_mask = (1ull << N) - 1ull;
    const FaDerived &c = a.c;
    const int ns = a.nsteps;

    // buffer s & 1: state at the start of step s and the by-products of step s-1, published by the chain wave before P(s-1)
    __shared__ double2 s_pos[2][FA_WAVE], s_vel[2][FA_WAVE]; // ((x, y) pairs side by side, see fa_step_pipe_kernel)
    __shared__ double s_ang[2][FA_WAVE], s_dd[2][FA_WAVE];
    __shared__ unsigned long long s_mask[2][2];          // [0] alive at the step's start, [1] done of the step before
    __shared__ double2 s_trig[2][FA_WAVE];               // [step parity][lane] = (cos, sin): wave 1 writes and reads it
    __shared__ int s_act[FA_ACT_BATCH][FA_WAVE];
    // the hand-offs to the chain wave: written and re-read without a barrier in between.  NOT volatile -- the backend
    // waits for every volatile LDS access on its own (lgkmcnt(0) each: 250 us) -- but fenced for the COMPILER by
    // FA_ORDER(): data before tag on the writing side, tags before data on the reading side; the hardware keeps a wave's
    // DS operations in order.
    __shared__ unsigned long long s_las[3][4];  // [step % 3][alive after the laser, hit, was hit]     (wave 1)
    __shared__ double2 s_W[FA_WAVE], s_U[FA_WAVE];             // wall force, decoded action (x, y)     (wave 2)
    __shared__ double s_rot[FA_WAVE];                          // ... and its rotation
    __shared__ double2 s_rp[FA_WAVE];                          // position of the lane's next reset     (wave 2)
    __shared__ double2 s_fm[N][FA_WAVE];                       // [partner j][lane]: pair force on the lane's agent
    __shared__ int s_tag[4];                                   // [wave]: the step its hand-off is complete for; [0]: the chain
                                                               // wave's -- the step whose start state is published
#define FA_ORDER() asm volatile("" ::: "memory")
    // FA_CHAIN_NOBAR: a helper starts step s when the chain wave's tag says state(s) is published (tag first, data behind it);
    // nothing else orders the waves inside the loop.  What a helper reads one step late (by-products, s_las) survives
    // because the chain wave publishes state(s+2) only after it has seen every helper's tag of step s+1, i.e. after
    // every helper finished step s entirely -- except s_las, which wave 1 rewrites two steps later WITHOUT waiting for
    // the others: three buffers.
    auto wait_state = [&](int s) {
        if (FA_CHAIN_NOBAR) {
            for (;;) {
                FA_ORDER();
                const int tf = s_tag[0];
                FA_ORDER();
                if (__builtin_amdgcn_readfirstlane(tf) >= s) break;
                if (FA_CHAIN_HSLEEP > 0) __builtin_amdgcn_s_sleep(FA_CHAIN_HSLEEP);
            }
        }
    };

    if (wave_id == 1) {
        // ---- wave 1: the laser (core.py:254-302) of step s -> chain wave; then sin/cos of the heading of step s+1 ----
        constexpr int KT = TG > TA ? TG : TA;
        const int n_opp = is_att ? G : A, opp0 = is_att ? 0 : G;
        const int team_idx = is_att ? i - G : i;
        const unsigned opp_bits = is_att ? ((1u << G) - 1u) : (((1u << A) - 1u) << G);
        int nh = 0, nwh = 0;
        if (a.track_counters) { nh = a.s.num_hit[idx]; nwh = a.s.num_was_hit[idx]; }
        double sn, cs, sn_g = 0.0, cs_g = 0.0, sn_a = 0.0, cs_a = 0.0;
        sincos_heading(a.s.ang[idx], sn, cs);
        if (ns > 1) { // headings after a reset (fortattack_env_v1.py:59)
            sincos_heading(c.ang_guard, sn_g, cs_g);
            sincos_heading(c.ang_attacker, sn_a, cs_a);
        }
        const double cs_ro = is_att ? cs_g : cs_a, sn_ro = is_att ? sn_g : sn_a;   // the opponents after a reset
        s_trig[0][lane] = make_double2(cs, sn);
        double k_size = c.agent_size, k_far = c.shoot_far, k_chw = c.cos_hw, k_shw = c.sin_hw;
        asm volatile("" : "+v"(k_size), "+v"(k_far), "+v"(k_chw), "+v"(k_shw));
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(nh), "+v"(nwh));
        const unsigned long long lane0_m = FA_M_EQ_U(lane, 0);
        FA_TICK_INIT
        FA_WG_BARRIER(); // P(-1)
        for (int s = 0; s < ns; ++s) {
            const int b = s & 1, b3 = s % 3;
            wait_state(s);
            FA_TICK(12)
            const int act = s_act[s & (FA_ACT_BATCH - 1)][lane];
            const unsigned long long alive0_m = s_mask[b][0];
            const unsigned long long reset_prev_m = (s > 0 && a.auto_reset != 0) ? s_mask[b][1] : 0ull;
            const double2 pos_ = s_pos[b][lane];
            const double px = pos_.x, py = pos_.y, ang = s_ang[b][lane];
            double oqx[KT], oqy[KT], ocs[KT], osn[KT];
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                const int j = gbase + opp0 + (k < n_opp ? k : 0);
                const double2 q_ = s_pos[b][j], tg_ = s_trig[b][j];
                oqx[k] = q_.x; oqy[k] = q_.y;
                ocs[k] = tg_.x; osn[k] = tg_.y;
            }
            if (__builtin_expect(reset_prev_m != 0ull, 0)) { // the env was reset at the end of the step before
                const bool rp = fa_lanes(reset_prev_m);
#pragma unroll
                for (int k = 0; k < KT; ++k) {
                    ocs[k] = rp ? cs_ro : ocs[k];
                    osn[k] = rp ? sn_ro : osn[k];
                }
                if (rp) { nh = 0; nwh = 0; }
            }
            const unsigned long long shooters_m = FA_M_EQ_U(act, 7) & alive0_m;
            unsigned long long hb[KT];
#pragma unroll
            for (int k = 0; k < KT; ++k) hb[k] = 0ull;
            int hit_cnt = 0, was_hit_cnt = 0;
            if (shooters_m != 0ull) {
                const unsigned gw_sh = (unsigned)(shooters_m >> gbase);
#pragma unroll
                for (int k = 0; k < KT; ++k) {
                    const unsigned long long cand_m = (k < n_opp ? FA_M_NE_U(gw_sh & (1u << (opp0 + k)), 0) : 0ull) & alive0_m;
                    double u, lhs, rhs;
                    fa_wedge(k_size, k_chw, k_shw, px, py, oqx[k], oqy[k], ocs[k], osn[k], u, lhs, rhs);
                    hb[k] = cand_m & FA_M_LE_D(u, k_far) & FA_M_LE_D(lhs, rhs);
                }
                int tix = team_idx;
                asm volatile("" : "+v"(tix));
                unsigned sel = (unsigned)(hb[0] >> gbase);
#pragma unroll
                for (int k = 1; k < KT; ++k) sel = (k == tix) ? (unsigned)(hb[k] >> gbase) : sel;
                hit_cnt = __popc(sel & opp_bits);
#pragma unroll
                for (int k = 0; k < KT; ++k) was_hit_cnt += fa_lanes(hb[k]) ? 1 : 0;
            }
            unsigned long long was_hit_m = hb[0];
#pragma unroll
            for (int k = 1; k < KT; ++k) was_hit_m |= hb[k];
            const unsigned long long hit_m = FA_M_NE_U(hit_cnt, 0) & shooters_m;
            const unsigned long long alive1_m = alive0_m & ~was_hit_m;        // :293-302 one shot kills
            if (fa_lanes(lane0_m)) {
                s_las[b3][0] = alive1_m;
                s_las[b3][1] = hit_m;
                s_las[b3][2] = was_hit_m;
            }
            FA_ORDER();
            if (fa_lanes(lane0_m)) s_tag[1] = s;   // behind the data: in order
            nh += hit_cnt;
            nwh += was_hit_cnt; // one per shooter that hit (core.py:283)
            FA_TICK(10)
            // the heading of step s+1: it only changes by the action's rotation (core.py:336) or by a reset to a
            // constant (handled above); a dead agent's value is never used
            if (FA_CHAIN_TRIG_WAVE == 1 && s + 1 < ns) {
                double rot = 0.0;
                if (act == 5) rot = c.rot_pos;
                if (act == 6) rot = c.rot_neg;
                sincos_heading(ang + rot, sn, cs);
                s_trig[(s + 1) & 1][lane] = make_double2(cs, sn);
            }
            FA_TICK(11)
            if (!FA_CHAIN_NOBAR) FA_WG_BARRIER(); // P(s)
        }
        FA_WG_BARRIER(); // (epilogues of the emitting waves)
        FA_TICK_FLUSH(10, 13, 29)
        if (a.track_counters) {
            if (a.auto_reset != 0 && ((s_mask[ns & 1][1] >> lane) & 1ull)) { nh = 0; nwh = 0; }
            a.s.num_hit[idx] = nh;
            a.s.num_was_hit[idx] = nwh;
        }
        return;
    }
    if (wave_id == 2) {
        // ---- wave 2: decoded action + wall forces of step s -> chain wave; owns the reset stream; then the rewards
        // of step s-1 -----------------------------------------------------------------------------------------------
        double prev = a.s.prev[idx], ep_rew = 0.0;
        if (a.track_counters) ep_rew = a.s.ep_rew[idx];
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(prev), "+v"(ep_rew));
        int act_prev = 0;
        bool alive0_prev = false;
        float *p_rew = a.rew32 ? a.rew32 + idx : nullptr;
        long long row = (long long)idx;
        double k_fort = c.fort_dim, k_03 = 0.3, k_10 = 10.0, k_3 = 3.0, k_01 = 0.1;
        asm volatile("" : "+v"(p_rew), "+v"(row));
        asm volatile("" : "+v"(k_fort), "+v"(k_03), "+v"(k_10), "+v"(k_3), "+v"(k_01));
        // rewards of step se (called once per step, in order): laser masks in s_las[se & 1], done / door distance in
        // buffer (se + 1) & 1
        auto emit_rew = [&](int se) {
            const int pl = se % 3, pb = (se + 1) & 1;
            const unsigned long long m1 = s_las[pl][0];
            const bool alive1 = (m1 >> lane) & 1ull;
            const bool hit = (s_las[pl][1] >> lane) & 1ull;
            const bool was_hit = (s_las[pl][2] >> lane) & 1ull;
            const bool done = (s_mask[pb][1] >> lane) & 1ull;
            const double dist_door = sqrt_rn(s_dd[pb][lane]);
            const bool alive0 = alive0_prev;
            const bool shoot = act_prev == 7;
            const int n_alive_att = __popcll(((m1 >> gbase) & grp_mask) >> G);
            const unsigned long long in_fort_b = fa_ballot(is_att && alive1 && dist_door < k_fort);
            const bool any_in_fort = ((in_fort_b >> gbase) & grp_mask) != 0ull;
            const bool just_died = alive0 && was_hit;
            const bool rewarded = (alive1 || just_died);
            const double rew = fa_reward(is_att, rewarded, prev, dist_door, shoot, hit, was_hit, n_alive_att, any_in_fort,
                                         k_fort, k_03, k_10, k_3, k_01);
            prev = rewarded ? dist_door : prev;
            if (a.track_counters) {
                ep_rew += alive0 ? rew : 0.0;
                if (done) {
                    a.s.ep_rew_sum[idx] += ep_rew;
                    if (alive1) a.s.alive_end[idx] += 1u;
                    ep_rew = 0.0;
                }
            }
            if (COLLECT) {
                *p_rew = (float)rew;
            } else {
                if (a.rew32) *p_rew = (float)rew;
                if (a.rew64) a.rew64[row] = rew;
                if (a.hit) a.hit[row] = hit ? 1 : 0;
                if (a.was_hit) a.was_hit[row] = was_hit ? 1 : 0;
            }
            p_rew += EN; row += (long long)EN;
        };
        ResetDraw rdA = {}, rdB = {};
        MtWords mw = {};
        bool need_b = false;
        auto wait_words = [&]() { if (a.rng_mode == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
        {
            rdA.base = a.rng_mode == 0 ? a.s.mt_pos[e] + 4 * i : (int)a.s.reset_count[e];
            rdB.base = draw_next_base(a, rdA.base, i, N);
            MtWords mwa = {}, mwb = {};
            draw_load(a, e, rdA.base, mwa);
            draw_load(a, e, rdB.base, mwb);
            draw_load(a, e, draw_next_base(a, rdB.base, i, N), mw);
            wait_words();
            draw_eval(a, e, i, is_att, mwa, rdA);
            draw_eval(a, e, i, is_att, mwb, rdB);
            s_rp[lane] = make_double2(rdA.px, rdA.py);
        }
        const unsigned long long lane0_m = FA_M_EQ_U(lane, 0);
        FA_TICK_INIT
        FA_WG_BARRIER(); // P(-1)
        for (int s = 0; s < ns; ++s) {
            const int b = s & 1;
            wait_state(s);
            FA_TICK(18)
            if (s > 0 && a.auto_reset != 0) {
                // envs that were reset at the end of step s-1 used draw A: commit it, promote B (ahead of this step's
                // tag: the chain wave reads s_rp behind it); the new B is drawn after the hand-off
                need_b = (s_mask[b][1] >> lane) & 1ull;
                if (need_b) {
                    draw_commit(a, e, i, N, rdA);
                    rdA = rdB;
                    s_rp[lane] = make_double2(rdA.px, rdA.py);
                }
            }
            const int act = s_act[s & (FA_ACT_BATCH - 1)][lane];
            const bool alive0 = (s_mask[b][0] >> lane) & 1ull;
            const double2 pos_ = s_pos[b][lane];
            double px = pos_.x, py = pos_.y;
            asm volatile("" : "+v"(px), "+v"(py));
            // fortattack.py:253-263,:289 _set_action (F starts as u + 0.0, core.py:221-228)
            double u0 = 0.0, u1 = 0.0, rot = 0.0;
            if (act == 1) u0 = +1.0;
            if (act == 2) u0 = -1.0;
            if (act == 3) u1 = +1.0;
            if (act == 4) u1 = -1.0;
            if (act == 5) rot = c.rot_pos;
            if (act == 6) rot = c.rot_neg;
            s_U[lane] = make_double2(u0 * c.accel + 0.0, u1 * c.accel + 0.0);
            s_rot[lane] = rot;
            double wx = 0.0, wy = 0.0;
            fa_wall_force_flat(c, px, py, wx, wy); // core.py:246-252 + :459-472
            s_W[lane] = make_double2(alive0 ? wx : 0.0, alive0 ? wy : 0.0);
            FA_ORDER();
            if (fa_lanes(lane0_m)) s_tag[2] = s;   // behind the wave's data writes: in order
            FA_ORDER();
            FA_TICK(16)
            if (need_b) {
                wait_words();
                rdB.base = draw_next_base(a, rdA.base, i, N);
                draw_eval(a, e, i, is_att, mw, rdB);
                draw_load(a, e, draw_next_base(a, rdB.base, i, N), mw);
            }
            need_b = false;
            if (s > 0) emit_rew(s - 1);
            act_prev = act;
            alive0_prev = alive0;
            FA_TICK(17)
            if (!FA_CHAIN_NOBAR) FA_WG_BARRIER(); // P(s)
        }
        FA_WG_BARRIER(); // the chain wave has published the last step's by-products
        FA_TICK_FLUSH(16, 19, 30)
        if (a.auto_reset != 0 && ((s_mask[ns & 1][1] >> lane) & 1ull)) draw_commit(a, e, i, N, rdA);
        emit_rew(ns - 1);
        a.s.prev[idx] = prev;
        if (a.track_counters) a.s.ep_rew[idx] = ep_rew;
        return;
    }
    if (wave_id == 3) {
        // ---- wave 3: the partner offsets DF+1..NOFF of the soft contacts of step s -> chain wave; then the observation,
        // done and mask rows + episode bookkeeping of step s-1 ----------------------------------------------------
        uint8_t *p_done = a.done ? a.done + e : nullptr;
        float *p_mask = a.mask32 ? a.mask32 + idx : nullptr;
        float *p_obs = a.obs32 ? a.obs32 + idx * 6 : nullptr;
        long long row6 = (long long)idx * 6;
        asm volatile("" : "+v"(p_done), "+v"(p_mask), "+v"(p_obs), "+v"(row6));
        bool alive0_prev = false;
        auto emit_obs = [&](int bo) { // the state after the step / reset (fortattack_env_v1.py:238)
            const bool alive_new = (s_mask[bo][0] >> lane) & 1ull;
            const double2 pos_ = s_pos[bo][lane], vel_ = s_vel[bo][lane];
            const double px = pos_.x, py = pos_.y, ang = s_ang[bo][lane];
            const double vx = vel_.x, vy = vel_.y;
            fa_store_obs((COLLECT || a.obs32) ? p_obs : nullptr, (!COLLECT && a.obs64) ? a.obs64 + row6 : nullptr, alive_new,
                         px, py, ang, vx, vy);
            p_obs += EN * 6; row6 += (long long)EN * 6;
        };
        auto emit_flags = [&](int se) { // done / mask rows and _get_done bookkeeping of step se (fortattack.py:202-225)
            const int pl = se % 3, pb = (se + 1) & 1;
            const unsigned long long m1 = s_las[pl][0];
            const bool alive1 = (m1 >> lane) & 1ull;
            const bool done = (s_mask[pb][1] >> lane) & 1ull;
            const int n_alive_att = __popcll(((m1 >> gbase) & grp_mask) >> G);
            const unsigned long long in_fort_b = fa_ballot(is_att && alive1 && s_dd[pb][lane] <= c.fort2_max);
            const bool any_in_fort = ((in_fort_b >> gbase) & grp_mask) != 0ull;
            if (i == 0) {
                if (done) {
                    const int which = any_in_fort ? 2 : (n_alive_att == 0 ? 0 : 1);
                    uint8_t *gr = a.s.game_result + (size_t)e * 3;
                    gr[0] = which == 0; gr[1] = which == 1; gr[2] = which == 2;
                    atomicAdd(a.s.result_count + (size_t)e * 3 + which, 1u);
                }
                if (COLLECT || a.done) *p_done = done ? 1 : 0;
            }
            const float mk = (alive0_prev || (done && a.auto_reset != 0)) ? 1.0f : 0.0f;
            if (COLLECT || a.mask32) *p_mask = mk;
            p_mask += EN; p_done += a.E;
        };
        const unsigned long long lane0_m = FA_M_EQ_U(lane, 0);
        FA_TICK_INIT
        FA_WG_BARRIER(); // P(-1)
        for (int s = 0; s < ns; ++s) {
            const int b = s & 1;
            wait_state(s);
            FA_TICK(14)
            const unsigned long long grp_alive0 = (s_mask[b][0] >> gbase) & grp_mask;
            const bool alive0 = (grp_alive0 >> i) & 1ull;
            const double2 pos_ = s_pos[b][lane];
            const double px = pos_.x, py = pos_.y;
            double qx[NOFF], qy[NOFF];
#pragma unroll
            for (int d = DF + 1; d <= NOFF; ++d) {
                int j = i + d;
                j = j >= N ? j - N : j;
                const double2 q_ = s_pos[b][gbase + j];
                qx[d - 1] = q_.x;
                qy[d - 1] = q_.y;
            }
#pragma unroll
            for (int d = DF + 1; d <= NOFF; ++d) {
                int j = i + d;
                j = j >= N ? j - N : j;
                const bool mine = (2 * d != N) || (i < N / 2); // the half offset: one side only
                const double dx = px - qx[d - 1], dy = py - qy[d - 1];
                const double d2 = dx * dx + dy * dy;
                double fxv = 0.0, fyv = 0.0;
                bool near = false;
                if (mine && alive0 && ((grp_alive0 >> j) & 1ull) && !(d2 > c.contact_skip_d2)) {
                    fa_contact_force(c, dx, dy, d2, fxv, fyv);
                    near = true;
                }
                if (mine) {
                    s_fm[j][lane] = make_double2(fxv, fyv);                                   // on agent i from partner j
                    s_fm[i][gbase + j] = make_double2(near ? -fxv : 0.0, near ? -fyv : 0.0); // on agent j from partner i: the exact negative
                }
            }
            FA_ORDER();
            if (fa_lanes(lane0_m)) s_tag[3] = s;   // behind the wave's data writes: in order
            FA_ORDER();
            FA_TICK(8)
            if (FA_CHAIN_TRIG_WAVE == 3 && s + 1 < ns) { // the heading of step s+1 (see wave 1)
                const int act = s_act[s & (FA_ACT_BATCH - 1)][lane];
                double rot = 0.0, sn, cs;
                if (act == 5) rot = c.rot_pos;
                if (act == 6) rot = c.rot_neg;
                sincos_heading(s_ang[b][lane] + rot, sn, cs);
                s_trig[(s + 1) & 1][lane] = make_double2(cs, sn);
            }
            if (s > 0) {
                emit_obs(b);
                emit_flags(s - 1);
            }
            alive0_prev = alive0;
            FA_TICK(9)
            if (!FA_CHAIN_NOBAR) FA_WG_BARRIER(); // P(s)
        }
        FA_WG_BARRIER(); // the chain wave has published the last step's by-products
        FA_TICK_FLUSH(8, 10, 31)
        FA_TICK_FLUSH(14, 15, 27)
        emit_obs(ns & 1);
        emit_flags(ns - 1);
        return;
    }

    // ---- wave 0: the chain ---------------------------------------------------------------------------------------
    __builtin_amdgcn_s_setprio(3);
    double px = a.s.px[idx], py = a.s.py[idx], vx = a.s.vx[idx], vy = a.s.vy[idx];
    double ang = a.s.ang[idx];
    unsigned long long alive_m = FA_M_NE_U(a.s.alive[idx], 0);
    int t = a.s.tstep[e];
    unsigned long long dirty_m = 0ull;
    const unsigned long long is_att_m = FA_M_NE_U(is_att ? 1u : 0u, 0), lane0_m = FA_M_EQ_U(lane, 0);
    const int64_t *act_ptr = a.actions + (int64_t)e * a.as_e + (int64_t)i * a.as_i;
    int av[FA_ACT_BATCH];
#pragma unroll
    for (int k = 0; k < FA_ACT_BATCH; ++k) av[k] = (k < ns) ? (int)act_ptr[(int64_t)k * a.as_t] : 0;
    constexpr unsigned grp_bits = (1u << N) - 1u;
#pragma unroll
    for (int k = 0; k < FA_ACT_BATCH; ++k) s_act[k][lane] = av[k];
#pragma unroll
    for (int k = 0; k < FA_ACT_BATCH; ++k)
        av[k] = (FA_ACT_BATCH + k < ns) ? (int)act_ptr[(int64_t)(FA_ACT_BATCH + k) * a.as_t] : 0;
    s_pos[0][lane] = make_double2(px, py);
    s_ang[0][lane] = ang;
    if (fa_lanes(lane0_m)) {
        s_mask[0][0] = alive_m;
        s_mask[0][1] = 0ull;
        s_tag[0] = 0; s_tag[1] = -1; s_tag[2] = -1; s_tag[3] = -1;
    }
    s_fm[i][lane] = make_double2(0.0, 0.0); // an agent exerts no force on itself: nobody writes the diagonal
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double qx[DF], qy[DF]; // the partners i+1 .. i+DF at the step's start: read back from this wave's own publish
#pragma unroll
    for (int d = 1; d <= DF; ++d) {
        int j = i + d;
        j = j >= N ? j - N : j;
        const double2 q_ = s_pos[0][gbase + j];
        qx[d - 1] = q_.x;
        qy[d - 1] = q_.y;
    }
    FA_WG_BARRIER(); // P(-1)

    double k_damp = c.one_minus_damping, k_dt = c.dt, k_sp2 = c.speed2_max, k_vmax = c.max_speed;
    double k_doorx = c.door_x, k_doory = c.door_y, k_fort2 = c.fort2_max;
    double k_ang_r = is_att ? c.ang_attacker : c.ang_guard;
    asm volatile("" : "+v"(k_damp), "+v"(k_dt), "+v"(k_sp2), "+v"(k_vmax));
    asm volatile("" : "+v"(k_doorx), "+v"(k_doory), "+v"(k_fort2), "+v"(k_ang_r));
    FA_TICK_INIT
    for (int s = 0; s < ns; ++s) {
        const int b = s & 1, nb = (s + 1) & 1;
        FA_TICK(4)
        const unsigned long long alive0_m = alive_m;
        const unsigned grp_alive0 = (unsigned)(alive0_m >> gbase) & grp_bits;
        const bool alive0 = fa_lanes(alive0_m);
        // ---- core.py:231-243, :440-456: this wave's partner offsets, once per unordered pair ----------------------
#pragma unroll
        for (int d = 1; d <= DF; ++d) {
            int j = i + d;
            j = j >= N ? j - N : j;
            const bool mine = (2 * d != N) || (i < N / 2);
            const double dx = px - qx[d - 1], dy = py - qy[d - 1];
            const double d2 = dx * dx + dy * dy;
            double fxv = 0.0, fyv = 0.0;
            bool near = false;
            if (mine && alive0 && ((grp_alive0 >> j) & 1u) && !(d2 > c.contact_skip_d2)) {
                fa_contact_force(c, dx, dy, d2, fxv, fyv);
                near = true;
            }
            if (mine) {
                s_fm[j][lane] = make_double2(fxv, fyv);
                s_fm[i][gbase + j] = make_double2(near ? -fxv : 0.0, near ? -fyv : 0.0);
            }
        }
        FA_TICK(0)
        // ---- the helpers' hand-offs of step s: tags first, data behind them, one burst; again while a tag is stale ----
        unsigned long long alive1_m;
        double fmx[N], fmy[N], wx, wy, u0, u1, rot;
        for (;;) {
            FA_ORDER();
            const int t1 = s_tag[1], t2 = s_tag[2], t3 = s_tag[3];
            FA_ORDER();
            alive1_m = s_las[s % 3][0];
            const double2 w_ = s_W[lane], u_ = s_U[lane];
            wx = w_.x; wy = w_.y;
            u0 = u_.x; u1 = u_.y; rot = s_rot[lane];
#pragma unroll
            for (int j = 0; j < N; ++j) { const double2 f_ = s_fm[j][lane]; fmx[j] = f_.x; fmy[j] = f_.y; }
            FA_ORDER();
            const bool ok = (t1 == s) & (t2 == s) & (t3 == s);
            if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
            if (FA_CHAIN_SLEEP > 0) __builtin_amdgcn_s_sleep(FA_CHAIN_SLEEP);
        }
        FA_TICK(1)
        const unsigned ga1 = (unsigned)(alive1_m >> gbase) & grp_bits;     // survivors of the lane's env
        const int n_alive_att = __popc(ga1 >> G);
        const bool restage = ((s + 1) & (FA_ACT_BATCH - 1)) == 0;
        // ---- core.py:221-252: F = u + 0, the pairs in the reference's order (partner j ascending), then the walls ----
        if (fa_lanes(alive1_m)) {
            double Fx = u0, Fy = u1;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const double m = (double)((ga1 >> j) & 1u);
                Fx = __fma_rn(fmx[j], m, Fx);
                Fy = __fma_rn(fmy[j], m, Fy);
            }
            Fx = wx + Fx;
            Fy = wy + Fy;
            // core.py:324-338 integrate_state (mass == 1.0: F/1.0 is exact)
            const double vdx = vx * k_damp, vdy = vy * k_damp;
            vx = vdx + Fx * k_dt;
            vy = vdy + Fy * k_dt;
            double speed2 = vx * vx + vy * vy;
            if (__builtin_expect(!(speed2 <= k_sp2), 0)) {
                if (speed2 != speed2) { // coincident agents: see fa_step_pipe_kernel
                    double Gx = u0, Gy = u1;
#pragma unroll
                    for (int j = 0; j < N; ++j)
                        if ((ga1 >> j) & 1u) { const double2 f_ = s_fm[j][lane]; Gx = f_.x + Gx; Gy = f_.y + Gy; }
                    Gx = wx + Gx;
                    Gy = wy + Gy;
                    vx = vdx + Gx * k_dt;
                    vy = vdy + Gy * k_dt;
                    speed2 = vx * vx + vy * vy;
                }
                if (speed2 > k_sp2) {
                    const double speed = sqrt_rn(speed2);
                    vx = div_rn(vx, speed) * k_vmax;
                    vy = div_rn(vy, speed) * k_vmax;
                }
            }
            ang += rot;
            px += vx * k_dt;
            py += vy * k_dt;
        }
        FA_TICK(2)
        // ---- what the next state needs of the reward / done logic (see fa_step_pipe_kernel) ------------------------
        const double ddx = px - k_doorx, ddy = py - k_doory;
        const double dd2 = ddx * ddx + ddy * ddy;
        const unsigned long long in_fort_m = FA_M_LE_D(dd2, k_fort2) & is_att_m & alive1_m;
        const unsigned gw_fort = (unsigned)(in_fort_m >> gbase) & grp_bits;
        const unsigned long long done_m = FA_M_NE_U(gw_fort, 0) | FA_M_EQ_U(n_alive_att, 0) | FA_M_EQ_U(t, a.max_t - 1);
        const unsigned long long reset_m = a.auto_reset != 0 ? done_m : 0ull;
        t += 1;                                                        // fortattack.py:171
        alive_m = alive1_m;
        dirty_m |= alive0_m;
        // ---- fortattack_env_v1.py:47-75 reset_world (prevDist is NOT reset: quirk Q1) ------------------------------
        if (__builtin_expect(reset_m != 0ull, 0)) {
            const double2 rp_ = s_rp[lane];
            const double rpx = rp_.x, rpy = rp_.y;
            if (fa_lanes(reset_m)) {
                px = rpx; py = rpy; vx = 0.0; vy = 0.0;
                ang = k_ang_r;
                t = 0;
            }
            alive_m |= reset_m;
            dirty_m |= reset_m;
        }
        // ---- publish state(s+1) and the by-products of step s -------------------------------------------------------
        s_pos[nb][lane] = make_double2(px, py);
        s_ang[nb][lane] = ang;
        s_vel[nb][lane] = make_double2(vx, vy);
        s_dd[nb][lane] = dd2;
        if (fa_lanes(lane0_m)) {
            s_mask[nb][0] = alive_m;
            s_mask[nb][1] = done_m;
        }
        if (__builtin_expect(restage, 0)) { // (every helper has read this step's action: their tags said so)
#pragma unroll
            for (int k = 0; k < FA_ACT_BATCH; ++k) s_act[k][lane] = av[k];
        }
        // the partners' positions of the next step: this wave's own writes, in order -- they land while the barrier's LDS
        // wait drains, so nothing stands between P(s) and the pair forces of step s+1
#pragma unroll
        for (int d = 1; d <= DF; ++d) {
            int j = i + d;
            j = j >= N ? j - N : j;
            const double2 q_ = s_pos[nb][gbase + j];
            qx[d - 1] = q_.x;
            qy[d - 1] = q_.y;
        }
        if (FA_CHAIN_NOBAR) { // state(s+1) is complete: tag behind the data
            FA_ORDER();
            if (fa_lanes(lane0_m)) s_tag[0] = s + 1;
            FA_ORDER();
        }
        FA_TICK(3)
        if (!FA_CHAIN_NOBAR) FA_WG_BARRIER(); // P(s)
        if (__builtin_expect(restage, 0)) {
#pragma unroll
            for (int k = 0; k < FA_ACT_BATCH; ++k)
                av[k] = (s + 1 + FA_ACT_BATCH + k < ns) ? (int)act_ptr[(int64_t)(s + 1 + FA_ACT_BATCH + k) * a.as_t] : 0;
        }
    }
    FA_WG_BARRIER(); // (epilogues of the emitting waves)
    FA_TICK_FLUSH(0, 5, 28)
    if (fa_lanes(dirty_m)) {
        a.s.px[idx] = px; a.s.py[idx] = py; a.s.vx[idx] = vx; a.s.vy[idx] = vy;
        a.s.ang[idx] = ang;
        a.s.alive[idx] = fa_lanes(alive_m) ? 1 : 0;
    }
    if (i == 0) a.s.tstep[e] = t;
#undef FA_ORDER
}


// ---- launcher: what the product dispatch calls when a variant library carries this translation unit ----
extern "C" hipError_t fa_launch_step_experiment(const FaStepArgs &a, int which, bool collect, hipStream_t st) {
    if (which == FA_KERNEL_EXP_CHAIN) {
        const int epw = FA_WAVE / (a.G + a.A);
        const int grid = (a.E + epw - 1) / epw;
        if (a.G == 3 && a.A == 3) {
            if (collect) hipLaunchKernelGGL((fa_step_chain_kernel<3, 3, true>), dim3(grid), dim3(4 * FA_WAVE), 0, st, a);
            else hipLaunchKernelGGL((fa_step_chain_kernel<3, 3, false>), dim3(grid), dim3(4 * FA_WAVE), 0, st, a);
        } else if (a.G == 5 && a.A == 5) {
            if (collect) hipLaunchKernelGGL((fa_step_chain_kernel<5, 5, true>), dim3(grid), dim3(4 * FA_WAVE), 0, st, a);
            else hipLaunchKernelGGL((fa_step_chain_kernel<5, 5, false>), dim3(grid), dim3(4 * FA_WAVE), 0, st, a);
        } else {
            return hipErrorInvalidValue;
        }
    } else if (which == FA_KERNEL_EXP_PAIRS && a.G == 3 && a.A == 3) {
        const int g2 = (a.E + 1) / 2; // two envs per wave
        if (collect) {
            if (a.choice_k > 0) hipLaunchKernelGGL((fa_step_pair_kernel<3, 3, true, true>), dim3(g2), dim3(FA_WAVE), 0, st, a);
            else hipLaunchKernelGGL((fa_step_pair_kernel<3, 3, true, false>), dim3(g2), dim3(FA_WAVE), 0, st, a);
        } else {
            if (a.choice_k > 0) hipLaunchKernelGGL((fa_step_pair_kernel<3, 3, false, true>), dim3(g2), dim3(FA_WAVE), 0, st, a);
            else hipLaunchKernelGGL((fa_step_pair_kernel<3, 3, false, false>), dim3(g2), dim3(FA_WAVE), 0, st, a);
        }
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
