// fa_step_classic.hip -- fa_step_kernel: the whole env-step on one wave (or with one / two stateless helper waves): single-step
// launches (the closed loop), run-time team sizes, the ensemble path's choice, grids beyond 768 workgroups, fa_reset;
// fa_seed_kernel; the launch dispatch of all step kernels; the divide / sqrt self-test.
#include "fa_step_common.h"
#include "fortattack.h" // FA_KERNEL_*
#include "experiments/fa_step_experiments.h"

// CHOICE: the ensemble path's np.random.choice after every reset (fa_set_reset_choice) is compiled in.  A
// template parameter, not a run-time test: the draw loop's loads and stores inside the reset block cost the
// ordinary build 13 % at large E even when never executed.
template <int TG, int TA, bool RESET_ONLY, bool COLLECT, int NW, bool CHOICE>
__global__ __launch_bounds__(NW * FA_WAVE) void fa_step_kernel(FaStepArgs a) {
    constexpr bool TWO = NW >= 2;    // wave 1: contact forces (+ walls when NW == 2)
    constexpr bool THREE = NW >= 3;  // wave 2: wall forces
    const int G = TG ? TG : a.G, A = TA ? TA : a.A;
    const int N = G + A;
    const int EPW = FA_WAVE / N;          // envs per wave
    const int lane = threadIdx.x & (FA_WAVE - 1);
    const int wave_id = threadIdx.x / FA_WAVE;
    const bool force_wave = TWO && wave_id >= 1;
    const int slot = lane / N;            // env slot inside the wave
    const int i = lane - slot * N;        // agent index
    const int gbase = slot * N;           // first lane of this env's group
    const int e = blockIdx.x * EPW + slot;
    if (!((slot < EPW) && (e < a.E))) return; // padding lanes leave: ballots count live lanes only
    const bool is_att = i >= G;
    const size_t idx = (size_t)e * N + i;
    const size_t EN = (size_t)a.E * N;
    const unsigned long long grp_mask = (1ull << N) - 1ull;
    const FaDerived &c = a.c;

    __shared__ double2 s_pos[FA_WAVE], s_trig[FA_WAVE]; // positions; (cos, sin) of the shooters' headings -- (x, y) pairs side by
                                                         // side: one 16-byte LDS operation per pair (see fa_step_pipe_kernel)
    __shared__ int s_act[FA_ACT_BATCH][FA_WAVE];
    __shared__ double2 s_F[TWO ? FA_WAVE : 1];   // TWO: total force per lane, from the force wave
    __shared__ double2 s_W[THREE ? FA_WAVE : 1]; // THREE: wall force per lane, from the wall wave
    __shared__ unsigned long long s_mask[2];        // TWO: ballots of alive-before / alive-after-laser

    if constexpr (TWO) {
        if (force_wave) {
            constexpr int NT = TG + TA;
            const int ns = a.nsteps;
            auto wall_force = [&](bool alive0, double px, double py, double &wx, double &wy) {
                wx = 0.0;
                wy = 0.0;
                if (alive0) fa_wall_force(c, px, py, wx, wy);
            };
            if (THREE && wave_id == 2) {
                // ---- the wall wave ---------------------------------------------------------------
                for (int s = 0; s < ns; ++s) {
                    FA_WG_BARRIER(); // (1)
                    const bool alive0 = (s_mask[0] >> lane) & 1ull;
                    double wx, wy;
                    const double2 pos_ = s_pos[lane];
                    wall_force(alive0, pos_.x, pos_.y, wx, wy);
                    s_W[lane] = make_double2(wx, wy);
                    FA_WG_BARRIER(); // (2)
                    FA_WG_BARRIER(); // (3)
                }
                return;
            }
            // ---- the force wave: core.py:221-252 for its lane's agent, every step ----------------
            for (int s = 0; s < ns; ++s) {
                FA_WG_BARRIER(); // (1) actions, positions and the alive-before ballot are staged
                const int act = s_act[s & (FA_ACT_BATCH - 1)][lane];
                double u0 = 0.0, u1 = 0.0;
                if (act == 1) u0 = +1.0;
                if (act == 2) u0 = -1.0;
                if (act == 3) u1 = +1.0;
                if (act == 4) u1 = -1.0;
                u0 *= c.accel;
                u1 *= c.accel;
                const unsigned long long grp_alive0 = (s_mask[0] >> gbase) & grp_mask;
                const bool alive0 = (grp_alive0 >> i) & 1ull;
                const double2 pos_ = s_pos[lane];
                const double px = pos_.x, py = pos_.y;
                // candidate pair force against every partner alive BEFORE the laser (a partner the
                // laser kills this step is masked out below); exactly +0.0 when out of range, so
                // that adding it is a no-op (F is never -0.0)
                double fxj[NT], fyj[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const double2 q_ = s_pos[gbase + j];
                    const double dx = px - q_.x, dy = py - q_.y;
                    const double d2 = dx * dx + dy * dy;
                    fxj[j] = 0.0;
                    fyj[j] = 0.0;
                    if (alive0 && j != i && ((grp_alive0 >> j) & 1ull) && !(d2 > c.contact_skip_d2)) {
                        fa_contact_force(c, dx, dy, d2, fxj[j], fyj[j]);
                    }
                }
                double wx = 0.0, wy = 0.0;
                if (!THREE) wall_force(alive0, px, py, wx, wy);
                FA_WG_BARRIER(); // (2) the alive-after-laser ballot is published
                const unsigned long long grp_alive1 = (s_mask[1] >> gbase) & grp_mask;
                if (THREE) { const double2 w_ = s_W[lane]; wx = w_.x; wy = w_.y; }
                double Fx = u0 + 0.0, Fy = u1 + 0.0;   // core.py:221-228
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    if ((grp_alive1 >> j) & 1ull) { // ascending partner order == the reference's pair order
                        Fx = fxj[j] + Fx;
                        Fy = fyj[j] + Fy;
                    }
                Fx = wx + Fx;
                Fy = wy + Fy;
                s_F[lane] = make_double2(Fx, Fy);
                FA_WG_BARRIER(); // (3) forces are published
            }
            return;
        }
    }

    // ---- load state once per launch (coalesced: lane-contiguous) --------------------
    double px = 0, py = 0, vx = 0, vy = 0, ang = 0, prev = 0;
    bool alive = false;
    int t = 0, nh = 0, nwh = 0;
    double ep_rew = 0.0; // episode return so far (reward * alive-before mask), track_counters only
    {
        px = a.s.px[idx]; py = a.s.py[idx]; vx = a.s.vx[idx]; vy = a.s.vy[idx];
        ang = a.s.ang[idx]; prev = a.s.prev[idx];
        alive = a.s.alive[idx] != 0;
        t = a.s.tstep[e];
        if (a.track_counters) { nh = a.s.num_hit[idx]; nwh = a.s.num_was_hit[idx]; ep_rew = a.s.ep_rew[idx]; }
    }
    bool dirty = false; // state changed => write it back
    int mt_base = (a.rng_mode == 0 ? a.s.mt_pos[e] : 0) + 4 * i; // cursor + 4*i of this lane's next reset draw

    const int nsteps = RESET_ONLY ? 1 : a.nsteps;
    const int64_t *act_ptr = RESET_ONLY ? nullptr : a.actions + (int64_t)e * a.as_e + (int64_t)i * a.as_i;
    // Actions reach the step through LDS in batches of FA_ACT_BATCH steps: gfx9 vector memory
    // returns in order, so a per-step action load would make every step wait (s_waitcnt vmcnt)
    // for the acknowledgement of its predecessor's stores.  One batch = 16 loads in flight,
    // one wait per 16 steps; inside a batch the step only touches LDS (lgkmcnt).
    // (double buffered: the loads of batch b+1 are issued while batch b is being stepped, so
    // the wait at a batch boundary only sees the most recent stores, not a load round trip)
    int av[FA_ACT_BATCH];
#pragma unroll
    for (int k = 0; k < FA_ACT_BATCH; ++k)
        av[k] = (!RESET_ONLY && k < nsteps) ? (int)act_ptr[(int64_t)k * a.as_t] : 0; // uniform
    for (int s = 0; s < nsteps; ++s) {
        if (!RESET_ONLY && (s & (FA_ACT_BATCH - 1)) == 0) {
#pragma unroll
            for (int k = 0; k < FA_ACT_BATCH; ++k) s_act[k][lane] = av[k];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < FA_ACT_BATCH; ++k)
                av[k] = (s + FA_ACT_BATCH + k < nsteps) ? (int)act_ptr[(int64_t)(s + FA_ACT_BATCH + k) * a.as_t] : 0;
        }
        bool do_reset;
        if (RESET_ONLY) {
            do_reset = (a.reset_mask == nullptr || a.reset_mask[e] != 0);
        } else {
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

            // ---- stage positions + the shooters' heading sin/cos in LDS (core.py:373-382) ------------
            s_pos[lane] = make_double2(px, py);
            if constexpr (TWO) {
                const unsigned long long alive0_b = __ballot(alive0);
                if (lane == 0) s_mask[0] = alive0_b;
                FA_WG_BARRIER(); // (1) the force wave starts on this step's contacts and walls
            }
            const bool shooter = alive0 && shoot;
            if (shooter) { // the laser test needs the shooter's position and sin/cos of its heading (fa_wedge)
                double sn, cs;
                sincos_heading(ang, sn, cs);
                s_trig[lane] = make_double2(cs, sn);
            }
            const unsigned long long shooters_b = __ballot(shooter);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            // partner deltas for the contact test, fetched now (they do not depend on the laser):
            // the LDS latency and the 5 flops per partner overlap with the laser tests below
            // (only for small teams: at N = 10 the 30 extra live doubles push the kernel past
            // 256 VGPRs and into scratch)
            constexpr int NT = (TG != 0) ? TG + TA : 0;
            constexpr bool HOIST = !TWO && NT != 0 && NT <= 8;
            double dxs[NT ? NT : 1], dys[NT ? NT : 1], d2s[NT ? NT : 1];
            if constexpr (HOIST) {
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const double2 q_ = s_pos[gbase + j];
                    dxs[j] = px - q_.x;
                    dys[j] = py - q_.y;
                    d2s[j] = dxs[j] * dxs[j] + dys[j] * dys[j];
                }
            }

            // ---- core.py:254-302 apply_laser_effect ------------------------------------
            // iteration k: every lane tests the triangle of its k-th opponent; the ballot
            // of the results gives shooter k of either team its hit list.
            bool was_hit = false;
            int hit_cnt = 0, was_hit_cnt = 0;
            if (shooters_b != 0ull) {
                const int n_opp = is_att ? G : A, opp0 = is_att ? 0 : G;
                const int team_idx = is_att ? i - G : i;
                const unsigned long long opp_mask =
                    is_att ? ((1ull << G) - 1ull) : (((1ull << A) - 1ull) << G);
                constexpr int KT = TG > TA ? TG : TA;
                if constexpr (KT != 0) {
                    // compile-time team sizes: all opponents are fetched from LDS in one batch
                    double tr[KT][4];
                    bool hk[KT];
#pragma unroll
                    for (int k = 0; k < KT; ++k) {
                        const int j = gbase + opp0 + (k < n_opp ? k : 0);
                        const double2 q_ = s_pos[j], tg_ = s_trig[j];
                        tr[k][0] = q_.x; tr[k][1] = q_.y; tr[k][2] = tg_.x; tr[k][3] = tg_.y;
                    }
#pragma unroll
                    for (int k = 0; k < KT; ++k) {
                        const int j = gbase + opp0 + k;
                        const bool cand = alive0 && k < n_opp && ((shooters_b >> j) & 1ull);
                        double u, lhs, rhs;
                        fa_wedge(c.agent_size, c.cos_hw, c.sin_hw, px, py, tr[k][0], tr[k][1], tr[k][2], tr[k][3], u, lhs, rhs);
                        hk[k] = cand & (u <= c.shoot_far) & (lhs <= rhs);
                    }
                    // the hit list of shooter k of either team is ballot k: pick the lane's own
                    // (uniform values, per-lane select), then one shift / mask / popcount
                    unsigned long long my_hb = 0ull;
#pragma unroll
                    for (int k = 0; k < KT; ++k) {
                        const unsigned long long hb = __ballot(hk[k]);
                        my_hb = (k == team_idx) ? hb : my_hb;
                        was_hit = was_hit | hk[k];
                        was_hit_cnt += hk[k] ? 1 : 0;
                    }
                    hit_cnt = __popcll((my_hb >> gbase) & opp_mask);
                } else {
                    const int KMAX = G > A ? G : A;
                    for (int k = 0; k < KMAX; ++k) {
                        const int j = gbase + opp0 + k;
                        bool h = false;
                        if (alive0 && k < n_opp && ((shooters_b >> j) & 1ull)) {
                            double u, lhs, rhs;
                            const double2 q_ = s_pos[j], tg_ = s_trig[j];
                            fa_wedge(c.agent_size, c.cos_hw, c.sin_hw, px, py, q_.x, q_.y, tg_.x, tg_.y, u, lhs, rhs);
                            h = (u <= c.shoot_far) & (lhs <= rhs);
                        }
                        const unsigned long long hb = __ballot(h);
                        if (k == team_idx) hit_cnt = __popcll((hb >> gbase) & opp_mask);
                        was_hit = was_hit || h;
                        was_hit_cnt += h ? 1 : 0;
                    }
                }
            }
            const bool hit = shooter && hit_cnt > 0;
            const bool alive1 = alive0 && !was_hit;       // :293-302 one shot kills
            const bool just_died = alive0 && was_hit;
            const unsigned long long alive1_b = __ballot(alive1);
            const unsigned long long grp_alive1 = (alive1_b >> gbase) & grp_mask;
            const int n_alive_att = __popcll(grp_alive1 >> G);
            if constexpr (TWO) {
                if (lane == 0) s_mask[1] = alive1_b;
                FA_WG_BARRIER(); // (2) the force wave masks its candidates with the survivors
                FA_WG_BARRIER(); // (3) and has published the total force of every lane
            }

            // ---- forces + integration for agents alive after the laser ----------------
            if (alive1) {
                double Fx = u0 + 0.0, Fy = u1 + 0.0;      // core.py:221-228
                if constexpr (TWO) {
                    const double2 f_ = s_F[lane];
                    Fx = f_.x;
                    Fy = f_.y;
                } else {
                // core.py:231-243 + :440-456.  Reference order: pairs (a,b), a<b, lexicographic;
                // for agent i that is partner j ascending, with f_i = +f for j>i and
                // -(f(j,i)) for j<i, which is bitwise the same number as f computed from
                // i's side (negation commutes exactly with *, / and the sqrt argument).
                if constexpr (NT != 0) {
                    if constexpr (!HOIST) {
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            const double2 q_ = s_pos[gbase + j];
                            dxs[j] = px - q_.x;
                            dys[j] = py - q_.y;
                            d2s[j] = dxs[j] * dxs[j] + dys[j] * dys[j];
                        }
                    }
                    // only partners actually in range take the slow path
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        // exact skip: farther than dist_min + 1000*margin => t < -1000 =>
                        // exp(t) == +0 => penetration == +0.0 => force == +-0.0, and F (never
                        // -0.0) is unchanged by adding it.
                        if (j == i || !((grp_alive1 >> j) & 1ull) || d2s[j] > c.contact_skip_d2) continue;
                        double fx, fy;
                        fa_contact_force(c, dxs[j], dys[j], d2s[j], fx, fy);
                        Fx = fx + Fx;
                        Fy = fy + Fy;
                    }
                } else {
                    for (int j = 0; j < N; ++j) {
                        if (j == i || !((grp_alive1 >> j) & 1ull)) continue;
                        const double2 q_ = s_pos[gbase + j];
                        const double dx = px - q_.x, dy = py - q_.y;
                        const double d2 = dx * dx + dy * dy;
                        if (d2 > c.contact_skip_d2) continue; // exact skip, see above
                        double fx, fy;
                        fa_contact_force(c, dx, dy, d2, fx, fy);
                        Fx = fx + Fx;
                        Fy = fy + Fy;
                    }
                }
                // core.py:246-252 + :459-472 walls
                {
                    double wx, wy;
                    fa_wall_force(c, px, py, wx, wy);
                    Fx = wx + Fx;
                    Fy = wy + Fy;
                }
                } // !TWO
                // core.py:324-338 integrate_state (mass == 1.0: F/1.0 is exact)
                vx = vx * c.one_minus_damping;
                vy = vy * c.one_minus_damping;
                vx += Fx * c.dt;
                vy += Fy * c.dt;
                // `sqrt(v.v) > max_speed` decided without the sqrt: speed2_max is the largest
                // double whose correctly rounded sqrt is <= max_speed (found on the host), so
                // the comparison below is the reference's comparison, exactly.
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
            const unsigned long long in_fort_b =
                __ballot(is_att && alive1 && dist_door < c.fort_dim);
            const bool any_in_fort = ((in_fort_b >> gbase) & grp_mask) != 0ull;
            const bool rewarded = (alive1 || just_died);
            const double rew = fa_reward(is_att, rewarded, prev, dist_door, shoot, hit, was_hit, n_alive_att, any_in_fort,
                                         c.fort_dim, 0.3, 10.0, 3.0, 0.1);
            prev = rewarded ? dist_door : prev;

            // ---- fortattack.py:202-225 _get_done, :171 time_step += 1 ------------------
            const bool timeout = t == a.max_t - 1;
            const bool done = any_in_fort || n_alive_att == 0 || timeout;
            if (i == 0) {
                if (done) {
                    const int which = any_in_fort ? 2 : (n_alive_att == 0 ? 0 : 1);
                    uint8_t *gr = a.s.game_result + (size_t)e * 3;
                    gr[0] = which == 0; gr[1] = which == 1; gr[2] = which == 2;
                    atomicAdd(a.s.result_count + (size_t)e * 3 + which, 1u); // no-return atomic: no wait
                }
                if (COLLECT || a.done) a.done[(size_t)s * a.E + e] = done ? 1 : 0;
            }
            t += 1;
            // evaluation statistics (test_fortattack_v2.py:88-101): episode_rewards += reward*mask;
            // at the end of an episode: who is alive, and the episode's return per agent
            if (a.track_counters) {
                ep_rew += alive0 ? rew : 0.0;
                if (done) {
                    a.s.ep_rew_sum[idx] += ep_rew;
                    if (alive1) a.s.alive_end[idx] += 1u;
                    ep_rew = 0.0;
                    dirty = true;
                }
            }
            do_reset = done && a.auto_reset != 0;
            dirty = dirty || alive0;
            alive = alive1;
            nh += hit_cnt;
            nwh += was_hit_cnt; // one per shooter that hit (core.py:283)

            // step-level outputs (the reset below must not touch them)
            {
                const size_t o = (size_t)s * EN + idx;
                // trainer mask (train_fortattack.py:53,87): alive BEFORE the step; an env that is
                // reset here gets the post-reset mask 1 (initialize_new_episode, rlagent.py:31)
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
        }

        // ---- fortattack_env_v1.py:47-75 reset_world --------------------------------------
        // (prevDist and the action are NOT reset: SURVEY quirk Q1)
        if (__ballot(do_reset) != 0ull) {
            double npx = px, npy = py;
            reset_agent(a, e, i, N, is_att, do_reset, mt_base, npx, npy);
            if (do_reset) {
                px = npx; py = npy; vx = 0.0; vy = 0.0;
                ang = is_att ? c.ang_attacker : c.ang_guard;
                alive = true;
                t = 0;
                nh = 0; nwh = 0;
                ep_rew = 0.0; // a new episode starts: an explicit reset mid-episode must not leak its partial return
                dirty = true;
                if (i == 0) {
                    reset_advance(a, e, mt_base);
                    if (RESET_ONLY) { uint8_t *gr = a.s.game_result + (size_t)e * 3; gr[0] = gr[1] = gr[2] = 0; }
                }
            }
            // ---- ensemble path: master.sample_attacker() after every env.reset() -- np.random.choice(k)
            // on the SAME stream (learner.py:119-121, train_fortattack_v2.py:29-35,104-111; quirk Q14).
            // Legacy RandomState.choice -> randint(0, k): genrand_int32() & mask until <= k - 1.
            if constexpr (CHOICE) {
                int extra = 0; // MT words the env's choice consumed
                if (do_reset && i == 0) {
                    const uint32_t rng = (uint32_t)(a.choice_k - 1);
                    uint32_t mask = rng, v = 0;
                    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
                    if (rng != 0u) {
                        if (a.rng_mode == 0) {
                            uint32_t *mt = a.s.mt + (size_t)e * FA_MT_N;
                            int c = mt_base; // lane 0: its draw base IS the env's cursor (< 624)
                            do {
                                const uint32_t nw = mt_twist(mt[c], mt[mt_wrap(c + 1)], mt[mt_wrap(c + FA_MT_M)]);
                                mt[c] = nw;
                                v = mt_temper(nw) & mask;
                                c = mt_wrap(c + 1);
                                ++extra;
                            } while (v > rng);
                            a.s.mt_pos[e] = c;
                        } else { // Philox mode: counter-based, keyed like the reset draw with agent index 255
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
                    extra = __shfl(extra, gbase); // from the env's lane 0
                    if (do_reset) mt_base = (mt_base - 4 * i + extra) % FA_MT_N + 4 * i;
                }
            }
        }

        // ---- observation row (fortattack_env_v1.py:238) ------------------------------------
        if ((!RESET_ONLY || do_reset)) {
            const size_t o6 = ((size_t)s * EN + idx) * 6;
            fa_store_obs((COLLECT || a.obs32) ? a.obs32 + o6 : nullptr, (!COLLECT && a.obs64) ? a.obs64 + o6 : nullptr,
                         alive, px, py, ang, vx, vy);
        }
        // next iteration restages LDS: keep its writes behind this iteration's reads
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    // ---- write back state once per launch ----------------------------------------------------
    if (dirty) {
        a.s.px[idx] = px; a.s.py[idx] = py; a.s.vx[idx] = vx; a.s.vy[idx] = vy;
        a.s.ang[idx] = ang; a.s.prev[idx] = prev;
        a.s.alive[idx] = alive ? 1 : 0;
        if (a.track_counters) { a.s.num_hit[idx] = nh; a.s.num_was_hit[idx] = nwh; a.s.ep_rew[idx] = ep_rew; }
    }
    if (i == 0 && (!RESET_ONLY || dirty)) a.s.tstep[e] = t;
}

// ---- np.random.seed(int): init_genrand, then discard the construction draws ----------
__global__ void fa_seed_kernel(FaState s, int E, uint64_t base_seed, int64_t env_offset, int skip_words) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    uint32_t *mt = s.mt + (size_t)e * FA_MT_N;
    uint32_t x = (uint32_t)(base_seed + (uint64_t)(env_offset + e));
    mt[0] = x;
    for (int k = 1; k < FA_MT_N; ++k) {
        x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)k;
        mt[k] = x;
    }
    int pos = 0;
    for (int w = 0; w < skip_words; ++w) {
        mt[pos] = mt_twist(mt[pos], mt[mt_wrap(pos + 1)], mt[mt_wrap(pos + FA_MT_M)]);
        pos = mt_wrap(pos + 1);
    }
    s.mt_pos[e] = pos;
    s.reset_count[e] = 0u;
}

// ---- launchers --------------------------------------------------------------------------
// the pipelined kernel lives in fa_step_pipe.hip
hipError_t fa_launch_step_pipe(const FaStepArgs &a, bool collect, bool three_per_cu, hipStream_t st);
// the round-4 experiment kernels (csrc/experiments/fa_step_experiments.hip) exist in variant libraries only
// (tools/build_variant.py): a weak symbol, null in the product library -- fa_create refuses their step_kernel values there
extern "C" __attribute__((weak)) hipError_t fa_launch_step_experiment(const FaStepArgs &a, int which, bool collect, hipStream_t st);
int fa_step_experiments_linked() { return fa_launch_step_experiment != nullptr; }

// which step kernel a launch of `nsteps` env-steps uses: 0 = pipelined (two workgroups per CU build),
// -3 = pipelined (three per CU build), 1/2/3 = fa_step_kernel with that many cooperating waves, >= FA_KERNEL_EXP_FIRST =
// an experiment kernel of a variant library.
// `forced` = the handle's fa_config.step_kernel (FA_KERNEL_*; tests pin every instantiation with it).
static int step_variant(int G, int A, int E, int nsteps, bool reset_only, int forced, bool choice = false) {
    const int epw = FA_WAVE / (G + A);
    const int grid = (E + epw - 1) / epw;
    const bool sized = (G == 3 && A == 3) || (G == 5 && A == 5);
    if (reset_only || !sized) return 1;
    // (the pipelined kernel draws resets ahead of time: it does not interleave the ensemble path's choice)
    switch (forced) {
    case FA_KERNEL_PIPE: if (nsteps >= 2 && !choice) return 0; break;   // (its prologue assumes a second step may follow)
    case FA_KERNEL_PIPE3: if (nsteps >= 2 && !choice) return -3; break;
    case FA_KERNEL_WAVES1: return 1;
    case FA_KERNEL_WAVES2: return 2;
    case FA_KERNEL_WAVES3: return 3;
    case FA_KERNEL_EXP_PAIRS: if (G == 3 && A == 3 && fa_launch_step_experiment) return forced; break;   // lane = (agent, partner)
    case FA_KERNEL_EXP_CHAIN: if (nsteps >= 2 && !choice && fa_launch_step_experiment) return forced; break; // one barrier per step
    default: break;
    }
    if (!choice && nsteps >= FA_PIPE_MIN_STEPS && grid <= FA_PIPE_MAX_GRID) return grid <= 2 * 256 ? 0 : -3;
    return grid <= FA_THREE_WAVE_MAX_GRID ? 3 : (grid <= FA_TWO_WAVE_MAX_GRID ? 2 : 1);
}
const char *fa_step_variant_name(int G, int A, int E, int nsteps, int forced, bool choice) {
    switch (step_variant(G, A, E, nsteps, false, forced, choice)) {
    case FA_KERNEL_EXP_PAIRS: return "fa_step_pair_kernel";
    case FA_KERNEL_EXP_CHAIN: return "fa_step_chain_kernel";
    case 0: return "fa_step_pipe_kernel";
    case -3: return "fa_step_pipe_kernel/3 per CU";
    case 3: return "fa_step_kernel/3 waves";
    case 2: return "fa_step_kernel/2 waves";
    default: return "fa_step_kernel/1 wave";
    }
}

template <bool RESET_ONLY, bool COLLECT>
static hipError_t launch_step_t(const FaStepArgs &a, hipStream_t st) {
    const int N = a.G + a.A;
    const int epw = FA_WAVE / N;
    const int grid = (a.E + epw - 1) / epw;
    // Latency regime (few workgroups per CU): cooperating waves per workgroup.  Rollout launches of
    // compile-time team sizes that fit the GPU in one round of 3 workgroups per CU use the
    // pipelined kernel; short launches (its prologue draws two resets ahead and evaluates three
    // sin/cos) and everything else use fa_step_kernel with 3 / 2 / 1 waves by grid size.
    const int nw = step_variant(a.G, a.A, a.E, a.nsteps, RESET_ONLY, a.step_kernel, a.choice_k > 0);
#define FA_LAUNCH(TG_, TA_, NW_)                                                                                      \
    do {                                                                                                              \
        if (a.choice_k > 0)                                                                                           \
            hipLaunchKernelGGL((fa_step_kernel<TG_, TA_, RESET_ONLY, COLLECT, RESET_ONLY ? 1 : NW_, true>), dim3(grid), \
                               dim3((RESET_ONLY ? 1 : NW_) * FA_WAVE), 0, st, a);                                      \
        else                                                                                                          \
            hipLaunchKernelGGL((fa_step_kernel<TG_, TA_, RESET_ONLY, COLLECT, RESET_ONLY ? 1 : NW_, false>), dim3(grid), \
                               dim3((RESET_ONLY ? 1 : NW_) * FA_WAVE), 0, st, a);                                      \
    } while (0)
    if (nw >= FA_KERNEL_EXP_FIRST) return RESET_ONLY ? hipErrorInvalidValue : fa_launch_step_experiment(a, nw, COLLECT, st);
    // up to two workgroups per CU the pipelined build may use 256 VGPRs; three per CU need <= 168
    if (nw == 0 || nw == -3) return RESET_ONLY ? hipErrorInvalidValue : fa_launch_step_pipe(a, COLLECT, nw == -3, st);
    if (a.G == 3 && a.A == 3) {
        if (nw == 3) FA_LAUNCH(3, 3, 3); else if (nw == 2) FA_LAUNCH(3, 3, 2); else FA_LAUNCH(3, 3, 1);
    } else if (a.G == 5 && a.A == 5) {
        if (nw == 3) FA_LAUNCH(5, 5, 3); else if (nw == 2) FA_LAUNCH(5, 5, 2); else FA_LAUNCH(5, 5, 1);
    } else {
        FA_LAUNCH(0, 0, 1);
    }
#undef FA_LAUNCH
    return hipGetLastError();
}

hipError_t fa_launch_step(const FaStepArgs &a, hipStream_t st) {
    const bool collect = a.obs32 && a.rew32 && a.mask32 && a.done && !a.obs64 && !a.rew64 && !a.hit && !a.was_hit;
    return collect ? launch_step_t<false, true>(a, st) : launch_step_t<false, false>(a, st);
}
hipError_t fa_launch_reset(const FaStepArgs &a, hipStream_t st) { return launch_step_t<true, false>(a, st); }
hipError_t fa_launch_seed(const FaState &s, int E, uint64_t base_seed, int64_t env_offset,
                          int skip_words, hipStream_t st) {
    hipLaunchKernelGGL(fa_seed_kernel, dim3((E + 255) / 256), dim3(256), 0, st, s, E, base_seed,
                       env_offset, skip_words);
    return hipGetLastError();
}

// ---- device self-test of div_rn / sqrt_rn against the compiler's `/` and sqrt() ----------------
__global__ void fa_selftest_kernel(unsigned long long n_per_thread, unsigned long long seed,
                                   unsigned long long *mismatch) {
    unsigned long long x = seed ^ (0x9E3779B97F4A7C15ull * (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x + 1));
    auto next = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    auto mag = [&](int lo, int hi) {   // log-uniform magnitude 10^lo .. 10^hi, random mantissa
        const double u = (next() >> 11) * (1.0 / 9007199254740992.0);
        const double v = (next() >> 11) * (1.0 / 9007199254740992.0);
        return exp10(lo + (hi - lo) * u) * (1.0 + v);
    };
    unsigned long long bad_div = 0, bad_sqrt = 0;
    double max_ulp = 0.0;
    for (unsigned long long k = 0; k < n_per_thread; ++k) {
        const double a = ((next() & 1ull) ? -1.0 : 1.0) * mag(-17, 11), b = mag(-17, 11);
        const double q1 = a / b, q2 = div_rn(a, b);
        bad_div += __double_as_longlong(q1) != __double_as_longlong(q2);
        const double bk = (k & 1ull) ? 1e-10 : b;   // the divisor the step uses most
        bad_div += __double_as_longlong(a / bk) != __double_as_longlong(div_rn(a, bk));
        const double s1 = sqrt(b), s2 = sqrt_rn(b);
        bad_sqrt += __double_as_longlong(s1) != __double_as_longlong(s2);
        // headings: 1.5 pi + multiples of the two rotation steps, up to ~1e3 rad
        const double ang = 4.71238898038469 + (double)(next() % 120) * 0.17 + (double)(next() % 120) * 6.113185307179586;
        double ls, lc, fs, fc;
        sincos(ang, &ls, &lc);
        sincos_heading(ang, fs, fc);
        const double us = fabs(fs - ls) / (fabs(ls) * 2.220446049250313e-16 + 1e-300);
        const double uc = fabs(fc - lc) / (fabs(lc) * 2.220446049250313e-16 + 1e-300);
        const double u = us > uc ? us : uc;
        max_ulp = u > max_ulp ? u : max_ulp;
    }
    atomicAdd(&mismatch[0], bad_div);
    atomicAdd(&mismatch[1], bad_sqrt);
    atomicMax(&mismatch[2], (unsigned long long)(max_ulp * 1000.0));   // milli-ulp vs the device libm
}

hipError_t fa_launch_selftest(unsigned long long n_per_thread, unsigned long long seed, unsigned long long *mismatch,
                              hipStream_t st) {
    hipLaunchKernelGGL(fa_selftest_kernel, dim3(1024), dim3(256), 0, st, n_per_thread, seed, mismatch);
    return hipGetLastError();
}
