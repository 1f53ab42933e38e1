// fa_step_pipe.hip -- fa_step_pipe_kernel: the pipelined four-wave step of the latency regime (rollout launches of compile-time
// team sizes on at most 768 workgroups) -- the dominant kernel of BASELINE config 2.
#include "fa_step_common.h"
#include "fa_probe.h" // FA_TICK* / FA_PROBE_*: no-ops in the product build (tools/make_timing_build.py)

// ---- the pipelined multi-wave step (latency regime, compile-time team sizes) ------------------
// Same arithmetic as fa_step_kernel, cut differently.  At E = 4096 there are fewer waves than
// SIMDs and a rollout is one long dependent chain per wave, so the workgroup spends idle SIMDs
// of its CU on shortening that chain.  Per step only this is left on wave 0:
//     triangles -> laser tests -> (B2) -> ordered force sum -> integrate -> done -> reset -> publish -> (P)
// and everything else runs beside it:
//   * state(s+1) and the by-products of step s are published in LDS at the end of step s
//     (double buffered by step parity; barrier P), so helpers work on step s+1 / finish step s
//     while wave 0 is busy;
//   * pair waves 1..NPW: the soft-contact pair forces are issue-bound fp64 (sqrt, three
//     divisions, exp/log per pair in range).  They are computed once per UNORDERED pair --
//     f(j,i) is bitwise -f(i,j) -- wave h taking the partner offsets d with (d-1) % NPW == h-1
//     (lane i, offset d <-> pair (i, (i+d) mod N); d = 1..N/2, the last one only for i < N/2)
//     and written from both sides into a per-lane partner row (s_fm[j][lane] = force on the
//     lane's agent from partner j), which wave 0 sums in the reference's order after the laser;
//   * the last pair wave also turns the published by-products of the PREVIOUS step into
//     rewards and rollout rows (reward select chain, f64->f32, every global store, the episode
//     statistics): none of that feeds the next state;
//   * the last wave computes the wall forces and sin/cos of the NEXT step's heading: the heading
//     only changes by the action's rotation (core.py:336) or by a reset to a constant, so it
//     does not wait for this step's forces (a dead agent's value is never used; a reset agent
//     takes the constant pair).
// Two workgroup barriers per step: B2 (pair + wall forces of step s are in LDS) and P.
// which pair wave (1..NPW) computes partner offset d: alternating; at N = 6 the half offset (2d == N, half
// the lanes) goes to the LAST pair wave instead -- wave 1 also owns the reset stream and the reward rows and
// with two offsets it was the wave everybody waited for at B2 (156 vs 161 us; at N = 10 the alternating
// split {1,3,5} / {2,4} is the faster one).  Handing the half offset to the walls wave was slower still.
template <int N, int NPW>
__device__ constexpr int fa_pair_wave(int d) { return (N == 6 && 2 * d == N) ? NPW : 1 + (d - 1) % NPW; }
template <int TG, int TA, bool COLLECT, int NPW, int MINW>
__global__ __launch_bounds__((NPW + 2) * FA_WAVE, MINW) void fa_step_pipe_kernel(FaStepArgs a) {
    constexpr int G = TG, A = TA, N = TG + TA;
    constexpr int NOFF = N / 2; // partner offsets that cover every unordered pair once
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
    FA_PROBE_HWID(lane, wave_id)

    // buffer s & 1: state at the start of step s (+ by-products of step s-1)
    // (x, y) pairs live side by side: whoever reads one reads the other, and one 16-byte LDS operation per pair halves the
    // number of LDS instructions queued behind each barrier (the reads behind P take ~330 cycles: four waves at once)
    __shared__ double2 s_pos[2][FA_WAVE], s_vel[2][FA_WAVE];
    __shared__ double s_ang[2][FA_WAVE], s_dd[2][FA_WAVE];
    __shared__ unsigned long long s_mask[2][8]; // ballots: 0 alive, 1 alive after laser, 2 hit, 3 was hit, 4 done
    __shared__ double2 s_trig[2][FA_WAVE]; // [step parity][lane] = (cos, sin): heading of the step's start state
    __shared__ double2 s_W[FA_WAVE];
    __shared__ double2 s_U[FA_WAVE];       // decoded action of the step: accel*u + 0.0 (x, y)
    __shared__ double s_rot[FA_WAVE];      // ... and its rotation
    __shared__ double2 s_fm[N][FA_WAVE];   // [partner j][lane]: pair force on the lane's agent
    __shared__ int s_act[FA_ACT_BATCH][FA_WAVE];
    __shared__ double2 s_rp[FA_WAVE]; // position of the lane's next reset (drawn ahead by wave 1)

    if (wave_id == NPW + 1) {
        // ---- last wave: walls of step s, sin/cos of the heading of step s+1, and the done / mask rows +
        // end-of-episode bookkeeping of the previous step ----------------------------------------------
        uint8_t *p_done = a.done ? a.done + e : nullptr;
        float *p_mask = a.mask32 ? a.mask32 + idx : nullptr;
        asm volatile("" : "+v"(p_done), "+v"(p_mask));
        bool alive0_prev = false;
        // in a VGPR pair: as an SGPR pair it is one of a spilled 16-register tuple of the kernel arguments, and the allocator
        // reloads the whole tuple (16 v_readlane per step) to use these two
        double k_fort2 = c.fort2_max;
        asm volatile("" : "+v"(k_fort2));
        // a step finished: buffer bo holds its by-products (called once per step, in order)
        auto emit_flags = [&](int bo) {
            const unsigned long long m1 = s_mask[bo][1];
            const bool alive1 = (m1 >> lane) & 1ull;
            const bool done = (s_mask[bo][4] >> lane) & 1ull;
            const int n_alive_att = __popcll(((m1 >> gbase) & grp_mask) >> G);
            // dist_door < fort_dim decided on the square, as wave 0 does (FaDerived::fort2_max)
            const unsigned long long in_fort_b = fa_ballot(is_att && alive1 && s_dd[bo][lane] <= k_fort2);
            const bool any_in_fort = ((in_fort_b >> gbase) & grp_mask) != 0ull;
            // ---- fortattack.py:202-225 _get_done bookkeeping --------------------------------------
            if (i == 0) {
                if (done) {
                    const int which = any_in_fort ? 2 : (n_alive_att == 0 ? 0 : 1);
                    uint8_t *gr = a.s.game_result + (size_t)e * 3;
                    gr[0] = which == 0; gr[1] = which == 1; gr[2] = which == 2;
                    atomicAdd(a.s.result_count + (size_t)e * 3 + which, 1u);
                }
                if (COLLECT || a.done) fa_gstore<uint8_t>(p_done, done ? 1 : 0);
            }
            // trainer mask (train_fortattack.py:53,87): alive BEFORE the step; an env that is
            // reset here gets the post-reset mask 1 (initialize_new_episode, rlagent.py:31)
            const float mk = (alive0_prev || (done && a.auto_reset != 0)) ? 1.0f : 0.0f;
            if (COLLECT || a.mask32) fa_gstore(p_mask, mk);
            p_mask += EN; p_done += a.E;
        };
        FA_TICK_INIT
        FA_WG_BARRIER(); // P(-1)
        for (int s = 0; s < ns; ++s) {
            const int b = s & 1;
            FA_TICK(16)
            const int act = s_act[s & (FA_ACT_BATCH - 1)][lane];
            const double ang = s_ang[b][lane];
            const bool alive0 = (s_mask[b][0] >> lane) & 1ull;
            const double2 pos_ = s_pos[b][lane];
            double px = pos_.x, py = pos_.y;
            asm volatile("" : "+v"(px), "+v"(py)); // read with the rest: one LDS round trip, not two
            // fortattack.py:253-263,:289 _set_action, for wave 0 (F starts as u + 0.0, core.py:221-228)
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
            wx = alive0 ? wx : 0.0;
            wy = alive0 ? wy : 0.0;
            s_W[lane] = make_double2(wx, wy);
            FA_TICK(17)
            FA_WG_BARRIER(); // B2(s)
            FA_TICK(18)
            if (s + 1 < ns) {
                double sn, cs;
                sincos_heading(ang + rot, sn, cs); // == wave 0's `ang += rot` for a survivor
                s_trig[(s + 1) & 1][lane] = make_double2(cs, sn);
            }
            if (s > 0) emit_flags(b);
            alive0_prev = alive0;
            FA_TICK(19)
            FA_WG_BARRIER(); // P(s)
        }
        FA_WG_BARRIER(); // (wave 0 publishes the last step's by-products)
        emit_flags(ns & 1);
        FA_TICK_FLUSH(16, 20, 30)
        return;
    }
    if (wave_id >= 1) {
        // ---- pair waves: soft contact (core.py:231-243, :440-456), once per unordered pair; the
        // last of them also emits the rewards and rollout rows of the previous step ---------------
        // the rows of a finished step are emitted by two waves: rewards / masks / done and the
        // episode bookkeeping by wave 1, the observation rows by the last pair wave
        const bool rew_wave = wave_id == 1, out_wave = wave_id == NPW;
        double prev = 0.0, ep_rew = 0.0;
        // the finished-episode statistics stay in registers for the launch: as read-modify-writes of global memory inside the
        // loop they put two load round trips (s_waitcnt vmcnt(0) each, ~1 500 cycles) into every step in which an env of the
        // wave ends its episode -- on the wave that arrives last at P.  Same additions in the same order: same bits.
        double ep_sum = 0.0;
        unsigned alive_end = 0u;
        if (rew_wave) {
            prev = a.s.prev[idx];
            if (a.track_counters) { ep_rew = a.s.ep_rew[idx]; ep_sum = a.s.ep_rew_sum[idx]; alive_end = a.s.alive_end[idx]; }
            // complete the loads here: first used inside the loop, they would put a vmcnt(0) --
            // which on gfx9 also drains every store in flight -- into each iteration
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(prev), "+v"(ep_rew), "+v"(ep_sum), "+v"(alive_end));
        }
        int act_prev = 0;
        bool alive0_prev = false;
        // the output rows are walked with per-lane pointers and the reward constants sit in VGPRs:
        // base pointers, strides and fp64 literals as SGPRs overflow the scalar file (see wave 0)
        float *p_rew = a.rew32 ? a.rew32 + idx : nullptr;
        long long row = (long long)idx; // row of the step being emitted in the optional (E, N) outputs
        double k_fort = c.fort_dim, k_03 = 0.3, k_10 = 10.0, k_3 = 3.0, k_01 = 0.1;
        double k_skip = c.contact_skip_d2; // (as an SGPR pair: spilled, reloaded with two v_readlane per partner offset)
        asm volatile("" : "+v"(k_skip));
        asm volatile("" : "+v"(p_rew), "+v"(row));
        asm volatile("" : "+v"(k_fort), "+v"(k_03), "+v"(k_10), "+v"(k_3), "+v"(k_01));
        // a step finished: buffer bo holds the state after it and its by-products
        // (called once per step, in order)
        auto emit_rew = [&](int bo) {
            const unsigned long long m1 = s_mask[bo][1];
            const bool alive1 = (m1 >> lane) & 1ull;
            const bool hit = (s_mask[bo][2] >> lane) & 1ull;
            const bool was_hit = (s_mask[bo][3] >> lane) & 1ull;
            const bool done = (s_mask[bo][4] >> lane) & 1ull;
            const double dist_door = sqrt_rn(s_dd[bo][lane]);
            const bool alive0 = alive0_prev;
            const bool shoot = act_prev == 7;
            const int n_alive_att = __popcll(((m1 >> gbase) & grp_mask) >> G);
            const unsigned long long in_fort_b = fa_ballot(is_att && alive1 && dist_door < k_fort);
            const bool any_in_fort = ((in_fort_b >> gbase) & grp_mask) != 0ull;
            // ---- rewards (fortattack_env_v1.py:87-188), after World.step ----------------------
            // (the done / mask rows and the end-of-episode bookkeeping of the step: the last wave's emit_flags)
            const bool just_died = alive0 && was_hit;
            const bool rewarded = (alive1 || just_died);
            const double rew = fa_reward(is_att, rewarded, prev, dist_door, shoot, hit, was_hit, n_alive_att, any_in_fort,
                                         k_fort, k_03, k_10, k_3, k_01);
            prev = rewarded ? dist_door : prev;
            // evaluation statistics (test_fortattack_v2.py:88-101)
            if (a.track_counters) {
                ep_rew += alive0 ? rew : 0.0;
                if (done) {
                    ep_sum += ep_rew;
                    if (alive1) alive_end += 1u;
                    ep_rew = 0.0;
                }
            }
            if (COLLECT) {
                fa_gstore(p_rew, (float)rew);
            } else {
                if (a.rew32) fa_gstore(p_rew, (float)rew);
                if (a.rew64) a.rew64[row] = rew;
                if (a.hit) a.hit[row] = hit ? 1 : 0;
                if (a.was_hit) a.was_hit[row] = was_hit ? 1 : 0;
            }
            p_rew += EN; row += (long long)EN;
        };
        float *p_obs = a.obs32 ? a.obs32 + idx * 6 : nullptr;
        long long row6 = (long long)idx * 6;
        asm volatile("" : "+v"(p_obs), "+v"(row6));
        auto emit_obs = [&](int bo) {
            // observation row (fortattack_env_v1.py:238): the state after the step / reset
            const bool alive_new = (s_mask[bo][0] >> lane) & 1ull;
            const double2 pos_ = s_pos[bo][lane], vel_ = s_vel[bo][lane];
            const double px = pos_.x, py = pos_.y, ang = s_ang[bo][lane];
            const double vx = vel_.x, vy = vel_.y;
            fa_store_obs((COLLECT || a.obs32) ? p_obs : nullptr, (!COLLECT && a.obs64) ? a.obs64 + row6 : nullptr, alive_new,
                         px, py, ang, vx, vy);
            p_obs += EN * 6; row6 += (long long)EN * 6;
        };
        // wave 1 owns the env's reset stream during the launch (see ResetDraw)
        const bool rng_wave = wave_id == 1;
        ResetDraw rdA = {}, rdB = {};
        MtWords mw = {};
        bool need_b = false;
        auto wait_words = [&]() { if (a.rng_mode == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
        if (rng_wave) {
            // one load round trip for the cursor, one for the words of all three draws
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
        FA_TICK_INIT
        FA_WG_BARRIER(); // P(-1)
        for (int s = 0; s < ns; ++s) {
            const int b = s & 1;
            FA_TICK(10)
            if (rng_wave && s > 0 && a.auto_reset != 0) {
                // envs that were reset at the end of step s-1 used draw A: commit it, promote B (in
                // LDS before B2(s), i.e. before wave 0 can need it); the new B is drawn after B2
                need_b = (s_mask[b][4] >> lane) & 1ull;
                if (need_b) {
                    draw_commit(a, e, i, N, rdA);
                    rdA = rdB;
                    s_rp[lane] = make_double2(rdA.px, rdA.py);
                }
            }
            const unsigned long long grp_alive0 = (s_mask[b][0] >> gbase) & grp_mask;
            const bool alive0 = (grp_alive0 >> i) & 1ull;
            const double2 pos_ = s_pos[b][lane];
            const double px = pos_.x, py = pos_.y;
            const int act_cur = s_act[s & (FA_ACT_BATCH - 1)][lane];
            // the partners' positions of all this wave's offsets in one LDS round trip (small teams:
            // at N = 10 the extra live registers push the 168-VGPR build into scratch)
            constexpr bool HOISTQ = NOFF <= 3;
            double qx[NOFF], qy[NOFF];
            if constexpr (HOISTQ) {
#pragma unroll
                for (int d = 1; d <= NOFF; ++d) {
                    if (fa_pair_wave<N, NPW>(d) != wave_id) continue; // uniform per wave
                    int j = i + d;
                    j = j >= N ? j - N : j;
                    const double2 q_ = s_pos[b][gbase + j];
                    qx[d - 1] = q_.x;
                    qy[d - 1] = q_.y;
                }
            }
#pragma unroll
            for (int d = 1; d <= NOFF; ++d) {
                if (fa_pair_wave<N, NPW>(d) != wave_id) continue; // uniform per wave
                int j = i + d;
                j = j >= N ? j - N : j;
                const bool mine = (2 * d != N) || (i < N / 2); // the half offset: one side only
                // candidate against a partner alive BEFORE the laser (one the laser kills this step
                // is masked out in the sum); exactly +0.0 when out of range, so adding it is a no-op
                if constexpr (!HOISTQ) {
                    const double2 q_ = s_pos[b][gbase + j];
                    qx[d - 1] = q_.x;
                    qy[d - 1] = q_.y;
                }
                const double dx = px - qx[d - 1], dy = py - qy[d - 1];
                const double d2 = dx * dx + dy * dy;
                double fxv = 0.0, fyv = 0.0;
                bool near = false;
                if (mine && alive0 && ((grp_alive0 >> j) & 1ull) && !(d2 > k_skip)) {
                    fa_contact_force(c, dx, dy, d2, fxv, fyv);
                    near = true;
                }
                if (mine) {
                    s_fm[j][lane] = make_double2(fxv, fyv);                                       // on agent i from partner j
                    s_fm[i][gbase + j] = make_double2(near ? -fxv : 0.0, near ? -fyv : 0.0);     // on agent j from partner i: the exact negative
                }
            }
            FA_TICK(11)
            FA_WG_BARRIER(); // B2(s)
            FA_TICK(12)
            if (rng_wave && need_b) {
                wait_words();
                rdB.base = draw_next_base(a, rdA.base, i, N);
                draw_eval(a, e, i, is_att, mw, rdB);
                draw_load(a, e, draw_next_base(a, rdB.base, i, N), mw);
            }
            need_b = false;
            if (rew_wave && s > 0) emit_rew(b);
            if (out_wave && s > 0) emit_obs(b);
            act_prev = act_cur;
            alive0_prev = alive0;
            FA_TICK(13)
            FA_WG_BARRIER(); // P(s)
        }
        FA_WG_BARRIER(); // wave 0 has published the last step's by-products
        if (rng_wave && a.auto_reset != 0 && ((s_mask[ns & 1][4] >> lane) & 1ull)) draw_commit(a, e, i, N, rdA);
        if (rew_wave) {
            emit_rew(ns & 1);
            a.s.prev[idx] = prev;
            if (a.track_counters) { a.s.ep_rew[idx] = ep_rew; a.s.ep_rew_sum[idx] = ep_sum; a.s.alive_end[idx] = alive_end; }
        }
        if (out_wave) emit_obs(ns & 1);
        if (FA_TICK_WAVE1 ? rew_wave : out_wave) { FA_TICK_FLUSH(10, 14, 29) }
        return;
    }

    // ---- wave 0 ----------------------------------------------------------------------------------
    // the chain of the launch: where two workgroups share a CU it out-prioritises the helper wave of the
    // other workgroup that sits on its SIMD
    __builtin_amdgcn_s_setprio(3);
    FA_PROBE_WAVE0_BEGIN
    double px = a.s.px[idx], py = a.s.py[idx], vx = a.s.vx[idx], vy = a.s.vy[idx];
    double ang = a.s.ang[idx];
    unsigned long long alive_m = FA_M_NE_U(a.s.alive[idx], 0); // wave mask of the living (see FA_M_*)
    int t = a.s.tstep[e], nh = 0, nwh = 0;
    if (a.track_counters) { nh = a.s.num_hit[idx]; nwh = a.s.num_was_hit[idx]; }
    unsigned long long dirty_m = 0ull;
    const unsigned long long is_att_m = FA_M_NE_U(is_att ? 1u : 0u, 0), lane0_m = FA_M_EQ_U(lane, 0);
    const int64_t *act_ptr = a.actions + (int64_t)e * a.as_e + (int64_t)i * a.as_i;
    int av[FA_ACT_BATCH];
    // rows past the launch's last step are never consumed: the index is clamped instead of the load being skipped (a branch per
    // row -- 32 blocks in the prologue, each reloading a spilled 16-SGPR tuple of the kernel arguments for the stride)
    int64_t as_t = a.as_t;
    asm volatile("" : "+s"(as_t));
    const int last_row = ns - 1;
#pragma unroll
    for (int k = 0; k < FA_ACT_BATCH; ++k) av[k] = (int)act_ptr[(int64_t)(k < last_row ? k : last_row) * as_t];
    // ---- the laser test (fa_wedge): a target lane needs position and heading sin/cos of its opponents.
    // Positions go through LDS inside this wave; sin/cos of the next heading comes from the last wave
    // (constants after a reset).
    constexpr int KT = TG > TA ? TG : TA;
    const int n_opp = is_att ? G : A, opp0 = is_att ? 0 : G;
    const int team_idx = is_att ? i - G : i;
    const unsigned opp_bits = is_att ? ((1u << G) - 1u) : (((1u << A) - 1u) << G); // the opponents in the group word
    constexpr unsigned grp_bits = (1u << N) - 1u;
    double sn, cs, sn_g = 0.0, cs_g = 0.0, sn_a = 0.0, cs_a = 0.0;
    sincos_heading(ang, sn, cs);
    if (ns > 1) { // headings after a reset (fortattack_env_v1.py:59)
        sincos_heading(c.ang_guard, sn_g, cs_g);
        sincos_heading(c.ang_attacker, sn_a, cs_a);
    }
    const double cs_ro = is_att ? cs_g : cs_a, sn_ro = is_att ? sn_g : sn_a;   // the opponents
    double oqx[KT], oqy[KT], ocs[KT], osn[KT]; // the opponents' position and heading at the step's start
    s_trig[0][lane] = make_double2(cs, sn);
    int act = av[0];
#pragma unroll
    for (int k = 0; k < FA_ACT_BATCH; ++k) s_act[k][lane] = av[k];
#pragma unroll
    for (int k = 0; k < FA_ACT_BATCH; ++k)
        av[k] = (int)act_ptr[(int64_t)(FA_ACT_BATCH + k < last_row ? FA_ACT_BATCH + k : last_row) * as_t];
    s_pos[0][lane] = make_double2(px, py);
    s_ang[0][lane] = ang;
    if (fa_lanes(lane0_m)) s_mask[0][0] = alive_m;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        const int j = gbase + opp0 + (k < n_opp ? k : 0);
        const double2 q_ = s_pos[0][j], tg_ = s_trig[0][j];
        oqx[k] = q_.x; oqy[k] = q_.y;
        ocs[k] = tg_.x; osn[k] = tg_.y;
    }
    unsigned long long reset_prev_m = 0ull;
    s_fm[i][lane] = make_double2(0.0, 0.0); // an agent exerts no force on itself: the pair waves never write the diagonal
    FA_WG_BARRIER(); // P(-1)
    FA_PROBE_WAVE0_LOOP_BEGIN(lane)

    // The loop's fp64 constants live in VGPRs: as SGPR pairs they (with the lane masks and the
    // write-back pointers) overflow the scalar file, and every spilled SGPR costs the lone wave a
    // v_readlane issue slot per use.
    double k_size = c.agent_size, k_far = c.shoot_far, k_chw = c.cos_hw, k_shw = c.sin_hw;
    double k_damp = c.one_minus_damping, k_dt = c.dt, k_sp2 = c.speed2_max, k_vmax = c.max_speed;
    double k_doorx = c.door_x, k_doory = c.door_y, k_fort2 = c.fort2_max;
    double k_ang_r = is_att ? c.ang_attacker : c.ang_guard;
    asm volatile("" : "+v"(k_size), "+v"(k_far), "+v"(k_chw), "+v"(k_shw), "+v"(k_damp), "+v"(k_dt));
    asm volatile("" : "+v"(k_sp2), "+v"(k_vmax), "+v"(k_doorx), "+v"(k_doory), "+v"(k_fort2), "+v"(k_ang_r));
    FA_TICK_INIT
    for (int s = 0; s < ns; ++s) {
        const int nb = (s + 1) & 1;
        const unsigned long long alive0_m = alive_m;
        // (the action is decoded by the last wave, fortattack.py:253-263; only `shoot` is needed here)
        const unsigned long long shooters_m = FA_M_EQ_U(act, 7) & alive0_m;
        if (s > 0) { // sin/cos of the opponents' headings: the last wave's, constants after a reset
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                const int j = gbase + opp0 + (k < n_opp ? k : 0);
                const double2 tg_ = s_trig[s & 1][j];
                ocs[k] = tg_.x;
                osn[k] = tg_.y;
            }
            if (__builtin_expect(reset_prev_m != 0ull, 0)) { // rare blocks out of line: a taken skip costs ~27 cycles
                const bool rp = fa_lanes(reset_prev_m);
#pragma unroll
                for (int k = 0; k < KT; ++k) {
                    ocs[k] = rp ? cs_ro : ocs[k];
                    osn[k] = rp ? sn_ro : osn[k];
                }
            }
        }
        FA_TICK(0)

        // ---- core.py:254-302 apply_laser_effect ------------------------------------------------
        // test k: every lane against its k-th opponent; hb[k] = the lanes hit by shooter k of either
        // team.  "Group word" = a wave mask shifted down to the lane's own env (bit j = agent j).
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
            // a shooter's hit list is the ballot of its team index, restricted to the opponents of its env
            int tix = team_idx;
            asm volatile("" : "+v"(tix)); // compare in the loop: KT hoisted lane masks cost 2 SGPRs each
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
        const unsigned ga1 = (unsigned)(alive1_m >> gbase) & grp_bits;     // survivors of the lane's env
        const int n_alive_att = __popc(ga1 >> G);
        FA_TICK(1)
        FA_WG_BARRIER(); // B2(s): pair and wall forces of this step are in LDS
        FA_TICK(2)

        // ---- core.py:221-252: F = u + 0, the pairs in the reference's order (for agent i:
        // partner j ascending), then the walls -------------------------------------------------
        double fmx[N], fmy[N];
#pragma unroll
        for (int j = 0; j < N; ++j) { const double2 f_ = s_fm[j][lane]; fmx[j] = f_.x; fmy[j] = f_.y; }
        const double2 w_ = s_W[lane], u_ = s_U[lane];
        const double wx = w_.x, wy = w_.y;
        const double u0 = u_.x, u1 = u_.y, rot = s_rot[lane];
        const bool restage = ((s + 1) & (FA_ACT_BATCH - 1)) == 0;
        const int act_lds = s_act[(s + 1) & (FA_ACT_BATCH - 1)][lane];
        int act_next = restage ? av[0] : act_lds;
        if (fa_lanes(alive1_m)) {
            // masked by the survivors with one FMA per term: fma(f, 1, F) == f + F and fma(f, 0, F) == F
            // bit for bit (F is never -0.0; f is finite unless two agents coincide exactly)
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
            // rare block, one test for two cases: the speed limit (sqrt(v.v) > max_speed decided on the
            // square, see fa_step_kernel) and a NaN -- written !(<=) so that the NaN takes it too
            if (__builtin_expect(!(speed2 <= k_sp2), 0)) {
                if (speed2 != speed2) {
                    // Two agents of the env coincide exactly: their pair force is NaN (0/0, as in the
                    // reference).  If the partner was shot in this very step the reference skips the
                    // pair (core.py:233-236 only walks the living) while fma(NaN, 0, F) is NaN: redo
                    // this lane's sum with the dead partners skipped, the rows are still in LDS.
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
        // (pinned here: selected after the barrier P, av[0] would still be live when the next batch
        // is loaded and the loop would carry a copy of a pending load -- a vmcnt(0) every step)
        asm volatile("" : "+v"(act_next));
        FA_TICK(3)
        // ---- what the next state needs of the reward / done logic ------------------------------
        // (`dist_door < fort_dim` decided on the squared distance, see FaDerived::fort2_max; the
        // square root itself is only needed by the rewards and is taken by the output wave)
        const double ddx = px - k_doorx, ddy = py - k_doory;
        const double dd2 = ddx * ddx + ddy * ddy;
        const unsigned long long in_fort_m = FA_M_LE_D(dd2, k_fort2) & is_att_m & alive1_m;
        const unsigned gw_fort = (unsigned)(in_fort_m >> gbase) & grp_bits;
        // fortattack.py:202-225: an attacker in the fort, no attacker left, or the time limit
        const unsigned long long done_m = FA_M_NE_U(gw_fort, 0) | FA_M_EQ_U(n_alive_att, 0) | FA_M_EQ_U(t, a.max_t - 1);
        const unsigned long long reset_m = a.auto_reset != 0 ? done_m : 0ull;
        t += 1;                                                        // fortattack.py:171
        alive_m = alive1_m;
        nh += hit_cnt;
        nwh += was_hit_cnt; // one per shooter that hit (core.py:283)
        dirty_m |= alive0_m;
        FA_TICK(4)
        // ---- fortattack_env_v1.py:47-75 reset_world (prevDist is NOT reset: quirk Q1) ----------
        // (the positions were drawn ahead by wave 1, see ResetDraw)
        if (__builtin_expect(reset_m != 0ull, 0)) { // wave-uniform: most steps reset no env of the wave
            const double2 rp_ = s_rp[lane];
            const double rpx = rp_.x, rpy = rp_.y;
            if (fa_lanes(reset_m)) {
                px = rpx; py = rpy; vx = 0.0; vy = 0.0;
                ang = k_ang_r;
                t = 0;
                nh = 0; nwh = 0;
            }
            alive_m |= reset_m;
            dirty_m |= reset_m;
        }
        reset_prev_m = reset_m;
        FA_TICK(5)
        // ---- publish state(s+1): what the helper waves need to start on step s+1 ----------------
        s_pos[nb][lane] = make_double2(px, py);
        s_ang[nb][lane] = ang;
        if (fa_lanes(lane0_m)) {
            s_mask[nb][0] = alive_m;
            s_mask[nb][4] = done_m;
        }
        if (__builtin_expect(restage, 0)) {
#pragma unroll
            for (int k = 0; k < FA_ACT_BATCH; ++k) s_act[k][lane] = av[k];
        }
        // the by-products of step s are only read by the emitting waves after B2(s+1): they are written
        // behind the barrier, while the helpers already work on step s+1 (those of the last step are
        // followed by one more barrier after the loop)
        auto publish_byproducts = [&]() {
            s_vel[nb][lane] = make_double2(vx, vy);
            s_dd[nb][lane] = dd2;
            if (fa_lanes(lane0_m)) {
                s_mask[nb][1] = alive1_m;
                s_mask[nb][2] = hit_m;
                s_mask[nb][3] = was_hit_m;
            }
        };
        FA_TICK(6)
        FA_WG_BARRIER(); // P(s)
        FA_TICK(7)
        publish_byproducts();
        if (__builtin_expect(restage, 0)) {
#pragma unroll
            for (int k = 0; k < FA_ACT_BATCH; ++k)
                av[k] = (s + 1 + FA_ACT_BATCH + k < ns) ? (int)act_ptr[(int64_t)(s + 1 + FA_ACT_BATCH + k) * as_t] : 0;
        }
        // the opponents' positions for the next step (this wave's own writes: in order)
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            const int j = gbase + opp0 + (k < n_opp ? k : 0);
            const double2 q_ = s_pos[nb][j];
            oqx[k] = q_.x;
            oqy[k] = q_.y;
        }
        act = act_next;
    }
    FA_WG_BARRIER(); // the by-products of the last step are published
    FA_TICK_FLUSH(0, 8, 28)
    FA_PROBE_WAVE0_LOOP_END(lane)

    if (fa_lanes(dirty_m)) {
        a.s.px[idx] = px; a.s.py[idx] = py; a.s.vx[idx] = vx; a.s.vy[idx] = vy;
        a.s.ang[idx] = ang;
        a.s.alive[idx] = fa_lanes(alive_m) ? 1 : 0;
        if (a.track_counters) { a.s.num_hit[idx] = nh; a.s.num_was_hit[idx] = nwh; }
    }
    if (i == 0) a.s.tstep[e] = t;
    FA_PROBE_WAVE0_END(lane)
}


// ---- launcher (the dispatch is fa_step_classic.hip's launch_step_t) ---------------------------
hipError_t fa_launch_step_pipe(const FaStepArgs &a, bool collect, bool three_per_cu, hipStream_t st) {
    const int epw = FA_WAVE / (a.G + a.A);
    const int grid = (a.E + epw - 1) / epw;
#define FA_LAUNCH_PIPE(TG_, TA_, NPW_, MINW_)                                                                                 \
    do {                                                                                                                      \
        if (collect) hipLaunchKernelGGL((fa_step_pipe_kernel<TG_, TA_, true, NPW_, MINW_>), dim3(grid), dim3((NPW_ + 2) * FA_WAVE), 0, st, a); \
        else hipLaunchKernelGGL((fa_step_pipe_kernel<TG_, TA_, false, NPW_, MINW_>), dim3(grid), dim3((NPW_ + 2) * FA_WAVE), 0, st, a);         \
    } while (0)
    if (a.G == 3 && a.A == 3) {
        if (three_per_cu) FA_LAUNCH_PIPE(3, 3, 2, 3); else FA_LAUNCH_PIPE(3, 3, 2, 2);
    } else if (a.G == 5 && a.A == 5) {
        if (three_per_cu) FA_LAUNCH_PIPE(5, 5, 2, 3); else FA_LAUNCH_PIPE(5, 5, 2, 2);
    } else {
        return hipErrorInvalidValue;
    }
#undef FA_LAUNCH_PIPE
    return hipGetLastError();
}
