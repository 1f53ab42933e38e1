// fa_step_common.h -- what the FortAttack step kernels share: the reset stream (numpy's MT19937 in incremental form,
// Philox), correctly rounded fp64 divide / sqrt cores, heading sin / cos, the soft-contact, wall, laser and reward pieces
// of World.step, observation rows, wave-mask helpers.  Included by fa_step_classic.hip (fa_step_kernel), fa_step_pipe.hip
// (fa_step_pipe_kernel) and csrc/experiments/fa_step_experiments.hip (the two round-4 experiment kernels, not part of
// the product library).
//
// One launch advances every env of the handle by `nsteps` env-steps (1 = closed loop, the policy
// runs between launches; K = open-loop rollout with the world state held in registers).
// Mapping: lane = agent, a wave64 = EPW = 64/N whole envs, one workgroup per EPW envs (E=4096,
// N=6 -> 410 workgroups over 256 CUs: the launch is latency bound, so the work is spread thin;
// in that regime a second, cooperating wave per workgroup computes the contact/wall forces --
// see TWO below).  Cross-agent data moves two ways, both inside the wave:
//   * positions and laser triangles are staged in LDS and read back with per-lane
//     addresses (broadcast reads inside an env); actions reach the loop through LDS too;
//   * every flag reduction (who shoots, who is alive, who was hit by whom, attackers in
//     the fort) is a 64-bit wave ballot shifted to the env's lane group + popcount.
// Float semantics: every fp64 operation is written in the order the reference evaluates
// it (file:line cited per block); built with -ffp-contract=off so nothing is fused (the
// explicit fma() calls below are the compiler's own divide / sqrt / polynomial sequences).
// The exact shortcuts (skipping a contact whose soft penalty is exactly 0.0, sqrt-free speed
// test, wrapper-free divide/sqrt) are argued where they are taken.
#pragma once
#include "fa_device.h"

// ---- numpy legacy RandomState (MT19937), incremental form --------------------------
// Matsumoto-Nishimura genrand regenerates all 624 words at once; word k of the new block
// depends only on old[k], old[k+1] and (k+397)%624 (old for k<227, new otherwise), so
// drawing word `pos` = twist it in place, temper, advance.  Identical stream, O(1) work
// per draw, no 624-word stall inside a step.
__device__ __forceinline__ uint32_t mt_twist(uint32_t cur, uint32_t nxt, uint32_t far_) {
    uint32_t y = (cur & 0x80000000u) | (nxt & 0x7fffffffu);
    return far_ ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}
__device__ __forceinline__ int mt_wrap(int k) { return k >= FA_MT_N ? k - FA_MT_N : k; }

// genrand_res53 (numpy mt19937_next_double)
__device__ __forceinline__ double res53(uint32_t w0, uint32_t w1) {
    uint32_t a = w0 >> 5, b = w1 >> 6;
    return (a * 67108864.0 + b) / 9007199254740992.0;
}

// Philox4x32-10 (Salmon et al. 2011), perf-mode reset stream.
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

// fortattack_env_v1.py:47-75 reset_world, for the lane's agent.  Agent i consumes the
// 2 doubles (4 words) number 2i, 2i+1 of this reset.  All lanes of the env call together.
// `base` = env cursor + 4*i, kept in a register across the launch (one dependent load less per
// reset); on return it is the base of the lane's next draw.
__device__ __forceinline__ void reset_agent(const FaStepArgs &a, int e, int i, int N, bool is_att,
                                            bool active, int &base, double &px, double &py) {
    uint32_t w[4] = {0, 0, 0, 0};
    if (active) {
        if (a.rng_mode == 0) {
            uint32_t *mt = a.s.mt + (size_t)e * FA_MT_N; // base < 624 + 64
            uint32_t cur[5], far_[4];
#pragma unroll
            for (int k = 0; k < 5; ++k) cur[k] = mt[mt_wrap(base + k)];
#pragma unroll
            for (int k = 0; k < 4; ++k) far_[k] = mt[mt_wrap(mt_wrap(base + k) + FA_MT_M)];
            // every load above is complete (data dependence) before any lane stores below
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t nw = mt_twist(cur[k], cur[k + 1], far_[k]);
                mt[mt_wrap(base + k)] = nw;
                w[k] = mt_temper(nw);
            }
            base = mt_wrap(base - 4 * i + 4 * N) + 4 * i;
        } else {
            const uint64_t genv = (uint64_t)(a.env_offset + e);
            uint32_t c[4] = {(uint32_t)genv, (uint32_t)(genv >> 32), a.s.reset_count[e], (uint32_t)i};
            philox4x32_10(c, (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
            w[0] = c[0]; w[1] = c[1]; w[2] = c[2]; w[3] = c[3];
        }
    }
    const double u1 = res53(w[0], w[1]), u2 = res53(w[2], w[3]);
    if (is_att) { // :66
        px = a.c.att_x_lo + a.c.att_x_rng * u1;
        py = a.c.att_y_lo + a.c.att_y_rng * u2;
    } else {      // :70
        px = a.c.grd_x_lo + a.c.grd_x_rng * u1;
        py = a.c.grd_y_lo + a.c.grd_y_rng * u2;
    }
}

// after every lane of the env has drawn: advance the env's cursor (one lane per env)
// (lane i == 0: its draw base IS the env's cursor)
__device__ __forceinline__ void reset_advance(const FaStepArgs &a, int e, int next_base) {
    if (a.rng_mode == 0) a.s.mt_pos[e] = next_base;
    else a.s.reset_count[e] += 1u;
}

// The pipelined kernel draws reset positions AHEAD of the reset, on a helper wave: a draw only
// depends on the RNG stream.  Two draws are pending per lane -- A, published in LDS for the next
// reset, and B, which replaces A the moment A is used (an env can be reset in consecutive steps) --
// plus the 9 MT words the draw after B needs, loaded early.  The twisted words of a draw and the
// env's cursor are stored only when a reset really uses the draw, so the state in HBM always is
// the stream position after the resets that took place.  Safe for N <= 18: the words of the two
// following draws (cursor + 4N .. cursor + 12N) and their +397 partners are not written by A.
struct ResetDraw {
    double px, py;
    uint32_t nw[4]; // MT: twisted words of the draw
    int base;       // MT: cursor + 4*i of the draw;  Philox: the env's reset counter for the draw
};
struct MtWords { uint32_t cur[5], far_[4]; };
__device__ __forceinline__ int draw_next_base(const FaStepArgs &a, int base, int i, int N) {
    return a.rng_mode == 0 ? mt_wrap(base - 4 * i + 4 * N) + 4 * i : base + 1;
}
__device__ __forceinline__ void draw_load(const FaStepArgs &a, int e, int base, MtWords &w) {
    if (a.rng_mode != 0) return;
    const uint32_t *mt = a.s.mt + (size_t)e * FA_MT_N;
#pragma unroll
    for (int k = 0; k < 5; ++k) w.cur[k] = mt[mt_wrap(base + k)];
#pragma unroll
    for (int k = 0; k < 4; ++k) w.far_[k] = mt[mt_wrap(mt_wrap(base + k) + FA_MT_M)];
}
// positions of the draw at d.base from the loaded words (MT) or the counter (Philox); reset_world
// fortattack_env_v1.py:47-75 as in reset_agent()
__device__ __forceinline__ void draw_eval(const FaStepArgs &a, int e, int i, bool is_att, const MtWords &mw,
                                          ResetDraw &d) {
    uint32_t w[4];
    if (a.rng_mode == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            d.nw[k] = mt_twist(mw.cur[k], mw.cur[k + 1], mw.far_[k]);
            w[k] = mt_temper(d.nw[k]);
        }
    } else {
        const uint64_t genv = (uint64_t)(a.env_offset + e);
        uint32_t c[4] = {(uint32_t)genv, (uint32_t)(genv >> 32), (uint32_t)d.base, (uint32_t)i};
        philox4x32_10(c, (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
        w[0] = c[0]; w[1] = c[1]; w[2] = c[2]; w[3] = c[3];
    }
    const double u1 = res53(w[0], w[1]), u2 = res53(w[2], w[3]);
    if (is_att) { // :66
        d.px = a.c.att_x_lo + a.c.att_x_rng * u1;
        d.py = a.c.att_y_lo + a.c.att_y_rng * u2;
    } else {      // :70
        d.px = a.c.grd_x_lo + a.c.grd_x_rng * u1;
        d.py = a.c.grd_y_lo + a.c.grd_y_rng * u2;
    }
}
// the draw was used by a reset: make it part of the stream in HBM
__device__ __forceinline__ void draw_commit(const FaStepArgs &a, int e, int i, int N, const ResetDraw &d) {
    const int nb = draw_next_base(a, d.base, i, N);
    if (a.rng_mode == 0) {
        uint32_t *mt = a.s.mt + (size_t)e * FA_MT_N;
#pragma unroll
        for (int k = 0; k < 4; ++k) mt[mt_wrap(d.base + k)] = d.nw[k];
        if (i == 0) a.s.mt_pos[e] = nb;
    } else {
        if (i == 0) a.s.reset_count[e] = (uint32_t)nb;
    }
}

// ---- correctly rounded fp64 divide / sqrt without the range-scaling wrappers -----------------
// hipcc expands a/b into v_div_scale x2 + v_rcp + 2 Newton FMAs + mul + residual FMA +
// v_div_fmas + v_div_fixup (11 instructions, serialised through VCC) and sqrt(x) into a
// scale/ldexp/class wrapper around v_rsq + a 9-FMA Goldschmidt core (17 instructions).  The
// wrappers only matter when a quotient or root leaves the normal range or an input is 0/inf/nan;
// every operand on the step's slow paths is a normal number of magnitude 1e-17 .. 1e11 (wall /
// contact clearances over k = 1e-10, forces over distances, door and pair distances), so the
// cores alone produce the same correctly rounded bits with 8 resp. 10 instructions and no VCC
// dependency (independent divisions can overlap).  fa_selftest_math() checks them bit for bit
// against `/` and sqrt() on the device.
__device__ __forceinline__ double div_rn(double a, double b) {
    double r = __builtin_amdgcn_rcp(b);
    double e = fma(-b, r, 1.0);
    r = fma(r, e, r);
    e = fma(-b, r, 1.0);
    r = fma(r, e, r);
    const double q = a * r;
    const double e2 = fma(-b, q, a);
    return fma(e2, r, q);
}
__device__ __forceinline__ double sqrt_rn(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    double d = fma(-g, g, x);
    g = fma(d, h, g);
    d = fma(-g, g, x);
    return fma(d, h, g);
}

// ---- sin and cos of a heading ---------------------------------------------------------------
// The heading is an unbounded sum of +0.17 / +(2pi - 0.17) steps (quirk Q3): |x| < ~1e3.
// Cody-Waite reduction by pi/2 in three FMA steps (exact to < 1 ulp of the reduced argument for
// |x| < 1e5), then the fdlibm __kernel_sin / __kernel_cos minimax polynomials on [-pi/4, pi/4]
// (< 1 ulp).  ~40 instructions instead of the ~100 of the general-purpose sincos (whose
// Payne-Hanek path for huge arguments is dead weight here).  Like any libm it differs from
// glibc's results in the last ulp now and then; that reaches only the laser triangle's vertices
// (see the note at the call site).  fa_selftest_math() reports the largest deviation from the
// device library.
__device__ __forceinline__ void sincos_heading(double x, double &sn, double &cs) {
    const double kf = rint(x * 0.63661977236758134308);          // x * 2/pi
    double r = fma(-kf, 1.5707963267948966, x);                    // pi/2 = P1 + P2 + P3
    r = fma(-kf, 6.123233995736766e-17, r);
    r = fma(-kf, -1.4973849048591698e-33, r);
    const double z = r * r;
    // __kernel_sin
    const double ps = fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
                                        2.75573137070700676789e-06), -1.98412698298579493134e-04),
                          8.33333333332248946124e-03);
    const double s = fma(z * r, fma(z, ps, -1.66666666666666324348e-01), r);
    // __kernel_cos
    const double pc = z * fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
                                                   -2.75573143513906633035e-07), 2.48015872894767294178e-05),
                                     -1.38888888888741095749e-03), 4.16666666666666019037e-02);
    const double hz = 0.5 * z, w = 1.0 - hz;
    const double c = w + (((1.0 - w) - hz) + z * pc);
    const int q = (int)kf & 3;
    const double s2 = (q & 1) ? c : s, c2 = (q & 1) ? s : c;
    sn = (q & 2) ? -s2 : s2;
    cs = ((q + 1) & 2) ? -c2 : c2;
}

// np.logaddexp(0, t) * k, numpy npy_logaddexp with x = 0 (core.py:452, :469).
//   t >= 40  : t + log1p(exp(-t)) == t exactly (exp(-t) <= 4.3e-18 < ulp(40)/2)
//   t < -746 : exp underflows to +0, log1p(0) = 0
// only the band in between needs libm.
// the libm band, kept out of line: it runs for ~1e-5 of contacts but would otherwise be
// inlined (exp + log1p, twice) at every one of the ~10 call sites
// (Taking exp(t) alone below t = -40 -- log1p of an x < 2^-57 is x itself -- is bit-identical on the device down to
// exp(t) = 2^-1021 and to the reference's libm everywhere (tools/probes/band_probe.hip), and measured SLOWER: 150.9 against
// 147.5 us per 128-step launch; the select keeps both results live across the call.)
__device__ __attribute__((noinline)) double softplus_band(double t) {
    if (t == 0.0) return 0.0 + 0.693147180559945309417232121458176568; // NPY_LOGE2
    if (t < 0.0) return 0.0 + log1p(exp(t));
    return t + log1p(exp(-t));
}
__device__ __forceinline__ double softplus_pen(double t, double k) {
    double v;
    if (t >= 40.0) v = t;
    else if (t < -746.0) v = 0.0;
    else v = softplus_band(t);
    return v * k;
}

// ---- the laser test (core.py:373-390) in the shooter's frame ------------------------------------
// The reference's triangle (get_tri_pts_arr) is the isosceles wedge with its apex at
// q + size*(cos a, sin a), half-angle shootWin/2 about the heading a and its far edge perpendicular to
// the heading at shootRad*cos(shootWin/2); laser_hit asks whether the target's barycentric coordinates
// in it are all >= 0 (an SVD solve in the reference, Cramer's rule in the oracle).  With d = target -
// apex, u = d.(cos a, sin a), v = d x (cos a, sin a):
//     inside  <=>  u <= shootRad*cos(w/2)  and  |v| cos(w/2) <= u sin(w/2)
// -- the same closed triangle, evaluated from (position, cos a, sin a) of the shooter instead of three
// vertices: nothing but sin/cos of the heading to stage, 14 flops per test.  It can differ from the
// vertex form only for a target within rounding (1e-16) of an edge, the class of deviation the heading
// sin/cos already has; 0 differing flags against the goldens and the oracle.
// Returns u and the two sides of the wedge inequality; hit = (u <= c.shoot_far) & (lhs <= rhs).
__device__ __forceinline__ void fa_wedge(double size, double cos_hw, double sin_hw, double px, double py, double qx, double qy,
                                         double cs, double sn, double &u, double &lhs, double &rhs) {
    const double ax = qx + size * cs, ay = qy + size * sn; // == pt1 of core.py:375
    const double dx = px - ax, dy = py - ay;
    u = dx * cs + dy * sn;
    const double v = dy * cs - dx * sn;
    lhs = fabs(v) * cos_hw;
    rhs = u * sin_hw;
}

// ---- pieces of World.step shared by the step kernels -----------------------------------------
// core.py:440-456 get_collision_force for one pair within range (d2 = dx*dx + dy*dy of the pair):
// force on the agent at the +delta end; the partner's is its exact negative.
__device__ __forceinline__ void fa_contact_force(const FaDerived &c, double dx, double dy, double d2, double &fx, double &fy) {
    const double dist = sqrt_rn(d2);
    const double pen = softplus_pen(div_rn(-(dist - c.dist_min), c.contact_margin), c.contact_margin);
    fx = div_rn(c.contact_force * dx, dist) * pen;
    fy = div_rn(c.contact_force * dy, dist) * pen;
}
// core.py:246-252 + :459-472 wall force of a living agent: (fx1 - fx2, fy1 - fy2), exactly +0.0 off the
// walls.  A wall whose clearance is > 1000*margin contributes exactly +0.0 and its division is skipped;
// measured: per-wall branches beat four unconditional ILP divisions (typically only one or two walls are
// touched by some lane of the wave).
__device__ __forceinline__ void fa_wall_force(const FaDerived &c, double px, double py, double &wx, double &wy) {
    wx = 0.0;
    wy = 0.0;
    const double k = c.contact_margin, size = c.agent_size;
    const double d0 = px - size - c.wall_xmin, d1 = c.wall_xmax - px - size;
    const double d2 = py - size - c.wall_ymin, d3 = c.wall_ymax - py - size;
    const bool w0 = !(d0 > c.wall_skip), w1 = !(d1 > c.wall_skip);
    const bool w2 = !(d2 > c.wall_skip), w3 = !(d3 > c.wall_skip);
    if (w0 || w1 || w2 || w3) {
        double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
        if (w0) p0 = softplus_pen(div_rn(-d0, k), k);
        if (w1) p1 = softplus_pen(div_rn(-d1, k), k);
        if (w2) p2 = softplus_pen(div_rn(-d2, k), k);
        if (w3) p3 = softplus_pen(div_rn(-d3, k), k);
        wx = c.contact_force * p0 - c.contact_force * p1;
        wy = c.contact_force * p2 - c.contact_force * p3;
    }
}
// The same wall force without per-wall branches, for the helper wave whose B2 arrival it decides (a
// taken branch costs a lone wave far more than the ~7 instructions it skips): the softplus of every
// wall is selected from its two closed-form ends, t >= 40 -> t and t < -746 -> +0.0 (which covers
// every wall farther than 1000*margin), and only if some lane sits in the band between them does the
// wave take the libm path for those lanes.
__device__ __forceinline__ void fa_wall_force_flat(const FaDerived &c, double px, double py, double &wx, double &wy) {
    const double k = c.contact_margin, size = c.agent_size;
    const double d[4] = {px - size - c.wall_xmin, c.wall_xmax - px - size, py - size - c.wall_ymin, c.wall_ymax - py - size};
    double t[4], v[4];
    bool band = false;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        t[q] = div_rn(-d[q], k);
        v[q] = t[q] >= 40.0 ? t[q] : 0.0;
        band = band | ((t[q] < 40.0) & !(t[q] < -746.0));
    }
    if (__builtin_amdgcn_ballot_w64(band) != 0ull) {
        bool bq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[q] = (t[q] < 40.0) & !(t[q] < -746.0);
#ifndef FA_WALLS_FOUR_CALLS
        // An agent is in the band of at most ONE wall per axis unless the arena is narrower than two band widths
        // (the band is 0.786 wide, the reference's arena 1.9 x 1.5 between the agents' surfaces): the axis' one
        // candidate goes through the libm path -- two calls per step instead of four, the same function on the same
        // operand, so the same bits.  (A wave with a lane between two bands of one axis takes the general path.)
        if (__builtin_amdgcn_ballot_w64((bq[0] & bq[1]) | (bq[2] & bq[3])) == 0ull) {
#pragma unroll
            for (int ax = 0; ax < 2; ++ax) {
                const bool any = bq[2 * ax] | bq[2 * ax + 1];
                if (any) {
                    const double vb = softplus_band(bq[2 * ax] ? t[2 * ax] : t[2 * ax + 1]);
                    if (bq[2 * ax]) v[2 * ax] = vb;
                    else v[2 * ax + 1] = vb;
                }
            }
        } else
#endif
        {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (bq[q]) v[q] = softplus_band(t[q]);
        }
    }
    wx = c.contact_force * (v[0] * k) - c.contact_force * (v[1] * k);
    wy = c.contact_force * (v[2] * k) - c.contact_force * (v[3] * k);
}
// fortattack_env_v1.py:87-188 reward of one agent after World.step.  attacker_reward (:94-128) and
// guard_reward (:130-188) as one select chain: both are a sum of six terms added left to right --
// attacker r0..r5; guard r0, r3..r7 (its r1, r2, r8 are literal zeros and x + 0.0 == x) -- so the
// per-team terms are selected and the additions are shared.  No divergent team branch.
// `prev`: prevDist (NaN == None).  The literals come in as arguments so that a caller can keep them in
// VGPRs (fort_dim, 0.3, 10, 3, 0.1).
__device__ __forceinline__ double fa_reward(bool is_att, bool rewarded, double prev, double dist_door, bool shoot, bool hit,
                                            bool was_hit, int n_alive_att, bool any_in_fort, double k_fort, double k_03,
                                            double k_10, double k_3, double k_01) {
    const bool has_prev = !(prev != prev);
    const double g0 = ((dist_door > k_03) & (prev <= k_03)) ? -1.0 : (((dist_door <= k_03) & (prev > k_03)) ? 1.0 : 0.0);
    const double t0 = has_prev ? (is_att ? 2 * (prev - dist_door) : g0) : 0.0;
    const bool c1 = is_att ? (dist_door < k_fort) : ((n_alive_att != 0) & any_in_fort);
    const double t1 = c1 ? (is_att ? k_10 : -k_10) : 0.0;
    const double t2 = shoot ? (is_att ? -1.0 : -k_01) : 0.0;
    const double t3 = hit ? k_3 : 0.0;
    const double t4 = was_hit ? -k_3 : 0.0;
    const double t5 = (n_alive_att == 0) ? (is_att ? -k_10 : k_10) : 0.0;
    return rewarded ? (t0 + t1 + t2 + t3 + t4 + t5) : 0.0;
}
// Stores to the caller's output rows name the GLOBAL address space.  The row pointers are nullable selects and are pinned in
// VGPRs through empty asm statements (see the pipelined kernel), which hides from the compiler that they point to global
// memory: it then emits FLAT stores, and a flat instruction counts on lgkmcnt as well as vmcnt -- the `s_waitcnt lgkmcnt(0)`
// in front of every workgroup barrier would wait for the row stores of the step.
#define FA_GLOBAL __attribute__((address_space(1)))
template <typename T> __device__ __forceinline__ void fa_gstore(T *p, T v) { *(FA_GLOBAL T *)p = v; }
// observation row (fortattack_env_v1.py:238): [alive, px, py, ang, vx, vy]
__device__ __forceinline__ void fa_store_obs(float *o32, double *o64, bool alive, double px, double py, double ang, double vx,
                                             double vy) {
    typedef float fa_v2f __attribute__((ext_vector_type(2)));
    typedef double fa_v2d __attribute__((ext_vector_type(2)));
    const double al = alive ? 1.0 : 0.0;
    if (o32) {
        FA_GLOBAL fa_v2f *o = (FA_GLOBAL fa_v2f *)o32;
        o[0] = fa_v2f{(float)al, (float)px};
        o[1] = fa_v2f{(float)py, (float)ang};
        o[2] = fa_v2f{(float)vx, (float)vy};
    }
    if (o64) {
        FA_GLOBAL fa_v2d *o = (FA_GLOBAL fa_v2d *)o64;
        o[0] = fa_v2d{al, px};
        o[1] = fa_v2d{py, ang};
        o[2] = fa_v2d{vx, vy};
    }
}

// COLLECT: the four trainer rows (obs32, rew32, mask32, done) are all present and nothing
// else is: their stores are then unconditional, which lets the compiler wait for the
// prefetched action with vmcnt(#stores) instead of draining the store queue every step.
// TWO: two cooperating waves per workgroup (compile-time team sizes only).  Wave 0 is the step
// as described above minus the contact/wall forces; wave 1 (the "force wave", stateless) computes
// them for the same lanes from the positions wave 0 stages in LDS, concurrently with wave 0's
// sin/cos + laser tests, and hands back F through LDS.  At E = 4096 there are fewer waves than
// SIMDs and the step is a latency chain, so running its two longest independent pieces
// (laser ~30 %, forces ~45 % of the chain) side by side on two SIMDs shortens it; the arithmetic
// and its order are unchanged.  (A two-barrier variant in which wave 0 sums the candidates itself
// measured 4 % slower: wave 0 is the critical path, the force wave has slack.)  Three workgroup barriers per step (raw s_barrier behind an LDS
// wait -- __syncthreads() would also drain the global stores).
// the lane mask of a condition as it already sits in an SGPR pair (HIP's __ballot() goes through an
// int: v_cndmask 0/1 + v_cmp_ne, two issue slots per ballot)
__device__ __forceinline__ unsigned long long fa_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// Lane conditions as 64-bit wave masks: a VALU compare delivers its lane mask (zero in inactive lanes)
// in an SGPR pair, mask algebra is scalar, and fa_lanes() hands a mask back as a lane predicate without
// an instruction.  (A `bool` that is ANDed / ORed and then balloted goes through v_cndmask + v_cmp.)
#define FA_M_EQ_U(a, b) __builtin_amdgcn_uicmp((unsigned)(a), (unsigned)(b), 32)  /* ICMP_EQ */
#define FA_M_NE_U(a, b) __builtin_amdgcn_uicmp((unsigned)(a), (unsigned)(b), 33)  /* ICMP_NE */
#define FA_M_LE_D(a, b) __builtin_amdgcn_fcmp((double)(a), (double)(b), 5)        /* FCMP_OLE */
__device__ __forceinline__ bool fa_lanes(unsigned long long m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
#define FA_WG_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
