// fa_step.hip -- the FortAttack env.step / reset kernels for CDNA4 (gfx950).
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
#include "fa_device.h"
#include "fa_probe.h" // FA_TICK* / FA_PROBE_*: no-ops in the product build (tools/make_timing_build.py)

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
// observation row (fortattack_env_v1.py:238): [alive, px, py, ang, vx, vy]
__device__ __forceinline__ void fa_store_obs(float *o32, double *o64, bool alive, double px, double py, double ang, double vx,
                                             double vy) {
    const double al = alive ? 1.0 : 0.0;
    if (o32) {
        float2 *o = reinterpret_cast<float2 *>(o32);
        o[0] = make_float2((float)al, (float)px);
        o[1] = make_float2((float)py, (float)ang);
        o[2] = make_float2((float)vx, (float)vy);
    }
    if (o64) {
        double2 *o = reinterpret_cast<double2 *>(o64);
        o[0] = make_double2(al, px);
        o[1] = make_double2(py, ang);
        o[2] = make_double2(vx, vy);
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
        // a step finished: buffer bo holds its by-products (called once per step, in order)
        auto emit_flags = [&](int bo) {
            const unsigned long long m1 = s_mask[bo][1];
            const bool alive1 = (m1 >> lane) & 1ull;
            const bool done = (s_mask[bo][4] >> lane) & 1ull;
            const int n_alive_att = __popcll(((m1 >> gbase) & grp_mask) >> G);
            // dist_door < fort_dim decided on the square, as wave 0 does (FaDerived::fort2_max)
            const unsigned long long in_fort_b = fa_ballot(is_att && alive1 && s_dd[bo][lane] <= c.fort2_max);
            const bool any_in_fort = ((in_fort_b >> gbase) & grp_mask) != 0ull;
            // ---- fortattack.py:202-225 _get_done bookkeeping --------------------------------------
            if (i == 0) {
                if (done) {
                    const int which = any_in_fort ? 2 : (n_alive_att == 0 ? 0 : 1);
                    uint8_t *gr = a.s.game_result + (size_t)e * 3;
                    gr[0] = which == 0; gr[1] = which == 1; gr[2] = which == 2;
                    atomicAdd(a.s.result_count + (size_t)e * 3 + which, 1u);
                }
                if (COLLECT || a.done) *p_done = done ? 1 : 0;
            }
            // trainer mask (train_fortattack.py:53,87): alive BEFORE the step; an env that is
            // reset here gets the post-reset mask 1 (initialize_new_episode, rlagent.py:31)
            const float mk = (alive0_prev || (done && a.auto_reset != 0)) ? 1.0f : 0.0f;
            if (COLLECT || a.mask32) *p_mask = mk;
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
        if (rew_wave) {
            prev = a.s.prev[idx];
            if (a.track_counters) ep_rew = a.s.ep_rew[idx];
            // complete the loads here: first used inside the loop, they would put a vmcnt(0) --
            // which on gfx9 also drains every store in flight -- into each iteration
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(prev), "+v"(ep_rew));
        }
        int act_prev = 0;
        bool alive0_prev = false;
        // the output rows are walked with per-lane pointers and the reward constants sit in VGPRs:
        // base pointers, strides and fp64 literals as SGPRs overflow the scalar file (see wave 0)
        float *p_rew = a.rew32 ? a.rew32 + idx : nullptr;
        long long row = (long long)idx; // row of the step being emitted in the optional (E, N) outputs
        double k_fort = c.fort_dim, k_03 = 0.3, k_10 = 10.0, k_3 = 3.0, k_01 = 0.1;
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
                if (mine && alive0 && ((grp_alive0 >> j) & 1ull) && !(d2 > c.contact_skip_d2)) {
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
            if (a.track_counters) a.s.ep_rew[idx] = ep_rew;
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
#pragma unroll
    for (int k = 0; k < FA_ACT_BATCH; ++k) av[k] = (k < ns) ? (int)act_ptr[(int64_t)k * a.as_t] : 0;
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
        av[k] = (FA_ACT_BATCH + k < ns) ? (int)act_ptr[(int64_t)(FA_ACT_BATCH + k) * a.as_t] : 0;
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
                av[k] = (s + 1 + FA_ACT_BATCH + k < ns) ? (int)act_ptr[(int64_t)(s + 1 + FA_ACT_BATCH + k) * a.as_t] : 0;
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
// which step kernel a launch of `nsteps` env-steps uses: 0 = pipelined (two workgroups per CU build),
// -3 = pipelined (three per CU build), 1/2/3 = fa_step_kernel with that many cooperating waves.
// `forced` = the handle's fa_config.step_kernel (FA_KERNEL_*; tests pin every instantiation with it).
static int step_variant(int G, int A, int E, int nsteps, bool reset_only, int forced, bool choice = false) {
    const int epw = FA_WAVE / (G + A);
    const int grid = (E + epw - 1) / epw;
    const bool sized = (G == 3 && A == 3) || (G == 5 && A == 5);
    if (reset_only || !sized) return 1;
    // (the pipelined kernel draws resets ahead of time: it does not interleave the ensemble path's choice)
    switch (forced) {
    case 1: if (nsteps >= 2 && !choice) return 0; break;   // FA_KERNEL_PIPE (its prologue assumes a second step may follow)
    case 2: if (nsteps >= 2 && !choice) return -3; break;  // FA_KERNEL_PIPE3
    case 3: return 1;
    case 4: return 2;
    case 5: return 3;
    case 6: if (G == 3 && A == 3) return 6; break;         // FA_KERNEL_PAIRS (experiment): lane = (agent, partner)
    case 7: if (nsteps >= 2 && !choice) return 7; break;   // FA_KERNEL_CHAIN (experiment): one barrier per step
    default: break;
    }
    if (!choice && nsteps >= FA_PIPE_MIN_STEPS && grid <= FA_PIPE_MAX_GRID) return grid <= 2 * 256 ? 0 : -3;
    return grid <= FA_THREE_WAVE_MAX_GRID ? 3 : (grid <= FA_TWO_WAVE_MAX_GRID ? 2 : 1);
}
const char *fa_step_variant_name(int G, int A, int E, int nsteps, int forced, bool choice) {
    switch (step_variant(G, A, E, nsteps, false, forced, choice)) {
    case 6: return "fa_step_pair_kernel";
    case 7: return "fa_step_chain_kernel";
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
#define FA_LAUNCH_PIPE(TG_, TA_, NPW_, MINW_) \
    hipLaunchKernelGGL((fa_step_pipe_kernel<TG_, TA_, COLLECT, NPW_, MINW_>), dim3(grid), dim3((NPW_ + 2) * FA_WAVE), 0, st, a)
    if (nw == 7) {
        if constexpr (!RESET_ONLY) {
            if (a.G == 3) hipLaunchKernelGGL((fa_step_chain_kernel<3, 3, COLLECT>), dim3(grid), dim3(4 * FA_WAVE), 0, st, a);
            else hipLaunchKernelGGL((fa_step_chain_kernel<5, 5, COLLECT>), dim3(grid), dim3(4 * FA_WAVE), 0, st, a);
        }
    } else if (nw == 6) {
        if constexpr (!RESET_ONLY) {
            const int g2 = (a.E + 1) / 2; // two envs per wave
            if (a.choice_k > 0) hipLaunchKernelGGL((fa_step_pair_kernel<3, 3, COLLECT, true>), dim3(g2), dim3(FA_WAVE), 0, st, a);
            else hipLaunchKernelGGL((fa_step_pair_kernel<3, 3, COLLECT, false>), dim3(g2), dim3(FA_WAVE), 0, st, a);
        }
    } else if (a.G == 3 && a.A == 3) {
        // up to two workgroups per CU the build may use 256 VGPRs; three per CU need <= 168
        if (nw == 0) FA_LAUNCH_PIPE(3, 3, 2, 2);
        else if (nw == -3) FA_LAUNCH_PIPE(3, 3, 2, 3);
        else if (nw == 3) FA_LAUNCH(3, 3, 3); else if (nw == 2) FA_LAUNCH(3, 3, 2); else FA_LAUNCH(3, 3, 1);
    } else if (a.G == 5 && a.A == 5) {
        if (nw == 0) FA_LAUNCH_PIPE(5, 5, 2, 2);
        else if (nw == -3) FA_LAUNCH_PIPE(5, 5, 2, 3);
        else if (nw == 3) FA_LAUNCH(5, 5, 3); else if (nw == 2) FA_LAUNCH(5, 5, 2); else FA_LAUNCH(5, 5, 1);
    } else {
        FA_LAUNCH(0, 0, 1);
    }
#undef FA_LAUNCH
#undef FA_LAUNCH_PIPE
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
