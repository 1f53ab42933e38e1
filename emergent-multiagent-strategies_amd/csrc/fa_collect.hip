// fa_collect.hip -- PPO rollout-collector kernels (gfx950): GAE scan, advantage
// statistics, advantage normalisation.  All tensors are the caller's joint
// RolloutStorage buffers, (T[+1], E, N) float32, column c = e*N + i contiguous.
//
// These are HBM-streaming kernels: GAE reads r, V, m (12 B) and writes ret (4 B) per
// (t, e, i); the statistics read ret, V (8 B).  One lane per column, consecutive lanes
// on consecutive columns => every row access is a coalesced span.
#include "fa_device.h"

// Learner.wrap_horizon (learner.py:191-211) + RolloutStorage.compute_returns
// (storage.py:59-66) as one backward pass with per-env episode boundaries.
// `done[t*E + e]` != 0  <=>  the episode of env e ended at rollout step t, i.e. t+1 is one
// of that env's end_pts; index end_pt < T is never visited by the reference (the next
// segment starts at end_pt + 1, learner.py:211): its `returns` entry keeps its old value
// and the accumulator restarts at 0 for the segment below it.
// float32 throughout, operation order of storage.py:63-66; gamma and gamma*tau are
// rounded to float32 once (torch scalar * tensor).
#define FA_GAE_CHUNK 16
struct FaGaeChunk { float r[FA_GAE_CHUNK], v[FA_GAE_CHUNK], m[FA_GAE_CHUNK]; uint8_t d[FA_GAE_CHUNK]; };
__global__ __launch_bounds__(64) void fa_gae_kernel(const float *__restrict__ rewards,
                                                    const float *__restrict__ value_preds,
                                                    const float *__restrict__ masks,
                                                    float *__restrict__ returns,
                                                    const uint8_t *__restrict__ done, int T, int E,
                                                    int N, float g32, float gt32) {
    const long long EN = (long long)E * N;
    const long long col = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= EN) return;
    const int e = (int)(col / N);
    float gae = 0.0f;
    float v_next = value_preds[(long long)T * EN + col];
    float m_next = masks[(long long)T * EN + col];
    // The scan is a short dependent chain per step; the loads do not depend on it.  With one wave per
    // SIMD (24 576 columns = 384 waves) the kernel is a sequence of load round trips, so the chunks are
    // software-pipelined: chunk c+1's 4 x 16 loads are in flight while chunk c is scanned out of
    // registers.  Loads are unconditional and clamped (a short-circuit on the done flag becomes a
    // branch + vmcnt(0) per step and serialises the whole chunk).
    auto load = [&](FaGaeChunk &c, int t0) {
#pragma unroll
        for (int k = 0; k < FA_GAE_CHUNK; ++k) {
            const int t = t0 - k;
            const long long o = (long long)(t >= 0 ? t : 0) * EN + col;
            c.r[k] = rewards[o];
            c.v[k] = value_preds[o];
            c.m[k] = masks[o];
            c.d[k] = done[(long long)(t > 0 ? t - 1 : 0) * E + e];
        }
    };
    auto scan = [&](const FaGaeChunk &c, int t0) {
#pragma unroll
        for (int k = 0; k < FA_GAE_CHUNK; ++k) {
            const int t = t0 - k;
            if (t >= 0) {
                const float delta = c.r[k] + g32 * v_next * m_next - c.v[k];
                const float g = delta + gt32 * m_next * gae;
                if ((t > 0) & (c.d[k] != 0)) {
                    gae = 0.0f;
                } else {
                    gae = g;
                    returns[(long long)t * EN + col] = g + c.v[k];
                }
                v_next = c.v[k];
                m_next = c.m[k];
            }
        }
    };
    FaGaeChunk ca, cb;
    load(ca, T - 1);
    for (int t0 = T - 1; t0 >= 0; t0 -= 2 * FA_GAE_CHUNK) {
        load(cb, t0 - FA_GAE_CHUNK);
        scan(ca, t0);
        load(ca, t0 - 2 * FA_GAE_CHUNK);
        scan(cb, t0 - FA_GAE_CHUNK);
    }
}

// Cooperative form for small batches: at E*N = 24 576 columns the one-wave kernel above has 384
// waves, every wave can keep 63 vector-memory operations in flight (vmcnt), and Little's law pins
// it at ~3 TB/s.  Here a workgroup of four waves shares 64 columns: waves 1..3 stream rewards /
// value_preds / masks + done flags a 32-step chunk ahead into LDS (each with its own queue), wave 0
// scans the previous chunk out of LDS and stores the returns.  Same arithmetic, same order.
// (Two chunks in flight per loader: 3x slower through __syncthreads() in round 2 -- its fence drains
// vmcnt -- and still 18.8 us against 16.0 with raw barriers and statically named register sets in
// round 4 -- in that form the masks + done wave has 128 requests outstanding, twice what a wave can have in
// flight, and every barrier waits for it.)
#define FA_GAEC_CHUNK 32
__global__ __launch_bounds__(256) void fa_gae_coop_kernel(const float *__restrict__ rewards,
                                                          const float *__restrict__ value_preds,
                                                          const float *__restrict__ masks,
                                                          float *__restrict__ returns,
                                                          const uint8_t *__restrict__ done, int T, int E,
                                                          int N, float g32, float gt32) {
    __shared__ float s_r[2][FA_GAEC_CHUNK][64], s_v[2][FA_GAEC_CHUNK][64], s_m[2][FA_GAEC_CHUNK][64];
    __shared__ uint8_t s_d[2][FA_GAEC_CHUNK][64];
    const long long EN = (long long)E * N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long colv = (long long)blockIdx.x * 64 + lane;
    const bool valid = colv < EN;
    const long long col = valid ? colv : EN - 1; // idle lanes shadow the last column, store nothing
    const int e = (int)(col / N);
    const int nc = (T + FA_GAEC_CHUNK - 1) / FA_GAEC_CHUNK;
    // chunk c holds steps t = T-1 - c*CHUNK - k, k = 0..CHUNK-1 (clamped loads below t = 0)
    auto load_chunk = [&](int c) {
        const int t0 = T - 1 - c * FA_GAEC_CHUNK, b = c & 1;
        if (wave == 3) {
            float x[FA_GAEC_CHUNK];
            uint8_t d[FA_GAEC_CHUNK];
#pragma unroll
            for (int k = 0; k < FA_GAEC_CHUNK; ++k) {
                const int t = t0 - k;
                x[k] = masks[(long long)(t >= 0 ? t : 0) * EN + col];
                d[k] = done[(long long)(t > 0 ? t - 1 : 0) * E + e];
            }
#pragma unroll
            for (int k = 0; k < FA_GAEC_CHUNK; ++k) { s_m[b][k][lane] = x[k]; s_d[b][k][lane] = d[k]; }
        } else {
            const float *src = wave == 1 ? rewards : value_preds;
            float x[FA_GAEC_CHUNK];
#pragma unroll
            for (int k = 0; k < FA_GAEC_CHUNK; ++k) {
                const int t = t0 - k;
                x[k] = src[(long long)(t >= 0 ? t : 0) * EN + col];
            }
            float(*dst)[64] = wave == 1 ? s_r[b] : s_v[b];
#pragma unroll
            for (int k = 0; k < FA_GAEC_CHUNK; ++k) dst[k][lane] = x[k];
        }
    };
    float gae = 0.0f, v_next = 0.0f, m_next = 0.0f;
    if (wave == 0) {
        v_next = value_preds[(long long)T * EN + col];
        m_next = masks[(long long)T * EN + col];
    } else {
        load_chunk(0);
    }
    __syncthreads();
    for (int c = 0; c < nc; ++c) {
        if (wave != 0) {
            if (c + 1 < nc) load_chunk(c + 1);
        } else {
            const int t0 = T - 1 - c * FA_GAEC_CHUNK, b = c & 1;
            // the chunk comes out of LDS in one batch (a read per scan step would put an LDS round
            // trip on every step of the chain)
            float r[FA_GAEC_CHUNK], v[FA_GAEC_CHUNK], m[FA_GAEC_CHUNK];
            uint8_t d[FA_GAEC_CHUNK];
#pragma unroll
            for (int k = 0; k < FA_GAEC_CHUNK; ++k) {
                r[k] = s_r[b][k][lane]; v[k] = s_v[b][k][lane]; m[k] = s_m[b][k][lane]; d[k] = s_d[b][k][lane];
            }
#pragma unroll
            for (int k = 0; k < FA_GAEC_CHUNK; ++k) {
                const int t = t0 - k;
                if (t >= 0) {
                    const float delta = r[k] + g32 * v_next * m_next - v[k];
                    const float g = delta + gt32 * m_next * gae;
                    if ((t > 0) & (d[k] != 0)) {
                        gae = 0.0f;
                    } else {
                        gae = g;
                        if (valid) returns[(long long)t * EN + col] = g + v[k];
                    }
                    v_next = v[k];
                    m_next = m[k];
                }
            }
        }
        __syncthreads();
    }
}


// ---- the collector tail in two launches: GAE + advantage moments, fold + normalisation --------------
// fa_gae_mom_kernel = the cooperative scan whose scanning wave also leaves the workgroup's one-pass
// advantage moments (ppo.py:121-123): S = sum(A - P), Q = sum((A - P)^2) per agent in fp64, with
// A = returns[t] - value_preds[t] taken in float32 from the value the scan is about to store -- the
// returns / value_preds re-read of fa_adv_onepass_vec_kernel and its launch disappear.  The entries at
// the episode ends are left stale by the scan (storage.py:59-66 never visits them) but belong to the
// statistics: their OLD returns (and value_preds) are gathered sparsely by the scanning wave behind the
// chunk's scan, FA_GAEC_GATHER per lane and trip -- the wave is ahead of the loaders, the round trip hides
// behind the next barrier.  (A dense prefetch of the old returns costs the 12.6 MB it reads: 21 us against
// 15.4 for the plain scan; round 4's first attempt, a fifth wave + the sums inside the scan loop, 29 us;
// the done flags requested by the scanning wave itself instead of a loader: 19.5 us -- a wave that stores
// should not also wait for loads, the counter is one.)  The workgroup barrier is a raw s_barrier behind an
// LDS wait, not __syncthreads(): the scanning wave's stores and gathers stay in flight across it.  Rows are
// addressed through buffer descriptors (scalar row offset + the lane's 32-bit byte offset).
// Pivot P_i: agent i's first GAE term of env 0, from the inputs alone, so that every workgroup and the
// fold agree on it without a grid-wide exchange (any sample-sized value does: it only keeps S*S/n from
// cancelling against Q).
#define FA_GAEC_GATHER 4
// the workgroup barrier between chunks: LDS traffic complete, then s_barrier.  NOT __syncthreads(): its fence also
// waits for every global load and store in flight (vmcnt(0)) -- the loaders' next chunks and the scanning wave's
// stores -- which is what pinned the first form of this kernel to one chunk in flight and a store drain per chunk.
#define FA_GAEC_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
__device__ __forceinline__ float fa_gae_pivot(const float *__restrict__ rewards, const float *__restrict__ value_preds,
                                              const float *__restrict__ masks, int T, long long EN, int i, float g32) {
    const long long o = (long long)(T - 1) * EN + i, o1 = (long long)T * EN + i;
    const float vp = value_preds[o];
    const float delta = rewards[o] + g32 * value_preds[o1] * masks[o1] - vp;
    return (delta + vp) - vp;
}

__global__ __launch_bounds__(256, 2) void fa_gae_mom_kernel(const float *__restrict__ rewards,
                                                             const float *__restrict__ value_preds,
                                                             const float *__restrict__ masks, float *__restrict__ returns,
                                                             const uint8_t *__restrict__ done, int T, int E, int N, float g32,
                                                             float gt32, double *__restrict__ partial) {
    __shared__ float s_x[3][2][FA_GAEC_CHUNK][64]; // [rewards, value_preds, masks][chunk parity][step][column]
    __shared__ uint8_t s_d[2][FA_GAEC_CHUNK][64];
    __shared__ double s_sq[2][64];
    const long long EN = (long long)E * N;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // wave: in a scalar register
    const long long colv = (long long)blockIdx.x * 64 + lane;
    const bool valid = colv < EN;
    const long long col = valid ? colv : EN - 1; // idle lanes shadow the last column, store nothing
    const int e = (int)(col / N);
    const int nc = (T + FA_GAEC_CHUNK - 1) / FA_GAEC_CHUNK;
    // chunk c holds steps t = T-1 - c*CHUNK - k, k = 0..CHUNK-1 (clamped loads below t = 0)
    // Rows are addressed through buffer descriptors: a wave-uniform row offset in a scalar register + the lane's
    // 32-bit byte offset.  (As 64-bit vector addresses the loads a loader keeps in flight cost it two registers each.)
    // Offsets fit 32 bits: (T + 1) * EN * 4 < 2^31 on this path.
    const unsigned rowb = (unsigned)EN * 4u;                                    // bytes per step of a (T, E, N) tensor
    const unsigned coff = (unsigned)col * 4u;
    const float *src = wave == 1 ? rewards : (wave == 2 ? value_preds : masks);
    const __amdgpu_buffer_rsrc_t rs_src =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, (int)((unsigned)(T + 1) * rowb), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_ret = __builtin_amdgcn_make_buffer_rsrc(returns, 0, (int)((unsigned)(T + 1) * rowb), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_val =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(value_preds), 0, (int)((unsigned)(T + 1) * rowb), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_done =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(done), 0, (int)((unsigned)T * (unsigned)E), 0x00020000);
    auto ldf = [&](const __amdgpu_buffer_rsrc_t &rs, unsigned voff, unsigned soff) -> float {
        // (readfirstlane: the row offset IS uniform, but computed with a vector clamp it gets a waterfall loop per load)
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, __builtin_amdgcn_readfirstlane((int)soff), 0));
    };
    // a loader's chunk: 32 floats of its tensor; wave 3 also the 32 done flags of the lane's env
    struct Chunk { float x[FA_GAEC_CHUNK]; uint8_t d[FA_GAEC_CHUNK]; };
    auto issue = [&](Chunk &q, int c) {
        const int t0 = T - 1 - c * FA_GAEC_CHUNK;
#pragma unroll
        for (int k = 0; k < FA_GAEC_CHUNK; ++k) {
            const int t = t0 - k;
            q.x[k] = ldf(rs_src, coff, (unsigned)(t >= 0 ? t : 0) * rowb);
        }
        if (wave == 3) {
#pragma unroll
            for (int k = 0; k < FA_GAEC_CHUNK; ++k) {
                const int t = t0 - k;
                q.d[k] = __builtin_amdgcn_raw_buffer_load_b8(rs_done, e, __builtin_amdgcn_readfirstlane((int)((unsigned)(t > 0 ? t - 1 : 0) * (unsigned)E)), 0);
            }
        }
    };
    auto stash = [&](const Chunk &q, int c) {
        float(*dst)[64] = s_x[wave - 1][c & 1];
#pragma unroll
        for (int k = 0; k < FA_GAEC_CHUNK; ++k) dst[k][lane] = q.x[k];
        if (wave == 3) {
#pragma unroll
            for (int k = 0; k < FA_GAEC_CHUNK; ++k) s_d[c & 1][k][lane] = q.d[k];
        }
    };
    if (wave != 0) {
        // ---- loaders (the two roles run their own copies of the chunk loop, each with its own barrier instructions:
        // the counts match)
        Chunk q;
        issue(q, 0);
        stash(q, 0);
        FA_GAEC_BARRIER(); // chunk 0 is in LDS
        for (int c = 0; c < nc; ++c) {
            if (c + 1 < nc) {
                issue(q, c + 1);
                stash(q, c + 1);
            }
            FA_GAEC_BARRIER();
        }
    } else {
        // ---- the scanning wave
        float gae = 0.0f;
        float v_next = value_preds[(long long)T * EN + col];
        float m_next = masks[(long long)T * EN + col];
        double accS[2] = {0.0, 0.0}, accQ[2] = {0.0, 0.0};
        const double piv = (double)fa_gae_pivot(rewards, value_preds, masks, T, EN, (int)(col % N), g32);
        auto add_stale = [&](float o, float v) {
            const double dd = (double)(o - v) - piv;
            accS[0] += dd;
            accQ[0] = __fma_rn(dd, dd, accQ[0]);
        };
        float go[FA_GAEC_GATHER], gv[FA_GAEC_GATHER]; // the previous chunk's requested stale entries: old returns, value_preds
        unsigned gmask = 0u;                            // ... which of them are real
#pragma unroll
        for (int j = 0; j < FA_GAEC_GATHER; ++j) { go[j] = 0.0f; gv[j] = 0.0f; }
        FA_GAEC_BARRIER(); // chunk 0 is in LDS
        auto scan_chunk = [&](int c) {
            const int t0 = T - 1 - c * FA_GAEC_CHUNK, b = c & 1;
#pragma unroll
            for (int j = 0; j < FA_GAEC_GATHER; ++j)   // requested behind the previous chunk's scan: a barrier's time to arrive
                if ((gmask >> j) & 1u) add_stale(go[j], gv[j]);
            // the chunk comes out of LDS in one batch (a read per scan step would put an LDS round
            // trip on every step of the chain)
            float r[FA_GAEC_CHUNK], v[FA_GAEC_CHUNK], m[FA_GAEC_CHUNK];
            uint8_t d[FA_GAEC_CHUNK];
#pragma unroll
            for (int k = 0; k < FA_GAEC_CHUNK; ++k) {
                r[k] = s_x[0][b][k][lane]; v[k] = s_x[1][b][k][lane]; m[k] = s_x[2][b][k][lane]; d[k] = s_d[b][k][lane];
            }
            unsigned stale = 0u;
#pragma unroll
            for (int k = 0; k < FA_GAEC_CHUNK; ++k) {
                const int t = t0 - k;
                if (t >= 0) {
                    const float delta = r[k] + g32 * v_next * m_next - v[k];
                    const float g = delta + gt32 * m_next * gae;
                    if ((t > 0) & (d[k] != 0)) {
                        gae = 0.0f;
                        stale |= 1u << k;
                    } else {
                        gae = g;
                        const float ret = g + v[k];
                        if (valid) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ret), rs_ret, (int)coff, __builtin_amdgcn_readfirstlane((int)((unsigned)t * rowb)), 0);
                        const double dd = (double)(ret - v[k]) - piv;
                        accS[k & 1] += dd;
                        accQ[k & 1] = __fma_rn(dd, dd, accQ[k & 1]);
                    }
                    v_next = v[k];
                    m_next = m[k];
                }
            }
            {
                // the chunk's stale entries (a few per cent of all): old returns and value_preds straight from memory.  The
                // first FA_GAEC_GATHER per lane are only REQUESTED here and added at the top of the next chunk -- the raw barrier
                // leaves them in flight, and waiting for them here would hold the whole workgroup's barrier back by a memory
                // round trip per chunk (20.4 us for the kernel instead of 17) -- the rest (rare) in a loop on the spot.
                unsigned rem = stale;
                gmask = 0u;
#pragma unroll
                for (int j = 0; j < FA_GAEC_GATHER; ++j) {
                    const int k = rem ? __builtin_ctz(rem) : 0;
                    gmask |= (rem ? 1u : 0u) << j;
                    rem &= rem - 1u;
                    const unsigned o = (unsigned)(t0 - k) * rowb + coff; // t0 - k >= 1 for a stale k, t0 >= 0 otherwise
                    go[j] = ldf(rs_ret, o, 0u);
                    gv[j] = ldf(rs_val, o, 0u);
                }
                while (__builtin_amdgcn_ballot_w64(rem != 0u) != 0ull) {
                    const bool has = rem != 0u;
                    const int k = has ? __builtin_ctz(rem) : 0;
                    rem &= rem - 1u;
                    const unsigned o = (unsigned)(t0 - k) * rowb + coff;
                    const float o_ = ldf(rs_ret, o, 0u), v_ = ldf(rs_val, o, 0u);
                    if (has) add_stale(o_, v_);
                }
            }
        };
        for (int c = 0; c < nc; ++c) {
            scan_chunk(c);
            FA_GAEC_BARRIER();
        }
#pragma unroll
        for (int j = 0; j < FA_GAEC_GATHER; ++j)   // the last chunk's
            if ((gmask >> j) & 1u) add_stale(go[j], gv[j]);
        s_sq[0][lane] = valid ? accS[0] + accS[1] : 0.0;
        s_sq[1][lane] = valid ? accQ[0] + accQ[1] : 0.0;
    }
    {
        __syncthreads();
        // partial[block][agent] = {S, Q}: the lanes of an agent in ascending order
        if ((int)threadIdx.x < 2 * N) {
            const int i = threadIdx.x >> 1, cmp = threadIdx.x & 1;
            const int first = (int)(((long long)blockIdx.x * 64) % N); // agent of lane 0
            double acc = 0.0;
            for (int l = (i - first + N) % N; l < 64; l += N) acc += s_sq[cmp][l];
            partial[((long long)blockIdx.x * N + i) * 2 + cmp] = acc;
        }
    }
}

// One wave folds agent i's partials: lane l takes workgroups l, l + 64, ... in order (16-byte loads,
// issued together), then a fixed shuffle tree; every lane returns the same (mean, M2).  Used by the
// one-workgroup fold AND by every workgroup of the fused normalisation, so both produce the same bits.
#define FA_GAEM_FOLD 8 // partial workgroups per lane: nblocks <= 64 * FA_GAEM_FOLD
__device__ __forceinline__ void fa_gae_mom_fold(const double *__restrict__ partial, int nblocks, int N, int i, int lane,
                                                double n_rows, double piv, double &mean, double &m2) {
    const double2 *p2 = reinterpret_cast<const double2 *>(partial);
    double2 p[FA_GAEM_FOLD];
#pragma unroll
    for (int k = 0; k < FA_GAEM_FOLD; ++k) {
        const int b = lane + 64 * k;
        p[k] = p2[(long long)(b < nblocks ? b : 0) * N + i];
    }
    double s = 0.0, q = 0.0;
#pragma unroll
    for (int k = 0; k < FA_GAEM_FOLD; ++k) {
        const bool in = lane + 64 * k < nblocks;
        s += in ? p[k].x : 0.0;
        q += in ? p[k].y : 0.0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_down(s, off, 64); q += __shfl_down(q, off, 64); }
    s = __shfl(s, 0, 64);
    q = __shfl(q, 0, 64);
    mean = piv + s / n_rows;
    m2 = q - s * (s / n_rows);
}

__global__ void fa_gae_mom_final_kernel(const double *__restrict__ partial, int nblocks, int N,
                                        const float *__restrict__ rewards, const float *__restrict__ value_preds,
                                        const float *__restrict__ masks, int T, long long EN, float g32, double n_rows,
                                        double *__restrict__ moments_out, double *__restrict__ mean_out,
                                        double *__restrict__ std_out) {
    const int i = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (i >= N) return;
    const double piv = (double)fa_gae_pivot(rewards, value_preds, masks, T, EN, i, g32);
    double mean, m2;
    fa_gae_mom_fold(partial, nblocks, N, i, lane, n_rows, piv, mean, m2);
    if (lane == 0) {
        if (moments_out) { moments_out[i * 3 + 0] = n_rows; moments_out[i * 3 + 1] = mean; moments_out[i * 3 + 2] = m2; }
        if (mean_out) mean_out[i] = mean;
        if (std_out) std_out[i] = sqrt(m2 / (n_rows - 1.0));
    }
}

// The fold and ppo.py:123 in one launch (one rank: nothing is exchanged between them): every workgroup
// folds the partials itself -- one wave per agent, 6 KB per agent out of L2 at config 2 -- and normalises its
// share of the advantages, 16 bytes per lane and trip, the first trip requested ahead of the fold.  (A - (float)mean) / ((float)std + 1e-5f) as fa_adv_norm_kernel; workgroup 0
// also leaves moments / mean / std.  `total` = T*E*N.
__global__ __launch_bounds__(512) void fa_gae_mom_norm_kernel(const double *__restrict__ partial, int nblocks, int N,
                                                              const float *__restrict__ rewards,
                                                              const float *__restrict__ value_preds,
                                                              const float *__restrict__ masks,
                                                              const float *__restrict__ returns, int T, long long EN, float g32,
                                                              double n_rows, long long total, float *__restrict__ out,
                                                              double *__restrict__ moments_out, double *__restrict__ mean_out,
                                                              double *__restrict__ std_out, int piv_from_returns,
                                                              const double *__restrict__ gathered, int W) {
    // gathered != null (several ranks): no fold -- every workgroup merges the W ranks' (n, mean, M2) triples as fa_adv_merge_kernel
    // does (one lane per agent, rank order: same bits everywhere) and normalises with the result.
    // piv_from_returns: the partials are fa_adv_onepass[_vec]_kernel's (shapes beyond the fused scan), whose pivot is agent i's
    // advantage in row 0 AFTER the scan; same {S, Q} layout, same fold as fa_adv_onepass_final_kernel: same bits
    __shared__ float s_mean[FA_MAX_AGENTS_DEV], s_den[FA_MAX_AGENTS_DEV];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long quads = total / 4;
    const bool vec = (((uintptr_t)returns | (uintptr_t)value_preds | (uintptr_t)out) & 15) == 0;
    const float4 *r4 = reinterpret_cast<const float4 *>(returns), *v4 = reinterpret_cast<const float4 *>(value_preds);
    // the lane's first trip is requested before the fold: its round trip and the fold's run side by side
    float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), va = ra, rb = ra, vb = ra;
    if (vec && gid < quads) {
        const long long q1 = gid + stride;
        ra = r4[gid]; va = v4[gid];
        rb = r4[q1 < quads ? q1 : gid]; vb = v4[q1 < quads ? q1 : gid];
    }
    if (gathered) {
        if ((int)threadIdx.x < N) {
            const int i = threadIdx.x;
            double n = 0.0, mean = 0.0, m2 = 0.0;
            for (int r = 0; r < W; ++r) { // Chan-Golub-LeVeque in rank order: fa_adv_merge_kernel's loop
                const double *g = gathered + ((long long)r * N + i) * 3;
                const double nr = g[0], mr = g[1], m2r = g[2];
                if (nr <= 0.0) continue;
                const double nn = n + nr, delta = mr - mean;
                mean += delta * (nr / nn);
                m2 += m2r + delta * delta * (n * nr / nn);
                n = nn;
            }
            const double sd = sqrt(m2 / (n - 1.0));
            s_mean[i] = (float)mean;
            s_den[i] = (float)sd + 1e-5f;
            if (blockIdx.x == 0) {
                if (mean_out) mean_out[i] = mean;
                if (std_out) std_out[i] = sd;
            }
        }
    } else
    for (int i = wave; i < N; i += 8) { // eight waves: one agent each up to 4v4, one round trip
        const double piv = piv_from_returns ? (double)(returns[i] - value_preds[i])
                                            : (double)fa_gae_pivot(rewards, value_preds, masks, T, EN, i, g32);
        double mean, m2;
        fa_gae_mom_fold(partial, nblocks, N, i, lane, n_rows, piv, mean, m2);
        const double sd = sqrt(m2 / (n_rows - 1.0));
        if (lane == 0) {
            s_mean[i] = (float)mean;
            s_den[i] = (float)sd + 1e-5f;
            if (blockIdx.x == 0) {
                if (moments_out) { moments_out[i * 3 + 0] = n_rows; moments_out[i * 3 + 1] = mean; moments_out[i * 3 + 2] = m2; }
                if (mean_out) mean_out[i] = mean;
                if (std_out) std_out[i] = sd;
            }
        }
    }
    __syncthreads();
    if (vec) {
        float4 *o4 = reinterpret_cast<float4 *>(out);
#ifndef FA_NORM_PLAIN_STORES
        // write-through (sc1) 16-byte stores: nothing of `out` stays dirty in L2 for the kernel boundary behind this launch to
        // flush (29.4 against 30.1 us for the two launches, 177.4 against 178.4 behind the rollout; 16-byte sc1 stores cost what
        // plain ones do -- the 8-byte observation rows of the step kernel as sc1 stores: slower, 180.0 against 177.8)
        typedef int v4i __attribute__((ext_vector_type(4)));
        const bool small = total * 4 < (1LL << 31); // buffer offsets are 32 bits
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(out, 0, small ? (int)(total * 4) : 0, 0x00020000);
        auto st4 = [&](long long q, const float4 &v) {
            if (small) {
                v4i w = {__builtin_bit_cast(int, v.x), __builtin_bit_cast(int, v.y), __builtin_bit_cast(int, v.z), __builtin_bit_cast(int, v.w)};
                __builtin_amdgcn_raw_buffer_store_b128(w, rs_out, (int)(q * 16), 0, 16);
            } else {
                o4[q] = v;
            }
        };
#else
        auto st4 = [&](long long q, const float4 &v) { o4[q] = v; };
#endif
        for (long long q0 = gid; q0 < quads; q0 += 2 * stride) {
            const long long q1 = q0 + stride;
            const bool two = q1 < quads;
            if (q0 != gid) {
                ra = r4[q0]; va = v4[q0];
                rb = r4[two ? q1 : q0]; vb = v4[two ? q1 : q0];
            }
            auto norm4 = [&](const float4 &r, const float4 &v, long long q) {
                int i = (int)((4 * q) % N);
                float4 o;
                o.x = (r.x - v.x - s_mean[i]) / s_den[i]; i = i + 1 == N ? 0 : i + 1;
                o.y = (r.y - v.y - s_mean[i]) / s_den[i]; i = i + 1 == N ? 0 : i + 1;
                o.z = (r.z - v.z - s_mean[i]) / s_den[i]; i = i + 1 == N ? 0 : i + 1;
                o.w = (r.w - v.w - s_mean[i]) / s_den[i];
                return o;
            };
            st4(q0, norm4(ra, va, q0));
            if (two) st4(q1, norm4(rb, vb, q1));
        }
        for (long long k = quads * 4 + gid; k < total; k += stride) {
            const int i = (int)(k % N);
            out[k] = (returns[k] - value_preds[k] - s_mean[i]) / s_den[i];
        }
    } else {
        for (long long k = gid; k < total; k += stride) {
            const int i = (int)(k % N);
            out[k] = (returns[k] - value_preds[k] - s_mean[i]) / s_den[i];
        }
    }
}

// Four adjacent columns per lane: 16-byte loads (1 KiB per wave instruction instead of
// 256 B) and four independent scan chains per lane.  Same arithmetic per column as above.
#define FA_GAE4_CHUNK 16
__global__ __launch_bounds__(64) void fa_gae4_kernel(const float *__restrict__ rewards,
                                                     const float *__restrict__ value_preds,
                                                     const float *__restrict__ masks,
                                                     float *__restrict__ returns,
                                                     const uint8_t *__restrict__ done, int T, int E,
                                                     int N, float g32, float gt32) {
    const long long EN = (long long)E * N;
    const long long col = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (col >= EN) return;
    int e[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) e[q] = (int)((col + q) / N);
    float gae[4] = {0.f, 0.f, 0.f, 0.f};
    float4 v_next = *reinterpret_cast<const float4 *>(value_preds + (long long)T * EN + col);
    float4 m_next = *reinterpret_cast<const float4 *>(masks + (long long)T * EN + col);
    for (int t0 = T - 1; t0 >= 0; t0 -= FA_GAE4_CHUNK) {
        float4 r[FA_GAE4_CHUNK], v[FA_GAE4_CHUNK], m[FA_GAE4_CHUNK];
        unsigned skip[FA_GAE4_CHUNK];
#pragma unroll
        for (int k = 0; k < FA_GAE4_CHUNK; ++k) {
            const int t = t0 - k;
            const bool in = t >= 0;
            const long long o = (long long)(in ? t : 0) * EN + col;
            r[k] = *reinterpret_cast<const float4 *>(rewards + o);
            v[k] = *reinterpret_cast<const float4 *>(value_preds + o);
            m[k] = *reinterpret_cast<const float4 *>(masks + o);
            unsigned sk = 0;
            const uint8_t *d = done + (long long)(t > 0 ? t - 1 : 0) * E;
#pragma unroll
            for (int q = 0; q < 4; ++q) sk |= ((unsigned)(t > 0) & (unsigned)(d[e[q]] != 0)) << q;
            skip[k] = sk;
        }
#pragma unroll
        for (int k = 0; k < FA_GAE4_CHUNK; ++k) {
            const int t = t0 - k;
            if (t >= 0) {
                const float rr[4] = {r[k].x, r[k].y, r[k].z, r[k].w};
                const float vv[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
                const float vn[4] = {v_next.x, v_next.y, v_next.z, v_next.w};
                const float mn[4] = {m_next.x, m_next.y, m_next.z, m_next.w};
                float out[4];
                float *ret = returns + (long long)t * EN + col;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float delta = rr[q] + g32 * vn[q] * mn[q] - vv[q];
                    const float g = delta + gt32 * mn[q] * gae[q];
                    const bool sk = (skip[k] >> q) & 1u;
                    gae[q] = sk ? 0.0f : g;
                    out[q] = g + vv[q];
                }
                if (skip[k] == 0u) {
                    *reinterpret_cast<float4 *>(ret) = make_float4(out[0], out[1], out[2], out[3]);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (!((skip[k] >> q) & 1u)) ret[q] = out[q];
                }
                v_next = v[k];
                m_next = m[k];
            }
        }
    }
}

// rlcore/algo/ppo.py:121-123 statistics.  A = returns[t] - value_preds[t] (float32
// subtraction as in the reference), accumulated in fp64.  Stage 1: per-workgroup
// partial sums per agent, fixed order; stage 2: one workgroup folds the partials in a
// fixed order => bitwise reproducible run to run.
// partial layout: [block][N][2] = {sum, sum_sq_dev}
template <int PASS>
__global__ __launch_bounds__(256) void fa_adv_partial_kernel(const float *__restrict__ returns,
                                                             const float *__restrict__ value_preds,
                                                             const double *__restrict__ mean, long long rows,
                                                             int N, double *__restrict__ partial) {
    __shared__ double red[4][FA_MAX_AGENTS_DEV];
    double acc[FA_MAX_AGENTS_DEV];
#pragma unroll
    for (int i = 0; i < FA_MAX_AGENTS_DEV; ++i) acc[i] = 0.0;
    double mu[FA_MAX_AGENTS_DEV];
#pragma unroll
    for (int i = 0; i < FA_MAX_AGENTS_DEV; ++i) mu[i] = (PASS == 1 && i < N) ? mean[i] : 0.0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += stride) {
        const float *ret = returns + r * N, *vp = value_preds + r * N;
#pragma unroll
        for (int i = 0; i < FA_MAX_AGENTS_DEV; ++i) {
            if (i < N) {
                const double adv = (double)(ret[i] - vp[i]);
                if (PASS == 0) acc[i] += adv;
                else { const double d = adv - mu[i]; acc[i] += d * d; }
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < FA_MAX_AGENTS_DEV; ++i) {
        double v = acc[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < N) {
        const int i = threadIdx.x;
        partial[((long long)blockIdx.x * N + i)] = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
    }
}

// Same reduction for a compile-time even N with 16-byte loads: a lane takes two whole rows
// (2N floats = N/2 float4, 48 B at N = 6) per trip, so the agent of every loaded element is a
// compile-time constant and the wave reads one contiguous span.  (The row-per-lane kernel above
// issues N scalar loads with a 4N-byte lane stride and reaches only ~1.5 TB/s.)
template <int PASS, int TN>
__global__ __launch_bounds__(256) void fa_adv_partial_vec_kernel(const float *__restrict__ returns,
                                                                 const float *__restrict__ value_preds,
                                                                 const double *__restrict__ mean, long long row_pairs,
                                                                 double *__restrict__ partial) {
    constexpr int NV = TN / 2; // float4 per row pair
    __shared__ double red[4][TN];
    double acc[TN], mu[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) { acc[i] = 0.0; mu[i] = PASS == 1 ? mean[i] : 0.0; }
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < row_pairs; p += stride) {
        const float4 *r4 = reinterpret_cast<const float4 *>(returns + p * 2 * TN);
        const float4 *v4 = reinterpret_cast<const float4 *>(value_preds + p * 2 * TN);
        float4 rr[NV], vv[NV];
#pragma unroll
        for (int q = 0; q < NV; ++q) { rr[q] = r4[q]; vv[q] = v4[q]; }
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const float a4[4] = {rr[q].x - vv[q].x, rr[q].y - vv[q].y, rr[q].z - vv[q].z, rr[q].w - vv[q].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = (4 * q + j) % TN; // folds to a constant after unrolling
                const double adv = (double)a4[j];
                if (PASS == 0) acc[i] += adv;
                else { const double d = adv - mu[i]; acc[i] += d * d; }
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        double v = acc[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < TN) {
        const int i = threadIdx.x;
        partial[((long long)blockIdx.x * TN + i)] = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
    }
}

// ---- advantage moments in ONE pass ---------------------------------------------------------------
// (n, mean, M2) per agent from a single sweep over returns / value_preds: every lane accumulates
// S = sum(A - P) and Q = sum((A - P)^2) in fp64 around a pivot P_i common to the whole grid -- agent
// i's advantage in row 0, an actual sample, so that S*S/n does not cancel against Q -- the workgroup
// folds its lanes into partial[block][i] = {S, Q}, and a one-workgroup kernel folds the workgroups in
// a fixed order: mean = P + S/n, M2 = Q - S*S/n.  Bitwise reproducible; relative error of M2 ~
// 1e-16 * (1 + ((P - mean)/std)^2) times the accumulation factor.  (Folding in the same launch by
// the last workgroup to finish -- ticket counter + device-scope fences -- measured 10x slower: an
// agent-scope release / acquire is an L2 write-back / invalidate on this multi-XCD part.)
__device__ __forceinline__ void fa_adv_onepass_finish(double (&acc_s)[FA_MAX_AGENTS_DEV], double (&acc_q)[FA_MAX_AGENTS_DEV],
                                                      int N, double *__restrict__ partial) {
    __shared__ double red[4][FA_MAX_AGENTS_DEV][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < FA_MAX_AGENTS_DEV; ++i) {
        if (i < N) {
            double s = acc_s[i], q = acc_q[i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { s += __shfl_down(s, off, 64); q += __shfl_down(q, off, 64); }
            if (lane == 0) { red[wave][i][0] = s; red[wave][i][1] = q; }
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < 2 * N) {
        const int i = threadIdx.x >> 1, c = threadIdx.x & 1;
        partial[((long long)blockIdx.x * N + i) * 2 + c] = ((red[0][i][c] + red[1][i][c]) + red[2][i][c]) + red[3][i][c];
    }
}

// one wave per agent: lane l takes workgroups l, l+64, ... in order (loads issued together), then a
// fixed shuffle tree
#define FA_ADV_FOLD 16 // partial workgroups per lane of the final fold: nblocks <= 64 * FA_ADV_FOLD
__global__ void fa_adv_onepass_final_kernel(const double *__restrict__ partial, int nblocks, int N,
                                            const float *__restrict__ returns, const float *__restrict__ value_preds,
                                            double n_rows, double *__restrict__ moments_out,
                                            double *__restrict__ mean_out, double *__restrict__ std_out) {
    const int i = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (i >= N) return;
    double ps[FA_ADV_FOLD], pq[FA_ADV_FOLD];
#pragma unroll
    for (int k = 0; k < FA_ADV_FOLD; ++k) {
        const int b = lane + 64 * k;
        const bool in = b < nblocks;
        const double *p = partial + ((long long)(in ? b : 0) * N + i) * 2;
        ps[k] = in ? p[0] : 0.0;
        pq[k] = in ? p[1] : 0.0;
    }
    double s = 0.0, q = 0.0;
#pragma unroll
    for (int k = 0; k < FA_ADV_FOLD; ++k) { s += ps[k]; q += pq[k]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_down(s, off, 64); q += __shfl_down(q, off, 64); }
    if (lane == 0) {
        const double piv = (double)(returns[i] - value_preds[i]); // the pivot of the sweep: row 0
        const double mean = piv + s / n_rows, m2 = q - s * (s / n_rows);
        if (moments_out) { moments_out[i * 3 + 0] = n_rows; moments_out[i * 3 + 1] = mean; moments_out[i * 3 + 2] = m2; }
        if (mean_out) mean_out[i] = mean;
        if (std_out) std_out[i] = sqrt(m2 / (n_rows - 1.0));
    }
}

__global__ __launch_bounds__(256) void fa_adv_onepass_kernel(const float *__restrict__ returns,
                                                             const float *__restrict__ value_preds, long long rows, int N,
                                                             double *__restrict__ partial) {
    double acc_s[FA_MAX_AGENTS_DEV], acc_q[FA_MAX_AGENTS_DEV], piv[FA_MAX_AGENTS_DEV];
#pragma unroll
    for (int i = 0; i < FA_MAX_AGENTS_DEV; ++i) {
        acc_s[i] = 0.0; acc_q[i] = 0.0;
        piv[i] = i < N ? (double)(returns[i] - value_preds[i]) : 0.0;
    }
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += stride) {
        const float *ret = returns + r * N, *vp = value_preds + r * N;
#pragma unroll
        for (int i = 0; i < FA_MAX_AGENTS_DEV; ++i) {
            if (i < N) {
                const double d = (double)(ret[i] - vp[i]) - piv[i];
                acc_s[i] += d;
                acc_q[i] = __fma_rn(d, d, acc_q[i]);
            }
        }
    }
    fa_adv_onepass_finish(acc_s, acc_q, N, partial);
}

// compile-time even N with 16-byte loads: two whole rows per lane per trip (see fa_adv_partial_vec_kernel)
template <int TN>
__global__ __launch_bounds__(256) void fa_adv_onepass_vec_kernel(const float *__restrict__ returns,
                                                                 const float *__restrict__ value_preds, long long row_pairs,
                                                                 double *__restrict__ partial) {
    constexpr int NV = TN / 2;
    constexpr int UNR = 4; // row pairs in flight per lane
    double acc_s[FA_MAX_AGENTS_DEV], acc_q[FA_MAX_AGENTS_DEV], piv[FA_MAX_AGENTS_DEV];
#pragma unroll
    for (int i = 0; i < FA_MAX_AGENTS_DEV; ++i) {
        acc_s[i] = 0.0; acc_q[i] = 0.0;
        piv[i] = i < TN ? (double)(returns[i] - value_preds[i]) : 0.0;
    }
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p0 = (long long)blockIdx.x * blockDim.x + threadIdx.x; p0 < row_pairs; p0 += stride * UNR) {
        float4 rr[UNR][NV], vv[UNR][NV];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long long p = p0 + u * stride;
            const long long pc = p < row_pairs ? p : p0; // clamped: loaded, not accumulated
            const float4 *r4 = reinterpret_cast<const float4 *>(returns + pc * 2 * TN);
            const float4 *v4 = reinterpret_cast<const float4 *>(value_preds + pc * 2 * TN);
#pragma unroll
            for (int q = 0; q < NV; ++q) { rr[u][q] = r4[q]; vv[u][q] = v4[q]; }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (p0 + u * stride < row_pairs) {
#pragma unroll
                for (int q = 0; q < NV; ++q) {
                    const float a4[4] = {rr[u][q].x - vv[u][q].x, rr[u][q].y - vv[u][q].y, rr[u][q].z - vv[u][q].z,
                                         rr[u][q].w - vv[u][q].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int i = (4 * q + j) % TN; // folds to a constant after unrolling
                        const double d = (double)a4[j] - piv[i];
                        acc_s[i] += d;
                        acc_q[i] = __fma_rn(d, d, acc_q[i]);
                    }
                }
            }
        }
    }
    fa_adv_onepass_finish(acc_s, acc_q, TN, partial);
}

// stats[i] = {n, sum, 0} (PASS 0) / stats[i][2] = ssd (PASS 1).  One wave per agent: lane l
// folds partials l, l+64, ... in order, then a fixed shuffle tree => reproducible.
template <int PASS>
__global__ void fa_adv_final_kernel(const double *__restrict__ partial, int nblocks, int N, double n_rows,
                                    double *__restrict__ stats, double *__restrict__ derived) {
    const int i = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (i >= N) return;
    double s = 0.0;
    for (int b = lane; b < nblocks; b += 64) s += partial[(long long)b * N + i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) {
        if (PASS == 0) {
            stats[i * 3 + 0] = n_rows; stats[i * 3 + 1] = s; stats[i * 3 + 2] = 0.0;
            if (derived) derived[i] = s / n_rows;                           // local mean
        } else {
            stats[i * 3 + 2] = s;
            if (derived) derived[i] = sqrt(s / (n_rows - 1.0));             // local unbiased std
        }
    }
}

// ppo.py:123: (A - mean) / (std + 1e-5), float32 arithmetic with float32 mean/std.
__global__ __launch_bounds__(256) void fa_adv_norm_kernel(const float *__restrict__ returns,
                                                          const float *__restrict__ value_preds,
                                                          const double *__restrict__ mean,
                                                          const double *__restrict__ std_, long long total,
                                                          int N, float *__restrict__ out) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += stride) {
        const int i = (int)(k % N);
        const float adv = returns[k] - value_preds[k];
        out[k] = (adv - (float)mean[i]) / ((float)std_[i] + 1e-5f);
    }
}

// Chan-Golub-LeVeque merge of per-rank (n, mean, M2) triples in rank order: exact combination
// of two-pass statistics, identical on every rank, one collective instead of two.
// gathered: (W, N, 3); one lane per agent.
__global__ void fa_adv_merge_kernel(const double *__restrict__ gathered, int W, int N,
                                    double *__restrict__ mean_out, double *__restrict__ std_out) {
    const int i = threadIdx.x;
    if (i >= N) return;
    double n = 0.0, mean = 0.0, m2 = 0.0;
    for (int r = 0; r < W; ++r) {
        const double *g = gathered + ((long long)r * N + i) * 3;
        const double nr = g[0], mr = g[1], m2r = g[2];
        if (nr <= 0.0) continue;
        const double nn = n + nr, delta = mr - mean;
        mean += delta * (nr / nn);
        m2 += m2r + delta * delta * (n * nr / nn);
        n = nn;
    }
    mean_out[i] = mean;
    std_out[i] = sqrt(m2 / (n - 1.0));
}

// (n, sum, ssd) -> (n, mean, M2) in place, one lane per agent
__global__ void fa_adv_moments_kernel(double *__restrict__ stats, int N) {
    const int i = threadIdx.x;
    if (i >= N) return;
    stats[i * 3 + 1] = stats[i * 3 + 1] / stats[i * 3 + 0];
}

hipError_t fa_launch_adv_merge(const double *gathered, int W, int N, double *mean_out, double *std_out,
                               hipStream_t st) {
    hipLaunchKernelGGL(fa_adv_merge_kernel, dim3(1), dim3(64), 0, st, gathered, W, N, mean_out, std_out);
    return hipGetLastError();
}
hipError_t fa_launch_adv_moments_fix(double *stats, int N, hipStream_t st) {
    hipLaunchKernelGGL(fa_adv_moments_kernel, dim3(1), dim3(64), 0, st, stats, N);
    return hipGetLastError();
}

hipError_t fa_launch_gae(const float *rewards, const float *value_preds, const float *masks, float *returns,
                         const uint8_t *done, int T, int E, int N, double gamma, double tau, hipStream_t st) {
    const long long EN = (long long)E * N;
    // four columns per lane pay off only once there are enough columns to fill the chip with
    // 16-byte lanes; below that the one-column kernel has 4x the waves in flight
    const bool vec4 = (EN >= 4LL * 64 * 1024) && (EN % 4 == 0) && ((((uintptr_t)rewards | (uintptr_t)value_preds | (uintptr_t)masks |
                                          (uintptr_t)returns) & 15) == 0);
    if (!vec4 && EN <= 64LL * 2048) {
        // few columns: four cooperating waves per 64 columns (see fa_gae_coop_kernel)
        const int grid = (int)((EN + 63) / 64);
        hipLaunchKernelGGL(fa_gae_coop_kernel, dim3(grid), dim3(256), 0, st, rewards, value_preds, masks, returns,
                           done, T, E, N, (float)gamma, (float)(gamma * tau));
        return hipGetLastError();
    }
    if (vec4) {
        const int grid = (int)((EN / 4 + 63) / 64);
        hipLaunchKernelGGL(fa_gae4_kernel, dim3(grid), dim3(64), 0, st, rewards, value_preds, masks, returns,
                           done, T, E, N, (float)gamma, (float)(gamma * tau));
    } else {
        const int grid = (int)((EN + 63) / 64);
        hipLaunchKernelGGL(fa_gae_kernel, dim3(grid), dim3(64), 0, st, rewards, value_preds, masks, returns,
                           done, T, E, N, (float)gamma, (float)(gamma * tau));
    }
    return hipGetLastError();
}

// ---- GAE + moments (fa_gae_mom_kernel) and its two possible second launches ---------------------------
// Eligible up to 512 workgroups = 32 768 (env, agent) columns -- two workgroups per CU.  (Beyond that a CU hosts three and
// the scanning wave, which carries the fp64 sums and the gathers, falls behind its loaders: 41 us against 34 for the
// separate kernels at 5v5 x 4096 and 3v3 x 8192; at 3v3 x 4096 / x 1024: 23.7 against 26.0 / 19.0 against 22.1.)  `partial`
// (partial_cap doubles) holds a {S, Q} pair per (workgroup, agent).  Returns the number of workgroups or 0.
int fa_gae_mom_blocks(const float *rewards, const float *value_preds, const float *masks, const float *returns, int T, int E,
                      int N, long long partial_cap) {
    const long long EN = (long long)E * N;
    if (EN > 64LL * 512 || T < 1 || (long long)(T + 1) * EN * 4 >= (1LL << 31)) return 0;
    const long long nb = (EN + 63) / 64;
    if (nb > 64 * FA_GAEM_FOLD || nb * N * 2 > partial_cap) return 0;
    return (int)nb;
}
hipError_t fa_launch_gae_mom(const float *rewards, const float *value_preds, const float *masks, float *returns,
                             const uint8_t *done, int T, int E, int N, double gamma, double tau, double *partial, int nblocks,
                             hipStream_t st) {
    hipLaunchKernelGGL(fa_gae_mom_kernel, dim3(nblocks), dim3(256), 0, st, rewards, value_preds, masks, returns, done, T, E, N,
                       (float)gamma, (float)(gamma * tau), partial);
    return hipGetLastError();
}
hipError_t fa_launch_gae_mom_final(const double *partial, int nblocks, const float *rewards, const float *value_preds,
                                   const float *masks, int T, int E, int N, double gamma, double *moments_out, double *mean_out,
                                   double *std_out, hipStream_t st) {
    const long long EN = (long long)E * N;
    const double n_rows = (double)T * (double)E;
    hipLaunchKernelGGL(fa_gae_mom_final_kernel, dim3(1), dim3(64 * N), 0, st, partial, nblocks, N, rewards, value_preds, masks, T,
                       EN, (float)gamma, n_rows, moments_out, mean_out, std_out);
    return hipGetLastError();
}
hipError_t fa_launch_gae_mom_norm(const double *partial, int nblocks, const float *rewards, const float *value_preds,
                                  const float *masks, const float *returns, int T, int E, int N, double gamma, float *out,
                                  double *moments_out, double *mean_out, double *std_out, int grid, int piv_from_returns,
                                  hipStream_t st) {
    const long long EN = (long long)E * N, total = (long long)T * EN;
    const double n_rows = (double)T * (double)E;
    if (grid <= 0) {
        // a lane takes two 16-byte quads per trip; two workgroups of eight waves per CU (measured at config 2's 38 MB:
        // 128 / 256 / 512 / 1024 workgroups -> 31.6 / 29.7 / 30.0 / 31.6 us for the two launches, 512 the best behind the rollout)
        long long want = (total / 4 + 1023) / 1024;
        grid = (int)(want < 512 ? (want < 1 ? 1 : want) : 512);
    }
    hipLaunchKernelGGL(fa_gae_mom_norm_kernel, dim3(grid), dim3(512), 0, st, partial, nblocks, N, rewards, value_preds, masks,
                       returns, T, EN, (float)gamma, n_rows, total, out, moments_out, mean_out, std_out, piv_from_returns,
                       (const double *)nullptr, 0);
    return hipGetLastError();
}
// several ranks: fa_adv_merge + fa_adv_normalize as ONE launch (every workgroup merges the gathered triples itself)
hipError_t fa_launch_adv_merge_norm(const double *gathered, int W, int N, const float *returns, const float *value_preds,
                                    long long total, float *out, double *mean_out, double *std_out, hipStream_t st) {
    long long want = (total / 4 + 1023) / 1024;
    const int grid = (int)(want < 512 ? (want < 1 ? 1 : want) : 512);
    hipLaunchKernelGGL(fa_gae_mom_norm_kernel, dim3(grid), dim3(512), 0, st, (const double *)nullptr, 0, N, (const float *)nullptr,
                       value_preds, (const float *)nullptr, returns, 0, 0LL, 0.0f, 0.0, total, out, (double *)nullptr, mean_out,
                       std_out, 0, gathered, W);
    return hipGetLastError();
}

// one-pass moments; `partial` must hold nblocks * N * 2 doubles, nblocks <= 64 * FA_ADV_FOLD
// (final = false: the sweep alone -- its partials then go to fa_launch_gae_mom_norm(..., piv_from_returns = 1); needs nblocks <=
// 64 * FA_GAEM_FOLD)
hipError_t fa_launch_adv_onepass(const float *returns, const float *value_preds, long long rows, int N, double *partial,
                                 int nblocks, double *moments_out, double *mean_out, double *std_out, hipStream_t st,
                                 bool final = true) {
    const bool vec = (rows % 2 == 0) && ((((uintptr_t)returns | (uintptr_t)value_preds) & 15) == 0);
    if (nblocks > 64 * FA_ADV_FOLD) nblocks = 64 * FA_ADV_FOLD;
    if (vec && N == 6)
        hipLaunchKernelGGL((fa_adv_onepass_vec_kernel<6>), dim3(nblocks), dim3(256), 0, st, returns, value_preds, rows / 2,
                           partial);
    else if (vec && N == 10)
        hipLaunchKernelGGL((fa_adv_onepass_vec_kernel<10>), dim3(nblocks), dim3(256), 0, st, returns, value_preds, rows / 2,
                           partial);
    else
        hipLaunchKernelGGL(fa_adv_onepass_kernel, dim3(nblocks), dim3(256), 0, st, returns, value_preds, rows, N, partial);
    if (final)
        hipLaunchKernelGGL(fa_adv_onepass_final_kernel, dim3(1), dim3(64 * N), 0, st, partial, nblocks, N, returns, value_preds,
                           (double)rows, moments_out, mean_out, std_out);
    return hipGetLastError();
}

template <int PASS>
static void launch_adv_partial(const float *returns, const float *value_preds, const double *mean, long long rows,
                               int N, double *partial, int nblocks, hipStream_t st) {
    const bool vec = (rows % 2 == 0) && ((((uintptr_t)returns | (uintptr_t)value_preds) & 15) == 0);
    if (vec && N == 6)
        hipLaunchKernelGGL((fa_adv_partial_vec_kernel<PASS, 6>), dim3(nblocks), dim3(256), 0, st, returns, value_preds,
                           mean, rows / 2, partial);
    else if (vec && N == 10)
        hipLaunchKernelGGL((fa_adv_partial_vec_kernel<PASS, 10>), dim3(nblocks), dim3(256), 0, st, returns, value_preds,
                           mean, rows / 2, partial);
    else
        hipLaunchKernelGGL(fa_adv_partial_kernel<PASS>, dim3(nblocks), dim3(256), 0, st, returns, value_preds, mean,
                           rows, N, partial);
}

hipError_t fa_launch_adv_stats(int pass, const float *returns, const float *value_preds, const double *mean,
                               long long rows, int N, double *partial, int nblocks, double *stats,
                               double *derived, hipStream_t st) {
    if (pass == 0) {
        launch_adv_partial<0>(returns, value_preds, mean, rows, N, partial, nblocks, st);
        hipLaunchKernelGGL(fa_adv_final_kernel<0>, dim3(1), dim3(64 * N), 0, st, partial, nblocks, N,
                           (double)rows, stats, derived);
    } else {
        launch_adv_partial<1>(returns, value_preds, mean, rows, N, partial, nblocks, st);
        hipLaunchKernelGGL(fa_adv_final_kernel<1>, dim3(1), dim3(64 * N), 0, st, partial, nblocks, N,
                           (double)rows, stats, derived);
    }
    return hipGetLastError();
}

hipError_t fa_launch_adv_norm(const float *returns, const float *value_preds, const double *mean,
                              const double *std_, long long total, int N, float *out, hipStream_t st) {
    long long want = (total + 255) / 256;
    const int grid = (int)(want < 2048 ? (want < 1 ? 1 : want) : 2048);
    hipLaunchKernelGGL(fa_adv_norm_kernel, dim3(grid), dim3(256), 0, st, returns, value_preds, mean, std_,
                       total, N, out);
    return hipGetLastError();
}
