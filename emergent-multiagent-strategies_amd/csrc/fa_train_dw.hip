// fa_train_dw.hip -- the weight gradients of one team's PPO minibatch as ONE split-K MFMA GEMM over all rows of the
// minibatch, and the fixed-order reductions that finish fa_ppo_grad.
//
// fa_train_kernel (fa_train.hip) differentiates a 32-row tile down to dL/dX of every layer and leaves, per tile, the
// OPERANDS of the weight-gradient products dW = X^T dY in global memory (fa_train.h FA_RECA_* / FA_RECB_*):
//   record A, one per (tile, round):  dW7 (256 x 128) += [h_in | hmix]^T dZ ;  dA_m (128 x 128) += h_in^T dg
//   record B, one per tile:           dW8 (128 x 256) = h3^T [dP | dV] ;  dB_o (64 x 64) = mix_o^T de_opp ;
//                                     dA_o (64 x 64) = h1^T dg_o
// Here 256 workgroups (one per CU, eight waves) each walk a contiguous range of records: a record (64 / 80 KB) is
// staged through registers into one of TWO LDS images while the previous record's MFMAs run out of the other (global ->
// registers at the top of a stage, registers -> LDS between its last MFMAs), every wave keeps 6 (A) / 5 (B) 32 x 32 accumulator tiles
// for the whole range -- 48 / 40 tiles per workgroup -- and both operands of an MFMA are 4-byte LDS reads of one row
// segment (A[i][kk] = X[row kk][i], B[kk][j] = dY[row kk][j]: bank-conflict free, 7-8 reads per 6 MFMAs).
// A workgroup ends with ONE partial slab (192 / 160 KB): 47 MB per minibatch where rounds 2-3 wrote (and re-read)
// 311 MB of per-tile slabs, and no accumulator lives in fa_train_kernel across its rounds.  (What this kernel READS is the
// larger number: the tiles' records, 320 KB per 32-row tile = 524 MB per call at config 3 -- fa_train.hip's header.)
// Roofline: MFMA-bound -- 3.1 k MFMAs per record A against 64 KB of loads (5.3 B per CU-cycle, HBM gives ~10).
// No atomics; every sum has a fixed order: bitwise reproducible.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fa_train.h"

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int DW_NT = 512;

// accumulator tile -> plain row-major global matrix with row length C
__device__ __forceinline__ void store_tile_global(float *dst, int C, const f32x16 &acc, int lane) {
    const int col = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) dst[((reg & 3) + 8 * (reg >> 2) + 4 * hh) * C + col] = acc[reg];
}

__device__ __forceinline__ void load_tile_global(const float *src, int C, f32x16 &acc, int lane) {
    const int col = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) acc[reg] = src[((reg & 3) + 8 * (reg >> 2) + 4 * hh) * C + col];
}

// JOB 0: records A (FA_RECA_FLOATS floats), JOB 1: records B (FA_RECB_FLOATS)
template <int JOB>
__device__ __forceinline__ void dw_range(const float *__restrict__ rec, int q0, int q1, float *slab, float *sT, bool accumulate) {
    constexpr int RF = JOB == 0 ? FA_RECA_FLOATS : FA_RECB_FLOATS;
    constexpr int NS = RF / 4 / DW_NT; // 16-byte pieces per thread and record: 8 / 10
    constexpr int NACC = JOB == 0 ? 6 : 5;
    static_assert(RF % (4 * DW_NT) == 0, "record size");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hh = lane >> 5, u = wave & 3, up = wave >> 2;
    f32x16 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t) acc[t] = f32x16{};
    if (accumulate) { // a later chunk of the minibatch: continue from the slab the earlier chunks left (fixed chunk order)
        if (JOB == 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c) load_tile_global(slab + ((up * 4 + u) * 32) * 128 + c * 32, 128, acc[c], lane);
            load_tile_global(slab + 256 * 128 + (u * 32) * 128 + (up * 2) * 32, 128, acc[4], lane);
            load_tile_global(slab + 256 * 128 + (u * 32) * 128 + (up * 2 + 1) * 32, 128, acc[5], lane);
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) load_tile_global(slab + (u * 32) * 256 + (up * 4 + c) * 32, 256, acc[c], lane);
            load_tile_global(slab + 128 * 256 + up * 64 * 64 + ((u >> 1) * 32) * 64 + (u & 1) * 32, 64, acc[4], lane);
        }
    }
    // this wave's operand columns inside the LDS image of a record (row stride RS floats)
    //   A: waves 0..3: X = h_in block u with dZ blocks 0..3 (dW7 rows 32u..) and with dg blocks 0, 1 (dA_m rows 32u..);
    //      waves 4..7: X = hmix block u with dZ blocks 0..3 (dW7 rows 128 + 32u..), h_in block u with dg blocks 2, 3
    //   B: wave (u, up): X = h3 block u with [dP | dV] blocks 4 up .. 4 up + 3;  up = 0: mix_o block (u >> 1) with de_opp
    //      block (u & 1);  up = 1: h1 block (u >> 1) with dg_o block (u & 1)
    const float *xa, *xb, *ya, *yb;
    int rsx_a, rsx_b, rsy_a, rsy_b;
    if (JOB == 0) {
        xa = sT + (up ? FA_RECA_HMIX : FA_RECA_HIN) + u * 32; rsx_a = 128;
        xb = sT + FA_RECA_HIN + u * 32; rsx_b = 128;
        ya = sT + FA_RECA_DZ; rsy_a = 128;
        yb = sT + FA_RECA_DG + up * 64; rsy_b = 128;
    } else {
        xa = sT + FA_RECB_H3 + u * 32; rsx_a = 128;
        ya = sT + FA_RECB_DPV + up * 128; rsy_a = 256;
        xb = sT + (up ? FA_RECB_H1 : FA_RECB_MO) + (u >> 1) * 32; rsx_b = 64;
        yb = sT + (up ? FA_RECB_DGO : FA_RECB_DE) + (u & 1) * 32; rsy_b = 64;
    }
    const int r0 = hh * (FA_TR_ROWS / 2); // lane half hh walks rows 16 hh .. 16 hh + 15
    xa += r0 * rsx_a + li; xb += r0 * rsx_b + li; ya += r0 * rsy_a + li; yb += r0 * rsy_b + li;

    // Two LDS images of a record: while record q is multiplied out of one, record q + 1 travels global -> registers (requested
    // at the top of the stage, pinned there: the scheduler otherwise sinks the loads to their use and the whole HBM round
    // trip is exposed once per stage) -> the OTHER image, written between the MFMAs of the stage's last k-steps; one
    // barrier per record.  (The staging registers are loaded and written unconditionally -- the last stage re-reads its own
    // record -- so that they stay registers: behind a condition the compiler kept the array in scratch memory.)
    constexpr int WRITE_AT = FA_TR_ROWS / 2 - 4; // k-step at which the next record's registers go to LDS
    f32x4 stage[NS];
    const f32x4 *src = reinterpret_cast<const f32x4 *>(rec + (size_t)q0 * RF);
    if (q0 < q1) {
#pragma unroll
        for (int j = 0; j < NS; ++j) stage[j] = src[tid + j * DW_NT];
#pragma unroll
        for (int j = 0; j < NS; ++j) reinterpret_cast<f32x4 *>(sT)[tid + j * DW_NT] = stage[j];
    }
    __syncthreads();
    for (int q = q0; q < q1; ++q) {
        const int cur = ((q - q0) & 1) * FA_RECB_FLOATS, nxt = FA_RECB_FLOATS - cur; // float offsets of the two images
        src = reinterpret_cast<const f32x4 *>(rec + (size_t)(q + 1 < q1 ? q + 1 : q) * RF);
#pragma unroll
#if FA_REC_NT
        for (int j = 0; j < NS; ++j) stage[j] = __builtin_nontemporal_load(src + tid + j * DW_NT); // (read once)
#else
        for (int j = 0; j < NS; ++j) stage[j] = src[tid + j * DW_NT];
#endif
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < FA_TR_ROWS / 2; ++t) {
            const float a0 = xa[cur + t * rsx_a], a1 = xb[cur + t * rsx_b];
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, ya[cur + t * rsy_a + c * 32], acc[c], 0, 0, 0);
            if (JOB == 0) {
                acc[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, yb[cur + t * rsy_b], acc[4], 0, 0, 0);
                acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, yb[cur + t * rsy_b + 32], acc[5], 0, 0, 0);
            } else {
                acc[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, yb[cur + t * rsy_b], acc[4], 0, 0, 0);
            }
            if (t == WRITE_AT) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < NS; ++j) reinterpret_cast<f32x4 *>(sT + nxt)[tid + j * DW_NT] = stage[j];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads(); // the next image is complete, and every wave has read this one
    }
    if (JOB == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) store_tile_global(slab + ((up * 4 + u) * 32) * 128 + c * 32, 128, acc[c], lane);
        store_tile_global(slab + 256 * 128 + (u * 32) * 128 + (up * 2) * 32, 128, acc[4], lane);
        store_tile_global(slab + 256 * 128 + (u * 32) * 128 + (up * 2 + 1) * 32, 128, acc[5], lane);
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) store_tile_global(slab + (u * 32) * 256 + (up * 4 + c) * 32, 256, acc[c], lane);
        store_tile_global(slab + 128 * 256 + up * 64 * 64 + ((u >> 1) * 32) * 64 + (u & 1) * 32, 64, acc[4], lane);
    }
}

__global__ __launch_bounds__(DW_NT) void fa_train_dw_kernel(const float *__restrict__ rec_a, const float *__restrict__ rec_b, int tiles,
                                                            float *dw_slabs, bool accumulate) {
    __shared__ __attribute__((aligned(16))) float sT[2 * FA_RECB_FLOATS]; // 2 x 80 KB: two records (all of a CU's LDS)
    const int b = blockIdx.x;
    if (b < FA_DW_WGS_A) {
        const int cnt = tiles * 3, per = (cnt + FA_DW_WGS_A - 1) / FA_DW_WGS_A;
        const int q0 = min(b * per, cnt), q1 = min(q0 + per, cnt);
        dw_range<0>(rec_a, q0, q1, dw_slabs + (size_t)b * FA_DWA_FLOATS, sT, accumulate);
    } else {
        const int w = b - FA_DW_WGS_A, per = (tiles + FA_DW_WGS_B - 1) / FA_DW_WGS_B;
        const int q0 = min(w * per, tiles), q1 = min(q0 + per, tiles);
        dw_range<1>(rec_b, q0, q1, dw_slabs + (size_t)FA_DW_WGS_A * FA_DWA_FLOATS + (size_t)w * FA_DWB_FLOATS, sT, accumulate);
    }
}

// Stage 1 of the small gradients: part p sums the tiles [p * per, (p + 1) * per) of mslab (coalesced: a thread per
// element), FA_MRED_PARTS parts.
__global__ __launch_bounds__(256) void fa_train_mred_kernel(const float *__restrict__ mslab, int tiles, float *__restrict__ mpart) {
    const int k = blockIdx.x * 256 + threadIdx.x, p = blockIdx.y;
    if (k >= FA_MSLAB_FLOATS) return;
    const int per = (tiles + FA_MRED_PARTS - 1) / FA_MRED_PARTS;
    const int t0 = min(p * per, tiles), t1 = min(t0 + per, tiles);
    float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    int t = t0;
    for (; t + 4 <= t1; t += 4) {
#pragma unroll
        for (int v = 0; v < 4; ++v) s[v] += mslab[(size_t)(t + v) * FA_MSLAB_FLOATS + k];
    }
    for (; t < t1; ++t) s[0] += mslab[(size_t)t * FA_MSLAB_FLOATS + k];
    mpart[p * FA_MSLAB_FLOATS + k] = (s[0] + s[1]) + (s[2] + s[3]);
}

// out[k], k < FA_SLAB_LOSS + 8: the plain-layout gradient of every kernel-facing matrix + the loss sums, from the
// partial slabs of fa_train_dw_kernel and the stage-1 sums of the small gradients.  FA_RED_U interleaved partial sums
// = loads in flight per lane; fixed order.
#ifndef FA_RED_U
#define FA_RED_U 16
#endif
__device__ __forceinline__ float sum_strided(const float *__restrict__ p, size_t stride, int n) {
    float s[FA_RED_U];
#pragma unroll
    for (int v = 0; v < FA_RED_U; ++v) s[v] = 0.0f;
    int t = 0;
    for (; t + FA_RED_U <= n; t += FA_RED_U) {
#pragma unroll
        for (int v = 0; v < FA_RED_U; ++v) s[v] += p[(size_t)(t + v) * stride];
    }
    for (; t < n; ++t) s[0] += p[(size_t)t * stride];
#pragma unroll
    for (int w = FA_RED_U / 2; w >= 1; w >>= 1)
#pragma unroll
        for (int v = 0; v < w; ++v) s[v] += s[v + w];
    return s[0];
}

__global__ __launch_bounds__(128) void fa_train_reduce_kernel(const float *__restrict__ mpart, const float *__restrict__ dw_slabs,
                                                              float *__restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= FA_SLAB_LOSS + 8) return;
    const float *sa = dw_slabs, *sb = dw_slabs + (size_t)FA_DW_WGS_A * FA_DWA_FLOATS;
    float v = 0.0f;
    if (k >= FA_POFF_W7 && k < FA_POFF_W7 + 256 * 128) v = sum_strided(sa + (k - FA_POFF_W7), FA_DWA_FLOATS, FA_DW_WGS_A);
    else if (k >= FA_POFF_AM && k < FA_POFF_AM + 128 * 128) v = sum_strided(sa + 256 * 128 + (k - FA_POFF_AM), FA_DWA_FLOATS, FA_DW_WGS_A);
    else if (k >= FA_POFF_W8 && k < FA_POFF_W8 + 128 * 256) v = sum_strided(sb + (k - FA_POFF_W8), FA_DWB_FLOATS, FA_DW_WGS_B);
    else if (k >= FA_POFF_BO && k < FA_POFF_BO + 64 * 64) v = sum_strided(sb + 128 * 256 + (k - FA_POFF_BO), FA_DWB_FLOATS, FA_DW_WGS_B);
    else if (k >= FA_POFF_AO && k < FA_POFF_AO + 64 * 64) v = sum_strided(sb + 128 * 256 + 64 * 64 + (k - FA_POFF_AO), FA_DWB_FLOATS, FA_DW_WGS_B);
    else {
        int m = -1; // index into a tile's small slab
        if (k < FA_POFF_AO) m = k; // encoders: the same offsets
        else if (k >= FA_POFF_BU && k < FA_POFF_BU + 128) m = FA_MSLAB_BU + (k - FA_POFF_BU);
        else if (k >= FA_POFF_B8 && k < FA_POFF_B8 + 256) m = FA_MSLAB_B8 + (k - FA_POFF_B8);
        else if (k >= FA_POFF_B9 && k < FA_POFF_B9 + 32) m = FA_MSLAB_B9 + (k - FA_POFF_B9);
        else if (k >= FA_SLAB_LOSS && k < FA_SLAB_LOSS + 4) m = FA_MSLAB_LOSS + (k - FA_SLAB_LOSS);
        else if (k >= FA_POFF_W9 && k < FA_POFF_W9 + 256 * 32) { // W9 is block diagonal: logits weights, value weights
            const int r = (k - FA_POFF_W9) >> 5, c = (k - FA_POFF_W9) & 31;
            if (r < 128 && c < 8) m = FA_MSLAB_W9C + r * 8 + c;
            else if (r >= 128 && c == 8) m = FA_MSLAB_W9C + 1024 + (r - 128);
        }
        if (m >= 0) v = sum_strided(mpart + m, FA_MSLAB_FLOATS, FA_MRED_PARTS);
    }
    out[k] = v;
}
} // namespace

hipError_t fa_launch_train_dw(const float *rec_a, const float *rec_b, int tiles, float *dw_slabs, bool accumulate, hipStream_t st) {
    hipLaunchKernelGGL(fa_train_dw_kernel, dim3(FA_DW_WGS_A + FA_DW_WGS_B), dim3(DW_NT), 0, st, rec_a, rec_b, tiles, dw_slabs, accumulate);
    return hipGetLastError();
}

hipError_t fa_launch_train_reduce(const float *mslab, int tiles, float *mpart, const float *dw_slabs, float *out, hipStream_t st) {
    hipLaunchKernelGGL(fa_train_mred_kernel, dim3((FA_MSLAB_FLOATS + 255) / 256, FA_MRED_PARTS), dim3(256), 0, st, mslab, tiles, mpart);
    hipLaunchKernelGGL(fa_train_reduce_kernel, dim3((FA_SLAB_LOSS + 8 + 127) / 128), dim3(128), 0, st, mpart, dw_slabs, out);
    return hipGetLastError();
}
