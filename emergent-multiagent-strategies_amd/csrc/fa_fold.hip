// fa_fold.hip -- the small dense algebra around the fused PPO step, as two kernels instead of ~120 tiny
// PyTorch launches per optimizer step (each ~5 us inside a hipGraph: a third of the step):
//   * fa_task_kernel: a list of strided matrix products / copies C = alpha * op(A) op(B), FA_TASK_SLICES
//     workgroups per task.  The host (mpnn_pack.FlatPolicy) writes the lists once: FOLD the module's parameters into the
//     kernel-facing matrices (A_o = norm W_key W_query^T, ... see mpnn_pack.py) and UNFOLD the gradients of
//     those matrices back onto the parameters (the chain rule of the same products).
//   * fa_pack_kernel: plain row-major matrices -> the MFMA B-operand lane order of fa_policy.h, forward and
//     transposed packs in one launch.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fa_train.h"

#define FA_TASK_SLICES 64
namespace {
// C[i*ldc + j] = alpha * sum_k A[i*a_rs + k*a_cs] * B[k*b_rs + j*b_cs]   (type 0)
// C[i*ldc + j] = alpha * A[i*a_rs + j*a_cs]                               (type 1)
__global__ __launch_bounds__(256) void fa_task_kernel(const fa_task *__restrict__ tasks) {
    // blockIdx.y: FA_TASK_SLICES workgroups share a task's outputs.  A product's output is owned by EIGHT adjacent
    // lanes, lane q taking the terms k = q (mod 8): the dependent chain of loads per lane is K / 8 long instead of
    // K (these launches are latency chains, not bandwidth: the largest product is 128 x 128 x 128), and the eight
    // partial sums meet in a fixed order (three xor-shuffles): reproducible.
    const fa_task t = tasks[blockIdx.x];
    const int total = t.M * t.N;
    const int gid = blockIdx.y * 256 + threadIdx.x;
    if (t.type == 1) {
        for (int o = gid; o < total; o += 256 * FA_TASK_SLICES) {
            const int i = o / t.N, j = o - i * t.N;
            t.C[(size_t)i * t.ldc + j] = t.alpha * t.A[(size_t)i * t.a_rs + (size_t)j * t.a_cs];
        }
        return;
    }
    const int q = gid & 7;
    for (int o0 = 0; o0 < total; o0 += 32 * FA_TASK_SLICES) { // (uniform trip count: the shuffles see whole groups)
        const int o = o0 + (gid >> 3);
        const bool live = o < total;
        const int i = live ? o / t.N : 0, j = live ? o - i * t.N : 0;
        const float *a = t.A + (size_t)i * t.a_rs, *b = t.B + (size_t)j * t.b_cs;
        float p0 = 0.0f, p1 = 0.0f;
        int k = q;
        for (; k + 8 < t.K; k += 16) {
            p0 = fmaf(a[(size_t)k * t.a_cs], b[(size_t)k * t.b_rs], p0);
            p1 = fmaf(a[(size_t)(k + 8) * t.a_cs], b[(size_t)(k + 8) * t.b_rs], p1);
        }
        if (k < t.K) p0 = fmaf(a[(size_t)k * t.a_cs], b[(size_t)k * t.b_rs], p0);
        float acc = p0 + p1;
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        acc += __shfl_xor(acc, 4);
        if (live && q == 0) t.C[(size_t)i * t.ldc + j] = t.alpha * acc;
    }
}

struct PackSpec { int src, K, C, dst; }; // plain offset, rows, columns, packed offset
__device__ __forceinline__ void pack_one(const float *__restrict__ plain, float *__restrict__ out, PackSpec s, bool transpose, int idx) {
    // element idx of the packed (K x C) matrix W (or of W^T when `transpose`): float4 index (cb*K/8 + t4)*64 + lane,
    // component q  <->  W[k = (lane >> 5)*K/2 + 4*t4 + q][c = 32*cb + (lane & 31)]
    const int K = transpose ? s.C : s.K, C = transpose ? s.K : s.C;
    const int q = idx & 3, f4 = idx >> 2, lane = f4 & 63, rest = f4 >> 6, t4 = rest % (K / 8), cb = rest / (K / 8);
    const int k = (lane >> 5) * (K / 2) + 4 * t4 + q, c = 32 * cb + (lane & 31);
    out[s.dst + idx] = transpose ? plain[s.src + c * s.C + k] : plain[s.src + k * s.C + c];
}
// element (k, c) of the (K x C) matrix -> its three bf16 terms in the bf16x3 pack at float offset dst3 (fa_policy.h FA_POFF3_*)
__device__ __forceinline__ unsigned pk_rne_hi(float x) {
    const unsigned u = __float_as_uint(x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
}
__device__ __forceinline__ void pack_x3(float x, float *__restrict__ w, int dst3, int K, int k, int c) {
    const unsigned h = pk_rne_hi(x);
    const float r1 = x - __uint_as_float(h);
    const unsigned m = pk_rne_hi(r1);
    const unsigned l = __float_as_uint(r1 - __uint_as_float(m));
    const int hh = k / (K / 2), kk = k - hh * (K / 2), s = kk >> 3, j = kk & 7, cb = c >> 5, li = c & 31;
    uint16_t *w16 = reinterpret_cast<uint16_t *>(w + dst3);
    const size_t base = ((size_t)(cb * (K / 16) + s) * 3 * 64 + hh * 32 + li) * 8 + j;
    w16[base] = (uint16_t)(h >> 16);
    w16[base + 64 * 8] = (uint16_t)(m >> 16);
    w16[base + 2 * 64 * 8] = (uint16_t)(l >> 16);
}
__global__ __launch_bounds__(256) void fa_pack_kernel(const float *__restrict__ plain, float *__restrict__ w, float *__restrict__ wt) {
    const int dst3[6] = {FA_POFF3_AO, FA_POFF3_BO, FA_POFF3_AM, FA_POFF3_W7, FA_POFF3_W8, FA_POFF3_W9};
    const PackSpec fwd[6] = {{FA_POFF_AO, 64, 64, FA_POFF_AO}, {FA_POFF_BO, 64, 64, FA_POFF_BO}, {FA_POFF_AM, 128, 128, FA_POFF_AM},
                             {FA_POFF_W7, 256, 128, FA_POFF_W7}, {FA_POFF_W8, 128, 256, FA_POFF_W8}, {FA_POFF_W9, 256, 32, FA_POFF_W9}};
    const int tdst[6] = {FA_TOFF_AOT, FA_TOFF_BOT, FA_TOFF_AMT, FA_TOFF_W7T, FA_TOFF_W8T, FA_TOFF_W9T};
    const int g = blockIdx.x * 256 + threadIdx.x;
    // the un-packed sections (encoders, biases) are copied as they are
    if (g < FA_POFF_AO) w[g] = plain[g];
    if (g >= FA_POFF_BU && g < FA_POFF_BU + 128) w[g] = plain[g];
    if (g >= FA_POFF_B8 && g < FA_POFF_B8 + 256) w[g] = plain[g];
    if (g >= FA_POFF_B9 && g < FA_POFF_B9 + 32) w[g] = plain[g];
#pragma unroll
    for (int m = 0; m < 6; ++m) {
        const int n = fwd[m].K * fwd[m].C;
        if (g >= fwd[m].src && g < fwd[m].src + n) {
            pack_one(plain, w, fwd[m], false, g - fwd[m].src);
            {   // plain element g - src = k * C + c
                const int e = g - fwd[m].src, k = e / fwd[m].C, c = e - k * fwd[m].C;
                pack_x3(plain[g], w, dst3[m], fwd[m].K, k, c);
            }
            PackSpec ts = fwd[m];
            ts.dst = tdst[m];
            pack_one(plain, wt, ts, true, g - fwd[m].src);
        }
    }
}
} // namespace

hipError_t fa_launch_tasks(const fa_task *tasks, int n, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(fa_task_kernel, dim3(n, FA_TASK_SLICES), dim3(256), 0, st, tasks);
    return hipGetLastError();
}
hipError_t fa_launch_pack(const float *plain, float *w, float *wt, hipStream_t st) {
    hipLaunchKernelGGL(fa_pack_kernel, dim3((FA_POLICY_PLAIN_FLOATS + 255) / 256), dim3(256), 0, st, plain, w, wt);
    return hipGetLastError();
}
