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
#include <stdlib.h>

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

// ---- the same GEMM on the bf16 matrix cores with fp32-class accuracy: the three-way split -------------------------------
// gfx950's v_mfma_f32_32x32x16_bf16 retires 16 x the MACs per cycle of v_mfma_f32_32x32x2_f32.  A float is the EXACT sum of
// three bf16 numbers -- hi = x rounded to nearest at 8 significant bits, mid = the remainder rounded likewise, lo = what is
// left (<= 6 bits): x = hi + mid + lo with no error, |mid| <= 2^-9 |x|, |lo| <= 2^-18 |x| -- so x y = the sum of nine
// bf16 x bf16 products, each exact in the MFMA's fp32 accumulator.  The six largest are issued (hi hi, hi mid, mid hi,
// hi lo, lo hi, mid mid); the three dropped ones (mid lo, lo mid, lo lo) are <= 2^-26 |x y| together and of either sign (a
// truncating split leaves them all with the sign of x y: a bias the size of one fp32 rounding per product, measured as
// 2-3 x the fp32 kernel's error against an fp64 GEMM) -- a quarter of ONE fp32 rounding of the product, which the fp32
// FMA chain of the kernel above commits at every step.  6 MFMAs of 32 cycles cover K = 16 where the fp32 form needs 8 of 64:
// 2.67 x less matrix-core time for the same sums.  The split happens once per element, in registers, on the way from global
// memory to the LDS image (the records in HBM stay fp32); the image holds the operands already in MFMA fragment order --
// [plane][k half][term][column][8 bf16 of consecutive rows] -- so that both operands of an MFMA are one 16-byte LDS read,
// conflict-free.  Half a record (16 rows = one K = 16 step) per stage, two images, one barrier per stage.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// x -> the bit patterns of (hi, mid, lo) in the HIGH halves of three words
__device__ __forceinline__ unsigned bf16_rne_hi(float x) { // round to nearest even at bit 16, result in the high half
    const unsigned u = __float_as_uint(x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
}
__device__ __forceinline__ void split3(float x, unsigned &h, unsigned &m, unsigned &l) {
#ifdef DW3_KO_SPLIT
    h = m = l = __float_as_uint(x);
    return;
#endif
    h = bf16_rne_hi(x);
    const float r1 = x - __uint_as_float(h);            // exact; |r1| <= 2^-9 |x|
    m = bf16_rne_hi(r1);
    l = __float_as_uint(r1 - __uint_as_float(m));       // exact, <= 2^-18 |x| on x's 2^-23 grid: <= 6 significant bits, a bf16 as it is
}
// the high halves of (a, b) -> one word, a in the low half (element 2j), b in the high half (element 2j + 1)
__device__ __forceinline__ unsigned pack_hi(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// image geometry: a "plane" is 128 columns wide; element (plane, khalf, term, col) = 16 bytes
#define DW3_PLANE_BYTES (2 * 3 * 128 * 16)
__device__ __forceinline__ int dw3_off(int plane, int khalf, int term, int col) { // in 16-byte units
    return ((plane * 2 + khalf) * 3 + term) * 128 + col;
}

// one unit of staging: 8 rows x 2 columns of fp32 (rows r0 .. r0 + 7 of a matrix with row stride `rs`) -> registers
struct Dw3Unit { f32x2 v[8]; };
__device__ __forceinline__ void dw3_load(Dw3Unit &u, const float *__restrict__ base, int rs) {
#pragma unroll
    for (int j = 0; j < 8; ++j) u.v[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x2 *>(base + j * rs));
}
// registers -> split -> the image: column col + c of (plane, khalf), c = 0 / 1 (two rows at a time: short live ranges)
__device__ __forceinline__ void dw3_store_col(const Dw3Unit &u, int c, u32x4 *img, int plane, int khalf, int col) {
    u32x4 H, M, L;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned h0, m0, l0, h1, m1, l1;
        split3(u.v[2 * j][c], h0, m0, l0);
        split3(u.v[2 * j + 1][c], h1, m1, l1);
        H[j] = pack_hi(h0, h1);
        M[j] = pack_hi(m0, m1);
        L[j] = pack_hi(l0, l1);
    }
    img[dw3_off(plane, khalf, 0, col + c)] = H;
    img[dw3_off(plane, khalf, 1, col + c)] = M;
    img[dw3_off(plane, khalf, 2, col + c)] = L;
}

struct Dw3Frag { bf16x8 h, m, l; };
__device__ __forceinline__ Dw3Frag dw3_frag(const u32x4 *img, int plane, int khalf, int col) {
    Dw3Frag f;
    f.h = __builtin_bit_cast(bf16x8, img[dw3_off(plane, khalf, 0, col)]);
    f.m = __builtin_bit_cast(bf16x8, img[dw3_off(plane, khalf, 1, col)]);
    f.l = __builtin_bit_cast(bf16x8, img[dw3_off(plane, khalf, 2, col)]);
    return f;
}
// acc += X^T Y over the 16 rows of the stage: the six largest of the nine cross products, small ones first
__device__ __forceinline__ void dw3_mma(f32x16 &acc, const Dw3Frag &x, const Dw3Frag &y) {
#ifdef DW3_KO_MFMA // (knock-out builds, tools/build_variant.py: where the kernel's time goes)
    acc[0] += __builtin_bit_cast(float, (unsigned)x.h[0]) + __builtin_bit_cast(float, (unsigned)y.l[7]);
    return;
#endif
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.h, y.l, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.l, y.h, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.m, y.m, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.h, y.m, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.m, y.h, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.h, y.h, acc, 0, 0, 0);
}

// JOB 0: records A = planes [h_in | hmix | dZ | dg], each 32 x 128.  JOB 1: records B as five 128-wide virtual planes
// [h3 | dPV[:, :128] | dPV[:, 128:] | mix_o, de_opp | h1, dg_o].  A stage = rows 16 s .. 16 s + 15 of a record.
template <int JOB>
__device__ __forceinline__ void dw3_range(const float *__restrict__ rec, int qfirst, int qstride, int nrec, float *slab, u32x4 *sI,
                                          bool accumulate) {
    constexpr int RF = JOB == 0 ? FA_RECA_FLOATS : FA_RECB_FLOATS;
    constexpr int NP = JOB == 0 ? 4 : 5;
    constexpr int IMG = NP * DW3_PLANE_BYTES / 16; // 16-byte units per image
    constexpr int NACC = JOB == 0 ? 6 : 5;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hh = lane >> 5, u = wave & 3, up = wave >> 2;
    f32x16 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t) acc[t] = f32x16{};
    if (accumulate) {
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
    // staging: thread -> unit (plane, khalf, column pair); JOB 1 has 640 units: threads 0..127 take one of plane 4 as well
    const int s_plane = tid >> 7, s_kh = (tid >> 6) & 1, s_cp = tid & 63;
    auto unit_src = [&](int plane, int kh, int cp, const float *r, int &rs) -> const float * { // first of the unit's 8 rows
        const int col = 2 * cp;
        if (JOB == 0) { rs = 128; return r + plane * FA_REC_PLANE + (kh * 8) * 128 + col; }
        if (plane == 0) { rs = 128; return r + FA_RECB_H3 + (kh * 8) * 128 + col; }
        if (plane <= 2) { rs = 256; return r + FA_RECB_DPV + (kh * 8) * 256 + (plane - 1) * 128 + col; }
        rs = 64; // two 32 x 64 matrices side by side
        const int basef = plane == 3 ? (col < 64 ? FA_RECB_MO : FA_RECB_DE) : (col < 64 ? FA_RECB_H1 : FA_RECB_DGO);
        return r + basef + (kh * 8) * 64 + (col & 63);
    };
    // the unit's first row inside a record and its row stride: loop invariant
    int rs, rs2 = 64;
    const float *ubase = unit_src(s_plane, s_kh, s_cp, rec, rs);
    const float *ubase2 = JOB == 1 ? unit_src(4, s_kh, s_cp, rec, rs2) : rec;
    const bool two = JOB == 1 && tid < 128;
    // Prefetch distance TWO stages, two register sets: the loads of stage s + 2 are requested at the top of stage s; what
    // stage s splits and writes to LDS (the image of stage s + 1) was requested a whole stage earlier, so the split's VALU
    // work interleaves with the MFMAs without anybody waiting for HBM (requested one stage ahead only, the scheduler hoists
    // the split in front of the MFMAs and every stage starts with an exposed HBM round trip: 148 us per call instead of ...).
    Dw3Unit s0, s1, t0, t1; // (t*: the second unit of JOB 1's threads 0..127)
    // The workgroup's records are qfirst, qfirst + qstride, ...: INTERLEAVED over the workgroups, so that at any moment the
    // 200 / 56 workgroups read one contiguous window of the record array (contiguous per-workgroup ranges put 256 streams at a
    // fixed 1.6 MB stride: a few HBM channels take most of the requests)
    const int st0 = 0, st1 = 2 * nrec;
    auto stage_load = [&](int st, Dw3Unit &a, Dw3Unit &b) { // st = 2 * (local record) + half
        st = st < st1 ? st : st1 - 1;
#ifdef DW3_KO_LOAD
        st &= 1; // every stage re-reads the workgroup's first record: L2 hits
#endif
        const size_t ro = (size_t)(qfirst + (st >> 1) * qstride) * RF;
        const int row0 = (st & 1) * 16;
        dw3_load(a, ubase + ro + row0 * rs, rs);
        if (two) dw3_load(b, ubase2 + ro + row0 * rs2, rs2);
    };
    // One stage: six groups of six MFMAs (five for JOB 1), the operands of group g + 1 read from LDS ahead of group g's MFMAs,
    // the split + LDS write of the NEXT stage's image spread over the groups a column at a time; scheduling fences between the
    // groups (left alone the scheduler hoists all LDS reads and the whole split to the top: 82 spilled registers)
    auto stage = [&](const u32x4 *cur, u32x4 *nxt, const Dw3Unit &a, const Dw3Unit &b) {
        const Dw3Frag xa = dw3_frag(cur, JOB == 0 ? (up ? 1 : 0) : 0, hh, u * 32 + li);
        const Dw3Frag xb = JOB == 0 ? dw3_frag(cur, 0, hh, u * 32 + li) : dw3_frag(cur, 3 + up, hh, (u >> 1) * 32 + li);
        auto yfrag = [&](int g) {
            if (JOB == 0) return g < 4 ? dw3_frag(cur, 2, hh, g * 32 + li) : dw3_frag(cur, 3, hh, (up * 2 + g - 4) * 32 + li);
            return g < 4 ? dw3_frag(cur, 1 + up, hh, g * 32 + li) : dw3_frag(cur, 3 + up, hh, 64 + (u & 1) * 32 + li);
        };
        Dw3Frag y = yfrag(0);
#pragma unroll
        for (int g = 0; g < NACC; ++g) {
            Dw3Frag yn = y;
            if (g + 1 < NACC) yn = yfrag(g + 1);
            dw3_mma(acc[g], g < 4 ? xa : xb, y);
            if (g == 0) dw3_store_col(a, 0, nxt, s_plane, s_kh, 2 * s_cp);
            if (g == 1) dw3_store_col(a, 1, nxt, s_plane, s_kh, 2 * s_cp);
            if (JOB == 1 && two && g == 2) dw3_store_col(b, 0, nxt, 4, s_kh, 2 * s_cp);
            if (JOB == 1 && two && g == 3) dw3_store_col(b, 1, nxt, 4, s_kh, 2 * s_cp);
            y = yn;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    u32x4 *img0 = sI, *img1 = sI + IMG;
    if (st0 < st1) {
        stage_load(st0, s0, t0);
        stage_load(st0 + 1, s1, t1);
        dw3_store_col(s0, 0, img0, s_plane, s_kh, 2 * s_cp);
        dw3_store_col(s0, 1, img0, s_plane, s_kh, 2 * s_cp);
        if (two) {
            dw3_store_col(t0, 0, img0, 4, s_kh, 2 * s_cp);
            dw3_store_col(t0, 1, img0, 4, s_kh, 2 * s_cp);
        }
    }
    __syncthreads();
    for (int st = st0; st < st1; st += 2) { // (a whole number of records: an even number of stages)
        stage_load(st + 2, s0, t0);
        __builtin_amdgcn_sched_barrier(0);
        stage(img0, img1, s1, t1);
        __syncthreads();
        stage_load(st + 3, s1, t1);
        __builtin_amdgcn_sched_barrier(0);
        stage(img1, img0, s0, t0);
        __syncthreads();
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

__global__ __launch_bounds__(DW_NT) void fa_train_dw3_kernel(const float *__restrict__ rec_a, const float *__restrict__ rec_b, int tiles,
                                                             float *dw_slabs, bool accumulate) {
    __shared__ __attribute__((aligned(16))) u32x4 sI[2 * 5 * DW3_PLANE_BYTES / 16]; // two images of <= 5 planes: 120 KB
    const int b = blockIdx.x;
    if (b < FA_DW_WGS_A) {
        const int cnt = tiles * 3;
        dw3_range<0>(rec_a, b, FA_DW_WGS_A, b < cnt ? (cnt - b + FA_DW_WGS_A - 1) / FA_DW_WGS_A : 0, dw_slabs + (size_t)b * FA_DWA_FLOATS, sI,
                     accumulate);
    } else {
        const int w = b - FA_DW_WGS_A;
        dw3_range<1>(rec_b, w, FA_DW_WGS_B, w < tiles ? (tiles - w + FA_DW_WGS_B - 1) / FA_DW_WGS_B : 0,
                     dw_slabs + (size_t)FA_DW_WGS_A * FA_DWA_FLOATS + (size_t)w * FA_DWB_FLOATS, sI, accumulate);
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
    // FA_DW_GEMM=f32: the fp32-MFMA form (rounds 4-5; the definition the split form is tested against)
    const char *v = getenv("FA_DW_GEMM");
    if (!(v && v[0] == 'f')) {
        hipLaunchKernelGGL(fa_train_dw3_kernel, dim3(FA_DW_WGS_A + FA_DW_WGS_B), dim3(DW_NT), 0, st, rec_a, rec_b, tiles, dw_slabs, accumulate);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(fa_train_dw_kernel, dim3(FA_DW_WGS_A + FA_DW_WGS_B), dim3(DW_NT), 0, st, rec_a, rec_b, tiles, dw_slabs, accumulate);
    return hipGetLastError();
}

hipError_t fa_launch_train_reduce(const float *mslab, int tiles, float *mpart, const float *dw_slabs, float *out, hipStream_t st) {
    hipLaunchKernelGGL(fa_train_mred_kernel, dim3((FA_MSLAB_FLOATS + 255) / 256, FA_MRED_PARTS), dim3(256), 0, st, mslab, tiles, mpart);
    hipLaunchKernelGGL(fa_train_reduce_kernel, dim3((FA_SLAB_LOSS + 8 + 127) / 128), dim3(128), 0, st, mpart, dw_slabs, out);
    return hipGetLastError();
}
