// tools/fa_probe_timing.h -- the timing implementation of csrc/fa_probe.h's hooks (experiments only;
// built by tools/make_timing_build.py, read by tools/timing_probe.py).  Sections accumulate
// shader-clock ticks per wave role into g_dbg; g_hw records where the hardware placed each wave.
#pragma once
#if defined(FA_PROBE_POLICY_TU) // ---- fa_policy.hip: wave 0 (and wave 4) of one workgroup record the shader clock at phase marks
__device__ unsigned long long g_pl[128];
#define FA_PL_TICK(k) if ((threadIdx.x & 255) == 0 && blockIdx.x == 100 && blockIdx.y == 0) g_pl[(k) + (threadIdx.x >> 8) * 64] = clock64();
extern "C" int fa_dbg_policy(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pl), sizeof(unsigned long long) * 128); }
#define FA_TR_TICK(k)
#elif !defined(FA_PROBE_TRAIN_TU) // ---- the pipelined step kernel's translation unit (fa_step_pipe.hip)
#ifndef FA_TICK_WAVE1
#define FA_TICK_WAVE1 0 // 1: report pair wave 1 instead of the last pair wave
#endif
__device__ unsigned long long g_dbg[32];
__device__ unsigned g_hw[4096];
#define FA_TICK_INIT unsigned long long tacc[24] = {0}; unsigned long long tlast = clock64();
#define FA_TICK(k) { const unsigned long long _n = clock64(); tacc[k] += _n - tlast; tlast = _n; }
#define FA_TICK_FLUSH(lo, hi, cnt) if (lane == 0) { for (int k = lo; k < hi; ++k) atomicAdd(&g_dbg[k], tacc[k]); atomicAdd(&g_dbg[cnt], 1ull); }
// where the hardware put this wave (HW_ID: simd [5:4], cu [11:8], se [15:13])
#define FA_PROBE_HWID(lane, wave_id) if (lane == 0 && blockIdx.x < 512) { unsigned hw; \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); g_hw[blockIdx.x * 8 + wave_id] = hw | 0x80000000u; }
#define FA_PROBE_WAVE0_BEGIN const unsigned long long tp0 = clock64();
#define FA_PROBE_WAVE0_LOOP_BEGIN(lane) const unsigned long long tk0 = clock64(), tw0 = wall_clock64(); \
    if (lane == 0) atomicAdd(&g_dbg[22], tk0 - tp0); // prologue: launch of the wave -> P(-1) passed
#define FA_PROBE_WAVE0_LOOP_END(lane) if (lane == 0) { atomicAdd(&g_dbg[20], clock64() - tk0); atomicAdd(&g_dbg[21], wall_clock64() - tw0); }
#define FA_PROBE_WAVE0_END(lane) if (lane == 0) atomicAdd(&g_dbg[23], clock64() - tk0); // P(-1) -> end of wave 0
extern "C" int fa_dbg_read(unsigned long long *out, int reset) {
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), sizeof(unsigned long long) * 32);
    if (reset) { unsigned long long z[32] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), z, sizeof(z)); }
    return 0;
}
extern "C" int fa_dbg_hw(unsigned *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hw), sizeof(unsigned) * 4096); }

#define FA_TR_TICK(k)
#define FA_PL_TICK(k)
#else // ---- fa_train.hip
#define FA_PL_TICK(k)
// fa_train.hip: workgroup 0 / thread 0 records the shader clock at phase marks
__device__ unsigned long long g_tr[64];
#define FA_TR_TICK(k) if (threadIdx.x == 0 && blockIdx.x == 7) g_tr[k] = clock64();
extern "C" int fa_dbg_train(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tr), sizeof(unsigned long long) * 64); }
#endif
