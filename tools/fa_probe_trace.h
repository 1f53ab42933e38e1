// tools/fa_probe_trace.h -- absolute shader-clock timestamps of the pipelined step kernel's marks, for two
// workgroups (a lone one and one that shares its CU), per wave role and step (experiments only; built with
// make_timing_build.py FA_PROBE=fa_probe_trace.h, read by tools/trace_probe.py through fa_dbg_trace).
#pragma once
#ifndef FA_PROBE_TRAIN_TU
#define FA_TICK_WAVE1 0
#define FA_TRACE_STEPS 96
__device__ unsigned long long g_trace[2][FA_TRACE_STEPS][24];
#define FA_TICK_INIT
#define FA_TICK(k) { if (lane == 0 && (blockIdx.x == 3 || blockIdx.x == 300) && s < FA_TRACE_STEPS) { \
    unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); \
    g_trace[blockIdx.x == 300][s][(k) + (wave_id == 1 ? 10 : 0)] = _t; } }
#define FA_TICK_FLUSH(lo, hi, cnt)
#define FA_PROBE_HWID(lane, wave_id)
#define FA_PROBE_WAVE0_BEGIN
#define FA_PROBE_WAVE0_LOOP_BEGIN(lane)
#define FA_PROBE_WAVE0_LOOP_END(lane)
#define FA_PROBE_WAVE0_END(lane)
extern "C" int fa_dbg_trace(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), sizeof(g_trace)); }
#define FA_TR_TICK(k)
#define FA_PL_TICK(k)
#else
#define FA_TR_TICK(k)
#define FA_PL_TICK(k)
#endif
