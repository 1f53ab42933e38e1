// fa_probe.h -- section-timing hooks of the step kernels.
// Product build: every hook expands to nothing.  tools/make_timing_build.py compiles the same
// sources with -DFA_PROBE_IMPL="<tools/fa_probe_timing.h>", which supplies clock64() counters per
// wave role and the fa_dbg_* readback entry points (not part of the C ABI).
#pragma once
#ifdef FA_PROBE_IMPL
#include FA_PROBE_IMPL
#else
#define FA_TICK_WAVE1 0
#define FA_TICK_INIT
#define FA_TICK(k)
#define FA_TICK_FLUSH(lo, hi, cnt)
#define FA_PROBE_HWID(lane, wave_id)
#define FA_PROBE_WAVE0_BEGIN
#define FA_PROBE_WAVE0_LOOP_BEGIN(lane)
#define FA_PROBE_WAVE0_LOOP_END(lane)
#define FA_PROBE_WAVE0_END(lane)
#define FA_TR_TICK(k) // fa_train.hip phase marks
#define FA_PL_TICK(k) // fa_policy.hip phase marks
#endif
