"""Experiment: per-section shader-clock breakdown of the pipelined step kernel, per wave role
(needs tools/_build/libfa_timing.so from tools/make_timing_build.py, which this script loads in
place of the product library).  usage: timing_probe.py [E] [G] [A]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import emergent_multiagent_strategies_amd as fa
TIMING_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", os.environ.get("FA_TIMING_LIB", "libfa_timing.so"))
fa._lib._build.LIB = TIMING_LIB   # same C ABI + fa_dbg_read / fa_dbg_hw
E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
G = int(sys.argv[2]) if len(sys.argv) > 2 else 3
A = int(sys.argv[3]) if len(sys.argv) > 3 else 3
T = 128
KERNEL = os.environ.get("FA_PROBE_KERNEL", "auto")     # "chain": the one-barrier experiment kernel's sections
eng = fa.BatchedFortAttack(E, G, A, 100, track_counters=False, step_kernel=KERNEL)
st = fa.JointRolloutStorage(T, E, G + A, device="cuda")
eng.bind_storage(st)
st.actions.copy_(torch.randint(0, 8, st.actions.shape, device="cuda"))
eng.collect_reset()
lib = C.CDLL(TIMING_LIB)
buf = (C.c_ulonglong * 32)()
for _ in range(3):
    eng.collect_rollout(0, T)
torch.cuda.synchronize()
lib.fa_dbg_read(buf, 1)
for _ in range(5):
    eng.collect_rollout(0, T)
torch.cuda.synchronize()
lib.fa_dbg_read(buf, 0)
names = {0: "w0 decode + triangles", 1: "w0 laser + ballots", 2: "w0 wait B2", 3: "w0 force sum + integrate",
         4: "w0 door distance + done", 5: "w0 reset", 6: "w0 publish", 7: "w0 wait P",
         10: "out wait P", 11: "out pair forces", 12: "out wait B2", 13: "out rewards + stores",
         16: "last wait P", 17: "last walls", 18: "last wait B2", 19: "last sin/cos"}
groups = ((0, 8, 28), (10, 14, 29), (16, 20, 30), (15, 16, 26))
names[15] = "last: LDS reads landed (probe build with the extra mark only)"
if KERNEL == "chain":
    names = {4: "chain wait P", 0: "chain own pair offsets", 1: "chain hand-off burst (+ waiting)", 2: "chain force sum + integrate",
             3: "chain done + reset + publish", 12: "laser wait P", 10: "laser read + tests + hand-off", 11: "laser next heading",
             18: "walls wait P", 16: "walls decode + walls + hand-off", 17: "walls rewards + reset stream",
             14: "pairs wait P", 8: "pairs rest offsets + hand-off", 9: "pairs obs / done / mask rows"}
    groups = ((0, 5, 28), (10, 13, 29), (16, 19, 30), (8, 10, 31), (14, 15, 27))
for lo, hi, cnt in groups:
    waves = max(buf[cnt], 1)
    tot = 0.0
    for k in range(lo, hi):
        v = buf[k] / waves / T
        tot += v
        print("%-28s %8.1f cycles/step" % (names[k], v))
    print("%-28s %8.1f cycles/step\n" % ("  total", tot))
print("wave 0 loop: %.1f shader-clock ticks per step, %.1f ns per step (100 MHz wall clock) -> %.3f ticks/ns" % (
    buf[20] / max(buf[28], 1) / T, buf[21] / max(buf[28], 1) / T * 10.0, buf[20] / max(buf[21], 1) / 10.0))
print("wave 0: prologue (wave start -> P(-1)) %.0f cycles, P(-1) -> end %.0f cycles" % (buf[22] / max(buf[28], 1), buf[23] / max(buf[28], 1)))
hw = (C.c_uint * 4096)()
lib.fa_dbg_hw(hw)
# HW_ID: wave slot [3:0], simd [5:4], cu [11:8], sh [12], se [15:13] (+ xcc from XCC_ID, not read: CUs are
# distinguished per (se, sh, cu) only, so 8 XCDs alias -- co-residency is judged by launch order instead)
import collections
place = {}
for b in range(min(512, (E + 9) // 10)):
    row = []
    for w in range(4):
        v = hw[b * 8 + w]
        if v & 0x80000000:
            row.append(((v >> 4) & 3, v & 15, (v >> 8) & 15, (v >> 12) & 1, (v >> 13) & 7))
    place[b] = row
pat = collections.Counter(tuple(r[0] for r in row) for row in place.values())
print("wave->simd patterns (w0,w1,w2,w3):", dict(pat))
slots = collections.Counter(tuple(r[1] for r in row) for row in place.values())
print("wave slot ids per workgroup:", dict(slots))
for b in list(range(4)) + list(range(256, 260)):
    if b in place:
        print("  wg %3d: %s" % (b, "  ".join("w%d->simd%d slot%d cu%d.sh%d.se%d" % ((w,) + r) for w, r in enumerate(place[b]))))
