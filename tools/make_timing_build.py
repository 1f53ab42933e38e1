#!/usr/bin/env python
"""Kernel experiment: build exp_libs/libfa_timing.so, a copy of the step kernel instrumented
with clock64() at its section boundaries (per-section shader cycles summed per wave; read by
tools/timing_probe.py).  The product sources are not modified."""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "emergent-multiagent-strategies_amd", "csrc")
s = open(os.path.join(CSRC, "fa_step.hip")).read()


def rep(old, new):
    global s
    assert old in s, old
    s = s.replace(old, new, 1)


rep('#include "fa_device.h"\n', '''#include "fa_device.h"
__device__ unsigned long long g_dbg[16];
#define TICK(k) { unsigned long long _n = clock64(); tacc[k] += _n - tlast; tlast = _n; }
''')
rep("    bool dirty = false; // state changed => write it back\n",
    "    bool dirty = false;\n    unsigned long long tacc[10] = {0,0,0,0,0,0,0,0,0,0}; unsigned long long tlast = clock64();\n")
rep("            const bool alive0 = alive;\n", "            TICK(0)\n            const bool alive0 = alive;\n")
rep("                FA_WG_BARRIER(); // (1) the force wave starts on this step's contacts and walls\n",
    "                TICK(1)\n                FA_WG_BARRIER(); // (1)\n                TICK(2)\n")
rep("            // partner deltas for the contact test, fetched now", "            TICK(3)\n            // partner deltas")
rep("            const bool hit = shooter && hit_cnt > 0;\n", "            TICK(4)\n            const bool hit = shooter && hit_cnt > 0;\n")
rep("                FA_WG_BARRIER(); // (2) the force wave masks its candidates with the survivors\n                FA_WG_BARRIER(); // (3) and has published the total force of every lane\n",
    "                TICK(5)\n                FA_WG_BARRIER();\n                FA_WG_BARRIER();\n                TICK(6)\n")
rep("            // ---- rewards (fortattack_env_v1.py:87-188), after World.step ---------------\n", "            TICK(7)\n")
rep("        // ---- fortattack_env_v1.py:47-75 reset_world --------------------------------------\n", "        TICK(8)\n")
rep("        // next iteration restages LDS: keep its writes behind this iteration's reads\n", "        TICK(9)\n")
rep("    // ---- write back state once per launch ----------------------------------------------------\n",
    "    if (lane == 0 && !RESET_ONLY) { for (int k = 0; k < 10; ++k) atomicAdd(&g_dbg[k], tacc[k]); atomicAdd(&g_dbg[10], 1ull); }\n")
s += '''
extern "C" int fa_dbg_read(unsigned long long *out, int reset) {
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), sizeof(unsigned long long) * 16);
    if (reset) { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), z, sizeof(z)); }
    return 0;
}
'''
tmp = tempfile.mkdtemp()
for f in ("fa_collect.hip", "fa_api.hip", "fa_device.h"):
    shutil.copy(os.path.join(CSRC, f), tmp)
open(os.path.join(tmp, "fa_step.hip"), "w").write(s)
os.makedirs(os.path.join(ROOT, "exp_libs"), exist_ok=True)
out = os.path.join(ROOT, "exp_libs", "libfa_timing.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
                       "-shared", "-I", os.path.join(ROOT, "include"), "-I", tmp] +
                      [os.path.join(tmp, f) for f in ("fa_step.hip", "fa_collect.hip", "fa_api.hip")] + ["-o", out])
print(out)
