#!/usr/bin/env python
"""Kernel experiment: build tools/_build/libfa_timing.so = the product sources with
-DFA_PROBE_IMPL=tools/fa_probe_timing.h, which turns on the clock64() section counters of the pipelined step kernel (summed per wave
role into g_dbg, read by tools/timing_probe.py through fa_dbg_read)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "emergent-multiagent-strategies_amd", "csrc")
os.makedirs(os.path.join(ROOT, "tools", "_build"), exist_ok=True)
out = os.path.join(ROOT, "tools", "_build", os.environ.get("FA_TIMING_LIB", "libfa_timing.so"))
subprocess.check_call(["/opt/rocm/bin/hipcc"] + sys.argv[1:] + [ "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
                       "-shared", '-DFA_PROBE_IMPL="%s"' % os.environ.get("FA_PROBE", "fa_probe_timing.h"), "-I", os.path.join(ROOT, "tools"), "-I", os.path.join(ROOT, "include")] +
                      [os.path.join(CSRC, f) for f in ("fa_step_pipe.hip", "fa_step_classic.hip", "fa_collect.hip", "fa_policy.hip", "fa_attend.hip", "fa_train.hip", "fa_train_dw.hip", "fa_fold.hip", "fa_rccl.hip", "fa_api.hip")] + ["-o", out])
print(out)
