"""Compile the HIP engine for gfx950 (in-tree, so the .so travels with the repo)."""
import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(CSRC, "libfortattack_hip.so")
SOURCES = ["fa_step.hip", "fa_collect.hip", "fa_policy.hip", "fa_attend.hip", "fa_train.hip", "fa_fold.hip", "fa_api.hip"]
# -ffp-contract=off: the fp64 step must evaluate every operation as the reference does
# (no fused multiply-add); no -ffast-math for the same reason.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]


def hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def needs_build():
    if not os.path.isfile(LIB):
        return True
    deps = [os.path.join(CSRC, f) for f in SOURCES + ["fa_device.h", "fa_probe.h", "fa_policy.h", "fa_mfma.h", "fa_train.h"]]
    deps.append(os.path.join(ROOT, "include", "fortattack.h"))
    return any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    cmd = [hipcc()] + FLAGS + ["-I", os.path.join(ROOT, "include")] + \
        [os.path.join(CSRC, f) for f in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
