#!/usr/bin/env python
"""Experiment helper: build tools/_build/lib_<name>.so = the product objects (csrc/_obj, from the last
product build) with ONE translation unit recompiled with extra flags / another source file.
usage: build_variant.py <name> <tu.hip> [--src other_source.hip] [--also other_tu.hip ...] [extra hipcc flags ...]
       build_variant.py <name> --add experiments/fa_step_experiments.hip [extra hipcc flags ...]
         (the product objects PLUS one more translation unit: the library with the round-4 experiment step kernels)
A/B the results on the GPU box with tools/ab_step.py / ab_policy.py / ab_train.py."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "emergent-multiagent-strategies_amd", "csrc")
OBJ = os.path.join(CSRC, "_obj")
SOURCES = ["fa_step_pipe.hip", "fa_step_classic.hip", "fa_collect.hip", "fa_policy.hip", "fa_attend.hip", "fa_train.hip", "fa_train_dw.hip", "fa_fold.hip", "fa_rccl.hip", "fa_api.hip"]
import shutil
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
name, tu = sys.argv[1], sys.argv[2]
extra = sys.argv[3:]
add = tu == "--add"
if add:
    tu, extra = extra[0], extra[1:]
src = os.path.join(CSRC, tu)
also = []
while "--also" in extra:          # further translation units recompiled with the same flags
    k = extra.index("--also")
    also.append(extra[k + 1])
    extra = extra[:k] + extra[k + 2:]
if "--src" in extra:
    k = extra.index("--src")
    src = os.path.abspath(extra[k + 1])
    extra = extra[:k] + extra[k + 2:]
bdir = os.path.join(ROOT, "tools", "_build")
os.makedirs(bdir, exist_ok=True)
obj = os.path.join(bdir, "%s_%s.o" % (name, os.path.splitext(os.path.basename(tu))[0]))
common = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
          "-I", os.path.join(ROOT, "include"), "-I", CSRC] + extra
subprocess.check_call(common + ["-c", src, "-o", obj])
if "--asm" in os.environ.get("FA_VARIANT", ""):
    subprocess.check_call(common + ["-S", "--cuda-device-only", src, "-o", obj[:-2] + ".s"])
repl = {tu: obj}
for t2 in also:
    o2 = os.path.join(bdir, "%s_%s.o" % (name, os.path.splitext(os.path.basename(t2))[0]))
    subprocess.check_call(common + ["-c", os.path.join(CSRC, t2), "-o", o2])
    repl[t2] = o2
objs = [repl.get(s, os.path.join(OBJ, os.path.splitext(s)[0] + ".o")) for s in SOURCES] + ([obj] if add else [])
out = os.path.join(bdir, "lib_%s.so" % name)
subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + ["-o", out])
for o in repl.values():
    os.remove(o)
print(out)
