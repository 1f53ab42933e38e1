#!/usr/bin/env python
"""Experiment helper: compile fa_step_pipe.hip to gfx950 assembly and print the instruction histogram of
the pipelined kernel's wave-0 loop (a lone wave issues one instruction of any kind per 4 cycles, so
the instruction count of this loop IS the step-to-step chain).  usage: isa_w0.py [G A COLLECT NPW MINW] [--dump]"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
G, A, COL, NPW, MINW = (args + ["3", "3", "1", "2", "2"][len(args):])[:5]
asm = "/tmp/fa_step.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I",
                       os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
                       os.path.join(ROOT, "emergent-multiagent-strategies_amd/csrc/fa_step_pipe.hip"), "-o", asm],
                      stderr=subprocess.DEVNULL)
name = "_Z19fa_step_pipe_kernelILi%sELi%sELb%sELi%sELi%sEEv10FaStepArgs" % (G, A, COL, NPW, MINW)
lines = open(asm).read().split("\n")
start = next(k for k, l in enumerate(lines) if l.startswith(name + ":"))
end = next(k for k in range(start, len(lines)) if "s_endpgm" in lines[k])
body = lines[start:end]
# a loop = its header block + every block the assembler annotates "in Loop: Header=<header>"
heads = [k for k, l in enumerate(body) if "Loop Header" in l]
best = None
for h in heads:
    tag = "Header=" + body[h].split(":")[0].lstrip(".L")
    last_blk = max([k for k, l in enumerate(body) if tag in l and re.match(r"^\.LBB", l)] + [h])
    endk = next((k for k in range(last_blk + 1, len(body)) if re.match(r"^\.LBB", body[k])), len(body))
    seg = body[h:endk]
    text = "\n".join(seg)
    if "--all" in sys.argv:
        ins = [l.split()[0] for l in seg if l.startswith("\t") and not l.strip().startswith(";") and not l.strip().startswith(".")]
        hh = collections.Counter(re.sub(r"_e(32|64)$", "", x) for x in ins)
        role = "wave 0" if ("ds_write2st64_b32" in text and "v_rcp_f64" in text) else "pair/output waves" if "global_store" in text else "last wave"
        print("loop @%s (%s): %d instructions; v_readlane %d, s_nop %d, global_store %d, ds %d" % (
            body[h].split(":")[0], role, len(ins), hh["v_readlane_b32"], hh["s_nop"],
            sum(v for k, v in hh.items() if k.startswith("global_store")), sum(v for k, v in hh.items() if k.startswith("ds_"))))
    if "ds_write2st64_b32" in text and "v_rcp_f64" in text:   # wave 0: restages actions and integrates
        best = seg
assert best is not None
ins = [l.split()[0] for l in best if l.startswith("\t") and not l.strip().startswith(";") and not l.strip().startswith(".")]
hist = collections.Counter(re.sub(r"_e(32|64)$", "", x) for x in ins)
if "--dump" in sys.argv:
    print("\n".join(best))
else:
    for k, v in hist.most_common(45):
        print("%5d %s" % (v, k))
    print("%5d total (incl. the 1-in-16 restage path: 16 global loads + their address arithmetic)" % len(ins))
    cls = collections.Counter()
    for k, v in hist.items():
        cls["valu" if k.startswith("v_") else "salu" if k.startswith("s_") else "lds" if k.startswith("ds_") else "vmem"] += v
    print(dict(cls))
