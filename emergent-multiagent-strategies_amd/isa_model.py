#!/usr/bin/env python
"""Issue model of the pipelined step kernel, read off the gfx950 assembly of the build (csrc/_obj/fa_step_pipe.s).

At BASELINE config 2's batch every wave of fa_step_pipe_kernel is (nearly) alone on its SIMD and a rollout is one
dependent chain of env-steps: a wave issues at most one instruction per 4 cycles (a wave64 VALU instruction occupies the
16-lane SIMD for 4 cycles; fp64 add / mul / fma run at full rate on gfx950), so the INSTRUCTION COUNT of a wave's step loop
x 4 cycles is the floor of its share of a step, and the step is two barrier-separated phases whose length is the slowest
wave's.  build.py calls `model()` after compiling, writes csrc/fa_isa_model.json next to the library (bench.py attaches it
to the JSON line as roofline.secondary, with the measured cycles per step beside it) and fails the build when a shipped
instantiation regresses (`check`): a FLAT memory instruction in the kernel, a spill reload in wave 0's or the walls wave's
step loop, more instructions in wave 0's loop than the recorded ceiling.

usage: isa_model.py [fa_step_pipe.s]            (prints the model as JSON)
"""
import collections
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ASM = os.path.join(HERE, "csrc", "_obj", "fa_step_pipe.s")
OUT = os.path.join(HERE, "csrc", "fa_isa_model.json")
# the instantiations AUTO launches at the bench's shapes: (G, A, COLLECT, NPW, MINW)
SHIPPED = {"3v3": (3, 3, 1, 2, 2), "3v3_3percu": (3, 3, 1, 2, 3), "5v5_3percu": (5, 5, 1, 2, 3), "5v5": (5, 5, 1, 2, 2)}
FP64 = re.compile(r"^v_(add|mul|fma|fmac|rcp|rsq|sqrt|div_fmas|div_fixup|div_scale|cmp\w*|cvt\w*|rndne|trunc|floor|fract|ldexp|frexp\w*|max|min)_?\w*f64")
W0_LOOP_MAX_INSTRS = {"3v3": 500, "5v5_3percu": 640}   # ceilings for check(): wave 0's step loop (491 / 626 when recorded)


def _ops(seg):
    return [l.split()[0] for l in seg if l.startswith("\t") and not l.strip().startswith((";", "."))]


def _classes(ops):
    c = collections.Counter()
    for op in ops:
        base = re.sub(r"_e(32|64)$", "", op)
        if base.startswith("v_"):
            c["valu"] += 1
            if FP64.match(base):
                c["valu_fp64"] += 1
        elif base.startswith("s_"):
            c["salu"] += 1
        elif base.startswith("ds_"):
            c["lds"] += 1
        elif base.startswith(("global_", "buffer_", "flat_", "scratch_")):
            c["vmem"] += 1
        if base == "v_readlane_b32":
            c["spill_reloads"] += 1
        if base == "s_barrier":
            c["barriers"] += 1
        if base.startswith("flat_"):
            c["flat"] += 1
    return dict(c)


def kernel_body(lines, sym):
    start = next(k for k, l in enumerate(lines) if l.startswith(sym + ":"))
    end = next(k for k in range(start, len(lines)) if "s_endpgm" in lines[k])
    return lines[start:end]


def loops(body):
    """[(role, segment)] of the kernel's step loops: header block + every block annotated `in Loop: Header=<it>`."""
    out = []
    for h in [k for k, l in enumerate(body) if "Loop Header" in l]:
        tag = "Header=" + body[h].split(":")[0].lstrip(".L")
        last = max([k for k, l in enumerate(body) if tag in l and re.match(r"^\.LBB", l)] + [h])
        endk = next((k for k in range(last + 1, len(body)) if re.match(r"^\.LBB", body[k])), len(body))
        seg = body[h:endk]
        text = "\n".join(seg)
        if "ds_write2st64_b32" in text and "v_rcp_f64" in text:
            role = "wave0"                       # restages actions, integrates (division in the speed clamp)
        elif "global_store" in text and ("v_rsq_f64" in text or "global_store_dwordx" in text) and text.count("s_barrier") >= 2 and "v_cvt_f32_f64" in text:
            role = "pairs"                       # pair forces, rewards / observation rows
        else:
            role = "walls"                       # decode, walls, next heading, done / mask rows
        out.append((role, seg))
    return out


def model(asm=ASM):
    lines = open(asm).read().split("\n")
    text = "\n".join(lines)
    res = {"source": os.path.relpath(asm, HERE), "cycles_per_instruction": 4,
           "note": "static instruction counts of each wave role's step loop; model_cycles = 4 x instructions of the every-step path "
                   "(wave 0: the loop minus 15/16 of the 1-in-16 action restage path); a step is two barrier-separated phases, so "
                   "the measured cycles per step also hold two barrier round trips and the LDS round trips behind them"}
    for tag, (G, A, col, npw, minw) in SHIPPED.items():
        sym = "_Z19fa_step_pipe_kernelILi%dELi%dELb%dELi%dELi%dEEv10FaStepArgs" % (G, A, col, npw, minw)
        try:
            body = kernel_body(lines, sym)
        except StopIteration:
            continue
        m = re.search(r"\.amdhsa_kernel " + re.escape(sym) + r"(.*?)\.end_amdhsa_kernel", text, re.S)
        meta = m.group(1) if m else ""
        spills = re.search(r"\.sgpr_spill_count:\s+(\d+)\n\s+\.symbol:\s+" + re.escape(sym), text)
        vspills = re.search(r"\.symbol:\s+" + re.escape(sym) + r"\.kd\n(?:.*\n){0,6}?\s+\.vgpr_spill_count:\s+(\d+)", text)
        k = {"symbol": sym, "vgprs": int(re.search(r"next_free_vgpr (\d+)", meta).group(1)) if meta else None,
             "scratch_bytes": int(re.search(r"private_segment_fixed_size (\d+)", meta).group(1)) if meta else None,
             "sgpr_spill_count": int(spills.group(1)) if spills else None,
             "vgpr_spill_count": int(vspills.group(1)) if vspills else None,
             "kernel_totals": _classes(_ops(body)), "loops": {}}
        k["kernel_totals"]["instructions"] = len(_ops(body))
        seen = collections.Counter()
        for role, seg in loops(body):
            seen[role] += 1
            name = role if role != "pairs" or seen[role] == 1 else "%s_%d" % (role, seen[role])
            ops = _ops(seg)
            rec = {"instructions": len(ops)}
            rec.update(_classes(ops))
            if role == "wave0":
                # the restage path: the blocks that request the next 16 steps' actions (global loads + their address arithmetic
                # + the 8 ds_write2st64 that hand the batch over); executed once per 16 steps
                blocks, cur = [], []
                for l in seg:
                    if re.match(r"^\.LBB", l) or l.startswith("; %bb."):
                        blocks.append(cur)
                        cur = []
                    cur.append(l)
                blocks.append(cur)
                restage = sum(len(_ops(b)) for b in blocks if any("global_load" in l or "ds_write2st64_b32" in l for l in b))
                rec["restage_path_instructions"] = restage
                rec["instructions_per_step"] = round(len(ops) - restage + restage / 16.0, 1)
                rec["fp64_instructions_per_step"] = rec.get("valu_fp64", 0)
                rec["model_cycles_per_step"] = round(4 * rec["instructions_per_step"], 1)
            k["loops"][name] = rec
        res[tag] = k
    return res


def check(m):
    """Regressions that fail the build (next to isa_lint)."""
    bad = []
    for tag, k in m.items():
        if not isinstance(k, dict) or "loops" not in k:
            continue
        if k["kernel_totals"].get("flat", 0):
            bad.append("%s: %d FLAT memory instruction(s) (row stores must name the global address space: fa_gstore)" % (tag, k["kernel_totals"]["flat"]))
        for role, rec in k["loops"].items():
            if role in ("wave0", "walls") and rec.get("spill_reloads", 0):
                bad.append("%s: %d v_readlane (spill reloads) in the %s step loop" % (tag, rec["spill_reloads"], role))
        w0 = k["loops"].get("wave0")
        if w0 and tag in W0_LOOP_MAX_INSTRS and w0["instructions"] > W0_LOOP_MAX_INSTRS[tag]:
            bad.append("%s: wave 0's step loop has %d instructions (ceiling %d)" % (tag, w0["instructions"], W0_LOOP_MAX_INSTRS[tag]))
        if k.get("vgpr_spill_count"):
            bad.append("%s: %d VGPR spills" % (tag, k["vgpr_spill_count"]))
    return bad


def write(asm=ASM, out=OUT):
    m = model(asm)
    with open(out, "w") as f:
        json.dump(m, f, indent=1, sort_keys=True)
        f.write("\n")
    return m


if __name__ == "__main__":
    if "--refresh" in sys.argv:          # rewrite the tracked csrc/fa_isa_model.json from the last build's assembly
        write()
        sys.argv.remove("--refresh")
    mm = model(sys.argv[1] if len(sys.argv) > 1 else ASM)
    print(json.dumps(mm, indent=1, sort_keys=True))
    for b in check(mm):
        print("REGRESSION:", b, file=sys.stderr)
