#!/usr/bin/env python
"""Lint gfx950 assembly for a register-allocator hazard seen with hipcc 7.2 on register-starved kernels: a live-range
split copy or spill (v_accvgpr_write / v_accvgpr_read / v_mov / scratch_store / scratch_load) placed at the top of a
join block BEFORE the `s_or_b64 exec, exec, s[..]` that re-enables the lanes masked off by the divergent region.
Lanes (or whole waves) that skipped the region are then not copied, and a later read gets garbage -- in
fa_train_kernel that was the lane's slab address (a wild store).  usage: isa_lint.py file.s [kernel-substring]"""
import re
import sys

VEC = re.compile(r"^(v_|scratch_|global_|ds_|buffer_|flat_)")


def lint(path, only=None):
    lines = [raw.split(";")[0].strip() for raw in open(path)]
    skip_targets = set()
    for line in lines:                    # blocks reached by waves / lanes that skipped a divergent region
        m = re.match(r"s_cbranch_execz\s+(\S+)", line)
        if m:
            skip_targets.add(m.group(1))
    bad = []
    kernel, block, pre = None, None, None
    for ln, line in enumerate(lines, 1):
        if not line:
            continue
        m = re.match(r"^([A-Za-z_.$][\w.$]*):", line)
        if m:
            name = m.group(1)
            if not name.startswith(".L"):
                kernel = name
            block, pre = name, ([] if name in skip_targets else None)
            continue
        if pre is None or line.startswith("."):
            continue
        op = line.split()[0]
        if re.match(r"s_or_b64\s+exec,\s*exec,", line) or re.match(r"s_or_b64\s+exec,\s*s\[\d+:\d+\],\s*exec", line):
            if only is None or (kernel and only in kernel):
                bad += [(kernel, block, p) for p in pre]
            pre = None
        elif VEC.match(op) and op not in ("v_writelane_b32", "v_readlane_b32", "v_readfirstlane_b32"):   # (these ignore exec)
            pre.append((ln, line))
        elif op.startswith("s_cbranch") or op.startswith("s_branch") or op in ("s_endpgm", "s_barrier") or "exec" in line:
            pre = None      # the block's prologue is over
    return bad


if __name__ == "__main__":
    bad = lint(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
    for kernel, block, (ln, line) in bad:
        print("%s %s line %d: %s" % (kernel, block, ln, line))
    print("%d vector instruction(s) ahead of an exec restore" % len(bad))
    sys.exit(1 if bad else 0)
