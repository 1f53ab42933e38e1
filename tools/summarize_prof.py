#!/usr/bin/env python
"""Condense a tools/profile_gpu.sh output directory (gpurun_out/prof_<tag>/) into the
tracked summaries under profiles/:
    profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats table (verbatim)
    profiles/<tag>_summary.json       per-kernel avg duration + HBM bytes per launch
HBM bytes follow MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE
come from separate --pmc passes, are in KiB, and on gfx950 FETCH_SIZE reports 1/2 of a
coalesced streaming read, so bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def counter_avg(path, name):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(path, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    tag = sys.argv[1]
    src = os.path.join("gpurun_out", "prof_" + tag)
    os.makedirs("profiles", exist_ok=True)
    stats = glob.glob(os.path.join(src, "trace", "*kernel_stats.csv"))[0]
    shutil.copy(stats, os.path.join("profiles", tag + "_kernel_stats.csv"))
    fetch = counter_avg(os.path.join(src, "pmc_fetch"), "FETCH_SIZE")
    write = counter_avg(os.path.join(src, "pmc_write"), "WRITE_SIZE")
    out = {"tag": tag, "source": src, "kernels": {},
           "hbm_bytes_formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024  [KiB counters, gfx950 FETCH_SIZE x2 correction]"}
    for r in csv.DictReader(open(stats)):
        k = r["Name"]
        if not k.startswith(("void fa_", "fa_")):
            continue
        ent = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
               "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3,
               "pct_of_gpu_time": float(r["Percentage"])}
        if k in fetch and k in write:
            ent.update(FETCH_SIZE_KiB=fetch[k], WRITE_SIZE_KiB=write[k],
                       hbm_bytes_per_launch=(2 * fetch[k] + write[k]) * 1024)
        out["kernels"][k] = ent
    for name in ("bench_trace.json", "bench_fetch.json", "bench_write.json"):
        p = os.path.join(src, name)
        if os.path.isfile(p) and os.path.getsize(p):
            try:
                out[name[:-5]] = json.loads(open(p).read().strip().splitlines()[-1])
            except Exception:
                pass
    with open(os.path.join("profiles", tag + "_summary.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    for k, e in out["kernels"].items():
        print("%-70s calls=%4d avg=%9.2f us  hbm/launch=%s" % (
            k[:70], e["calls"], e["avg_us"], "%.1f MB" % (e["hbm_bytes_per_launch"] / 1e6)
            if "hbm_bytes_per_launch" in e else "-"))


if __name__ == "__main__":
    main()
