#!/bin/bash
# Experiment (run on the GPU box): SQ / SQC counters of the fused step kernel, one --pmc pass per group.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_probe
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -f csv -d $OUT/g$i -o g -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-closed-loop --no-collector --min-seconds 0 ${PROBE_ARGS:-} > /dev/null 2> $OUT/g$i.err
done
cd $R
python - <<'PY'
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_probe/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fa_step" in r["Kernel_Name"]:
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in acc.items():
    print(k[:70])
    avg = {n: sum(v) / len(v) for n, v in d.items()}
    for n, v in sorted(avg.items()):
        print("   %-32s %16.1f" % (n, v))
    out[k] = avg
# per workgroup per env-step for the headline launch (410 workgroups x 128 steps), if present
E, N, T = 4096, 6, 128
wgs = (E + (64 // N) - 1) // (64 // N)
for k, avg in out.items():
    if "pipe" in k and "<3, 3, true" in k:
        rec = {"config": "3v3, E=4096, 128 env-steps per launch, %d workgroups x 4 waves (wave 0 + 3 helper waves)" % wgs,
               "kernel": k,
               "note": "sums over the 4 waves of a workgroup, per env-step of the workgroup (10 envs); SQ_WAVE_CYCLES / SQ_WAIT_* / "
                       "SQ_ACTIVE_INST_* / SQ_BUSY_CYCLES count quad-cycles (x4 for shader cycles); one --pmc pass per group of 4",
               "per_workgroup_per_env_step": {n: round(v / (wgs * T), 2) for n, v in sorted(avg.items())},
               "per_launch": {n: v for n, v in sorted(avg.items())}}
        json.dump(rec, open("gpurun_out/pmc_probe/sq_counters.json", "w"), indent=1)
PY
