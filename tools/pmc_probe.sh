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
  rocprofv3 --kernel-trace --pmc $grp -f csv -d $OUT/g$i -o g -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-collector ${PROBE_ARGS:-} > /dev/null 2> $OUT/g$i.err
done
cd $R
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_probe/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fa_step" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for n, v in sorted(d.items()):
        print("   %-32s %16.1f  (n=%d)" % (n, sum(v) / len(v), len(v)))
PY
