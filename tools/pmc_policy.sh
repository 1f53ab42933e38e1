#!/bin/bash
# Experiment (GPU box): SQ counters of the fused policy kernel, one --pmc pass per group (kernel-trace only).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_policy
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -f csv -d $OUT/g$i -o g -- python $R/bench_rollout_mpnn.py --iters 1 --update 0 --graph 0 > /dev/null 2> $OUT/g$i.err
done
cd $R
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_policy/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fa_policy" in r["Kernel_Name"]:
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k[:70])
    for n, v in sorted(d.items()):
        print("   %-32s %16.1f   (n=%d)" % (n, sum(v) / len(v), len(v)))
PY
