#!/bin/bash
# Experiment (GPU box): SQ counters and HBM traffic of fa_ppo_grad's kernels (tools/prof_grad.py: eager calls, each launch
# has the GPU to itself), one --pmc pass per group (kernel-trace only).  usage: bash tools/pmc_grad.sh [G A]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_grad
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -f csv -d $OUT/g$i -o g -- python $R/tools/prof_grad.py $1 $2 > /dev/null 2> $OUT/g$i.err
done
cd $R
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_grad/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fa_train" in r["Kernel_Name"]:
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k[:80])
    for n, v in sorted(d.items()):
        print("   %-32s %16.1f   (n=%d)" % (n, sum(v) / len(v), len(v)))
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        f_, w_ = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]), sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
        print("   HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 = %.1f MB" % ((2 * f_ + w_) * 1024 / 1e6))
PY
