#!/bin/bash
# Experiment (GPU box): SQ counters and HBM traffic of the fused PPO kernel (fa_train_kernel / fa_train_share_kernel),
# one --pmc pass per group (kernel-trace only).  The update runs with the teams one after the other
# (bench_rollout_mpnn.py --sequential-teams) so that a launch has the GPU to itself.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_train
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -f csv -d $OUT/g$i -o g -- python $R/bench_rollout_mpnn.py --iters 1 --epochs 1 --sequential-teams > /dev/null 2> $OUT/g$i.err
done
cd $R
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_train/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fa_train_kernel" in r["Kernel_Name"] or "fa_train_share" in r["Kernel_Name"] or "fa_train_reduce" in r["Kernel_Name"]:
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k[:80])
    for n, v in sorted(d.items()):
        print("   %-32s %16.1f   (n=%d)" % (n, sum(v) / len(v), len(v)))
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        f_, w_ = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]), sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
        print("   HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 = %.1f MB" % ((2 * f_ + w_) * 1024 / 1e6))
PY
