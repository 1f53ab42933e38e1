#!/bin/bash
# rocprofv3 kernel stats of the closed loop (rollout + PPO update): top kernels by total time.
# usage (on the GPU box, from the repo root): bash tools/prof_update.sh <tag>
tag=${1:-upd}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $out -o upd -- python $root/bench_rollout_mpnn.py --iters 2 > $out/run.log 2>&1
cd $root
f=$(ls $out/*kernel_stats.csv $out/*/*kernel_stats.csv 2>/dev/null | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot / 1e6)
for r in rows[:28]:
    print("%-90s calls %6s avg %9.1f us  total %8.2f ms  %5.1f%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
PY
tail -2 $out/run.log
