#!/bin/bash
# rocprofv3 kernel stats of eager fa_ppo_grad calls (tools/prof_grad.py) and of the whole update (two chains).
# usage (GPU box, repo root): bash tools/prof_grad.sh <tag>
tag=${1:-grad}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
show() { f=$(find $1 -name "*kernel_stats.csv" | head -1); python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]:
    print("%-84s calls %6s avg %9.1f us  total %8.2f ms  %5.1f%%" % (r["Name"][:84], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
PY
}
for cfg in "3 3" "3 3 share" "5 5"; do
  d=$out/g_$(echo $cfg | tr ' ' '_')
  rocprofv3 --kernel-trace --stats -f csv -d $d -o g -- python $root/tools/prof_grad.py $cfg > $d.log 2>&1
  tail -1 $d.log; show $d
done
rocprofv3 --kernel-trace --stats -f csv -d $out/upd -o upd -- python $root/bench_rollout_mpnn.py --iters 2 > $out/upd.log 2>&1
tail -1 $out/upd.log; show $out/upd
