#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04j}
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests/test_gpu_collector.py tests/test_gpu_policy.py tests/test_gpu_learner.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 600 python bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-closed-loop --no-esweep --no-5v5 > $O/bench_head.json 2> $O/bench.err; python - <<PY
import json
r = json.loads(open("$O/bench_head.json").read().strip().splitlines()[-1])
print("value %.4e  ms_per_step %.4f  launch %.1f us" % (r["value"], r["ms_per_step"], r["roofline"]["avg_launch_us"]))
PY
cd /tmp; rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o h -- python $R/bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-closed-loop --no-esweep --no-5v5 > /dev/null 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); head -8 "$f" | cut -d, -f1-4 | cut -c1-150
