#!/bin/bash
# Run ON THE GPU BOX via gpurun: GPU tests, bench, timing probe and a kernel trace of the closed loop.
# usage: tools/gpu_round.sh <tag> [what...]   what = tests bench probe closed
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-run}; shift
WHAT=${@:-tests bench probe closed}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd $R
for w in $WHAT; do
case $w in
tests) timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -15 $O/pytest.log;;
bench) timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-600 $O/bench.json; tail -3 $O/bench.err;;
probe) timeout 300 python tools/timing_probe.py > $O/timing_probe.txt 2>&1; cat $O/timing_probe.txt | head -40;;
closed) cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $O/closed_trace -o trace -- python $R/bench_rollout_mpnn.py --iters 2 --update 0 > $O/closed.json 2> $O/closed.err; cd $R;
   cat $O/closed.json; f=$(find $O/closed_trace -name "*kernel_stats.csv" | head -1); head -45 "$f" | cut -c1-200;;
esac
done
