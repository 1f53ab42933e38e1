#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel stats of the closed-loop rollout (fused policy kernel + step kernel)
# -> gpurun_out/<tag>/closed_trace; closed-loop bench lines for 3v3, 5v5 and the 5v5 ensemble.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-closed}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d $O/closed_trace -o trace -- python $R/bench_rollout_mpnn.py --iters 3 --update 0 > $O/closed_3v3_prof.json 2> $O/closed.err
cd $R
python bench_rollout_mpnn.py --iters 3 > $O/closed_3v3.json 2>> $O/closed.err
python bench_rollout_mpnn.py --iters 3 --guards 5 --attackers 5 > $O/closed_5v5.json 2>> $O/closed.err
python bench_rollout_mpnn.py --iters 3 --guards 5 --attackers 5 --ensemble 5 > $O/closed_5v5_ens5.json 2>> $O/closed.err
python bench_rollout_mpnn.py --iters 2 --guards 5 --attackers 5 --ensemble 5 --backend torch --update 0 > $O/closed_5v5_ens5_torch.json 2>> $O/closed.err
python bench_rollout_mpnn.py --iters 2 --backend torch --update 0 > $O/closed_3v3_torch.json 2>> $O/closed.err
cat $O/closed_*.json
f=$(find $O/closed_trace -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-180
