#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04c}
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 600 python tools/train_probe.py > $O/train_probe.txt 2>&1; cat $O/train_probe.txt
timeout 900 python tools/ab_train.py --no-tests product > $O/ab_train.txt 2>$O/ab_train.err; cat $O/ab_train.txt
timeout 1200 python -m pytest tests/test_gpu_policy.py tests/test_gpu_ppo_golden.py -m gpu -q -s -k "ppo_grad or fold or adam or golden or published" > $O/pytest_train.log 2>&1; echo "train tests rc=$?"; grep -E "passed|failed|FAILED|Error|within" $O/pytest_train.log | tail -12
