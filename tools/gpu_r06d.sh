#!/bin/bash
set -x
mkdir -p gpurun_out/r06d
python -m pytest tests/test_gpu_policy.py -x -q -s -m gpu -k "dw_gemm" > gpurun_out/r06d/tests.log 2>&1
tail -5 gpurun_out/r06d/tests.log
grep "f32.*bf16x3" gpurun_out/r06d/tests.log
python tools/ab_train.py --no-tests product > gpurun_out/r06d/ab_bf16x3.log 2>&1
tail -1 gpurun_out/r06d/ab_bf16x3.log
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/r06d/prof -o p -- python $GRAFT_REPO_ROOT/tools/prof_grad.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/r06d/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -8 {} | cut -c1-220'
