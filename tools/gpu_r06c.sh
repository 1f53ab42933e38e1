#!/bin/bash
# round 6, call C: the split-bf16 weight-gradient GEMM -- accuracy against an fp64 GEMM, every gradient test, timing A/B
set -x
mkdir -p gpurun_out/r06c
python -m pytest tests/test_gpu_policy.py tests/test_gpu_ppo_golden.py tests/test_gpu_learner.py -x -q -s -m gpu -k "dw_gemm or ppo or grad or fused_update or one_graph or fold" > gpurun_out/r06c/tests.log 2>&1
tail -12 gpurun_out/r06c/tests.log
grep "f32.*bf16x3" gpurun_out/r06c/tests.log
FA_DW_GEMM=f32 python tools/ab_train.py --no-tests product > gpurun_out/r06c/ab_f32.log 2>&1
python tools/ab_train.py --no-tests product > gpurun_out/r06c/ab_bf16x3.log 2>&1
tail -1 gpurun_out/r06c/ab_f32.log; tail -1 gpurun_out/r06c/ab_bf16x3.log
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r06c/prof -o p -- python $GRAFT_REPO_ROOT/tools/ab_train.py --no-tests product > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/r06c/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -12 {} | cut -c1-200'
