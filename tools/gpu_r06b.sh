#!/bin/bash
# round 6, call B: sharded-config + RCCL tests again; the chunked tile -> dW hand-off (FA_PPO_CHUNKS) A/B, with and without
# the non-temporal hint on the records
set -x
mkdir -p gpurun_out/r06b
python -m pytest tests/test_gpu_sharded_configs.py tests/test_gpu_rccl.py -x -q -s -m gpu > gpurun_out/r06b/tests.log 2>&1
tail -8 gpurun_out/r06b/tests.log
for C in 1 2 4 8; do
  for lib in product recplain; do
    echo "== chunks $C lib $lib" >> gpurun_out/r06b/chunks.log
    FA_PPO_CHUNKS=$C python tools/ab_train.py --no-tests $lib >> gpurun_out/r06b/chunks.log 2>> gpurun_out/r06b/chunks.err
  done
done
cat gpurun_out/r06b/chunks.log
FA_PPO_CHUNKS=4 python -m pytest tests/test_gpu_ppo_golden.py tests/test_gpu_policy.py tests/test_gpu_learner.py -x -q -m gpu -k "ppo or grad or fused_update" > gpurun_out/r06b/tests_chunks4.log 2>&1
tail -5 gpurun_out/r06b/tests_chunks4.log
