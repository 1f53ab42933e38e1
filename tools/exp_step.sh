#!/bin/bash
# Run ON THE GPU BOX: quick loop for step-kernel experiments: parity of the shipped builds vs the oracle,
# then the fused / 5v5 / per-step launch times.  usage: tools/exp_step.sh <tag> [probe]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-exp}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_shipped_kernels.py tests/test_gpu_collector.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for cfg in "" "--guards 5 --attackers 5" "--launch per-step" "--envs 7680"; do
  timeout 300 python bench.py --no-cpu-baseline --no-closed-loop --steps 300 --min-seconds 0 $cfg 2> $O/bench.err | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-28s %s  launch %.2f us  frac %.3f  value %.3e' % ('$cfg', r['roofline']['kernel'], r['roofline']['avg_launch_us'], r['roofline']['frac'], r['value']))"
done
if [ "$2" = "probe" ]; then
  for lib in libfa_timing.so libfa_timing_w1.so; do
    FA_TIMING_LIB=$lib timeout 300 python tools/timing_probe.py 2>&1 | grep -v "^  wg\|amdgpu\|wave placement" | head -24
  done
fi
