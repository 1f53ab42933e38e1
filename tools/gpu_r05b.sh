#!/bin/bash
# round 5: helper-wave spill / flat-store changes of fa_step_pipe_kernel -- A/B against the round-4 library, parity, phase probe
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05b
mkdir -p $O
export TMPDIR=/tmp
cd $R
python tools/ab_step.py product r04 product r04 > $O/ab_step.jsonl 2> $O/ab_step.err
cat $O/ab_step.jsonl
timeout 900 python -m pytest tests/test_gpu_shipped_kernels.py tests/test_gpu_experiment_kernels.py tests/test_gpu_parity.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
python tools/make_timing_build.py > $O/timing_build.log 2>&1 && python tools/timing_probe.py 4096 > $O/timing_probe_3v3.txt 2>&1; tail -32 $O/timing_probe_3v3.txt
