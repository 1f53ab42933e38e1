#!/bin/bash
# Round 4, first GPU call: new golden tests first, soak in measure mode, full GPU suite, default bench (driver's flags).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04a
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_ppo_golden.py tests/test_gpu_policy.py -m gpu -q -x -s -k "golden or published" > $O/pytest_new.log 2>&1; echo "new tests rc=$?"; tail -12 $O/pytest_new.log
timeout 900 python tools/soak_closed_loop.py 80 3 3 measure > $O/soak80_measure.jsonl 2> $O/soak.err; echo "soak rc=$?"; cut -c1-900 $O/soak80_measure.jsonl; tail -3 $O/soak.err
timeout 600 python tools/soak_closed_loop.py 20 5 5 measure > $O/soak20_5v5_measure.jsonl 2>> $O/soak.err; echo "soak5 rc=$?"; cut -c1-900 $O/soak20_5v5_measure.jsonl
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-400 $O/bench.json; tail -3 $O/bench.err
timeout 2700 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
