#!/bin/bash
# round 6, call A: sharded-config parity + the one-call several-rank tail (tests, then the bench line in four forms)
set -x
mkdir -p gpurun_out/r06a
python -m pytest tests/test_gpu_sharded_configs.py tests/test_gpu_rccl.py -x -q -s -m gpu > gpurun_out/r06a/tests.log 2>&1
tail -15 gpurun_out/r06a/tests.log
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-5v5 --no-esweep --no-closed-loop --no-live-traffic"
$B > gpurun_out/r06a/bench_plain.json 2> gpurun_out/r06a/bench_plain.err
$B --force-collective --exchange torch > gpurun_out/r06a/bench_fc_torch.json 2> gpurun_out/r06a/bench_fc_torch.err
$B --force-collective > gpurun_out/r06a/bench_fc_library.json 2> gpurun_out/r06a/bench_fc_library.err
$B --force-collective --graph-hot-path > gpurun_out/r06a/bench_fc_graph.json 2> gpurun_out/r06a/bench_fc_graph.err
$B --graph-hot-path > gpurun_out/r06a/bench_plain_graph.json 2> gpurun_out/r06a/bench_plain_graph.err
for f in plain fc_torch fc_library fc_graph plain_graph; do
  python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/r06a/bench_$f.json") if l.startswith("{")][-1])
    print("$f", "ms_per_step", round(d["ms_per_step"],5), "host_enqueue", round(d["host_enqueue_ms_per_step"],5), "steady", d["steady_state"] and round(d["steady_state"]["ms_per_step"],5), "launch_us", round(d["roofline"]["avg_launch_us"],2), d["collective"].get("library_exchange_equals_torch_route"), d["collective"]["rank_binding"][0].get("pin"))
except Exception as e:
    print("$f", "FAILED", e); print(open("gpurun_out/r06a/bench_$f.err").read()[-1500:])
PY
done
