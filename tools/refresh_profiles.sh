#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: everything profiles/ records for a round.
# Raw output -> gpurun_out/; tools/summarize_prof.py + the copy step at the end are run back in the
# build container (see profiles/README.md).
TAG=${1:-r01}
bash tools/profile_gpu.sh ${TAG}_fused "--steps 20 --warmup 3 --no-cpu-baseline"
bash tools/profile_gpu.sh ${TAG}_perstep "--steps 20 --warmup 3 --no-cpu-baseline --launch per-step"
bash tools/pmc_probe.sh > gpurun_out/pmc_probe.txt 2>&1
python tools/sweep.py > gpurun_out/${TAG}_esweep.jsonl 2> gpurun_out/esweep.err
python bench.py > gpurun_out/${TAG}_bench_fused.json 2> gpurun_out/bench_fused.err
python bench.py --launch per-step --no-cpu-baseline > gpurun_out/${TAG}_bench_perstep.json 2> gpurun_out/bench_perstep.err
python bench.py --guards 5 --attackers 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_5v5_fused.json 2> gpurun_out/bench_5v5.err
tail -2 gpurun_out/pmc_probe.txt; cat gpurun_out/${TAG}_bench_fused.json | cut -c1-300
