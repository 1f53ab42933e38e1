#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: everything profiles/ records for a round.
# Raw output -> gpurun_out/; tools/summarize_prof.py + the copy step at the end are run back in the
# build container (see profiles/README.md).
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/profile_gpu.sh ${TAG}_fused "--steps 20 --warmup 3 --no-cpu-baseline --no-closed-loop"
bash tools/profile_gpu.sh ${TAG}_perstep "--steps 20 --warmup 3 --no-cpu-baseline --no-closed-loop --launch per-step"
bash tools/pmc_probe.sh > gpurun_out/pmc_probe.txt 2>&1
bash tools/pmc_policy.sh > gpurun_out/pmc_policy.txt 2>&1
bash tools/prof_policy.sh ${TAG}_closed > gpurun_out/${TAG}_closed.txt 2>&1
bash tools/prof_update.sh ${TAG}_update > gpurun_out/${TAG}_update.txt 2>&1
python tools/sweep.py > gpurun_out/${TAG}_esweep.jsonl 2> gpurun_out/esweep.err
python tools/perstep_probe.py > gpurun_out/${TAG}_perstep_probe.jsonl 2> gpurun_out/perstep_probe.err
python bench.py > gpurun_out/${TAG}_bench_fused.json 2> gpurun_out/bench_fused.err
python bench.py --launch per-step --no-cpu-baseline --no-closed-loop > gpurun_out/${TAG}_bench_perstep.json 2> gpurun_out/bench_perstep.err
python bench.py --guards 5 --attackers 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_5v5_fused.json 2> gpurun_out/bench_5v5.err
python bench.py --gpus 2 --share-devices --backend gloo --no-cpu-baseline > gpurun_out/${TAG}_bench_2ranks_shared_gpu.json 2> gpurun_out/bench_2r.err
cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc
tail -2 gpurun_out/pmc_probe.txt; cut -c1-300 gpurun_out/${TAG}_bench_fused.json
