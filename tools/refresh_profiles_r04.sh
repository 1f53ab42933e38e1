#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: everything profiles/r04_* records.  Raw output -> gpurun_out/;
# tools/summarize_prof.py + the copy step are run back in the build container.
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
HEAD="--steps 200 --warmup 5 --no-cpu-baseline --no-closed-loop --no-esweep --no-5v5"
bash tools/profile_gpu.sh ${TAG}_fused "$HEAD"
bash tools/profile_gpu.sh ${TAG}_5v5_fused "$HEAD --guards 5 --attackers 5"
bash tools/pmc_grad.sh 3 3 > gpurun_out/${TAG}_grad_sq_counters_3v3.txt 2>&1
bash tools/prof_grad.sh ${TAG} > gpurun_out/${TAG}_grad_kernel_stats.txt 2>&1
bash tools/prof_policy.sh ${TAG}_closed > gpurun_out/${TAG}_closed.txt 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_final.json 2> gpurun_out/bench_final.err
python bench.py --gpus 2 --share-devices --backend gloo --no-cpu-baseline --no-5v5 --no-esweep > gpurun_out/${TAG}_bench_2ranks_shared_gpu.json 2> gpurun_out/bench_2r.err
python bench.py --gpus 8 --share-devices --backend gloo --envs 512 --no-cpu-baseline --closed-loop-rollouts 5 --closed-loop-updates 1 > gpurun_out/${TAG}_bench_8ranks_rehearsal.json 2> gpurun_out/bench_8r.err
python tools/ab_update.py 3 > gpurun_out/${TAG}_update_3v3.jsonl 2>/dev/null
python train_fortattack_amd.py --num-guards 3 --num-attackers 3 --num-processes 4096 --num-steps 128 --num-frames 41943040 --save-dir /tmp/fa_r04 > gpurun_out/${TAG}_train_curve_3v3.jsonl 2> gpurun_out/train_curve.err
tail -c 600 gpurun_out/${TAG}_bench_final.json | head -c 300; echo; tail -1 gpurun_out/${TAG}_train_curve_3v3.jsonl | cut -c1-200; tail -2 gpurun_out/bench_8r.err
