#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: `bash tools/refresh_profiles.sh r06` = everything profiles/<tag>_* records -- headline + 5v5 rocprofv3 stats / PMC
# passes, the step-kernel builds, the driver's bench command, the update's kernel stats, the closed-loop kernel trace, 2- and
# 8-rank shared-GPU runs (+ the --smoke form), the one-rank forced RCCL collective, parity soaks, an 80-update training run.  Raw output -> gpurun_out/; tools/summarize_prof.py + the copy step run in the build container.
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
HEAD="--steps 200 --warmup 5 --no-cpu-baseline --no-closed-loop --no-esweep --no-5v5 --no-live-traffic"
bash tools/profile_gpu.sh ${TAG}_fused "$HEAD"
bash tools/profile_gpu.sh ${TAG}_5v5_fused "$HEAD --guards 5 --attackers 5"
bash tools/prof_grad.sh ${TAG} > gpurun_out/${TAG}_grad_kernel_stats.txt 2>&1
python tools/soak_parity.py > gpurun_out/${TAG}_soak_parity.jsonl 2>/dev/null
python tools/step_variants.py 4096 640 2>/dev/null | grep -v amdgpu > gpurun_out/${TAG}_step_variants.jsonl
bash tools/prof_policy.sh ${TAG}_closed > gpurun_out/${TAG}_closed.txt 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_final.json 2> gpurun_out/bench_final.err
python bench.py --gpus 2 --share-devices --backend gloo --no-cpu-baseline --no-5v5 --no-esweep > gpurun_out/${TAG}_bench_2ranks_shared_gpu.json 2> gpurun_out/bench_2r.err
python bench.py --gpus 8 --share-devices --backend gloo --envs 512 --no-cpu-baseline --closed-loop-rollouts 5 --closed-loop-updates 1 > gpurun_out/${TAG}_bench_8ranks_rehearsal.json 2> gpurun_out/bench_8r.err
python bench.py --gpus 2 --share-devices --backend gloo --smoke > gpurun_out/${TAG}_bench_2ranks_smoke.json 2> gpurun_out/bench_2s.err
python bench.py --gpus 1 --force-collective --steps 2000 --no-cpu-baseline --no-5v5 --no-esweep --no-closed-loop > gpurun_out/${TAG}_bench_force_collective.json 2> gpurun_out/bench_fc.err
python bench.py --gpus 1 --force-collective --graph-hot-path --steps 2000 --no-cpu-baseline --no-5v5 --no-esweep --no-closed-loop > gpurun_out/${TAG}_bench_force_collective_graph.json 2> gpurun_out/bench_fcg.err
python bench.py --gpus 1 --force-collective --exchange torch --steps 2000 --no-cpu-baseline --no-5v5 --no-esweep --no-closed-loop > gpurun_out/${TAG}_bench_force_collective_torch_route.json 2> gpurun_out/bench_fct.err
python tools/soak_closed_loop.py 80 3 3 > gpurun_out/${TAG}_soak_closed_loop.jsonl 2> gpurun_out/soak_cl.err
python tools/soak_closed_loop.py 20 5 5 >> gpurun_out/${TAG}_soak_closed_loop.jsonl 2>> gpurun_out/soak_cl.err
python train_fortattack_amd.py --num-guards 3 --num-attackers 3 --num-processes 4096 --num-steps 128 --num-frames 41943040 --save-dir /tmp/fa_${TAG} > gpurun_out/${TAG}_train_curve_3v3.jsonl 2> gpurun_out/train_curve.err
tail -c 600 gpurun_out/${TAG}_bench_final.json | head -c 300; echo; tail -1 gpurun_out/${TAG}_train_curve_3v3.jsonl | cut -c1-200; tail -2 gpurun_out/bench_8r.err; tail -2 gpurun_out/bench_2r.err; tail -c 300 gpurun_out/${TAG}_bench_force_collective.json
