#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05c
mkdir -p $O
export TMPDIR=/tmp
cd $R
python tools/make_timing_build.py > $O/tb.log 2>&1
FA_TIMING_LIB=libfa_timing_w1.so python tools/make_timing_build.py -DFA_TICK_WAVE1=1 >> $O/tb.log 2>&1
for lib in libfa_timing.so libfa_timing_w1.so; do
  FA_TIMING_LIB=$lib timeout 300 python tools/timing_probe.py 2>&1 | grep -v "^  wg\|amdgpu" > $O/probe_$lib.txt
  head -24 $O/probe_$lib.txt
done
timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r05c/bench.json").read().strip().splitlines()[-1])
print("value %.4e ms/step %.5f frac %.4f launch_us %.2f" % (r["value"], r["ms_per_step"], r["roofline"]["frac"], r["roofline"]["avg_launch_us"]))
for k in ("closed_loop", "closed_loop_5v5", "closed_loop_5v5_ens5"):
    c = r.get(k)
    if c: print(k, "ms/env-step %.4f update_s %.4f policy frac %.3f train frac %.3f" % (c["ms_per_env_step_launch"], c["update_s"], c["roofline"]["frac"], c["update_roofline"]["frac"]))
print("fused_5v5", r["fused_5v5"]["ms_per_step"], r["fused_5v5"]["roofline"]["frac"])
print("facade", r["single_env_facade"]["us_per_step"])
PY
