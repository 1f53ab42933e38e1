#!/bin/bash
# round 5: full GPU suite + the bench line with live traffic / secondary / steady_state / facade
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05j
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r05j/bench.json").read().strip().splitlines()[-1])
ro = r["roofline"]
print("value %.4e ms/step %.5f frac %.4f launch_us %.2f" % (r["value"], r["ms_per_step"], ro["frac"], ro["avg_launch_us"]))
print("traffic", ro["traffic"], ro["traffic_source"])
print("secondary", json.dumps(ro["secondary"])[:900])
print("steady", r["steady_state"])
print("facade", r["single_env_facade"]["us_per_step"])
PY
timeout 3000 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
