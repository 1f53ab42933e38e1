#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: rocprofv3 kernel-trace stats of the
# bench command, then the two HBM PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a
# pass on gfx950: TCC has 4 slots, FETCH_SIZE takes 3, WRITE_SIZE 2).  Raw output goes to
# gpurun_out/prof_<tag>/; tools/summarize_prof.py condenses it into profiles/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r01}
ARGS=${2:---steps 20 --warmup 3 --no-cpu-baseline --no-closed-loop}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/trace" -o trace -- python "$R/bench.py" $ARGS > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d "$OUT/pmc_fetch" -o fetch -- python "$R/bench.py" $ARGS > "$OUT/bench_fetch.json" 2> "$OUT/fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d "$OUT/pmc_write" -o write -- python "$R/bench.py" $ARGS > "$OUT/bench_write.json" 2> "$OUT/write.err"
cd "$R"
find "$OUT" -name "*.csv" | head -20
