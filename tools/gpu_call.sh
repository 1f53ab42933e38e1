#!/bin/bash
# retry wrapper used from the build container: gpurun exits 3 when no GPU slot is free
# usage: tools/gpu_call.sh <timeout-seconds> '<command run on the GPU box>'
for k in 1 2 3 4 5 6 7 8; do
  /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"; rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
