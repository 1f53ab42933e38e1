#!/bin/bash
# kernel experiments: time the fused step kernel for each alternative build in exp_libs/
for f in exp_libs/libfa_exp*.so; do
  FA_LIB_OVERRIDE=$PWD/$f python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-collector > /tmp/ab.json 2>/dev/null
  python -c "import json; d=json.load(open('/tmp/ab.json')); print('$f', round(d['roofline']['avg_launch_us'],1))"
done
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-collector > /tmp/ab.json 2>/dev/null
python -c "import json; d=json.load(open('/tmp/ab.json')); print('base', round(d['roofline']['avg_launch_us'],1))"
