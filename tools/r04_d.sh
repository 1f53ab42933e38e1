#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04d}
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 300 python tools/debug_golden.py 5v5_g_clip > $O/debug_golden.txt 2>&1; cat $O/debug_golden.txt | grep -v amdgpu.ids
bash tools/prof_grad.sh ${1:-r04d} 2>&1 | grep -E "fa_|shape"
