#!/bin/bash
# knock-out timings of fa_train_dw3_kernel (rocprofv3 kernel stats over tools/prof_grad.py)
mkdir -p gpurun_out/r06e
cd /tmp && export TMPDIR=/tmp
for v in product dw3ko_MFMA dw3ko_LOAD dw3ko_SPLIT dw3ko_MFMA_LOAD; do
  if [ $v = product ]; then unset FA_LIBRARY; else export FA_LIBRARY=$GRAFT_REPO_ROOT/tools/_build/lib_$v.so; fi
  rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/r06e/$v -o p -- python $GRAFT_REPO_ROOT/tools/prof_grad.py > /dev/null 2>&1
  echo "$v: $(grep dw3_kernel $GRAFT_REPO_ROOT/gpurun_out/r06e/$v/p_kernel_stats.csv | cut -d, -f2-7)"
done
