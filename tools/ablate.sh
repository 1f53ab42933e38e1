#!/bin/bash
# Kernel experiment: build exp_libs/libfa_abl<k>.so with one piece of the pipelined step kernel
# compiled out (-DFA_ABL=bitmask; results are WRONG by construction) to read that piece's marginal cost
# off the end-to-end launch time.  bits: 1 output wave emit, 2 laser tests, 8 pair forces,
# 16 walls, 32 next-heading sin/cos, 64 ordered force sum.
set -e
cd "$(dirname "$0")/.."
mkdir -p exp_libs
C=emergent-multiagent-strategies_amd/csrc
for k in ${@:-1 2 8 16 32 64}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DFA_ABL=$k -I include \
    $C/fa_step.hip $C/fa_collect.hip $C/fa_api.hip -o exp_libs/libfa_abl$k.so 2>/dev/null &
done
wait
ls exp_libs
