#!/bin/bash
# Round 6, GPU session 19: which bit of the workgroup index should pick the half-tile-first class
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out; mkdir -p $O
for k in 2 10 11 12 13 14 15 16 17 2 10; do echo "--- layer512_tail = $k"; SS_LAYER512_TAIL=$k timeout 300 python tools/kbench_layer512.py --one --iters 400 --which fused 2>&1 | grep -E "layer512 fused"; done | tee $O/r06s19_kbench.log
