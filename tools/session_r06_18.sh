#!/bin/bash
# Round 6, GPU session 18: which workgroups take their half tile first (by XCD slot or by parity); stream loads + next DMA after [B3] (experiment build)
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out; mkdir -p $O
for lib in "" stylesinger_amd/_abl/libss_l512late.so; do
for k in 1 3 2; do echo "--- lib=${lib:-product} layer512_tail = $k"; SS_LIB_PATH=$lib SS_LAYER512_TAIL=$k timeout 300 python tools/kbench_layer512.py --one --iters 400 --which fused 2>&1 | grep -E "layer512 fused" ; SS_LIB_PATH=$lib SS_LAYER512_TAIL=$k timeout 300 python tools/kbench_layer512.py --iters 400 --which fused 2>&1 | grep -E "layer512 fused"; done
done | tee $O/r06s18_kbench.log
