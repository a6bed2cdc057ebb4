#!/bin/bash
# Round 6, GPU session 22: depth of the weight ring (the wave left alone in the conv loop is L2-latency bound at the sustained clock?)
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out; mkdir -p $O
for lib in "" stylesinger_amd/_abl/libss_l512_ring6_4.so stylesinger_amd/_abl/libss_l512_ring7_3.so stylesinger_amd/_abl/libss_l512_ring8_4.so ""; do
echo "--- lib=${lib:-product (ring 5 / 3)}"; SS_LIB_PATH=$lib timeout 300 python tools/kbench_layer512.py --one --iters 400 --which fused 2>&1 | grep -E "layer512 fused" ; SS_LIB_PATH=$lib timeout 300 python tools/kbench_layer512.py --iters 400 --which fused 2>&1 | grep -E "layer512 fused"
done | tee $O/r06s22_kbench.log
