#!/bin/bash
# Round 6, GPU session 23: are the gate weights an L2 hot spot (32 CUs of an XCD stream the same 98 KB per wave in step)? n copies, one per group of CUs
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out; mkdir -p $O
for n in 1 2 4 1; do
lib=""; [ $n -gt 1 ] && lib=stylesinger_amd/_abl/libss_l512_wcopies$n.so
echo "--- $n copies of the gate weights"; SS_LIB_PATH=$lib timeout 300 python tools/kbench_layer512.py --one --iters 400 --which fused --wcopies $n 2>&1 | grep -E "layer512 fused" ; SS_LIB_PATH=$lib timeout 300 python tools/kbench_layer512.py --iters 400 --which fused --wcopies $n 2>&1 | grep -E "layer512 fused"
done | tee $O/r06s23_kbench.log
