#!/bin/bash
# Round 6, GPU session 35: the two-group kernel with a deeper weight ring (one conv wave per SIMD has to hide the L2 latency of its weight stream alone)
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out; mkdir -p $O
for lib in "" stylesinger_amd/_abl/libss_l512g_ring4.so stylesinger_amd/_abl/libss_l512g_ring5.so; do echo "--- lib=${lib:-product (ring 3)}"; SS_LIB_PATH=$lib SS_LAYER512_GROUPS=1 timeout 120 python tools/kbench_layer512.py --one --e16 --iters 400 --which layer512 2>&1 | grep -E "layer512 (fused|gate)"; done | tee $O/r06s35_kbench.log
