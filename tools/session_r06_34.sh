#!/bin/bash
# Round 6, GPU session 34: which waves share a SIMD? the two-group kernel with three wave -> group maps
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out; mkdir -p $O
for lib in "" stylesinger_amd/_abl/libss_l512g_map1.so stylesinger_amd/_abl/libss_l512g_map2.so; do echo "--- lib=${lib:-product map 0 (wave >> 2)}"; SS_LIB_PATH=$lib SS_LAYER512_GROUPS=1 timeout 120 python tools/kbench_layer512.py --one --e16 --iters 400 --which layer512 2>&1 | grep -E "layer512 (fused|gate)"; done | tee $O/r06s34_kbench.log
SS_LIB_PATH=stylesinger_amd/_abl/libss_l512g_map1.so timeout 300 python -m pytest tests/test_gpu_layer512.py -q -x -m gpu -k "two_group" 2>&1 | tail -2
