#!/bin/bash
# Round 6, GPU session 20: E / P tiles with the eight waves' blocks side by side (8 KB contiguous per step) against one stream per wave
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_layer512.py -q -m gpu -k "not model" 2>&1 | tail -3 | tee $O/r06s20_tests.log
for lib in "" stylesinger_amd/_abl/libss_l512wavemajor.so "" stylesinger_amd/_abl/libss_l512wavemajor.so; do
for k in 17 2; do echo "--- lib=${lib:-product} layer512_tail = $k"; SS_LIB_PATH=$lib SS_LAYER512_TAIL=$k timeout 300 python tools/kbench_layer512.py --one --iters 400 --which fused 2>&1 | grep -E "layer512 fused" ; SS_LIB_PATH=$lib SS_LAYER512_TAIL=$k timeout 300 python tools/kbench_layer512.py --iters 400 --which fused 2>&1 | grep -E "layer512 fused"; done
done | tee $O/r06s20_kbench.log
