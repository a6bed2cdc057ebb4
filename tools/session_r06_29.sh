#!/bin/bash
# Round 6, GPU session 29: the next item's DMA before [B2] (under the fast waves' barrier wait) instead of at the end of the gate epilogue
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out; mkdir -p $O
for lib in "" stylesinger_amd/_abl/libss_l512_dmaearly.so "" stylesinger_amd/_abl/libss_l512_dmaearly.so; do echo "--- lib=${lib:-product}"; SS_LIB_PATH=$lib timeout 300 python tools/kbench_layer512.py --one --e16 --iters 400 --which fused 2>&1 | grep -E "layer512 fused"; SS_LIB_PATH=$lib timeout 300 python tools/kbench_layer512.py --iters 400 --which fused 2>&1 | grep -E "layer512 fused"; done | tee $O/r06s29_kbench.log
SS_LIB_PATH=stylesinger_amd/_abl/libss_l512trace.so timeout 200 python tools/trace_layer512.py --one --e16 2>&1 | grep -v "^  slot\|amdgpu" | tail -18 | tee $O/r06s29_trace.log
SS_LIB_PATH=stylesinger_amd/_abl/libss_l512_dmaearly.so timeout 600 python -m pytest tests/test_gpu_layer512.py -q -m gpu -k "not model" 2>&1 | tail -3
