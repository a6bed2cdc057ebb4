#!/bin/bash
# Round 6, GPU session 28: warm trace of the fp16sd layer launch in its final data forms (fp16 addend set, stream remainder); all addend blocks requested up front?
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out; mkdir -p $O
for lib in "" stylesinger_amd/_abl/libss_l512_eahead.so "" stylesinger_amd/_abl/libss_l512_eahead.so; do echo "--- lib=${lib:-product}"; SS_LIB_PATH=$lib timeout 300 python tools/kbench_layer512.py --one --e16 --iters 400 --which fused 2>&1 | grep -E "layer512 fused"; done | tee $O/r06s28_kbench.log
SS_LIB_PATH=stylesinger_amd/_abl/libss_l512trace.so timeout 200 python tools/trace_layer512.py --one --e16 2>&1 | grep -v "^  slot\|amdgpu" | tail -18 | tee $O/r06s28_trace.log
