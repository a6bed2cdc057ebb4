#!/bin/bash
# Round 6, GPU session 2: phase trace of layer512_kernel at the C4 shape (fused and gate only), then the kbench again
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out; mkdir -p $O
SS_LIB_PATH=stylesinger_amd/_abl/libss_l512trace.so timeout 200 python tools/trace_layer512.py 2>&1 | tail -12 | tee $O/r06s2_trace_fused.log
SS_LIB_PATH=stylesinger_amd/_abl/libss_l512trace.so timeout 200 python tools/trace_layer512.py --gate-only 2>&1 | tail -12 | tee $O/r06s2_trace_gate.log
timeout 300 python tools/kbench_layer512.py 2>&1 | tail -6 | tee $O/r06s2_kbench_layer512.log
