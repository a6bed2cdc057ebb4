#!/bin/bash
# Round 6, GPU session 5: is layer512's epilogue memory-starved because all CUs run their phases in lockstep? Trace with 64 of 256 CUs, and
# with odd workgroups started half a tile late.
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out; mkdir -p $O
export SS_LIB_PATH=stylesinger_amd/_abl/libss_l512trace.so
echo "--- all CUs" | tee $O/r06s5_trace_experiments.log
timeout 200 python tools/trace_layer512.py 2>&1 | tail -11 | tee -a $O/r06s5_trace_experiments.log
echo "--- 64 workgroups (a quarter of the CUs)" | tee -a $O/r06s5_trace_experiments.log
SS_L512_GRID=64 timeout 200 python tools/trace_layer512.py 2>&1 | tail -11 | tee -a $O/r06s5_trace_experiments.log
for st in 25 50; do
echo "--- odd workgroups start $st kilocycles late" | tee -a $O/r06s5_trace_experiments.log
timeout 200 python tools/trace_layer512.py --stagger $st 2>&1 | tail -11 | tee -a $O/r06s5_trace_experiments.log
done
unset SS_LIB_PATH
timeout 600 python -m pytest tests/test_gpu_layer512.py -q -m gpu -x 2>&1 | tail -5
