#!/bin/bash
# Round 6, GPU session 16: a layer512 launch on one time axis (workgroup start / end on the 100 MHz counter), cold and at the sustained clock
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSS_L512_TRACE -c stylesinger_amd/csrc/layer512.hip -o /tmp/l512t.o && hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libss_l512trace.so /tmp/l512t.o $(ls stylesinger_amd/_obj/*.o | grep -v layer512) || exit 1
SS_LIB_PATH=/tmp/libss_l512trace.so timeout 200 python tools/trace_layer512.py --one 2>&1 | tail -26 | tee $O/r06s16_trace_one_product.log
SS_LIB_PATH=/tmp/libss_l512trace.so timeout 200 python tools/trace_layer512.py --one --gate-only 2>&1 | tail -26 | tee $O/r06s16_trace_one_product_gate_only.log
SS_LIB_PATH=/tmp/libss_l512trace.so timeout 200 python tools/trace_layer512.py --one --warm 0 2>&1 | tail -26 | tee $O/r06s16_trace_one_product_cold.log
SS_LIB_PATH=/tmp/libss_l512trace.so timeout 200 python tools/trace_layer512.py 2>&1 | tail -26 | tee $O/r06s16_trace_two_products.log
