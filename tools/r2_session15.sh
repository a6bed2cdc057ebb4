#!/bin/bash
cd $GRAFT_REPO_ROOT
export SS_LIB_PATH=$GRAFT_REPO_ROOT/stylesinger_amd/_abl/libss_trace.so
timeout 200 python tools/wave_trace.py 2>&1 | tail -16
echo "--- B=32 (TN=2)"
timeout 200 python tools/wave_trace.py --B 32 2>&1 | tail -16
echo "--- B=32 TN=1"
SS_WINO_TN=1 timeout 200 python tools/wave_trace.py --B 32 2>&1 | tail -16
echo "--- 1 block/CU"
SS_WINO_LDS_PAD=90000 timeout 200 python tools/wave_trace.py 2>&1 | tail -16
