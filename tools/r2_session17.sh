#!/bin/bash
cd $GRAFT_REPO_ROOT
export SS_LIB_PATH=$GRAFT_REPO_ROOT/stylesinger_amd/_abl/libss_trace.so
for p in 0 1 2; do
echo "--- wave_prio $p"
SS_WAVE_PRIO=$p timeout 200 python tools/wave_trace.py 2>&1 | grep -E "cycles per chunk|wave lifetime|kernel span|launch:"
done
