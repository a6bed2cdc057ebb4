#!/bin/bash
# Timing ablations of the Winograd gate kernel (debug builds, run on the GPU box): which part of a K chunk bounds the loop?
#   SS_ABL=1 no global fetches in the loop, 2 no LDS stores, 3 no MFMAs, 4 no barrier. Results are wrong by design.
# Build first (in the container): for n in 0 1 2 3 4; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSS_EXPERIMENT_KNOBS -DSS_ABL=$n \
#   stylesinger_amd/csrc/*.hip -o stylesinger_amd/_abl/libss_abl$n.so; done
cd $GRAFT_REPO_ROOT
for abl in 0 1 2 3 4; do
  out=$GRAFT_REPO_ROOT/stylesinger_amd/_abl/libss_abl$abl.so
  echo "--- SS_ABL=$abl"
  SS_LIB_PATH=$out timeout 200 python tools/kbench.py --which wino --net mel --iters 40 2>&1 | tail -1
  SS_LIB_PATH=$out SS_WINO_LDS_PAD=90000 timeout 200 python tools/kbench.py --which wino --net mel --iters 40 2>&1 | tail -1
done
