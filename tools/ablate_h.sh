#!/bin/bash
# Timing ablations of the bf16-in-HBM GEMM at the C4 size (debug builds, run on the GPU box).
#   SS_HABL=1 no global fetches in the loop, 2 no MFMAs, 3 no addend loads in the epilogue, 4 no epilogue stores.
# Build first (in the container): for n in 1 2 3 4; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSS_HABL=$n \
#   stylesinger_amd/csrc/*.hip -o stylesinger_amd/_abl/libss_habl$n.so; done
cd $GRAFT_REPO_ROOT
echo "--- product"
timeout 300 python tools/kbench_h.py 2>&1 | tail -2
for abl in 1 2 3 4; do
  echo "--- SS_HABL=$abl"
  SS_LIB_PATH=$GRAFT_REPO_ROOT/stylesinger_amd/_abl/libss_habl$abl.so timeout 300 python tools/kbench_h.py 2>&1 | tail -2
done
