#!/bin/bash
# Timing ablations of the 16x16-tile Winograd gate kernel (debug builds; results wrong by design): what bounds a single-round launch?
#   build (container):  tools/ablate_g16.sh build      run (GPU box):  tools/ablate_g16.sh run
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd $R
if [ "$1" = build ]; then
  mkdir -p stylesinger_amd/_abl
  others=$(ls stylesinger_amd/_obj/*.o | grep -v wino43_gate16)
  for n in 1 2 3 4 5 6 7 8; do
    (hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DSS_G16_ABL=$n -c stylesinger_amd/csrc/wino43_gate16.hip -o /tmp/g16_abl$n.o &&
     hipcc --offload-arch=gfx950 -shared -fPIC -o stylesinger_amd/_abl/lib_g16abl$n.so $others /tmp/g16_abl$n.o) &
  done
  wait; ls -la stylesinger_amd/_abl
  exit 0
fi
names=("" "no global fetches in the loop" "no LDS stores" "no MFMAs" "no barriers" "no addend loads" "no activations/exchange" "no weight fetches in the loop" "no raw-row fetches in the loop")
echo "--- full kernel"; timeout 200 python tools/kbench.py --which wino43_16 --iters 60 --mt 3 2>&1 | tail -2
for n in ${ABLS:-1 2 3 4 5 6 7 8}; do
  echo "--- SS_G16_ABL=$n (${names[$n]})"
  SS_LIB_PATH=$R/stylesinger_amd/_abl/lib_g16abl$n.so timeout 200 python tools/kbench.py --which wino43_16 --iters 60 --mt 3 2>&1 | tail -2
done
