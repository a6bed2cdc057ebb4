#!/bin/bash
# Timing ablations of the residual-projection kernel (gemm16_res_kernel; debug builds, results wrong by design): what does a single-round
# launch of 10 us of matrix time spend its other 10 us on?
#   build (container):  tools/ablate_r16.sh build      run (GPU box):  tools/ablate_r16.sh run
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd $R
if [ "$1" = build ]; then
  mkdir -p stylesinger_amd/_abl
  others=$(ls stylesinger_amd/_obj/*.o | grep -v gemm16.hip)
  for n in 1 2 3 4 5 6; do
    (hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DSS_R16_ABL=$n -c stylesinger_amd/csrc/gemm16.hip -o /tmp/r16_abl$n.o &&
     hipcc --offload-arch=gfx950 -shared -fPIC -o stylesinger_amd/_abl/lib_r16abl$n.so $others /tmp/r16_abl$n.o) &
  done
  wait; ls -la stylesinger_amd/_abl
  exit 0
fi
names=("" "no weight preload" "no A DMA after the first two chunks" "no MFMAs" "no residual-stream loads" "no output stores" "no barriers in the loop")
echo "--- full kernel"; timeout 200 python tools/kbench.py --which res16 --iters 60 --mt 6 2>&1 | tail -2
for n in ${ABLS:-1 2 3 4 5 6}; do
  echo "--- SS_R16_ABL=$n (${names[$n]})"
  SS_LIB_PATH=$R/stylesinger_amd/_abl/lib_r16abl$n.so timeout 200 python tools/kbench.py --which res16 --iters 60 --mt 6 2>&1 | tail -2
done
