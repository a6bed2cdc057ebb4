#!/bin/bash
# Round 6, GPU session 21: four phases by a start delay on top of the half-tile-first classes?
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out; mkdir -p $O
for dk in 0 15 25 35 50 0; do echo "--- start delay $dk kilocycles for workgroups with bit 6 set (classes by bit 7)"; SS_L512_DELAY_KC=$dk SS_LAYER512_TAIL=17 timeout 300 python tools/kbench_layer512.py --one --iters 400 --which fused 2>&1 | grep -E "layer512 fused"; done | tee $O/r06s21_kbench.log
