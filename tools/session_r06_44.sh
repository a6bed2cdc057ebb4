#!/bin/bash
# Round 6, GPU session 44: package power and sclk under the final fp16sd layer launch (back to back), as r06_smi_power_under_load.log did for the fp16x2 form
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out; mkdir -p $O
( for i in $(seq 1 40); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | tr '\n' ';'; echo; sleep 0.25; done ) > $O/r06s44_smi.log 2>&1 &
timeout 120 python tools/kbench_layer512.py --one --e16 --iters 30000 --which fused 2>&1 | grep "layer512 fused"
wait
grep -o "sclk clock level: [0-9]: ([0-9]*Mhz)\|Power (W): [0-9.]*" $O/r06s44_smi.log | paste - - | sort | uniq -c | sort -rn | head -12
