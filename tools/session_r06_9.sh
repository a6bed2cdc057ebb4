#!/bin/bash
# Round 6, GPU session 9: layer512 with the half-tile tail: tests, kbench with the knob on / off, C4 end to end on / off (same box), long-run power readout
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out; mkdir -p $O
T0=$(date +%s)
stamp() { echo "=== [$(( $(date +%s) - T0 )) s] $*" | tee -a $O/r06s9_timeline.log; }
stamp "1 tests"
timeout 900 python -m pytest tests/test_gpu_layer512.py -q -s -m gpu -x -k "float64 or many_tiles or half_tile" 2>&1 | tail -6 | tee $O/r06s9_tests_layer512.log
stamp "2 kbench, tail split on / off (400 launches each)"
timeout 300 python tools/kbench_layer512.py --iters 400 --which fused 2>&1 | tail -1 | tee $O/r06s9_kbench_tail.log
SS_LAYER512_TAIL=0 timeout 300 python tools/kbench_layer512.py --iters 400 --which fused 2>&1 | tail -1 | tee -a $O/r06s9_kbench_tail.log
timeout 300 python tools/kbench_layer512.py --iters 400 --which pair 2>&1 | tail -1 | tee -a $O/r06s9_kbench_tail.log
stamp "3 power / clocks under a 12 s load"
( for i in $(seq 1 30); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | tr '\n' ';'; echo; sleep 0.3; done ) > $O/r06s9_smi_under_load.log 2>&1 &
SMI=$!
timeout 200 python tools/kbench_layer512.py --iters 30000 --which fused 2>&1 | tail -1 | tee -a $O/r06s9_kbench_tail.log
wait $SMI
sort $O/r06s9_smi_under_load.log | uniq -c | sort -rn | head -8
stamp "4 C4 end to end: tail on, tail off, two-launch form"
for v in "" "SS_LAYER512_TAIL=0" "SS_LAYER512=0"; do
  env $v timeout 600 python bench.py --config c4 --streams 1 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-roofline 2>&1 | tail -1 > $O/r06s9_c4_$(echo $v | tr -d '=' ).json
  python -c "import json;d=json.load(open('$O/r06s9_c4_$(echo $v | tr -d '=' ).json'));print('$v', d['value'], d['ms_per_step'])" | tee -a $O/r06s9_c4_compare.log
done
stamp done
