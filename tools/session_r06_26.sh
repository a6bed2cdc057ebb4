#!/bin/bash
# Round 6, GPU session 26: the conditioner addend of the fused layer launch as fp16 sigma-delta sets (fp16sd): unit + model parity, kbench, C4 end to end
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out; mkdir -p $O
T0=$(date +%s)
stamp() { echo "=== [$(( $(date +%s) - T0 )) s] $*" | tee -a $O/r06s26_timeline.log; }
stamp "1 tests"
timeout 1500 python -m pytest tests/test_gpu_layer512.py -q -s -m gpu -k "addend or fp16sd" 2>&1 | tail -14 | tee $O/r06s26_tests.log
timeout 900 python -m pytest tests/test_gpu_round5.py -q -s -m gpu -k "c4_batch_items and fp16sd" 2>&1 | tail -8 | tee -a $O/r06s26_tests.log
stamp "2 kbench"
for e in "" "--e16"; do echo "--- one product $e"; timeout 300 python tools/kbench_layer512.py --one $e --iters 400 --which layer512 2>&1 | grep -E "layer512 (fused|gate)"; done | tee $O/r06s26_kbench.log
stamp "3 C4 end to end: 8 addend sets (default), fp32 slab"
for es in 8 0; do
SS_SD_E_SETS=$es timeout 900 python bench.py --config c4 --streams 1 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary 2>&1 | tail -1 > $O/r06s26_c4_e$es.json
python -c "import json;d=json.load(open('$O/r06s26_c4_e$es.json'));print('e_sets=$es', d['value'], d['ms_per_step'], d['parity']['measured_in_this_run'], d['roofline'].get('us_per_launch'), d['roofline'].get('frac'), d['roofline'].get('clock_ghz'))"
done
stamp done
