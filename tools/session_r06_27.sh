#!/bin/bash
# Round 6, GPU session 27: the residual stream of ss_layer512 as (H, fp16 remainder) instead of an fp32 copy: tests, kbench, C4 end to end
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out; mkdir -p $O
T0=$(date +%s)
stamp() { echo "=== [$(( $(date +%s) - T0 )) s] $*" | tee -a $O/r06s27_timeline.log; }
stamp "1 tests"
timeout 1500 python -m pytest tests/test_gpu_layer512.py -q -s -m gpu 2>&1 | tail -24 | tee $O/r06s27_tests.log
timeout 900 python -m pytest tests/test_gpu_round5.py -q -s -m gpu -k "c4_batch_items" 2>&1 | tail -10 | tee -a $O/r06s27_tests.log
stamp "2 kbench"
for e in "--one --e16" "--one" ""; do echo "--- $e"; timeout 300 python tools/kbench_layer512.py $e --iters 400 --which layer512 2>&1 | grep -E "layer512 (fused|gate|entry)"; done | tee $O/r06s27_kbench.log
stamp "3 C4 end to end"
for cfg in c4 c4x2; do
timeout 900 python bench.py --config $cfg --streams 1 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary 2>&1 | tail -1 > $O/r06s27_$cfg.json
python -c "import json;d=json.load(open('$O/r06s27_$cfg.json'));print('$cfg', d['value'], d['ms_per_step'], d['parity']['measured_in_this_run'], d['roofline'].get('us_per_launch'), d['roofline'].get('bound'), d['roofline'].get('frac'), d['roofline'].get('clock_ghz'))"
done
stamp done
