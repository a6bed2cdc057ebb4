#!/bin/bash
# Round 6, GPU session 17: phase-shifted schedule of ss_layer512 (even workgroups run their half tile first)
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out; mkdir -p $O
T0=$(date +%s)
stamp() { echo "=== [$(( $(date +%s) - T0 )) s] $*" | tee -a $O/r06s17_timeline.log; }
stamp "1 tests"
timeout 1500 python -m pytest tests/test_gpu_layer512.py -q -s -m gpu -k "not model" 2>&1 | tail -14 | tee $O/r06s17_tests.log
stamp "2 kbench: phase shift on / off"
for k in 1 2; do echo "--- layer512_tail = $k"; SS_LAYER512_TAIL=$k timeout 300 python tools/kbench_layer512.py --one --iters 400 2>&1 | grep -E "layer512" ; SS_LAYER512_TAIL=$k timeout 300 python tools/kbench_layer512.py --iters 400 2>&1 | grep -E "layer512"; done | tee $O/r06s17_kbench.log
stamp "3 trace, warm, both schedules"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSS_L512_TRACE -c stylesinger_amd/csrc/layer512.hip -o /tmp/l512t.o && hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libss_l512trace.so /tmp/l512t.o $(ls stylesinger_amd/_obj/*.o | grep -v layer512) || exit 1
for k in 1 2; do echo "--- layer512_tail = $k"; SS_LAYER512_TAIL=$k SS_LIB_PATH=/tmp/libss_l512trace.so timeout 200 python tools/trace_layer512.py --one 2>&1 | tail -26; done | tee $O/r06s17_trace_one_product.log
echo "--- 64 workgroups, halves last"; SS_L512_GRID=64 SS_LAYER512_TAIL=2 SS_LIB_PATH=/tmp/libss_l512trace.so timeout 200 python tools/trace_layer512.py --one 2>&1 | tail -26 | tee -a $O/r06s17_trace_one_product.log
stamp "4 C4 end to end, fp16sd and fp16x2"
for cfg in c4sd c4; do
timeout 900 python bench.py --config $cfg --streams 1 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary 2>&1 | tail -1 > $O/r06s17_$cfg.json
python -c "import json;d=json.load(open('$O/r06s17_$cfg.json'));print(d['value'], d['ms_per_step'], d['parity']['measured_in_this_run'], d['roofline'].get('us_per_launch'), d['roofline'].get('frac'), d['roofline'].get('clock_ghz'))"
done
stamp done
