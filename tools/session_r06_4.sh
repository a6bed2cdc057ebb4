#!/bin/bash
# Round 6, GPU session 4: layer512 v2 (fp32 stream, one-rcp gate, deeper projection ring, no store wait at [B1]): tests, trace, kbench, C4
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out; mkdir -p $O
T0=$(date +%s)
stamp() { echo "=== [$(( $(date +%s) - T0 )) s] $*" | tee -a $O/r06s4_timeline.log; }
stamp "1 layer512 tests"
timeout 900 python -m pytest tests/test_gpu_layer512.py -q -s -m gpu -x 2>&1 | tail -25 | tee $O/r06s4_tests_layer512.log
stamp "2 trace"
SS_LIB_PATH=stylesinger_amd/_abl/libss_l512trace.so timeout 200 python tools/trace_layer512.py 2>&1 | tail -12 | tee $O/r06s4_trace_fused.log
stamp "3 kbench"
timeout 300 python tools/kbench_layer512.py 2>&1 | tail -6 | tee $O/r06s4_kbench_layer512.log
stamp "4 C4 end to end, fused"
timeout 600 python bench.py --config c4 --streams 1 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary 2>&1 | tail -1 > $O/r06s4_c4_fused.json
python -c "import json;d=json.load(open('$O/r06s4_c4_fused.json'));print(d['value'], d['ms_per_step'], d['parity']['measured_in_this_run'], d['roofline'].get('us_per_launch'), d['roofline'].get('clock_ghz'))"
stamp done
