#!/bin/bash
# Round 6, GPU session 6: layer512 with the addend prefetched two blocks ahead: unit tests, trace, kbench, C4; then the round-6 host-path tests
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out; mkdir -p $O
T0=$(date +%s)
stamp() { echo "=== [$(( $(date +%s) - T0 )) s] $*" | tee -a $O/r06s6_timeline.log; }
stamp "1 layer512 unit tests"
timeout 900 python -m pytest tests/test_gpu_layer512.py -q -s -m gpu -x -k "float64 or many_tiles" 2>&1 | tail -9 | tee $O/r06s6_tests_layer512.log
stamp "2 trace"
SS_LIB_PATH=stylesinger_amd/_abl/libss_l512trace.so timeout 200 python tools/trace_layer512.py 2>&1 | tail -12 | tee $O/r06s6_trace_fused.log
stamp "3 kbench"
timeout 300 python tools/kbench_layer512.py 2>&1 | tail -6 | tee $O/r06s6_kbench_layer512.log
stamp "4 C4 end to end, fused"
timeout 600 python bench.py --config c4 --streams 1 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary 2>&1 | tail -1 > $O/r06s6_c4_fused.json
python -c "import json;d=json.load(open('$O/r06s6_c4_fused.json'));print(d['value'], d['ms_per_step'], d['parity']['measured_in_this_run'])"
stamp "5 round-6 tests (8 ranks on one device, q4 guard, non-finite flag) + entry point tests"
timeout 1500 python -m pytest tests/test_gpu_round6.py -q -s -m gpu 2>&1 | tail -15 | tee $O/r06s6_tests_round6.log
timeout 900 python -m pytest tests/test_gpu_round5.py -q -s -m gpu -k "infer_once or example_run" 2>&1 | tail -15 | tee $O/r06s6_tests_entry.log
stamp done
