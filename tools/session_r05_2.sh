#!/bin/bash
# Round 5, GPU session 2: full GPU suite on the refactored tree (gate16 / gemm16 bodies, f0 tracker, fp16q4 parity), the dataflow-launch experiment,
# c4q, and the C2 headline for regression.
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out; mkdir -p $O
T0=$(date +%s)
stamp() { echo "=== [$(( $(date +%s) - T0 )) s] $*" | tee -a $O/r05s2_timeline.log; }
stamp "1 the new tests first"
timeout 900 python -m pytest tests/test_gpu_round5.py -q -s -m gpu 2>&1 | tail -60 > $O/r05s2_tests_round5.log
grep -E "passed|failed|C4 as spec|fp16q4|item |f0 tracker|speaker" $O/r05s2_tests_round5.log | tail -20
stamp "2 dataflow launch experiment (gate + projection in one launch)"
timeout 120 python tools/kbench_fused.py 2>&1 | tail -4 | tee $O/r05s2_kbench_fused_c2.log
timeout 120 python tools/kbench_fused.py --B 32 --T 1500 2>&1 | tail -4 | tee $O/r05s2_kbench_fused_b32.log
timeout 120 python tools/kbench_fused.py --B 1 --T 750 2>&1 | tail -4 | tee $O/r05s2_kbench_fused_b1.log
stamp "3 c4q"
timeout 400 python bench.py --config c4q --streams 1 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary 2>&1 | tail -1 > $O/r05s2_c4q.json
python -c "import json;d=json.load(open('$O/r05s2_c4q.json'));print(d['value'], d['ms_per_step'], d.get('parity'), d['roofline'].get('us_per_launch'), d['roofline'].get('clock_ghz'))"
stamp "4 C2 headline (regression check of the refactored kernels)"
timeout 400 python bench.py --steps 12 --warmup 3 --no-secondary --no-cpu-baseline 2>&1 | tail -1 > $O/r05s2_c2.json
python -c "import json;d=json.load(open('$O/r05s2_c2.json'));print(d['value'], d['ms_per_step'], d.get('one_batch_at_a_time'), d['roofline']['us_per_launch'], d['roofline']['frac'])"
stamp "5 the rest of the GPU suite"
timeout 900 python -m pytest tests -q -m gpu --deselect tests/test_gpu_round5.py 2>&1 | tail -15 > $O/r05s2_tests_rest.log
tail -5 $O/r05s2_tests_rest.log
stamp done
