#!/bin/bash
# Round 6, GPU session 3: ss_layer512 wired into the fp16x2 mel stack: unit + model parity tests, the B = 32 batch test, C4 end to end both ways
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out; mkdir -p $O
T0=$(date +%s)
stamp() { echo "=== [$(( $(date +%s) - T0 )) s] $*" | tee -a $O/r06s3_timeline.log; }
stamp "1 layer512 tests (unit + model vs the real reference)"
timeout 900 python -m pytest tests/test_gpu_layer512.py -q -s -m gpu 2>&1 | tail -25 | tee $O/r06s3_tests_layer512.log
stamp "2 B = 32 x T = 5625 through forward (now on the fused path)"
timeout 900 python -m pytest tests/test_gpu_round5.py -q -s -m gpu -k "c4_batch_items" 2>&1 | tail -12 | tee $O/r06s3_tests_c4_batch.log
stamp "3 C4 end to end, fused"
timeout 600 python bench.py --config c4 --streams 1 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary 2>&1 | tail -1 > $O/r06s3_c4_fused.json
python -c "import json;d=json.load(open('$O/r06s3_c4_fused.json'));print(d['value'], d['ms_per_step'], d.get('parity'), d['roofline'].get('us_per_launch'), d['roofline'].get('clock_ghz'))"
stamp "4 C4 end to end, two-launch form (SS_LAYER512=0)"
SS_LAYER512=0 timeout 600 python bench.py --config c4 --streams 1 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary 2>&1 | tail -1 > $O/r06s3_c4_pair.json
python -c "import json;d=json.load(open('$O/r06s3_c4_pair.json'));print(d['value'], d['ms_per_step'], d.get('parity'), d['roofline'].get('us_per_launch'), d['roofline'].get('clock_ghz'))"
stamp done
