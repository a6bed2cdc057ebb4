#!/bin/bash
# Round 5, GPU session 3: the f0 tracker test (own short timeout), round-5 records of the fp32 headline kernel (rocprofv3 kernel stats + PMC).
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out; mkdir -p $O
T0=$(date +%s)
stamp() { echo "=== [$(( $(date +%s) - T0 )) s] $*" | tee -a $O/r05s3_timeline.log; }
stamp "1 f0 tracker"
timeout 200 python -m pytest tests/test_gpu_round5.py -q -s -m gpu -k "f0_tracker" 2>&1 | tail -25 > $O/r05s3_test_f0.log
tail -8 $O/r05s3_test_f0.log
stamp "2 rocprofv3 kernel stats of the C2 step, one batch at a time"
timeout 300 bash tools/reproduce.sh profile > $O/r05s3_profile.log 2>&1
tail -12 $O/r05s3_profile.log | cut -c1-200
stamp "3 PMC passes on the fp32 gate (MT = 2)"
timeout 400 bash tools/reproduce.sh pmc-gate 2 > $O/r05s3_pmc_gate.log 2>&1
grep -A12 "wino43_gate16_kernel" $O/r05s3_pmc_gate.log | head -60
stamp done
