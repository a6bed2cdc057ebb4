#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
O=gpurun_out; mkdir -p $O
timeout 200 python tools/kbench_fused.py 2>&1 | tail -6 | tee $O/r05s6_kbench_fused_c2.log
timeout 200 python tools/kbench_fused.py --B 32 --T 1500 2>&1 | tail -6 | tee $O/r05s6_kbench_fused_b32.log
timeout 200 python tools/kbench_fused.py --B 1 --T 750 2>&1 | tail -6 | tee $O/r05s6_kbench_fused_b1.log
