#!/bin/bash
# Round 6, GPU session 1: first run of ss_layer512 (fused residual layer of the fp16x2 mel denoiser): unit tests, then the kernel against the
# launch pair it replaces at the BASELINE configs[3] shape.
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out; mkdir -p $O
T0=$(date +%s)
stamp() { echo "=== [$(( $(date +%s) - T0 )) s] $*" | tee -a $O/r06s1_timeline.log; }
stamp "1 layer512 unit tests"
timeout 600 python -m pytest tests/test_gpu_layer512.py -q -s -m gpu -x 2>&1 | tail -40 | tee $O/r06s1_tests_layer512.log
stamp "2 kbench: fused layer vs the launch pair, C4 shape"
timeout 300 python tools/kbench_layer512.py 2>&1 | tail -6 | tee $O/r06s1_kbench_layer512.log
stamp done
