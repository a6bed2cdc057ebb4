#!/bin/bash
# Round 6, GPU session 33: the two-group form of ss_layer512 (layer512g_kernel): bit-identity with layer512_kernel, kbench
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_layer512.py -q -s -x -m gpu -k "two_group" 2>&1 | grep -v amdgpu | tail -25 | tee $O/r06s33_tests.log
for gr in 0 1 0 1; do echo "--- layer512_groups = $gr"; SS_LAYER512_GROUPS=$gr timeout 120 python tools/kbench_layer512.py --one --e16 --iters 400 --which layer512 2>&1 | grep -E "layer512 (fused|gate)"; done | tee $O/r06s33_kbench.log
