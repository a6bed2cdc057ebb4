#!/bin/bash
# Round 6, GPU session 25: the fp16sd skip GEMM with 64 channels per step (both operands compact: every DMA lane live, half the barriers)
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_round6.py -q -s -m gpu -k "skip_gemm" 2>&1 | tail -6 | tee $O/r06s25_tests.log
for d in 1 0 1 0; do echo "--- skip_dense = $d"; SS_SKIP_DENSE=$d timeout 300 python tools/kbench_h.py --f16 --which skip --compact 2 --iters 200 2>&1 | grep "skip GEMM"; done | tee $O/r06s25_kbench.log
echo "--- two products, compact A"; timeout 300 python tools/kbench_h.py --f16 --which skip --compact 1 --iters 200 2>&1 | grep "skip GEMM" | tee -a $O/r06s25_kbench.log
echo "--- two products, pair layout"; timeout 300 python tools/kbench_h.py --f16 --which skip --iters 200 2>&1 | grep "skip GEMM" | tee -a $O/r06s25_kbench.log
