#!/bin/bash
cd $GRAFT_REPO_ROOT
SS_WINO_V3=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k winograd 2>&1 | tail -3
echo "--- v2"
timeout 200 python tools/kbench.py --which wino --net mel --iters 40 2>&1 | tail -1
echo "--- v2 at 2 blocks/CU"
SS_WINO_LDS_PAD=20000 timeout 200 python tools/kbench.py --which wino --net mel --iters 40 2>&1 | tail -1
echo "--- v3 (196 regs, 2 blocks/CU)"
SS_WINO_V3=1 timeout 200 python tools/kbench.py --which wino --net mel --iters 40 2>&1 | tail -1
echo "--- B=32: v2 / v2 2blk / v3"
timeout 200 python tools/kbench.py --which wino --net mel --B 32 --iters 20 2>&1 | tail -1
SS_WINO_LDS_PAD=20000 timeout 200 python tools/kbench.py --which wino --net mel --B 32 --iters 20 2>&1 | tail -1
SS_WINO_V3=1 timeout 200 python tools/kbench.py --which wino --net mel --B 32 --iters 20 2>&1 | tail -1
