#!/bin/bash
cd $GRAFT_REPO_ROOT
for pad in 0 40000 90000; do
  echo "--- LDS pad $pad"
  SS_WINO_LDS_PAD=$pad timeout 200 python tools/kbench.py --which wino --net mel --iters 40 2>&1 | tail -1
  SS_WINO_LDS_PAD=$pad timeout 200 python tools/kbench.py --which wino --net mel --iters 40 --B 32 2>&1 | tail -1
done
timeout 900 python -m pytest tests/test_gpu_round2.py -q -m gpu -s -k "1000_step" 2>&1 | grep -E "bf16 mode|passed|failed|Error" | tail -5
