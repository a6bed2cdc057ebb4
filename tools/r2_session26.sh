#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_round2.py -q -m gpu -s -k "winograd_f43 or c2_batch_item" 2>&1 | grep -E "mel L1|C2 item|passed|failed|Error" | tail -8
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -s 2>&1 | grep -E "L1|err|passed|failed" | tail -30
