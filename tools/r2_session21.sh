#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_round2.py -q -m gpu -k "bf16" 2>&1 | tail -3
timeout 300 python tools/kbench_h.py 2>&1 | tail -2
