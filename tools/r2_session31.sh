#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "--- SS_WINO_M=2"
SS_WINO_M=2 timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -3
echo "--- SS_WINO=0 (direct), goldens only"
SS_WINO=0 timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "golden" 2>&1 | tail -3
