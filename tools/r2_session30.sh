#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "winograd_f43" 2>&1 | tail -4
