#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
O=gpurun_out; mkdir -p $O
timeout 400 python -m pytest tests -q -s -m gpu -k "preprocess or frontend or emotion or speaker or f0_tracker or vad or front_end or wav2mel" 2>&1 | tail -25 | tee $O/r05s5_tests.log
