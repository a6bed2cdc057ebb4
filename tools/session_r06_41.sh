#!/bin/bash
# Round 6, GPU session 41: the mel sampler's draw as one Philox block per four frames of a bin: timing of the DDPM launch at C2 / C4 shapes, the tests that pin its consistency
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out; mkdir -p $O
timeout 300 python tools/kbench_final.py --B 8 --T 1500 --iters 300 2>&1 | grep -v amdgpu | grep "auto" | tee $O/r06s41_kbench.log
timeout 300 python tools/kbench_final.py --iters 100 2>&1 | grep -v amdgpu | grep "auto" | tee -a $O/r06s41_kbench.log
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_round4.py -q -m gpu -x 2>&1 | tail -4 | tee $O/r06s41_tests.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>&1 | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('c2', d['value'], d['ms_per_step'])"
