#!/bin/bash
cd $GRAFT_REPO_ROOT
export SS_WINO_M=4
for s in 2 4 5 6; do
  SS_BENCH_STREAMS=$s timeout 300 python bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-roofline 2>&1 | grep -E "^\{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wino_m 4 streams $s', d['value'], d['ms_per_step'])"
done
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -q -m gpu -x 2>&1 | tail -5
