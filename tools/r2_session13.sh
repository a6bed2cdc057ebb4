#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu 2>&1 | tail -3
timeout 200 python tools/kbench.py --which wino --net mel --iters 40 2>&1 | tail -1
timeout 200 python tools/kbench.py --which wino --net f0 --B 16 --iters 40 2>&1 | tail -1
timeout 200 python tools/kbench.py --which wino --net mel --B 32 --iters 20 2>&1 | tail -1
for s in 1 3; do
  SS_BENCH_STREAMS=$s timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | grep -E "^\{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "golden or c2_full" 2>&1 | tail -3
