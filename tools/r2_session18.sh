#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "winograd" 2>&1 | tail -5
for net in mel f0; do
  Bn=8; [ $net = f0 ] && Bn=16
  timeout 200 python tools/kbench.py --which wino --net $net --B $Bn --iters 40 2>&1 | tail -1
  timeout 200 python tools/kbench.py --which wino43 --net $net --B $Bn --iters 40 2>&1 | tail -1
done
timeout 200 python tools/kbench.py --which wino --net mel --B 32 --iters 20 2>&1 | tail -1
timeout 200 python tools/kbench.py --which wino43 --net mel --B 32 --iters 20 2>&1 | tail -1
for m in 2 4; do
  for s in 1 3; do
    SS_WINO_M=$m SS_BENCH_STREAMS=$s timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | grep -E "^\{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wino_m $m streams $s', d['value'], d['ms_per_step'])"
  done
done
