#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "winograd" 2>&1 | tail -2
for rep in 1 2; do
for v in prev new; do
  if [ $v = prev ]; then export SS_LIB_PATH=$GRAFT_REPO_ROOT/stylesinger_amd/_abl/libss_prev.so; else unset SS_LIB_PATH; fi
  echo "--- $v"
  timeout 200 python tools/kbench.py --which wino43 --net mel --iters 40 2>&1 | tail -1
  timeout 200 python tools/kbench.py --which wino43 --net f0 --B 16 --iters 40 2>&1 | tail -1
  timeout 200 python tools/kbench.py --which wino43 --net mel --B 32 --iters 20 2>&1 | tail -1
  SS_BENCH_STREAMS=3 timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | grep -E "^\{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams 3', d['value'], d['ms_per_step'], 'clock', d['clock_ghz_timed_region'])"
done
done
