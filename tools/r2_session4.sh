#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu 2>&1 | tail -5
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -q -m gpu -s 2>&1 | grep -E "passed|failed|FAILED|Error|mel L1|C2 item" | tail -40
echo "=== kbench: wino v1 vs v2, conv loops"
for v in "SS_WINO_V1=1" "SS_NOP=1"; do
  echo "--- $v"
  env $v timeout 200 python tools/kbench.py --which wino --net mel --iters 40 2>&1 | tail -1
  env $v timeout 200 python tools/kbench.py --which wino --net f0 --B 16 --iters 40 2>&1 | tail -1
done
timeout 200 python tools/kbench.py --which resskip --net mel --iters 60 --tile 3 2>&1 | tail -1
timeout 200 python tools/kbench.py --which voc --iters 30 2>&1 | tail -5
timeout 200 python tools/kbench_skip.py 2>&1 | tail -10
echo "=== bench"
for v in "SS_BENCH_STREAMS=1" "SS_BENCH_STREAMS=2" "SS_BENCH_STREAMS=2 SS_WINO_V1=1"; do
  echo "--- $v"
  env $v timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>&1 | grep -E "^\{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['step_overlap'], d.get('one_batch_at_a_time'))"
done
