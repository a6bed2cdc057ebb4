#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_round2.py -q -m gpu -x -k "gemm_bf16" 2>&1 | tail -15
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "bf16" 2>&1 | grep -E "bf16 mode|passed|failed|Error" | tail
echo "=== TN experiments (mel C2, streams 2)"
for v in "SS_WINO_TN=1" "SS_WINO_TN=2"; do
  echo "--- $v"
  env $v timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>&1 | grep -E "^\{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('one_batch_at_a_time'))"
done
for s in 3 4; do
  echo "--- streams $s"
  SS_BENCH_STREAMS=$s timeout 300 python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-roofline 2>&1 | grep -E "^\{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
echo "=== c4 quick (100 mel steps) old vs new bf16 path"
for v in "SS_BF16_HBM=0" "SS_BF16_HBM=1"; do
  echo "--- $v"
  env $v timeout 900 python bench.py --config c4 --diff-steps 100 --steps 1 --warmup 1 --streams 1 --no-cpu-baseline --no-roofline 2>&1 | grep -E "^\{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
