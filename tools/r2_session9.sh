#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_round2.py -q -m gpu -x -k "gemm_bf16" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "bf16" 2>&1 | grep -E "bf16 mode|passed|failed|Error" | tail
for v in 1; do
  (cd /tmp && export TMPDIR=/tmp && SS_BF16_HBM=$v SS_GRAPHS=off timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c4b_$v -o c4 -- python $R/bench.py --config c4 --diff-steps 10 --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_c4b_$v.log 2>&1)
  echo "=== SS_BF16_HBM=$v"; grep -E "^\{" gpurun_out/prof_c4b_$v.log | cut -c1-200
  f=$(find gpurun_out/prof_c4b_$v -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-200
done
for v in "SS_BF16_HBM=0" "SS_BF16_HBM=1"; do
  echo "--- $v"
  env $v timeout 900 python bench.py --config c4 --diff-steps 100 --steps 1 --warmup 1 --streams 1 --no-cpu-baseline --no-roofline 2>&1 | grep -E "^\{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
