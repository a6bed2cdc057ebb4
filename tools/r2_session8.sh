#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
for v in 0 1; do
  (cd /tmp && export TMPDIR=/tmp && SS_BF16_HBM=$v SS_GRAPHS=off timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c4_$v -o c4 -- python $R/bench.py --config c4 --diff-steps 10 --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_c4_$v.log 2>&1)
  echo "=== SS_BF16_HBM=$v"; grep -E "^\{" gpurun_out/prof_c4_$v.log | cut -c1-200
  f=$(find gpurun_out/prof_c4_$v -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-200
done
