#!/bin/bash
# record session: full GPU suite, smoke, bench lines for profiles/
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
echo "=== pytest"
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -4
echo "=== smoke"
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -1
echo "=== bench c2"
timeout 900 python bench.py --steps 9 --warmup 3 2>&1 | grep -E "^\{" > gpurun_out/r02_bench_c2.json; cut -c1-300 gpurun_out/r02_bench_c2.json
echo "=== rocprof 1 stream"
(cd /tmp && export TMPDIR=/tmp && SS_BENCH_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02c -o r02c -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_r02c.log 2>&1)
grep -E "^\{" gpurun_out/prof_r02c.log > gpurun_out/r02_bench_c2_1stream_under_rocprof.json
f=$(find gpurun_out/prof_r02c -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r02_bench_c2_1stream_kernel_stats.csv; head -6 "$f" | cut -c1-160
rm -f gpurun_out/prof_r02c/*kernel_trace.csv
echo "=== bench c5"
timeout 900 python bench.py --config c5 --no-cpu-baseline 2>&1 | grep -E "^\{" > gpurun_out/r02_bench_c5_sweep.json; cut -c1-200 gpurun_out/r02_bench_c5_sweep.json
echo "=== bench c4"
timeout 900 python bench.py --config c4 --steps 1 --warmup 1 --streams 1 --no-cpu-baseline 2>&1 | grep -E "^\{" > gpurun_out/r02_bench_c4_bf16.json; cut -c1-200 gpurun_out/r02_bench_c4_bf16.json
