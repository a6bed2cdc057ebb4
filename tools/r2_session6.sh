#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "=== c5"
timeout 900 python bench.py --config c5 --steps 2 --warmup 1 --streams 1 --no-cpu-baseline > gpurun_out/bench_c5.json 2> gpurun_out/bench_c5.err; cut -c1-700 gpurun_out/bench_c5.json; tail -3 gpurun_out/bench_c5.err
echo "=== c4 (bf16, 1000 mel steps)"
timeout 1500 python bench.py --config c4 --steps 1 --warmup 1 --streams 1 --no-cpu-baseline > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err; cut -c1-700 gpurun_out/bench_c4.json; tail -3 gpurun_out/bench_c4.err
echo "=== c2 default"
timeout 900 python bench.py --steps 6 --warmup 2 > gpurun_out/bench_c2_r02.json 2> gpurun_out/bench_c2_r02.err; cut -c1-400 gpurun_out/bench_c2_r02.json; tail -3 gpurun_out/bench_c2_r02.err
