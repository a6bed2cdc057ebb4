#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -4
echo "=== c4 full"
timeout 1500 python bench.py --config c4 --steps 1 --warmup 1 --streams 1 --no-cpu-baseline > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err; cut -c1-300 gpurun_out/bench_c4.json; tail -2 gpurun_out/bench_c4.err
echo "=== c2 default"
timeout 900 python bench.py --steps 9 --warmup 3 > gpurun_out/bench_c2_r02.json 2> gpurun_out/bench_c2_r02.err; cut -c1-300 gpurun_out/bench_c2_r02.json; tail -2 gpurun_out/bench_c2_r02.err
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
