#!/bin/bash
cd $GRAFT_REPO_ROOT
for s in 1 3; do
SS_BENCH_STREAMS=$s timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | grep -E "^\{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams $s', d['value'], d['ms_per_step'], 'clock', d['clock_ghz_timed_region'], d['config']['e2e_fraction_of_mfma_peak'])"
done
