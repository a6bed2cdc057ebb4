#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { SS_BENCH_STREAMS=3 timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | grep -E "^\{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
run base
SS_RES_TILE=4 run res128x64
SS_RES_TILE=2 run res64x128
SS_RES_TILE=1 run res128x128
SS_SKIP_TILE=4 run skip128x64
SS_SKIP_TILE=2 run skip64x128
SS_SKIP_TILE=1 run skip128x128
run base
