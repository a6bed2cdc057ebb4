#!/bin/bash
# Round 5, GPU session 4: the row-owner skeleton next to the two-launch / dataflow figures of the same box.
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 tools/ubench/rowowner_skeleton.hip -o /tmp/rowowner 2>/dev/null && timeout 60 /tmp/rowowner | tee $O/r05s4_rowowner_skeleton.log
timeout 120 python tools/kbench_fused.py 2>&1 | tail -3 | tee $O/r05s4_kbench_fused_c2.log
timeout 200 python bench.py --config c4q --streams 1 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary 2>&1 | tail -1 > $O/r05s4_c4q.json
python -c "import json;d=json.load(open('$O/r05s4_c4q.json'));r=d['roofline'];print(d['value'], d['ms_per_step'], r['kernel'][:40], r['us_per_launch'], r['clock_ghz'], r['frac'], r['traffic'])"
