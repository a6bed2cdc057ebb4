#!/bin/bash
# Round 5, final-tree records: the driver's own commands (full GPU suite, smoke, default bench line).
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out; mkdir -p $O
T0=$(date +%s)
stamp() { echo "=== [$(( $(date +%s) - T0 )) s] $*" | tee -a $O/r05f_timeline.log; }
stamp "1 full GPU suite"
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 > $O/r05f_tests.log
tail -4 $O/r05f_tests.log
stamp "2 smoke"
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee $O/r05f_smoke.log
stamp "3 default bench line as the driver runs it"
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/r05f_bench.err | tail -1 > $O/r05f_bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05f_bench.json"))
print("c2", d["value"], d["ms_per_step"], d.get("one_batch_at_a_time"), d["roofline"]["frac"], d["roofline"]["us_per_launch"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
for k, v in d.get("secondary", {}).items():
    print(k, {kk: v.get(kk) for kk in ("value", "ms_per_step", "wall_s_incl_setup", "error", "spread_frac", "item_order_restored")}, (v.get("parity") or {}).get("meets_north_star"))
PY
stamp done
