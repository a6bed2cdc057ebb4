#!/bin/bash
# Round 6, GPU session 11: the driver's command on the final tree (default line incl. secondary), the new tests, kbench of the final kernel
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out; mkdir -p $O
T0=$(date +%s)
stamp() { echo "=== [$(( $(date +%s) - T0 )) s] $*" | tee -a $O/r06s11_timeline.log; }
stamp "1 new tests"
timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_layer512.py -q -m gpu 2>&1 | tail -4 | tee $O/r06s11_tests.log
stamp "2 the driver's command"
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | tail -1 > $O/r06s11_bench_c2.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06s11_bench_c2.json"))
print(d["value"], d["ms_per_step"], d.get("one_batch_at_a_time"), d["roofline"]["frac"], d["roofline"]["us_per_launch"], d["cpu_baseline"]["value"])
for k, v in d.get("secondary", {}).items():
    print(" ", k, {kk: v.get(kk) for kk in ("value", "ms_per_step", "error")} if isinstance(v, dict) else v)
    if isinstance(v, dict) and "roofline" in v and v["roofline"]:
        print("     roofline:", v["roofline"].get("kernel", "")[:60], v["roofline"].get("frac"), v["roofline"].get("us_per_launch"), v["roofline"].get("clock_ghz"))
PY
stamp "3 kbench of the final kernel, 400 launches each"
timeout 300 python tools/kbench_layer512.py --iters 400 2>&1 | tail -5 | tee $O/r06s11_kbench_layer512.log
stamp done
