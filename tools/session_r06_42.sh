#!/bin/bash
# Round 6, GPU session 24 (re-run as the final validation after the fp16 addend sets and the stream remainder): the tree with fp16sd as configs[3]'s mode, compact skip operands and the phase-shifted layer schedule: whole GPU suite, smoke,
# rocprofv3 kernel stats of the C4 loops (fp16sd, fp16x2), the driver's own bench command.
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out; mkdir -p $O
T0=$(date +%s)
stamp() { echo "=== [$(( $(date +%s) - T0 )) s] $*" | tee -a $O/r06s42_timeline.log; }
rm -f $O/parity_measurements.jsonl
stamp "1 the whole GPU suite"
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -15 | tee $O/r06s42_gpu_suite.log
stamp "2 smoke"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/r06s42_smoke.log
for cfg in c4 c4x2; do
stamp "3 rocprofv3 kernel stats, $cfg loop (20 diffusion steps)"
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof42_$cfg -o $cfg -- \
   python $GRAFT_REPO_ROOT/bench.py --config $cfg --diff-steps 20 --streams 1 --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/$O/r06s42_prof_$cfg.log 2>&1)
cp "$(find $O/prof42_$cfg -name '*kernel_stats.csv' | head -1)" $O/r06s42_${cfg}_kernel_stats.csv; grep -E "layer512|tile256s_kernel<0" $O/r06s42_${cfg}_kernel_stats.csv | cut -c1-170
rm -rf $O/prof42_$cfg
done
stamp "4 the driver's command"
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | tail -1 > $O/r06s42_bench_c2.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06s42_bench_c2.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("cpu_baseline", {}).get("value"))
for k, v in d.get("secondary", {}).items():
    if isinstance(v, dict):
        print(k, v.get("value"), v.get("ms_per_step"), (v.get("roofline") or {}).get("bound"), (v.get("roofline") or {}).get("frac"), (v.get("parity") or {}).get("meets_north_star"), v.get("error"))
PY
stamp done
