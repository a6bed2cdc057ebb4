#!/bin/bash
# Round 6, GPU session 10: this round's profiles on the final tree - rocprofv3 kernel stats of C2 (one stream) and of the C4 loop, the launch floor of
# the C2 loop, the full per-GPU share of configs[4], then the whole GPU test suite.
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out; mkdir -p $O
T0=$(date +%s)
stamp() { echo "=== [$(( $(date +%s) - T0 )) s] $*" | tee -a $O/r06s10_timeline.log; }
stamp "1 rocprofv3 kernel stats, C2 one stream"
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_c2 -o c2 -- \
   python $GRAFT_REPO_ROOT/bench.py --streams 1 --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/$O/r06s10_prof_c2.log 2>&1)
grep -E "^\{" $O/r06s10_prof_c2.log | cut -c1-200
cp "$(find $O/prof_c2 -name '*kernel_stats.csv' | head -1)" $O/r06s10_c2_kernel_stats.csv; head -6 $O/r06s10_c2_kernel_stats.csv | cut -c1-160
stamp "2 rocprofv3 kernel stats, C4 loop (20 diffusion steps)"
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_c4b -o c4 -- \
   python $GRAFT_REPO_ROOT/bench.py --config c4 --diff-steps 20 --streams 1 --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/$O/r06s10_prof_c4.log 2>&1)
grep -E "^\{" $O/r06s10_prof_c4.log | cut -c1-200
cp "$(find $O/prof_c4b -name '*kernel_stats.csv' | head -1)" $O/r06s10_c4_kernel_stats.csv; head -5 $O/r06s10_c4_kernel_stats.csv | cut -c1-160
stamp "3 launch floor of the C2 mel loop"
timeout 200 python tools/launch_floor.py 2>&1 | tail -1 | tee $O/r06s10_launch_floor.log
timeout 200 python tools/launch_floor.py --B 1 --T 750 2>&1 | tail -1 | tee -a $O/r06s10_launch_floor.log
stamp "4 C5: the full share of one GPU (32 references x 256 targets = 8192 pairs)"
timeout 900 python bench.py --config c5 --refs 32 --targets 256 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary 2>&1 | tail -1 > $O/r06s10_c5_full_share.json
python -c "import json;d=json.load(open('$O/r06s10_c5_full_share.json'));print(d['value'], d['ms_per_step'], d.get('style_cache'), d.get('plan_cache'))"
stamp "5 the whole GPU suite"
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -12 | tee $O/r06s10_gpu_suite.log
stamp "6 smoke"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/r06s10_smoke.log
stamp done
