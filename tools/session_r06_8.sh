#!/bin/bash
# Round 6, GPU session 8: what does the chip report (power, clocks) while the C4 layer launches run back to back? Then rocprofv3 kernel stats of
# the C4 loop on the fused path, and PMC passes on layer512_kernel.
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out; mkdir -p $O
T0=$(date +%s)
stamp() { echo "=== [$(( $(date +%s) - T0 )) s] $*" | tee -a $O/r06s8_timeline.log; }
stamp "1 power / clocks under load"
( for i in $(seq 1 40); do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (edge|junction|hotspot)" | tr '\n' ';'; echo; sleep 0.25; done ) > $O/r06s8_smi_under_load.log 2>&1 &
SMI=$!
sleep 1
timeout 120 python tools/kbench_layer512.py --iters 4000 --which fused 2>&1 | tail -2 | tee $O/r06s8_kbench_long.log
wait $SMI
head -3 $O/r06s8_smi_under_load.log; tail -3 $O/r06s8_smi_under_load.log
rocm-smi --showmaxpower --showpower 2>/dev/null | grep -iE "power|cap" | head -5 | tee -a $O/r06s8_smi_under_load.log
stamp "2 rocprofv3 kernel stats of the C4 loop (20 diffusion steps), fused path"
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_c4 -o c4 -- \
   python $GRAFT_REPO_ROOT/bench.py --config c4 --diff-steps 20 --streams 1 --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/$O/r06s8_prof_c4.log 2>&1)
grep -E "^\{" $O/r06s8_prof_c4.log | cut -c1-300
F=$(find $O/prof_c4 -name '*kernel_stats.csv' | head -1); cp "$F" $O/r06s8_c4_kernel_stats.csv; head -12 $O/r06s8_c4_kernel_stats.csv | cut -c1-200
stamp "3 PMC passes on layer512_kernel"
bash tools/pmc_layer512.sh 2>&1 | tee $O/r06s8_pmc_layer512.log | tail -60
stamp done
