#!/bin/bash
# Round 6, GPU session 14: compact G (ss_layer512 g_compact -> ss_gemm_bf16 a_compact): tests, C4 fp16sd and fp16x2 end to end, kernel stats
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out; mkdir -p $O
T0=$(date +%s)
stamp() { echo "=== [$(( $(date +%s) - T0 )) s] $*" | tee -a $O/r06s14_timeline.log; }
stamp "1 tests"
timeout 900 python -m pytest tests/test_gpu_round6.py -q -s -m gpu -k "skip_gemm" 2>&1 | tail -6 | tee $O/r06s14_tests.log
timeout 1500 python -m pytest tests/test_gpu_layer512.py -q -s -m gpu 2>&1 | tail -16 | tee -a $O/r06s14_tests.log
for cfg in c4sd c4; do
stamp "2 C4 end to end, $cfg"
timeout 900 python bench.py --config $cfg --streams 1 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary 2>&1 | tail -1 > $O/r06s14_$cfg.json
python -c "import json;d=json.load(open('$O/r06s14_$cfg.json'));print(d['value'], d['ms_per_step'], d['parity']['measured_in_this_run'], d['roofline'].get('us_per_launch'), d['roofline'].get('frac'), d['roofline'].get('clock_ghz'))"
done
stamp "3 kernel stats of the fp16sd C4 loop (20 steps)"
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_c4sd -o c4sd -- \
   python $GRAFT_REPO_ROOT/bench.py --config c4sd --diff-steps 20 --streams 1 --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/$O/r06s14_prof_c4sd.log 2>&1)
cp "$(find $O/prof_c4sd -name '*kernel_stats.csv' | head -1)" $O/r06s14_c4sd_kernel_stats.csv; head -8 $O/r06s14_c4sd_kernel_stats.csv | cut -c1-170
stamp "4 kernel stats of the fp16x2 C4 loop (20 steps)"
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_c4 -o c4 -- \
   python $GRAFT_REPO_ROOT/bench.py --config c4 --diff-steps 20 --streams 1 --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/$O/r06s14_prof_c4.log 2>&1)
cp "$(find $O/prof_c4 -name '*kernel_stats.csv' | head -1)" $O/r06s14_c4_kernel_stats.csv; head -8 $O/r06s14_c4_kernel_stats.csv | cut -c1-170
stamp done
