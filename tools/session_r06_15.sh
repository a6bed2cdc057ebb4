#!/bin/bash
# Round 6, GPU session 15: compact one-term skip weights (one_product = 2); where a layer512 launch's time goes beyond its tile periods
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out; mkdir -p $O
T0=$(date +%s)
stamp() { echo "=== [$(( $(date +%s) - T0 )) s] $*" | tee -a $O/r06s15_timeline.log; }
stamp "1 tests"
timeout 900 python -m pytest tests/test_gpu_round6.py -q -s -m gpu -k "skip_gemm" 2>&1 | tail -6 | tee $O/r06s15_tests.log
timeout 1500 python -m pytest tests/test_gpu_layer512.py -q -s -m gpu -k "fp16sd" 2>&1 | tail -6 | tee -a $O/r06s15_tests.log
stamp "2 C4 end to end, fp16sd"
timeout 900 python bench.py --config c4sd --streams 1 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary 2>&1 | tail -1 > $O/r06s15_c4sd.json
python -c "import json;d=json.load(open('$O/r06s15_c4sd.json'));print(d['value'], d['ms_per_step'], d['parity']['measured_in_this_run'], d['roofline'].get('us_per_launch'), d['roofline'].get('frac'), d['roofline'].get('clock_ghz'))"
stamp "3 kernel stats of the fp16sd C4 loop (20 steps)"
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_c4sd15 -o c4sd -- \
   python $GRAFT_REPO_ROOT/bench.py --config c4sd --diff-steps 20 --streams 1 --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/$O/r06s15_prof_c4sd.log 2>&1)
cp "$(find $O/prof_c4sd15 -name '*kernel_stats.csv' | head -1)" $O/r06s15_c4sd_kernel_stats.csv; grep -E "layer512|tile256s_kernel<0" $O/r06s15_c4sd_kernel_stats.csv | cut -c1-170
rm -rf $O/prof_c4sd15
stamp "4 trace of the one-product kernel: the launch on its own time axis"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSS_L512_TRACE -c stylesinger_amd/csrc/layer512.hip -o /tmp/l512t.o && hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libss_l512trace.so /tmp/l512t.o $(ls stylesinger_amd/_obj/*.o | grep -v layer512) && SS_LIB_PATH=/tmp/libss_l512trace.so timeout 200 python tools/trace_layer512.py --one 2>&1 | tail -24 | tee $O/r06s15_trace_one_product.log
SS_LIB_PATH=/tmp/libss_l512trace.so timeout 200 python tools/trace_layer512.py 2>&1 | tail -24 | tee $O/r06s15_trace_two_products.log
stamp done
