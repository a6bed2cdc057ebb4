#!/bin/bash
# Round 6, GPU session 7: layer512 with the gate arithmetic ahead of [B2]; power experiment: lo terms with few mantissa bits / zero
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_layer512.py -q -s -m gpu -x -k "float64 or many_tiles" 2>&1 | tail -4 | tee $O/r06s7_tests_layer512.log
SS_LIB_PATH=stylesinger_amd/_abl/libss_l512trace.so timeout 200 python tools/trace_layer512.py 2>&1 | tail -12 | tee $O/r06s7_trace_fused.log
timeout 300 python tools/kbench_layer512.py 2>&1 | tail -6 | tee $O/r06s7_kbench_layer512.log
for b in 3 1; do
echo "--- lo terms with $b mantissa bits" | tee -a $O/r06s7_kbench_lo_bits.log
timeout 300 python tools/kbench_layer512.py --lo-bits $b 2>&1 | tail -5 | head -4 | tee -a $O/r06s7_kbench_lo_bits.log
done
echo "--- lo terms zero" | tee -a $O/r06s7_kbench_lo_bits.log
timeout 300 python tools/kbench_layer512.py --zero-lo 2>&1 | tail -5 | head -4 | tee -a $O/r06s7_kbench_lo_bits.log
timeout 600 python bench.py --config c4 --streams 1 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary 2>&1 | tail -1 > $O/r06s7_c4_fused.json
python -c "import json;d=json.load(open('$O/r06s7_c4_fused.json'));print(d['value'], d['ms_per_step'], d['parity']['measured_in_this_run'])"
