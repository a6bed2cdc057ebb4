#!/bin/bash
# Round 6, GPU session 12: "fp16sd" - one fp16 product per GEMM with noise-shaped weight sets: kernel test, model vs the real reference, kbench, C4 end to end
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out; mkdir -p $O
T0=$(date +%s)
stamp() { echo "=== [$(( $(date +%s) - T0 )) s] $*" | tee -a $O/r06s12_timeline.log; }
stamp "1 kernel: one product"
timeout 600 python -m pytest tests/test_gpu_layer512.py -q -s -m gpu -x -k "one_product" 2>&1 | tail -5 | tee $O/r06s12_test_one_product.log
stamp "2 model: fp16sd vs the real reference (T = 32 x 1000 steps generic + forced fused; T = 5625 x 1000 steps forced fused)"
timeout 1500 python -m pytest tests/test_gpu_layer512.py -q -s -m gpu -k "fp16sd" 2>&1 | tail -8 | tee $O/r06s12_test_fp16sd.log
stamp "3 kbench: one product vs two (400 launches)"
timeout 300 python tools/kbench_layer512.py --iters 400 --one 2>&1 | tail -5 | tee $O/r06s12_kbench_one_product.log
timeout 300 python tools/kbench_layer512.py --iters 400 --which fused 2>&1 | tail -1 | tee -a $O/r06s12_kbench_one_product.log
stamp "4 C4 end to end, fp16sd"
timeout 900 python bench.py --config c4sd --streams 1 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary 2>&1 | tail -1 > $O/r06s12_c4sd.json
python -c "import json;d=json.load(open('$O/r06s12_c4sd.json'));print(d['value'], d['ms_per_step'], d['parity']['measured_in_this_run'], d['roofline'].get('us_per_launch'), d['roofline'].get('frac'))"
stamp done
