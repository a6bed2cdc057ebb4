#!/bin/bash
# Round 6, GPU session 32: the output projection + DDPM update on 128x32 tiles at configs[3]'s size: bit-identity with the 128x128 tile, C4 end to end
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out; mkdir -p $O
timeout 300 python tools/kbench_final.py --iters 50 2>&1 | grep -v amdgpu | head -3
timeout 900 python -m pytest tests/test_gpu_round5.py -q -s -m gpu -k "c4_batch_items" 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_layer512.py -q -s -m gpu -k "fp16sd and 5625" 2>&1 | tail -4
timeout 900 python bench.py --config c4 --streams 1 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary 2>&1 | tail -1 > $O/r06s32_c4.json
python -c "import json;d=json.load(open('$O/r06s32_c4.json'));print('c4', d['value'], d['ms_per_step'], d['parity']['measured_in_this_run'], d['roofline'].get('us_per_launch'))"
