#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu 2>&1 | tail -3
timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline 2>&1 | grep -E "^\{" > gpurun_out/bench_c2_probe.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_c2_probe.json'))
print(d['value'], d['ms_per_step']); print(json.dumps(d['roofline'], indent=1))
PY
