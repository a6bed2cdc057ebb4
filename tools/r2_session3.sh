#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu.hip -o /tmp/mfma_valu 2>/dev/null && timeout 120 /tmp/mfma_valu | tee gpurun_out/ubench_mfma_valu.txt
for v in "SS_BENCH_STREAMS=2" "SS_BENCH_STREAMS=3" "SS_BENCH_STREAMS=4"; do
  echo "--- $v"
  env $v timeout 300 python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-roofline 2>&1 | grep -E "^\{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['step_overlap'])"
done
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "vocoder" 2>&1 | tail -3
