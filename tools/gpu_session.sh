#!/bin/bash
# The ONE scratch script for a gpurun call (overwritten per call; the commands worth keeping move to tools/reproduce.sh).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
tools/ubench/glds_probe.bin 2>&1 | tee gpurun_out/s4_glds_probe.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_round3.py -x -q -k "f43 or gate16" 2>&1 | tail -4 | tee gpurun_out/s4_tests.log
echo "=== kbench"
timeout 300 python tools/kbench.py --which wino43_16 --iters 60 --mt -1,3,2 2>&1 | tail -6 | tee gpurun_out/s4_kbench.txt
echo "=== ablations"
bash tools/ablate_g16.sh run 2>&1 | tee gpurun_out/s4_ablate_g16.txt
echo "=== bench"
pp() { grep -E "^\{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; o=d.get('one_batch_at_a_time') or {}; print(round(d['value']), round(d['ms_per_step'],1), d['config']['step_overlap'][:20], '| one-batch', o.get('ms_per_step'), '|', (r.get('kernel') or '')[:40], r.get('us_per_launch'), r.get('executed_mfma_frac'), r.get('clock_ghz'))"; }
echo "--- gate16 default, 1 stream"; timeout 300 python bench.py --streams 1 --steps 4 --warmup 1 --no-cpu-baseline --no-secondary 2>&1 | pp
echo "--- gate16 default, 3 streams, 12 steps"; timeout 300 python bench.py --streams 3 --steps 12 --warmup 3 --no-cpu-baseline --no-secondary 2>&1 | pp
