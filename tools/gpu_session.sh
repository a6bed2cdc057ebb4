#!/bin/bash
# The ONE scratch script for a gpurun call (overwritten per call; the commands worth keeping move to tools/reproduce.sh).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_round3.py -x -q -k "f43 or gate16" 2>&1 | tail -6 | tee gpurun_out/s2_tests.log
echo "=== kbench: default lib (SLP), shared transform"
timeout 300 python tools/kbench.py --which wino43_16 --iters 60 2>&1 | tail -6 | tee gpurun_out/s2_kbench_default.txt
echo "=== kbench: default lib, plain transform"
SS_GATE16_PLAIN=1 timeout 300 python tools/kbench.py --which wino43_16 --iters 60 --mt 3,2 2>&1 | tail -4 | tee gpurun_out/s2_kbench_plain.txt
echo "=== kbench: -fno-slp-vectorize lib, shared / plain"
SS_LIB_PATH=$R/stylesinger_amd/_abl/lib_noslp.so timeout 300 python tools/kbench.py --which wino43_16 --iters 60 --mt 3,2 2>&1 | tail -4 | tee gpurun_out/s2_kbench_noslp.txt
SS_LIB_PATH=$R/stylesinger_amd/_abl/lib_noslp.so SS_GATE16_PLAIN=1 timeout 300 python tools/kbench.py --which wino43_16 --iters 60 --mt 3 2>&1 | tail -2 | tee gpurun_out/s2_kbench_noslp_plain.txt
echo "=== kbench at 48000 frames (B=32)"
timeout 300 python tools/kbench.py --which wino43_16 --net mel --B 32 --iters 30 2>&1 | tail -3 | tee gpurun_out/s2_kbench_b32.txt
echo "=== PMC"
K="python $R/tools/kbench.py --which wino43_16 --net mel --iters 20 --mt 3"
timeout 200 bash tools/pmc.sh g16_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -- $K 2>&1 | tee gpurun_out/s2_pmc_sq1.txt
timeout 200 bash tools/pmc.sh g16_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS -- $K 2>&1 | tee gpurun_out/s2_pmc_lds.txt
timeout 200 bash tools/pmc.sh g16_grbm GRBM_GUI_ACTIVE -- $K 2>&1 | tee gpurun_out/s2_pmc_grbm.txt
echo "=== bench A/B on this box"
for v in "SS_GATE16=0 SS_BENCH_STREAMS=1" "SS_GATE16=1 SS_BENCH_STREAMS=1" "SS_GATE16=0 SS_BENCH_STREAMS=3" "SS_GATE16=1 SS_BENCH_STREAMS=3"; do
  echo "--- $v"
  env $v timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "^\{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print(d['value'], d['ms_per_step'], d['config']['step_overlap'], r.get('kernel'), r.get('us_per_launch'), r.get('executed_mfma_frac'))"
done 2>&1 | tee gpurun_out/s2_bench_ab.txt
