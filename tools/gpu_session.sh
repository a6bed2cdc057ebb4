#!/bin/bash
# The ONE scratch script for a gpurun call (overwritten per call; the commands worth keeping move to tools/reproduce.sh).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export SS_BF16_CHAIN_L1=1.0
python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py::test_single_utterance_entrypoint_matches_batched_path \
  tests/test_gpu_round2.py::test_c2_batch_item_matches_oracle_at_full_size_and_100_steps \
  tests/test_gpu_round2.py::test_bf16_mode_on_the_1000_step_golden_reports_its_distance_to_the_fp32_reference \
  -x -q -s 2>&1 | tail -25 | tee gpurun_out/s1_tests.log
tools/ubench/mfma16.bin 2>&1 | tee gpurun_out/s1_mfma16.txt
python bench.py --streams 1 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-600 | tee gpurun_out/s1_bench_1stream.json
SS_STREAMS=2 python bench.py --streams 1 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-300 | tee gpurun_out/s1_bench_split2.json
