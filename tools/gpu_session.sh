#!/bin/bash
# The ONE scratch script for a gpurun call (overwritten per call; the commands worth keeping move to tools/reproduce.sh).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
export SS_BF16_CHAIN_L1=0.0025
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 | tee gpurun_out/s3_tests.log
echo "=== ablations"
bash tools/ablate_g16.sh run 2>&1 | tee gpurun_out/s3_ablate_g16.txt
echo "=== PMC"
K="python $R/tools/kbench.py --which wino43_16 --net mel --iters 20 --mt 3"
timeout 200 bash tools/pmc.sh g16_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -- $K 2>&1 | tee gpurun_out/s3_pmc_sq1.txt
timeout 200 bash tools/pmc.sh g16_grbm GRBM_GUI_ACTIVE -- $K 2>&1 | tee gpurun_out/s3_pmc_grbm.txt
timeout 200 bash tools/pmc.sh g16_fetch FETCH_SIZE -- $K 2>&1 | tee gpurun_out/s3_pmc_fetch.txt
timeout 200 bash tools/pmc.sh g16_write WRITE_SIZE -- $K 2>&1 | tee gpurun_out/s3_pmc_write.txt
echo "=== bench"
pp() { grep -E "^\{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; o=d.get('one_batch_at_a_time') or {}; print(round(d['value']), round(d['ms_per_step'],1), d['config']['step_overlap'][:20], '| one-batch', o.get('ms_per_step'), '|', (r.get('kernel') or '')[:40], r.get('us_per_launch'), r.get('executed_mfma_frac'), r.get('clock_ghz'))"; }
echo "--- gate16 default, 1 stream"; timeout 300 python bench.py --streams 1 --steps 4 --warmup 1 --no-cpu-baseline 2>&1 | pp
echo "--- gate16 default, 3 streams, 12 steps"; timeout 300 python bench.py --streams 3 --steps 12 --warmup 3 --no-cpu-baseline 2>&1 | pp
echo "--- SS_GATE16=0, 3 streams, 12 steps"; SS_GATE16=0 timeout 300 python bench.py --streams 3 --steps 12 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | pp
echo "--- half batches (B=4) on 2 streams, 8 steps = 4 full batches"; timeout 300 python bench.py --batch 4 --streams 2 --steps 8 --warmup 2 --no-cpu-baseline --no-roofline 2>&1 | pp
echo "--- SS_GATE16=2 (mt=2), 1 stream"; SS_GATE16=2 timeout 300 python bench.py --streams 1 --steps 4 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | pp
