#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_round2.py -q -m gpu -k "bench_two" 2>&1 | tail -3
(cd /tmp && export TMPDIR=/tmp && SS_BENCH_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02 -o r02 -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_r02.log 2>&1)
grep -E "^\{" gpurun_out/prof_r02.log | cut -c1-300
f=$(find gpurun_out/prof_r02 -name "*kernel_stats.csv" | head -1); head -30 "$f" | cut -c1-230
K="python $R/tools/kbench.py --which wino --net mel --iters 20"
timeout 300 bash tools/pmc.sh v2_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -- $K
timeout 300 bash tools/pmc.sh v2_grbm GRBM_GUI_ACTIVE -- $K
timeout 300 bash tools/pmc.sh v2_fetch FETCH_SIZE -- $K
timeout 300 bash tools/pmc.sh v2_write WRITE_SIZE -- $K
