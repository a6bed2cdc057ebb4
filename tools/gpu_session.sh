#!/bin/bash
# The ONE scratch script for a gpurun call (overwritten per call; the commands worth keeping move to tools/reproduce.sh).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_round3.py -x -q -k "preprocess" -s 2>&1 | tail -12 | tee gpurun_out/s5_tests.log
echo "=== ablations 7 8"
ABLS="7 8" bash tools/ablate_g16.sh run 2>&1 | tee gpurun_out/s5_ablate_g16.txt
echo "=== counters available"
(cd /tmp && rocprofv3 -L 2>/dev/null | grep -oE "\b(TA_[A-Z_0-9a-z]+|TCP_[A-Z_0-9a-z]+|TCC_[A-Z_0-9a-z]+|TD_[A-Z_0-9a-z]+)\b" | sort -u | tr '\n' ' ' | cut -c1-6000) | tee gpurun_out/s5_counters.txt
echo
K="python $R/tools/kbench.py --which wino43_16 --net mel --iters 20 --mt 3"
timeout 200 bash tools/pmc.sh g16_ta TA_BUSY_avr TA_BUSY_max TA_TA_BUSY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum -- $K 2>&1 | tail -12 | tee gpurun_out/s5_pmc_ta.txt
timeout 200 bash tools/pmc.sh g16_tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum -- $K 2>&1 | tail -12 | tee gpurun_out/s5_pmc_tcc.txt
timeout 200 bash tools/pmc.sh g16_tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum -- $K 2>&1 | tail -12 | tee gpurun_out/s5_pmc_tcp.txt
