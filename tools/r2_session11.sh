#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"
SQ2="SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD"
K="python $R/tools/kbench.py --which voc --iters 10"
timeout 300 bash tools/pmc.sh voc_sq1 $SQ1 -- $K
timeout 300 bash tools/pmc.sh voc_sq2 $SQ2 -- $K
timeout 300 bash tools/pmc.sh voc_grbm GRBM_GUI_ACTIVE -- $K
K2="python $R/tools/kbench_skip.py"
timeout 300 bash tools/pmc.sh skip_sq1 $SQ1 -- $K2
timeout 300 bash tools/pmc.sh skip_grbm GRBM_GUI_ACTIVE -- $K2
timeout 600 python -m pytest tests/test_gpu_round2.py -q -m gpu -k "in_flight" 2>&1 | tail -2
echo "=== c5"
timeout 900 python bench.py --config c5 --steps 2 --warmup 1 --streams 1 --no-cpu-baseline > gpurun_out/bench_c5.json 2> gpurun_out/bench_c5.err; cut -c1-300 gpurun_out/bench_c5.json
