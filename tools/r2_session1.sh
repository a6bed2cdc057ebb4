#!/bin/bash
# GPU session 1 of round 2: new tests, C2 bench, kernel stats, PMC passes of the dominant kernels (one counter block per pass).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -s 2>&1 | grep -v "^$" | tail -150 > gpurun_out/tests.log
tail -15 gpurun_out/tests.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err; cut -c1-600 gpurun_out/bench_c2.json; tail -3 gpurun_out/bench_c2.err
(cd /tmp && export TMPDIR=/tmp && timeout 60 rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/counters.txt 2>&1); grep -c . gpurun_out/counters.txt
K="python tools/kbench.py --which wino --net mel --iters 20"
timeout 300 bash tools/pmc.sh wino_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -- $K
timeout 300 bash tools/pmc.sh wino_sq2 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD -- $K
timeout 300 bash tools/pmc.sh wino_grbm GRBM_GUI_ACTIVE -- $K
timeout 300 bash tools/pmc.sh wino_fetch FETCH_SIZE -- $K
timeout 300 bash tools/pmc.sh wino_write WRITE_SIZE -- $K
K2="python tools/kbench.py --which resskip --net mel --iters 20 --tile 3"
timeout 300 bash tools/pmc.sh res_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -- $K2
timeout 300 bash tools/pmc.sh res_grbm GRBM_GUI_ACTIVE -- $K2
