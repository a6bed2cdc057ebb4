#!/bin/bash
# GPU session 2 of round 2: prodiff tests, PMC passes (absolute paths), wave-priority A/B, multi-stream bench modes.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "prodiff or mel1000" 2>&1 | grep -v "^$" | tail -30 > gpurun_out/tests2.log; tail -8 gpurun_out/tests2.log
K="python $R/tools/kbench.py --which wino --net mel --iters 20"
timeout 300 bash tools/pmc.sh wino_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -- $K
timeout 300 bash tools/pmc.sh wino_sq2 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD -- $K
timeout 300 bash tools/pmc.sh wino_sq3 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES -- $K
timeout 300 bash tools/pmc.sh wino_grbm GRBM_GUI_ACTIVE -- $K
timeout 300 bash tools/pmc.sh wino_fetch FETCH_SIZE -- $K
timeout 300 bash tools/pmc.sh wino_write WRITE_SIZE -- $K
K2="python $R/tools/kbench.py --which resskip --net mel --iters 20 --tile 3"
timeout 300 bash tools/pmc.sh res_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -- $K2
timeout 300 bash tools/pmc.sh res_grbm GRBM_GUI_ACTIVE -- $K2
echo "=== wave priority A/B (kbench)"
for p in 0 1 2; do
  echo "--- prio $p"
  timeout 200 python tools/kbench.py --which wino --net mel --iters 40 --prio $p 2>&1 | tail -1
  timeout 200 python tools/kbench.py --which wino --net f0 --B 16 --iters 40 --prio $p 2>&1 | tail -1
  timeout 200 python tools/kbench.py --which resskip --net mel --iters 60 --tile 3 --prio $p 2>&1 | tail -1
  timeout 200 python tools/kbench.py --which voc --iters 30 --prio $p 2>&1 | tail -5
  timeout 200 python tools/kbench_skip.py $p 2>&1 | tail -2
done
echo "=== bench variants"
for v in "SS_WAVE_PRIO=0" "SS_WAVE_PRIO=1" "SS_WAVE_PRIO=2" "SS_BENCH_STREAMS=2" "SS_BENCH_PIPELINE=1" "SS_BENCH_STREAMS=2 SS_WAVE_PRIO=1"; do
  echo "--- $v"
  env $v timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>&1 | grep -E "^\{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['step_overlap'])"
done
