# PMC passes (one counter block per rocprofv3 run, --kernel-trace --pmc only) on layer512_kernel<true> at the BASELINE config 4 shape:
#   bash tools/pmc_layer512.sh
# matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128); clock under collection = GRBM_GUI_ACTIVE / 8 XCDs / wall
K="python $GRAFT_REPO_ROOT/tools/kbench_layer512.py --which fused --iters 20 $*"
cd $GRAFT_REPO_ROOT
$K
T=l512
bash tools/pmc.sh ${T}_sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -- $K
bash tools/pmc.sh ${T}_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS -- $K
bash tools/pmc.sh ${T}_grbm GRBM_GUI_ACTIVE -- $K
bash tools/pmc.sh ${T}_FETCH_SIZE FETCH_SIZE -- $K
bash tools/pmc.sh ${T}_WRITE_SIZE WRITE_SIZE -- $K
bash tools/pmc.sh ${T}_tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum -- $K
