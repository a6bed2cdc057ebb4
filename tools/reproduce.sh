#!/bin/bash
# How every measured number and profile in this repo is produced. Run ON THE GPU BOX from the repo root:
#     /usr/local/graft/bin/gpurun --timeout 1800 -- 'bash tools/reproduce.sh <section> [args]'
# Output goes to gpurun_out/ (scratch, merged back by gpurun); what is quoted in DESIGN.md / README.md is copied to profiles/.
# Sections:
#   tests            python -m pytest tests -m gpu            (the parity tests proper; measured distances -> parity_measurements.jsonl)
#   bench [args]     python bench.py [args]                   (the driver's line; default = c2 + cpu_baseline + secondary c5 / c4)
#   profile [args]   rocprofv3 --kernel-trace --stats of `bench.py --streams 1 --steps 4 --warmup 1 --no-cpu-baseline --no-secondary [args]`
#   pmc-gate         PMC passes on the dominant kernel launch (one counter block per pass, --kernel-trace --pmc only)
#   pmc-gate-c4x2    the same for the split-operand (bf16x2) 256x256 gate kernel at the C4 shape (tools/pmc_split.sh)
#   pmc-gate128 [--q4]  the same for the fp16x2 gate on 256x128 tiles, two workgroups per CU (tools/pmc_gate128.sh)
#   kbench           back-to-back timings of the hot launches (gate tilings, residual projection, vocoder convs direct vs grouped F(4,3))
#   kbench-c4        the 16-bit many-round launches at the BASELINE config 4 shape: fp16x2 gate on gate256 / gate128, residual projection on
#                    tile256 (tools/kbench_h.py)
#   c4-streams       BASELINE config 4 with two batches in flight (measured in round 5: slower)
#   fused            round 5: gate + residual projection of one layer as one dataflow launch vs two launches (DESIGN.md 7 lead 1)
#   c4q / c3-emulated  the fp16q4 line of config 4 / configs[2]'s 8 shards emulated on one device
#   pmc-gate128      PMC passes on gate128_kernel at the config-4 shape (profiles/r05_pmc_gate128.json)
#   ablate-gate16    timing ablations of the 16x16-tile gate kernel (debug builds: tools/ablate_g16.sh build, in the container)
#   ablate-res16     the same for the residual-projection kernel (tools/ablate_r16.sh build, in the container)
#   ubench           micro-benchmarks behind DESIGN.md §3.0 (VALU beside fp32 MFMA, 16x16x4 issue rate, DPP / LDS-DMA probes)
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
R=$PWD
mkdir -p gpurun_out
sec=$1; shift
case "$sec" in
  tests)
    python -m pytest tests -x -q -m gpu -s 2>&1 | tail -60 | tee gpurun_out/tests.log ;;
  bench)
    python bench.py "$@" 2>&1 | tail -3 | tee gpurun_out/bench.json ;;
  profile)
    (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- \
       python $R/bench.py --streams 1 --steps 4 --warmup 1 --no-cpu-baseline --no-secondary "$@" > $R/gpurun_out/prof_bench.log 2>&1)
    grep -E "^\{" gpurun_out/prof_bench.log | cut -c1-600
    head -30 "$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1)" | cut -c1-220 ;;
  pmc-gate)
    K="python $R/tools/kbench.py --which wino43_16 --net mel --iters 20 --e16 --mt ${1:-2}"
    bash tools/pmc.sh g16_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -- $K
    bash tools/pmc.sh g16_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS -- $K
    bash tools/pmc.sh g16_grbm GRBM_GUI_ACTIVE -- $K
    bash tools/pmc.sh g16_fetch FETCH_SIZE -- $K
    bash tools/pmc.sh g16_write WRITE_SIZE -- $K
    bash tools/pmc.sh g16_tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum -- $K ;;
  pmc-gate-c4x2)
    bash tools/pmc_split.sh ;;
  pmc-gate128)
    bash tools/pmc_gate128.sh "$@" ;;
  kbench)
    python tools/kbench.py --which wino43_16 --iters 60 --mt=-1,3,2
    python tools/kbench.py --which res16 --iters 60 --mt 6
    python tools/kbench.py --which voc --iters 30 ;;
  kbench-c4)
    SS_GATE128=0 python tools/kbench_h.py --which gate --f16
    python tools/kbench_h.py --which gate --f16 --gate128
    python tools/kbench_h.py --which gate --f16 --q4
    python tools/kbench_h.py --which res --f16 --pair-only
    python tools/kbench_h.py --which skip --f16 ;;
  fused)
    # round 5: one residual layer as ONE dataflow launch (gate + projection, per-row-tile counters; fences | write-through) vs the two launches, bit-identity
    # checked. The experiment is not in the library: tools/kbench_fused.py builds tools/experiments/libfused_gate_res.so (hipcc) when it is missing / stale
    python tools/kbench_fused.py
    python tools/kbench_fused.py --B 32 --T 1500
    python tools/kbench_fused.py --B 1 --T 750 ;;
  c4q)
    python bench.py --config c4q --streams 1 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary | tail -1 | cut -c1-600 ;;
  c3-emulated)
    python bench.py --config c2 --emulate-ranks 8 --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-secondary --no-roofline | tail -1 | cut -c1-900 ;;
  c4-streams)
    python bench.py --config c4 --streams 1 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-roofline | tail -1 | cut -c1-400
    python bench.py --config c4 --streams 2 --steps 2 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline | tail -1 | cut -c1-400 ;;
  ablate-gate16)
    bash tools/ablate_g16.sh run ;;
  ablate-res16)
    bash tools/ablate_r16.sh run ;;
  ubench)
    for f in mfma_valu mfma16 glds_probe l2bw soffset_probe mfma_mx_layout cvt_fp4_probe; do
      hipcc --offload-arch=gfx950 -O3 tools/ubench/$f.hip -o /tmp/$f 2>/dev/null && /tmp/$f
    done ;;
  *) echo "unknown section '$sec'"; sed -n 2,15p $0; exit 2 ;;
esac
