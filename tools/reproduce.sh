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
#   kbench-layer512  round 6: ss_layer512 (one launch per residual layer of the fp16x2 mel stack) against the launch pair it replaces, C4 shape
#   trace-layer512   per-phase shader-clock timing of layer512_kernel (builds the -DSS_L512_TRACE library first; hipcc works on the box)
#   pmc-layer512 [--one --e16]  PMC passes on layer512_kernel (profiles/r06_pmc_layer512.json; with --one --e16 the fp16sd form: r06_pmc_layer512sd.json)
#   profile-c4 [cfg] rocprofv3 kernel stats of the C4 loop at 20 diffusion steps (cfg = c4 (fp16sd, default) | c4x2: profiles/r06_bench_c4_fp16sd_20steps_kernel_stats.csv,
#                    r06_bench_c4x2_20steps_kernel_stats.csv)
#   phases-layer512  the layer launch with its half tiles first / last and as whole tiles (knob layer512_tail; profiles/r06_kbench_layer512_phase_shift.log)
#   kbench-final     the mel output projection + DDPM update per tile choice, with / without the in-kernel noise (profiles/r06_kbench_final_projection.log)
#   kbench-skip      the K = L C skip GEMM: pair layout, compact A, compact A + W (64-channel steps; SS_SKIP_DENSE=0: 32) (profiles/r06_kbench_skip_gemm_dense.log)
#   groups-layer512  EXPERIMENT (own shared object): the fp16sd layer launch as two wave groups half a period apart (profiles/r06_kbench_layer512_groups.log)
#   numerics-sd      CPU: the fp16sd numerics study (weight sets; --e-sets=N: the conditioner addend as N fp16 sets)
#   power            rocm-smi power / sclk sampled under 12 s of back-to-back layer launches (1400 W = the cap)
#   launch-floor     null-kernel hipGraph with the C2 mel loop's launch topology (tools/launch_floor.py)
#   c5-full          the full per-GPU share of BASELINE configs[4]: 32 references x 256 targets (profiles/r06_bench_c5_full_share.json)
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
  kbench-layer512)
    python tools/kbench_layer512.py --iters 400
    python tools/kbench_layer512.py --iters 400 --one --which layer512
    SS_LAYER512_TAIL=0 python tools/kbench_layer512.py --iters 400 --which fused ;;
  trace-layer512)
    mkdir -p stylesinger_amd/_abl
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSS_L512_TRACE -c stylesinger_amd/csrc/layer512.hip -o /tmp/l512t.o && \
    hipcc --offload-arch=gfx950 -shared -fPIC -o stylesinger_amd/_abl/libss_l512trace.so /tmp/l512t.o $(ls stylesinger_amd/_obj/*.o | grep -v layer512) && \
    SS_LIB_PATH=stylesinger_amd/_abl/libss_l512trace.so python tools/trace_layer512.py "$@" ;;
  pmc-layer512)
    bash tools/pmc_layer512.sh "$@" ;;
  kbench-final)
    python tools/kbench_final.py "$@" ;;
  kbench-skip)
    python tools/kbench_h.py --f16 --which skip --iters 200
    python tools/kbench_h.py --f16 --which skip --compact 1 --iters 200
    python tools/kbench_h.py --f16 --which skip --compact 2 --iters 200
    SS_SKIP_DENSE=0 python tools/kbench_h.py --f16 --which skip --compact 2 --iters 200 ;;
  groups-layer512)
    python tools/kbench_layer512_groups.py "$@" ;;
  numerics-sd)
    python -m oracle.dither_numerics 1 2 4 8 16 32
    python -m oracle.dither_numerics 32 --e-sets=1 --e-sets=4 --e-sets=8 --e-sets=16 ;;
  phases-layer512)
    for k in 1 2 0; do echo "--- layer512_tail = $k"; SS_LAYER512_TAIL=$k python tools/kbench_layer512.py --one --iters 400 --which fused; SS_LAYER512_TAIL=$k python tools/kbench_layer512.py --iters 400 --which fused; done ;;
  profile-c4)
    (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c4 -o c4 -- \
       python $R/bench.py --config ${1:-c4} --diff-steps 20 --streams 1 --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/prof_c4.log 2>&1)
    head -12 "$(find gpurun_out/prof_c4 -name '*kernel_stats.csv' | head -1)" | cut -c1-220 ;;
  power)
    ( for i in $(seq 1 30); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | tr '\n' ';'; echo; sleep 0.3; done ) > gpurun_out/smi_under_load.log 2>&1 &
    python tools/kbench_layer512.py --iters 30000 --which fused | tail -1
    wait; sort gpurun_out/smi_under_load.log | uniq -c | sort -rn | head -8 ;;
  launch-floor)
    python tools/launch_floor.py; python tools/launch_floor.py --B 1 --T 750 ;;
  c5-full)
    python bench.py --config c5 --refs 32 --targets 256 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary | tail -1 | cut -c1-1200 ;;
  ubench)
    for f in mfma_valu mfma16 glds_probe l2bw soffset_probe mfma_mx_layout cvt_fp4_probe; do
      hipcc --offload-arch=gfx950 -O3 tools/ubench/$f.hip -o /tmp/$f 2>/dev/null && /tmp/$f
    done ;;
  *) echo "unknown section '$sec'"; sed -n 2,15p $0; exit 2 ;;
esac
