#!/bin/bash
# final-record session for the F(4,3) default: PMC of the new dominant kernel, full GPU suite, bench lines, kernel stats
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
K="python $R/tools/kbench.py --which wino43 --net mel --iters 20"
timeout 300 bash tools/pmc.sh w43_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -- $K
timeout 300 bash tools/pmc.sh w43_grbm GRBM_GUI_ACTIVE -- $K
timeout 300 bash tools/pmc.sh w43_fetch FETCH_SIZE -- $K
timeout 300 bash tools/pmc.sh w43_write WRITE_SIZE -- $K
timeout 300 bash tools/pmc.sh w43_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- $K
echo "=== pytest"
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -4
echo "=== smoke"
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2
echo "=== bench c2"
timeout 900 python bench.py --steps 9 --warmup 3 2>&1 | grep -E "^\{" > gpurun_out/r02_bench_c2.json; cut -c1-400 gpurun_out/r02_bench_c2.json
echo "=== rocprof 1 stream"
(cd /tmp && export TMPDIR=/tmp && SS_BENCH_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02b -o r02b -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_r02b.log 2>&1)
grep -E "^\{" gpurun_out/prof_r02b.log > gpurun_out/r02_bench_c2_1stream_under_rocprof.json; cut -c1-200 gpurun_out/r02_bench_c2_1stream_under_rocprof.json
f=$(find gpurun_out/prof_r02b -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r02_bench_c2_1stream_kernel_stats.csv; head -12 "$f" | cut -c1-200
echo "=== bench c5"
timeout 900 python bench.py --config c5 --no-cpu-baseline 2>&1 | grep -E "^\{" > gpurun_out/r02_bench_c5_sweep.json; cut -c1-300 gpurun_out/r02_bench_c5_sweep.json
echo "=== bench c4"
timeout 900 python bench.py --config c4 --steps 1 --warmup 1 --streams 1 --no-cpu-baseline 2>&1 | grep -E "^\{" > gpurun_out/r02_bench_c4_bf16.json; cut -c1-300 gpurun_out/r02_bench_c4_bf16.json
