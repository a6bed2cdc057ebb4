#!/bin/bash
# Round 5, GPU session 1 (one gpurun call): the untested configurations first, then the C4 roofline evidence, then the cheap experiments.
# Every step has its own timeout and log under gpurun_out/ (merged back by gpurun).
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out; mkdir -p $O
T0=$(date +%s)
stamp() { echo "=== [$(( $(date +%s) - T0 )) s] $*" | tee -a $O/r05s1_timeline.log; }

stamp "1 round-5 tests + the 7 formerly skipped tests"
SS_TEST_FP16Q4=1 SS_TEST_SPEAKER=1 timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_fp16q4.py tests/test_gpu_speaker.py -q -s -m gpu 2>&1 | tail -70 > $O/r05s1_tests.log
tail -15 $O/r05s1_tests.log

stamp "2 kbench gate128 / gate128q back to back"
timeout 200 python tools/kbench_h.py --which gate --f16 --gate128 --iters 20 2>&1 | tail -4 > $O/r05s1_kbench_gate128.log
timeout 200 python tools/kbench_h.py --which gate --f16 --q4 --iters 20 2>&1 | tail -4 > $O/r05s1_kbench_gate128q.log
cat $O/r05s1_kbench_gate128.log $O/r05s1_kbench_gate128q.log

stamp "3 PMC passes on gate128_kernel"
timeout 600 bash tools/pmc_gate128.sh > $O/r05s1_pmc_gate128.log 2>&1
tail -40 $O/r05s1_pmc_gate128.log

stamp "4 rocprofv3 kernel stats of bench.py --config c4 --diff-steps 20"
( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_c4_20 -o c4 -- \
    python $GRAFT_REPO_ROOT/bench.py --config c4 --diff-steps 20 --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/$O/r05s1_prof_c4_20.json 2> /tmp/prof_c4.err )
tail -c 600 $O/r05s1_prof_c4_20.json
find $O/prof_c4_20 -name "*kernel_stats.csv" | head -1 | xargs -r head -12

stamp "5 C4 one batch vs two batches in flight"
timeout 400 python bench.py --config c4 --streams 1 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary 2>&1 | tail -1 > $O/r05s1_c4_1stream.json
cut -c1-300 $O/r05s1_c4_1stream.json
timeout 500 python bench.py --config c4 --streams 2 --steps 2 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline 2>&1 | tail -1 > $O/r05s1_c4_2streams.json
cut -c1-300 $O/r05s1_c4_2streams.json

stamp "6 C2 one batch at a time: batch halves on two streams inside the forward (SS_STREAMS=2) vs one stream"
SS_STREAMS=1 timeout 200 python bench.py --streams 1 --steps 8 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline 2>&1 | tail -1 | cut -c1-260 > $O/r05s1_c2_1batch_s1.json
SS_STREAMS=2 timeout 200 python bench.py --streams 1 --steps 8 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline 2>&1 | tail -1 | cut -c1-260 > $O/r05s1_c2_1batch_s2.json
timeout 200 python bench.py --batch 4 --streams 2 --steps 12 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline 2>&1 | tail -1 | cut -c1-260 > $O/r05s1_c2_b4x2.json
cat $O/r05s1_c2_1batch_s1.json $O/r05s1_c2_1batch_s2.json $O/r05s1_c2_b4x2.json

stamp "7 c3 emulated (8 shards on one device) + c4bf16x2 with the in-run parity block"
timeout 300 python bench.py --config c2 --emulate-ranks 8 --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-secondary --no-roofline 2>&1 | tail -1 > $O/r05s1_c3_emulated.json
python -c "import json;d=json.load(open('$O/r05s1_c3_emulated.json'));print(d.get('emulated'), d['ms_per_step'])"
timeout 400 python bench.py --config c4bf16x2 --streams 1 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary 2>&1 | tail -1 > $O/r05s1_c4bf16x2.json
python -c "import json;d=json.load(open('$O/r05s1_c4bf16x2.json'));print(d['value'], d['ms_per_step'], d.get('parity'))"
stamp "done"
