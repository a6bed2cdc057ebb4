#!/bin/bash
# helper run on the GPU box by gpurun: args = what to run
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
prof() {  # prof <name> <cmd...>: rocprofv3 kernel trace + stats of a command, first 30 rows of the kernel stats
  name=$1; shift
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$name -o $name -- "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$name.log 2>&1)
  f=$(find gpurun_out/prof_$name -name "*kernel_stats.csv" | head -1)
  grep -E "^\{" gpurun_out/prof_$name.log | cut -c1-400
  head -25 "$f" | cut -c1-200
}
case "$1" in
  tests) python -m pytest tests -x -q -m gpu -s 2>&1 | tail -80 | tee gpurun_out/tests.log ;;
  kernels) python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -40 | tee gpurun_out/kernels.log ;;
  parity) python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s 2>&1 | tail -60 | tee gpurun_out/parity.log ;;
  smoke) python __graft_entry__.py smoke 2>&1 | tail -5 | tee gpurun_out/smoke.log ;;
  bench) shift; python bench.py "$@" 2>&1 | tail -5 | tee gpurun_out/bench.log ;;
  prof) shift; prof r2 python $GRAFT_REPO_ROOT/bench.py "$@" ;;
  pmc) shift; bash tools/pmc.sh "$@" ;;
  *) "$@" ;;
esac
