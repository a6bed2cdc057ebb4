#!/bin/bash
# Round 6, GPU session 43: rocprofv3 kernel stats of C2 (one stream) on the final tree
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out; mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof43_c2 -o c2 -- \
   python $GRAFT_REPO_ROOT/bench.py --streams 1 --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/$O/r06s43_prof_c2.log 2>&1)
grep -E "^\{" $O/r06s43_prof_c2.log | cut -c1-160
cp "$(find $O/prof43_c2 -name '*kernel_stats.csv' | head -1)" $O/r06s43_c2_kernel_stats.csv; head -16 $O/r06s43_c2_kernel_stats.csv | cut -c1-170
rm -rf $O/prof43_c2
