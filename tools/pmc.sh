#!/bin/bash
# usage: tools/pmc.sh <outname> <counters...> -- <command...>   (run on the GPU box; separate pass per call)
out=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp
# the command runs from /tmp: give script paths relative to the repo root as $GRAFT_REPO_ROOT/...
rocprofv3 --kernel-trace --pmc "${ctrs[@]}" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$out -o $out -- "$@" > /tmp/pmc_$out.log 2>&1
grep -E "TF/s|rror|No such" /tmp/pmc_$out.log | tail -4
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_$out/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"][:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if any(t in k for t in ("conv_gemm", "attention", "wino", "gemm16", "gate256", "gate128", "tile256", "gemm_bf16", "layer512")):
        print(k)
        for c, v in d.items(): print(f"    {c:32s} avg/dispatch = {v / cnt[(k, c)]:.4g}   (n={cnt[(k,c)]})")
PY
