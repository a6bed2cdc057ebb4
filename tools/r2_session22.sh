#!/bin/bash
cd $GRAFT_REPO_ROOT
for lay in row layer; do
  echo "--- E layout $lay"
  timeout 200 python tools/kbench.py --which wino43 --net mel --iters 40 --e-layout $lay 2>&1 | tail -1
  timeout 200 python tools/kbench.py --which wino43 --net f0 --B 16 --iters 40 --e-layout $lay 2>&1 | tail -1
  timeout 200 python tools/kbench.py --which wino43 --net mel --B 32 --iters 20 --e-layout $lay 2>&1 | tail -1
  timeout 300 python tools/kbench_h.py --which gate --e-layout $lay 2>&1 | tail -1
done
