#!/bin/bash
cd $GRAFT_REPO_ROOT
for t in 128 64; do echo "--- SS_HTILE=$t"; SS_HTILE=$t timeout 300 python tools/kbench_h.py 2>&1 | tail -2; done
