"""Merge gpurun_out/parity_measurements.jsonl (written by the GPU tests through tests/conftest.py::record_measurement; gpurun merges the box's
file back, so it holds the LAST session's records only) into profiles/r06_parity.json (the round's file; --out picks another). A later record of the same name replaces the earlier one.
    python tools/collect_parity.py [--out profiles/r06_parity.json]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "profiles", "r06_parity.json")
if "--out" in sys.argv:
    out = sys.argv[sys.argv.index("--out") + 1]
src = os.path.join(ROOT, "gpurun_out", "parity_measurements.jsonl")
rec = {"source": "tests/conftest.py::record_measurement on MI355X (gpurun), round 6; merged by tools/collect_parity.py", "measurements": {}}
if os.path.exists(out):
    rec = json.load(open(out))
n = 0
for line in open(src):
    line = line.strip()
    if line:
        d = json.loads(line)
        rec["measurements"][d.pop("name")] = d
        n += 1
json.dump(rec, open(out, "w"), indent=1)
print(f"{n} records merged -> {out} ({len(rec['measurements'])} names)")
