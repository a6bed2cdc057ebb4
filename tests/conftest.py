import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def record_measurement(name, **vals):
    """Append a measured parity figure to gpurun_out/parity_measurements.jsonl (scratch on the GPU box, merged back by gpurun):
    the numbers DESIGN.md §4 tabulates come from here, not from reading test output."""
    import json
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "parity_measurements.jsonl"), "a") as fh:
        fh.write(json.dumps(dict(name=name, **vals)) + "\n")
