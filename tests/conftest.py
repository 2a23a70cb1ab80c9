import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_sessionfinish(session, exitstatus):
    """GPU sessions: the measured multiples of the reference's own fp32-vs-fp64 drift (tests/parity_common.py: MEASURED_SPREAD) are
    written next to the other records of the run, so that the bound OWN_SPREAD can be compared with what the hardware did."""
    try:
        from tests import parity_common as pc
        import torch
        rows = pc.MEASURED_SPREAD
        out_dir = os.path.join(ROOT, "gpurun_out")
        if rows and torch.cuda.is_available() and os.path.isdir(out_dir):
            import json
            worst = {}
            for r in rows:
                key = ("GEMM kernels" if r["pinned_gemm"] else "default kernels") + " / " + r["precision"]
                worst[key] = max(worst.get(key, 0.0), r["multiple"])
            json.dump(dict(bound=pc.OWN_SPREAD, largest_multiple_by_family=worst, rows=rows),
                      open(os.path.join(out_dir, "own_spread_measured.json"), "w"), indent=1)
    except Exception:       # a record, never a reason to fail a session
        pass
