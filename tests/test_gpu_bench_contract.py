"""The driver's contract with bench.py (-m gpu): one JSON line on stdout with the agreed keys, the iteration `roofline` and the
`cpu_baseline` objects; refusals are one line on stderr with exit code 2.  A small shape keeps this to a few seconds; the
numbers themselves are not asserted (profiles/ holds the measured ones)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
        "data", "config", "roofline", "cpu_baseline"}


def _run(*args):
    env = dict(os.environ, PYTHONPATH=ROOT)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, cwd=ROOT, env=env,
                          timeout=600)


def test_bench_prints_one_json_line_with_the_contract_keys():
    r = _run("--shape", "6000,200,1500", "--steps", "6", "--warmup", "2", "--no-alt")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert KEYS <= set(d), KEYS - set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] - 1000.0) < 1e-6 * 1000.0
    assert "workload" in d["config"] and "model" not in d["config"]
    roof = d["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(roof) and roof["bound"] in ("hbm", "mfma")
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9 and 0 < roof["frac"] < 1
    # round 6: the line describes the device state it was measured in
    assert d["value_long"]["steps"] >= 20 and d["value_long"]["value"] > 0 and abs(d["value_long"]["value"] * d["value_long"]["ms_per_step"] - 1000.0) < 1e-3
    assert 0.05 < roof["update_TBps_actual"] < 8.0 and "update_traffic_source" in roof
    cpu = d["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(cpu) and cpu["kind"] in ("port", "reference") and cpu["cores"] >= 1


def test_bench_refuses_more_ranks_than_gpus_with_one_line():
    import torch
    n = torch.cuda.device_count() + 1
    r = _run("--gpus", str(n), "--steps", "2", "--warmup", "1")
    assert r.returncode == 2 and r.stdout.strip() == ""
    msg = [l for l in r.stderr.splitlines() if l.startswith("bench.py:")]
    assert len(msg) == 1 and "GPUs" in msg[0]
