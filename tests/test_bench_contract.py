"""bench.py prints ONE JSON line with the fields the driver and the judge read (metric contract of the task)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_the_contract_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["higher_is_better"] is True and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert 0.3 < rf["frac"] < 1.0 and d["value"] > 50.0
    # the optional spectral-projection density solver is reported beside, on the same problem, with the same energy
    assert d["density_solver"]["name"] == "eigh"
    assert d["alt"]["density_solver"] == "sp2" and d["alt"]["value"] > 50.0 and abs(d["alt"]["energy_minus_eigh"]) < 1e-6


def test_bench_cpu_baseline_object_shape():
    """The cpu_baseline leg (the oracle on the host cores) on a tiny shape: keys of the contract."""
    sys.path.insert(0, ROOT)
    import bench
    cb = bench.cpu_baseline(24, 40, 4, budget_s=0.5)
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1
