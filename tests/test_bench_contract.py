"""The driver's bench contract: one JSON line with the keys the prompt names, a roofline object for
the dominant kernel and a cpu_baseline; checked on the committed line (CPU tier) and on a short live
run (GPU tier)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
        "scaling", "vs_baseline", "dtype", "data", "config", "roofline"}


def _check(line, need_cpu):
    assert KEYS <= set(line), KEYS - set(line)
    assert line["n_gpus"] == 1 and line["higher_is_better"] is True and line["scaling"] == "weak"
    assert line["vs_baseline"] is None and line["dtype"] == "f64" and line["data"] == "synthetic"
    assert "workload" in line["config"] and "model" not in line["config"]
    assert abs(line["value"] - 1e3 / line["ms_per_step"]) / line["value"] < 1e-6
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.3 < r["frac"] < 1.0
    assert r["traffic"] is None or r["traffic"] >= r["algorithmic_bytes"] * 0.99
    if need_cpu:
        c = line["cpu_baseline"]
        assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    for s in line.get("secondary", []):
        rr = s["roofline"]
        assert rr["bound"] in ("hbm", "mfma") and abs(rr["frac"] - rr["achieved"] / rr["peak"]) < 1e-9


def test_committed_bench_line_meets_the_contract():
    with open(os.path.join(ROOT, "profiles", "r02_bench_line.json")) as f:
        line = json.loads(f.read())
    _check(line, need_cpu=True)
    names = " ".join(s["config"] for s in line["secondary"])
    for cfg in ("cfg3b", "cfg3a", "cfg1b", "cfg4", "cfg5"):      # every other BASELINE config
        assert cfg in names, cfg


@pytest.mark.gpu
def test_live_bench_line_meets_the_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5",
                          "--no-secondary", "--no-cpu-baseline"], capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["steps"] == 20 and line["warmup"] == 5
    _check(line, need_cpu=False)
