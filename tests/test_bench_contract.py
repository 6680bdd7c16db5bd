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
        assert "error" not in s, s
        rr = s["roofline"]
        if rr["bound"] == "latency":      # config 4 with a vector state: us/step against the hand-off floor
            assert rr["us_per_step"] > 0 and rr["us_per_step_vs_handoff_floor"] > 0.5
            assert "frac" not in rr and 0 < rr["vs_restreamed_bound"] < 2
            continue
        assert rr["bound"] in ("hbm", "mfma") and abs(rr["frac"] - rr["achieved"] / rr["peak"]) < 1e-9


def test_committed_bench_line_meets_the_contract():
    with open(os.path.join(ROOT, "profiles", "r04_bench_line.json")) as f:
        line = json.loads(f.read())
    _check(line, need_cpu=True)
    names = " ".join(s["config"] for s in line["secondary"])
    for cfg in ("cfg3b", "cfg3a", "cfg1b", "cfg4", "cfg5", "placed"):    # every other BASELINE config
        assert cfg in names, cfg


def test_plain_python_launch_with_gpus_n_becomes_the_contract_launcher():
    """`python bench.py --gpus N` without WORLD_SIZE in the environment re-executes itself as
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py <the same arguments>` (shown, not run, under the dry-launch hook)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["AESARA_BENCH_DRY_LAUNCH"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20",
                          "--warmup", "5"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    cmd = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])["launch"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 0 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]


@pytest.mark.gpu
def test_plain_python_two_rank_launch_runs_on_one_device():
    """The same, for real: `python bench.py --gpus 2` (no launcher, no WORLD_SIZE) with the
    one-device gloo hooks — rank 0 prints the one JSON line of a 2-rank run."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(AESARA_BENCH_BACKEND="gloo", AESARA_BENCH_ONE_DEVICE="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "16",
                          "--warmup", "8", "--no-secondary"], capture_output=True, text=True, timeout=900,
                         cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["steps"] == 16
    assert line["transport"].startswith("torch.distributed (gloo)")       # RCCL needs one device per rank


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


@pytest.mark.gpu
def test_two_rank_bench_control_flow_on_one_device():
    """What the driver launches for N > 1 (`torch.distributed.run --nproc-per-node N bench.py
    --gpus N`), here with 2 ranks sharing cuda:0 over gloo (test hooks AESARA_BENCH_BACKEND /
    AESARA_BENCH_ONE_DEVICE): rotating inputs + ring slots + bucketed async all-reduce of the
    headline, the batch-sharded config-4 row, config 5 row-sharded with its packed fp64
    all-reduce, the placed-outputs row; max-over-ranks timing; ONE JSON line from rank 0."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, AESARA_BENCH_BACKEND="gloo", AESARA_BENCH_ONE_DEVICE="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "24", "--warmup", "8",
                          "--cfg5-log2n", "20"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 24 and line["scaling"] == "weak"
    assert abs(line["value"] - 2 * 1e3 / line["ms_per_step"]) / line["value"] < 1e-6     # whole-job evals/s
    assert "cpu_baseline" not in line
    rows = {r["config"].split(":")[0].split(" ")[0]: r for r in line["secondary"]}
    assert {"cfg4", "cfg5", "placed"} <= set(rows), list(rows)
    for r in line["secondary"]:
        assert "error" not in r, r
        assert r["n_gpus"] == 2 and r["scaling"] == "strong" and r["evals_per_s"] > 0
    assert rows["placed"]["placement"] == [[0, 2, 4, 6], [1, 3, 5, 7]]
    assert rows["cfg5"]["check"]["logp_rel_err"] <= 1e-6


@pytest.mark.gpu
def test_eight_rank_bench_control_flow_on_one_device():
    """The driver's 8-GPU launch line with all 8 ranks on cuda:0 over gloo: the headline's rotating
    inputs / ring slots / bucketed asynchronous all-reduce, config 5 row-sharded 8 ways with its
    packed fp64 all-reduce, the placed-outputs row (one tower per rank); max-over-ranks timing; ONE
    JSON line from rank 0 that says how many ranks the process group saw.  (The batch-sharded
    config-4 row is left to the 2-rank test: eight persistent kernels of eight processes cannot be
    co-resident on one device.)"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, AESARA_BENCH_BACKEND="gloo", AESARA_BENCH_ONE_DEVICE="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                          "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "24", "--warmup", "8",
                          "--cfg5-log2n", "20", "--only-secondary", "cfg5,placed", "--rotate", "2"],
                         capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["ranks_seen"] == 8 and line["steps"] == 24
    assert abs(line["value"] - 8 * 1e3 / line["ms_per_step"]) / line["value"] < 1e-6     # whole-job evals/s
    rows = {r["config"].split(":")[0].split(" ")[0]: r for r in line["secondary"]}
    assert {"cfg5", "placed"} <= set(rows), list(rows)
    for r in line["secondary"]:
        assert "error" not in r, r
        assert r["n_gpus"] == 8 and r["scaling"] == "strong" and r["evals_per_s"] > 0
    assert rows["placed"]["placement"] == [[k] for k in range(8)]
    assert rows["cfg5"]["ranks_seen"] == 8 and rows["cfg5"]["check"]["logp_rel_err"] <= 1e-6
