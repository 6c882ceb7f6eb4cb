"""bench.py's output contract (one JSON line on stdout, the keys the driver parses), on the tiny configuration so that it runs in
seconds; once directly and once under torch.distributed.run with the RCCL weight broadcast forced (the N > 1 start-up path on 1 GPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
        "config", "roofline")


def _check(line, steps, warmup):
    d = json.loads(line)
    for k in KEYS:
        assert k in d, k
    assert d["unit"] == "images/s" and d["n_gpus"] == 1 and d["steps"] == steps and d["warmup"] == warmup
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["value"] > 0 and abs(d["value"] - d["config"]["images_per_step_per_gpu"] * 1000.0 / d["ms_per_step"]) < 1e-6 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    return d


@pytest.mark.gpu
def test_bench_prints_one_contract_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", "tiny", "--steps", "2", "--warmup", "1", "--no_cpu_baseline"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    d = _check(lines[0], 2, 1)
    assert d["config"]["shards"] == [[0, 0, 3 * d["config"]["images_per_step_per_gpu"]]] and "stub" not in d


@pytest.mark.gpu
def test_bench_refuses_more_ranks_than_this_box_has_gpus():
    """`--gpus N` is honoured or refused, never ignored: on a box with fewer than 16 GPUs the plain entry point must say so (GPUs
    counted from sysfs by the launcher parent, before any GPU call) instead of measuring one GPU and printing n_gpus 1."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "16", "--config", "tiny", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode != 0 and "GPU(s) are visible" in out.stderr and out.stdout.strip() == ""


@pytest.mark.gpu
def test_bench_under_torchrun_with_the_weight_broadcast():
    env = dict(os.environ, DD_FORCE_BROADCAST="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", "tiny", "--steps", "1",
                          "--warmup", "1", "--no_cpu_baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    _check(lines[0], 1, 1)
    assert "broadcast" in out.stderr
