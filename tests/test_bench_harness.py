"""bench.py's N > 1 arithmetic on CPU (gloo, world 2 and 3): the helpers main() is built from -- `agree_batch`, `timed_steps`,
`job_rate` -- with a stub step instead of the engine, so that the first real multi-GPU run cannot fail on the harness:

  * every rank runs `warmup` untimed and EXACTLY `steps` timed steps, bracketed by barriers on both sides
  * the job's time is the MAX over ranks (one slow rank sets it), the same number on every rank
  * the static batch is the MIN over ranks of what each one's free HBM allows
  * value = steps x B x world / that time  (whole-job images per second, "weak" scaling: per-rank work fixed)
"""
import os
import sys
import time

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.join(os.path.dirname(__file__), "..")


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import bench
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world)})
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def reduce_min(v):
        t = torch.tensor([v], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return t.item()

    def reduce_max(v):
        t = torch.tensor([v], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # rank 1 has less free HBM: everybody must take its batch
    B = bench.agree_batch(16 if rank == 1 else 32, world, reduce_min)
    calls, hooks = [], []
    per_step = 0.05 * (1 + 2 * (rank == world - 1))      # the last rank is 3x slower

    def step(i):
        calls.append(i)
        hooks.append(i)
        time.sleep(per_step)
        return (i, B)

    steps, warmup = 4, 2
    dt, last = bench.timed_steps(step, steps, warmup, dist.barrier, reduce_max, (lambda: hooks.append("on"), lambda: hooks.append("off")))
    images, rate = bench.job_rate(steps, B, world, dt)
    q.put((rank, B, calls, hooks, dt, last, images, rate))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_bench_multi_rank_arithmetic(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 300) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    steps, warmup = 4, 2
    dts = {r[4] for r in res}
    assert len(dts) == 1, "the job time must be the same (max over ranks) number on every rank: %s" % (dts,)
    dt = dts.pop()
    slow = 0.15 * steps
    assert slow <= dt < slow + 0.5, dt                       # the slowest rank sets it (not the mean, not rank 0's own time)
    for rank, B, calls, hooks, _, last, images, rate in res:
        assert B == 16                                          # min over ranks
        assert calls == list(range(warmup + steps))             # warm-up steps 0..W-1, timed steps W..W+K-1, nothing else
        # per-op profiling brackets the LAST WARM-UP step only: the timed steps are exactly the product's steps
        assert hooks == [0, "on", 1, "off", 2, 3, 4, 5], hooks
        assert last == (warmup + steps - 1, 16)
        assert images == steps * 16 * world and abs(rate - images / dt) < 1e-9


def test_bench_single_rank_needs_no_process_group():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.agree_batch(32, 1, None) == 32
    seen = []
    dt, out = bench.timed_steps(lambda i: seen.append(i) or i, 3, 1, lambda: None, lambda v: v)
    assert seen == [0, 1, 2, 3] and out == 3 and dt >= 0
    assert bench.job_rate(3, 32, 1, 2.0) == (96, 48.0)


# ------------------------------------------------------------------------------------------------
# the real entry point: `python bench.py --gpus N` started plainly launches its own N ranks
# ------------------------------------------------------------------------------------------------
import json
import subprocess

BENCH = os.path.join(ROOT, "bench.py")


def _run_bench(args, env_extra=None, drop=("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)


@pytest.mark.parametrize("world", [2, 3])
def test_bench_entry_point_launches_its_own_ranks(world):
    """`python bench.py --gpus N` (the form of the driver's recorded command) without torch.distributed.run around it: N ranks of the
    same script under gloo with the stand-in engine (--harness_stub), ONE JSON line on stdout, n_gpus == N, one shard per rank that
    together cover steps x B x N images, and the value is the whole job's."""
    out = _run_bench(["--gpus", str(world), "--harness_stub", "--config", "tiny", "--steps", "3", "--warmup", "1", "--batch", "4"])
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == 3 and d["warmup"] == 1 and d["stub"] is True and d["data"] == "stub"
    assert "HARNESS STUB" in d["metric"]                       # cannot be mistaken for a measurement
    per_rank = 4 * (3 + 1)
    assert d["config"]["shards"] == [[r, r * per_rank, (r + 1) * per_rank] for r in range(world)]
    assert abs(d["value"] - 3 * 4 * world / (d["ms_per_step"] * 3 / 1000.0)) < 1e-6 * d["value"]


def test_bench_refuses_a_world_that_contradicts_gpus():
    """Under a launcher (RANK set) `--gpus` must equal WORLD_SIZE: a line with the wrong n_gpus is worse than no line."""
    out = _run_bench(["--gpus", "8", "--harness_stub", "--config", "tiny"], {"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"})
    assert out.returncode != 0 and "WORLD_SIZE=2" in out.stderr and out.stdout.strip() == ""


def test_bench_refuses_more_ranks_than_gpus():
    """No GPU call is needed to know: the launcher counts GPUs from the visibility variables / sysfs."""
    out = _run_bench(["--gpus", "4", "--config", "tiny"], {"HIP_VISIBLE_DEVICES": "0,1"})
    assert out.returncode != 0 and "2 GPU(s) are visible" in out.stderr and out.stdout.strip() == ""


def test_resolve_world():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.resolve_world(None, {}) == (1, False)
    assert bench.resolve_world(1, {}) == (1, False)
    assert bench.resolve_world(8, {}) == (8, True)                                  # plain start: launch 8 ranks
    assert bench.resolve_world(8, {"RANK": "3", "WORLD_SIZE": "8"}) == (8, False)     # torch.distributed.run / our own child
    assert bench.resolve_world(None, {"RANK": "0", "WORLD_SIZE": "2"}) == (2, False)
    with pytest.raises(SystemExit):
        bench.resolve_world(8, {"RANK": "0", "WORLD_SIZE": "1"})
    with pytest.raises(SystemExit):
        bench.resolve_world(0, {})
