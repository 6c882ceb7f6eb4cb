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
