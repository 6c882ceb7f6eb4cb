"""world_size-2 gloo test of the N>1 path: weight broadcast from rank 0 + disjoint, covering shards."""
import os
import sys

import torch
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    import torch.distributed as dist
    from distdiff_amd.config import tiny_config
    from distdiff_amd.launcher import broadcast_weights, shard_range
    from distdiff_amd.weights import synthetic_weights
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = tiny_config()
    w = synthetic_weights(cfg, seed=0, num_classes=5) if rank == 0 else None
    got = broadcast_weights(w, cfg, src=0, bucket_bytes=1 << 22)
    ref = synthetic_weights(cfg, seed=0, num_classes=5)
    ok = all(torch.equal(got[m][k], ref[m][k]) for m in ref for k in ref[m]) and all(len(got[m]) == len(ref[m]) for m in ref)
    mine = shard_range(11, world, rank)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    q.put((rank, ok, gathered))
    dist.destroy_process_group()


def test_broadcast_and_shards_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=180) for _ in ps]
    for p in ps:
        p.join(60)
    assert all(ok for _, ok, _ in res)
    shards = res[0][2]
    flat = [i for s in shards for i in s if i < 11]
    assert sorted(flat) == list(range(11)) and len(set(flat)) == len(flat)
