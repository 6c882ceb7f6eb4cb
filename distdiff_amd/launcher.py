"""Multi-GPU launcher pieces: the reference's shard function and the RCCL weight broadcast.

The reference fans out one OS process per GPU with `--split k --total_split N` (scripts/exps/expand_diff.sh:19-24,
generate_data.py:1003-1009) and every process loads its own copy of the weights from disk. Here rank 0 loads
(or synthesises) the weights once and broadcasts them over RCCL/xGMI; the unit of work (train image i,
expand index j) shards with the reference's own partition function and needs no data-path collective.
"""
import math

import torch


def shard_range(total, total_split, split):
    """Contiguous index range of `split`, identical to generate_data.py:1003-1007."""
    per = math.ceil(total / total_split)
    if split == total_split - 1 and total < per * (split + 1):
        return list(range(per * split, total))
    return list(range(per * split, per * (split + 1)))


def _layout(weights):
    return [(m, k, tuple(t.shape)) for m in ("unet", "vae", "guide") for k, t in sorted(weights[m].items())]


def broadcast_weights(weights, cfg, src=0, device=None, bucket_bytes=1 << 30, group=None):
    """Broadcasts the state dicts from `src` to every rank in ~1 GiB flat fp32 buckets (few, large collectives:
    xGMI ring/tree broadcasts are per-link bound, ~1.9 GB total for SD-1.x). Works on gloo (CPU tests) and nccl (=RCCL)."""
    import torch.distributed as dist
    rank = dist.get_rank(group)
    meta = [_layout(weights)] if rank == src else [None]
    dist.broadcast_object_list(meta, src=src, group=group)
    layout = meta[0]
    backend = dist.get_backend(group)
    dev = device if (backend == "nccl" and device is not None) else torch.device("cpu")
    out = {"unet": {}, "vae": {}, "guide": {}}
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        n = sum(int(torch.tensor(s).prod()) if len(s) else 1 for _, _, s in bucket)
        flat = torch.empty(n, dtype=torch.float32, device=dev)
        if rank == src:
            off = 0
            for m, k, s in bucket:
                t = weights[m][k].reshape(-1).float()
                flat[off:off + t.numel()] = t.to(dev)
                off += t.numel()
        dist.broadcast(flat, src=src, group=group)
        host = flat.cpu()
        off = 0
        for m, k, s in bucket:
            cnt = 1
            for d in s:
                cnt *= d
            out[m][k] = host[off:off + cnt].reshape(s).clone()
            off += cnt
        bucket, size = [], 0

    for m, k, s in layout:
        cnt = 1
        for d in s:
            cnt *= d
        if size + cnt * 4 > bucket_bytes and bucket:
            flush()
        bucket.append((m, k, s))
        size += cnt * 4
    flush()
    return out
