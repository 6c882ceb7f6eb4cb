"""The N-rank launcher of the expansion step.

The reference fans out one OS process per GPU by hand, `CUDA_VISIBLE_DEVICES=k python generate_data.py --split k --total_split N`
(single_exp.sh:4-8, scripts/exps/expand_diff.sh:19-24), and every process loads its own copy of the weights from disk and repeats the
prototype extraction over the whole training set.  Here `generate_data.py --gpus N` (or `expand_diff.sh <EXPAND_NUM>`) starts N ranks,
one per GPU, BEFORE any GPU call; rank 0 loads and packs the weights once and broadcasts the PACKED device buffers (bf16 MFMA layouts
+ fp32 guide / norm / embedding tables, 3.8 GB for SD-1.x incl. the input-gradient packings) over RCCL/xGMI in ~1 GiB buckets -- few, large collectives, because an
xGMI ring is per-link bound -- ; prototype features are computed on per-rank shards and all-gathered; the units of work
(train image i, expand index j) shard with the reference's own partition function (generate_data.py:1003-1007) and need no
collective on the data path; image counts and elapsed time are all-reduced at the end for the node-level images/s.
"""
import math
import os
import socket
import subprocess
import sys
import time

import torch


def shard_range(total, total_split, split):
    """Contiguous index range of `split`, identical to generate_data.py:1003-1007."""
    per = math.ceil(total / total_split)
    if split == total_split - 1 and total < per * (split + 1):
        return list(range(per * split, total))
    return list(range(per * split, per * (split + 1)))


# ------------------------------------------------------------------------------------------------
# process fan-out
# ------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def visible_gpu_count():
    """GPUs this process may use, WITHOUT loading the HIP / HSA runtime (the launcher parent must stay GPU-free: it forks the ranks,
    and `torch.cuda.device_count()` falls back to hipGetDeviceCount -- i.e. initialises the runtime -- whenever amdsmi is not
    usable).  The visibility variables win (the first one set, in the runtime's order of precedence); otherwise the KFD topology is
    read from sysfs: a node with simd_count > 0 is a GPU."""
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "GPU_DEVICE_ORDINAL"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""]) if v.strip() not in ("", "-1") else 0
    return kfd_gpu_count()


def kfd_gpu_count(root="/sys/class/kfd/kfd/topology/nodes"):
    n = 0
    try:
        nodes = sorted(os.listdir(root))
    except OSError:
        return 0
    for node in nodes:
        try:
            props = dict(line.split()[:2] for line in open(os.path.join(root, node, "properties")) if len(line.split()) >= 2)
        except OSError:
            continue
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    return n


PNG_THREADS_MAX = 8      # PNG writer threads of one rank, at most (one thread encodes ~18 512x512 files/s; a rank produces ~16/s)


def png_threads(n_ranks=1):
    """PNG writer threads of one rank when `n_ranks` ranks share this host: its share of the cores, at most PNG_THREADS_MAX.  The ONE
    place the number comes from: the launcher's environment for its ranks, the CLI's default and bench.py's output-stage extra."""
    return max(1, min(PNG_THREADS_MAX, (os.cpu_count() or 1) // max(1, n_ranks)))


def rank_thread_env(n):
    """Host threads of one rank: N ranks share the node's cores (PNG encoding, torch's intra-op pool), so each gets cpu_count // N
    instead of every rank spawning a pool sized for the whole machine."""
    per = max(1, (os.cpu_count() or 1) // max(1, n))
    return {"OMP_NUM_THREADS": str(per), "MKL_NUM_THREADS": str(per), "DD_PNG_THREADS": str(png_threads(n))}


def exit_code(rc):
    """Popen return code -> shell exit code: a rank killed by signal N (Popen reports -N: SIGSEGV, the NCCL watchdog's SIGABRT, an
    OOM SIGKILL) becomes 128 + N, never 0."""
    return 128 - rc if rc < 0 else rc


def spawn_ranks(n, argv, module="distdiff_amd.generate_data", env_extra=None, grace=10.0, poll=0.2, script=None):
    """Starts `n` worker processes of the CLI (`python -m module argv`; or of a script file, `python script argv`: bench.py launches
    itself this way) with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, rendezvous on 127.0.0.1, and supervises them: all children are polled together; as soon as one exits non-zero (signals included) the others
    -- which would otherwise sit in a collective until the RCCL timeout -- are terminated (SIGTERM, SIGKILL after `grace`
    seconds) and its code is returned (128 + N for signal N).  Returns 0 only if every rank returned 0.  The parent never
    touches the GPU."""
    port = _free_port()
    procs = []
    threads = rank_thread_env(n)
    for r in range(n):
        env = dict(os.environ)
        for k, v in threads.items():
            env.setdefault(k, v)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                    "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
        env.update(env_extra or {})
        head = [sys.executable, script] if script else [sys.executable, "-m", module]
        procs.append(subprocess.Popen(head + list(argv), env=env))
    failed = 0
    live = list(procs)
    try:
        while live and not failed:
            time.sleep(poll)
            for p in list(live):
                rc = p.poll()
                if rc is None:
                    continue
                live.remove(p)
                if rc != 0 and not failed:
                    failed = exit_code(rc)
                    print("[launcher] rank %d exited with %s: stopping the other ranks" % (procs.index(p), ("signal %d" % -rc) if rc < 0 else "code %d" % rc),
                          file=sys.stderr, flush=True)
    finally:
        if live:            # a failed sibling, or the parent itself is going down (KeyboardInterrupt): no orphans
            for p in live:
                p.terminate()
            deadline = time.time() + grace
            for p in live:
                try:
                    p.wait(max(0.0, deadline - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
    return failed


def init_distributed(device=None):
    """Joins the process group when launched with RANK/WORLD_SIZE set (by spawn_ranks or torchrun): nccl (= RCCL) on a GPU,
    gloo on the CPU.  Returns (rank, world)."""
    import torch.distributed as dist
    if "RANK" not in os.environ:
        return 0, 1
    if not dist.is_initialized():
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # the other ranks wait in a barrier while rank 0 lists the dataset, VAE-encodes the latent cache and runs the text encoder
        # (generate_data.main): the default collective timeout of 10 minutes is too short for a first run on a large dataset
        timeout = datetime.timedelta(seconds=float(os.environ.get("DD_DIST_TIMEOUT_S", 4 * 3600)))
        if device is not None and torch.device(device).type == "cuda":
            dist.init_process_group("nccl", device_id=torch.device(device), timeout=timeout)
        else:
            dist.init_process_group("gloo", timeout=timeout)
    return dist.get_rank(), dist.get_world_size()


# ------------------------------------------------------------------------------------------------
# weights: one load + pack on rank 0, one broadcast of the packed device buffers
# ------------------------------------------------------------------------------------------------
def broadcast_object(obj, src=0, group=None):
    import torch.distributed as dist
    box = [obj if dist.get_rank(group) == src else None]
    dist.broadcast_object_list(box, src=src, group=group)
    return box[0]


def broadcast_packed_weights(engine, src=0, bucket_bytes=1 << 30, group=None, device=None):
    """Broadcasts the packed weight buffers of rank `src`'s engine into every other rank's (shape-built) engine through a device
    staging bucket: export -> dist.broadcast -> import.  Returns the number of bytes moved."""
    import torch.distributed as dist
    rank = dist.get_rank(group)
    total = engine.packed_bytes()
    sizes = [None] * dist.get_world_size(group)
    dist.all_gather_object(sizes, total, group=group)
    if len(set(sizes)) != 1:
        raise RuntimeError("packed weight layouts differ across ranks: %s" % sizes)
    dev = device if device is not None else getattr(engine, "device", torch.device("cpu"))
    bucket = torch.empty(min(bucket_bytes, max(total, 1)), dtype=torch.uint8, device=dev)
    off = 0
    while off < total:
        n = min(bucket.numel(), total - off)
        view = bucket[:n]
        if rank == src:
            engine.export_packed(view, off)
        dist.broadcast(view, src=src, group=group)
        if rank != src:
            engine.import_packed(view, off)
        off += n
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    return total


def build_engine_distributed(make_cfg_and_weights, make_engine, src=0, group=None):
    """rank `src`: (cfg, weights) = make_cfg_and_weights(); engine = make_engine(cfg, weights, None).
    other ranks: receive (cfg, layout) and build engine = make_engine(cfg, None, layout) from the shapes; then one broadcast of the
    packed buffers.  Returns (cfg, engine)."""
    import torch.distributed as dist
    rank = dist.get_rank(group)
    if rank == src:
        cfg, weights = make_cfg_and_weights()
        eng = make_engine(cfg, weights, None)
        del weights
        broadcast_object((cfg, eng.weight_layout()), src, group)
    else:
        cfg, layout = broadcast_object(None, src, group)
        eng = make_engine(cfg, None, layout)
    t0 = time.time()
    nbytes = broadcast_packed_weights(eng, src, group=group)
    if rank == src:
        print("broadcast %.2f GB of packed weights to %d ranks in %.2f s" % (nbytes / 1e9, dist.get_world_size(group), time.time() - t0),
              flush=True)
    return cfg, eng


def broadcast_tensors(tensors, src=0, device=None, group=None):
    """Small fp32 tensors (prototypes, per-class text embeddings) from `src` to every rank, through the device for nccl."""
    import torch.distributed as dist
    rank = dist.get_rank(group)
    meta = broadcast_object([tuple(t.shape) for t in tensors] if rank == src else None, src, group)
    dev = device if (dist.get_backend(group) == "nccl" and device is not None) else torch.device("cpu")
    out = []
    for i, shp in enumerate(meta):
        t = tensors[i].float().to(dev).contiguous() if rank == src else torch.empty(shp, dtype=torch.float32, device=dev)
        dist.broadcast(t, src=src, group=group)
        out.append(t.cpu())
    return out


def all_gather_rows(rows, device=None, group=None):
    """Row-wise all-gather of per-rank fp32 matrices [n_r, D] with different n_r, in rank order (prototype features, f-1)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    counts = [None] * world
    dist.all_gather_object(counts, int(rows.shape[0]), group=group)
    dev = device if (dist.get_backend(group) == "nccl" and device is not None) else torch.device("cpu")
    D = rows.shape[1]
    nmax = max(counts)
    pad = torch.zeros(nmax, D, dtype=torch.float32, device=dev)
    pad[:rows.shape[0]] = rows.float().to(dev)
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:c].cpu() for p, c in zip(parts, counts)])


def reduce_run_stats(images, seconds, device=None, group=None):
    """End of run: total images (sum) and wall time (max over ranks) -> node-level images/s."""
    import torch.distributed as dist
    dev = device if (dist.get_backend(group) == "nccl" and device is not None) else torch.device("cpu")
    n = torch.tensor([float(images)], dtype=torch.float64, device=dev)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=dev)
    dist.all_reduce(n, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(n.item()), float(t.item())
