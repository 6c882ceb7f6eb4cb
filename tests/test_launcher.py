"""world_size-2 gloo tests of the N>1 path (distdiff_amd/launcher.py): the packed-weight broadcast from rank 0, the reference's shard
function giving disjoint covering PNG sets through the real CLI loop (run_expansion with a recording engine), the row-wise feature
all-gather of the prototype builder and the end-of-run statistics reduction."""
import os
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.join(os.path.dirname(__file__), "..")


class PackedFake:
    """Engine stand-in for the CPU: a flat packed-weight byte array with the export / import interface of distdiff_amd.engine.Engine,
    plus the calls run_expansion makes."""
    device = torch.device("cpu")

    def __init__(self, B, layout, data=None):
        self.B, self.layout = B, layout
        n = sum(int(torch.tensor(s).prod()) * 4 for _, _, s in layout)
        self.buf = data if data is not None else torch.zeros(n, dtype=torch.uint8)
        self.calls = []

    def weight_layout(self):
        return list(self.layout)

    def packed_bytes(self):
        return self.buf.numel()

    def export_packed(self, view, off):
        view.copy_(self.buf[off:off + view.numel()])

    def import_packed(self, view, off):
        self.buf[off:off + view.numel()].copy_(view)

    def set_prompt(self, emb):
        pass

    def expand(self, lat, noise, e, b, tg, si, gt, gfirst, gcount, want_image=True):
        self.calls.append(tg.tolist())
        return lat.clone(), torch.rand(lat.shape[0], 3, 8, 8), torch.tensor([1.0])


def _worker(rank, world, port, q, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from distdiff_amd import generate_data as G
    from distdiff_amd import launcher as LA
    from distdiff_amd.config import tiny_config
    from distdiff_amd.scheduler import DDIMSchedule
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world)})
    r, w = LA.init_distributed(None)
    assert (r, w) == (rank, world)
    # 1. packed weights: rank 0 "loads + packs", the other rank builds from the layout and receives the bytes in several buckets
    layout = [("unet", "a.weight", (64, 32, 3, 3)), ("vae", "b.bias", (1000,)), ("guide", "c.weight", (7, 5))]
    g = torch.Generator().manual_seed(0)
    ref = torch.randint(0, 256, (sum(int(torch.tensor(s).prod()) * 4 for _, _, s in layout),), generator=g, dtype=torch.uint8)

    def load():
        return tiny_config(max_batch=2), {"payload": ref}

    cfg, eng = LA.build_engine_distributed(load, lambda c, wts, lay: PackedFake(2, layout if wts is not None else lay,
                                                                                wts["payload"].clone() if wts is not None else None))
    ok_w = torch.equal(eng.buf, ref) and eng.weight_layout() == layout and cfg.max_batch == 2
    eng2 = PackedFake(2, layout, ref.clone() if rank == 0 else None)
    LA.broadcast_packed_weights(eng2, 0, bucket_bytes=4096)
    ok_w = ok_w and torch.equal(eng2.buf, ref)
    # 2. the CLI loop on this rank's shard
    ds = G.ExpansionDataset.synthetic(cfg, n=11, n_classes=3, seed=0)
    sched = DDIMSchedule(cfg.scheduler)
    sched.set_timesteps(50)
    args = G.parse_args(["--synthetic", "11", "--output_dir", out_dir, "--train_batch_size", "2", "--num_images_per_prompt", "2",
                         "--total_split", str(world), "--split", str(rank)])
    written = []
    n = G.run_expansion(args, eng, sched, ds, writer=lambda img, p: written.append(p))
    # 3. feature all-gather in dataset order, statistics
    feats = torch.arange(len(LA.shard_range(11, world, rank)), dtype=torch.float32)[:, None] + 100.0 * rank
    feats = feats[:len([i for i in LA.shard_range(11, world, rank) if i < 11])].expand(-1, 4).contiguous()
    allf = LA.all_gather_rows(feats)
    total, tmax = LA.reduce_run_stats(n, 1.0 + rank)
    small = LA.broadcast_tensors([torch.full((2, 3), 7.0)] if rank == 0 else None)
    q.put((rank, ok_w, written, allf[:, 0].tolist(), total, tmax, small[0].tolist()))
    dist.destroy_process_group()


def test_packed_broadcast_shards_gather_and_stats_world2(tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q, str(tmp_path / "out"))) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=240) for _ in ps)
    for p in ps:
        p.join(60)
    assert all(r[1] for r in res), "packed weights did not arrive intact"
    paths0, paths1 = set(res[0][2]), set(res[1][2])
    assert not (paths0 & paths1) and len(paths0 | paths1) == 11 * 2          # disjoint and covering: 11 images x 2 expansions
    # ceil(11/2) = 6 -> rank 0: images 0..5, rank 1: images 6..10 (generate_data.py:1003-1007)
    assert len(paths0) == 12 and len(paths1) == 10
    assert res[0][3] == res[1][3] == [0, 1, 2, 3, 4, 5, 100, 101, 102, 103, 104]
    assert res[0][4] == res[1][4] == 22 and res[0][5] == res[1][5] == 2.0
    assert res[0][6] == res[1][6] == [[7.0] * 3] * 2


def test_spawn_ranks_sets_the_rendezvous_environment(tmp_path):
    """The process fan-out of `generate_data.py --gpus N`: every child sees RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1."""
    sys.path.insert(0, ROOT)
    from distdiff_amd.launcher import spawn_ranks
    mod = tmp_path / "echo_rank.py"
    mod.write_text("import os, sys\nopen(os.path.join(sys.argv[1], os.environ['RANK']), 'w').write(' '.join(os.environ[k] for k in "
                   "('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR')))\n")
    rc = spawn_ranks(3, [str(tmp_path)], module="echo_rank", env_extra={"PYTHONPATH": str(tmp_path)})
    assert rc == 0
    got = sorted((tmp_path / str(r)).read_text() for r in range(3))
    assert got == ["0 0 3 127.0.0.1", "1 1 3 127.0.0.1", "2 2 3 127.0.0.1"]
