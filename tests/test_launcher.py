"""world_size-2 gloo tests of the N>1 path (distdiff_amd/launcher.py): the packed-weight broadcast from rank 0, the reference's shard
function giving disjoint covering PNG sets through the real CLI loop (run_expansion with a recording engine), the row-wise feature
all-gather of the prototype builder and the end-of-run statistics reduction."""
import os
import sys

import pytest
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


def _shard_worker(rank, world, port, q, out_dir, n_images):
    """The CLI loop of one rank of `world` on a dataset of n_images (ragged / empty shards) + the end-of-run reduction."""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from distdiff_amd import generate_data as G
    from distdiff_amd import launcher as LA
    from distdiff_amd.config import tiny_config
    from distdiff_amd.scheduler import DDIMSchedule
    torch.set_num_threads(1)
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world)})
    LA.init_distributed(None)
    cfg = tiny_config(max_batch=2)
    eng = PackedFake(2, [("unet", "a.weight", (2, 2))])
    ds = G.ExpansionDataset.synthetic(cfg, n=n_images, n_classes=3, seed=0)
    sched = DDIMSchedule(cfg.scheduler)
    sched.set_timesteps(50)
    args = G.parse_args(["--synthetic", str(n_images), "--output_dir", out_dir, "--train_batch_size", "2", "--num_images_per_prompt", "2",
                         "--total_split", str(world), "--split", str(rank)])
    written = []
    n = G.run_expansion(args, eng, sched, ds, writer=lambda img, p: written.append(p))
    feats = torch.full((len([i for i in LA.shard_range(n_images, world, rank) if i < n_images]), 3), float(rank))
    allf = LA.all_gather_rows(feats)                      # an empty shard contributes zero rows
    total, tmax = LA.reduce_run_stats(n, 1.0 + rank)
    q.put((rank, written, total, tmax, allf[:, 0].tolist()))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_images", [(4, 11), (8, 13), (8, 5)])
def test_ragged_and_empty_shards(tmp_path, world, n_images):
    """generate_data.py:1003-1007 on N not divisible by the rank count: ceil(N / world) images per rank, a short last non-empty rank,
    and -- for small N -- ranks whose range lies entirely beyond N (they write nothing, still join every collective)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 21000 + (os.getpid() * 7 + world * 13 + n_images) % 8000
    ps = [ctx.Process(target=_shard_worker, args=(r, world, port, q, str(tmp_path / "out"), n_images)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=300) for _ in ps)
    for p in ps:
        p.join(60)
    per = -(-n_images // world)
    sets = [set(r[1]) for r in res]
    assert sum(len(s) for s in sets) == len(set().union(*sets)) == 2 * n_images          # disjoint and covering, 2 expansions each
    for r, s in enumerate(sets):
        assert len(s) == 2 * max(0, min(n_images, per * (r + 1)) - per * r), (r, len(s))
    if (world, n_images) == (8, 5):
        assert [len(s) for s in sets] == [2, 2, 2, 2, 2, 0, 0, 0]
    assert all(r[2] == 2 * n_images and r[3] == float(world) for r in res)
    want = [float(r) for r in range(world) for _ in range(max(0, min(n_images, per * (r + 1)) - per * r))]
    assert all(r[4] == want for r in res)


def test_spawn_ranks_reports_a_failing_rank_and_stops_its_siblings(tmp_path):
    """A rank that exits non-zero or dies on a signal must fail the launch (never exit code 0), and its siblings -- blocked in a
    collective in the real run -- must be torn down instead of waiting for the RCCL timeout."""
    import signal
    import time
    sys.path.insert(0, ROOT)
    from distdiff_amd.launcher import exit_code, spawn_ranks
    mod = tmp_path / "misbehave.py"
    mod.write_text(
        "import os, signal, sys, time\n"
        "r, mode = int(os.environ['RANK']), sys.argv[1]\n"
        "open(os.path.join(sys.argv[2], 'pid%d' % r), 'w').write(str(os.getpid()))\n"
        "if r == 1 and mode == 'exit3': sys.exit(3)\n"
        "if r == 1 and mode == 'sigkill': os.kill(os.getpid(), signal.SIGKILL)\n"
        "if mode == 'ok': sys.exit(0)\n"
        "time.sleep(120)\n")
    env = {"PYTHONPATH": str(tmp_path)}
    t0 = time.time()
    assert spawn_ranks(3, ["exit3", str(tmp_path)], module="misbehave", env_extra=env, grace=5.0) == 3
    assert spawn_ranks(3, ["sigkill", str(tmp_path)], module="misbehave", env_extra=env, grace=5.0) == 128 + signal.SIGKILL
    assert time.time() - t0 < 60, "siblings were waited for instead of being terminated"
    for r in (0, 2):                                        # the sleeping siblings are gone
        pid = int((tmp_path / ("pid%d" % r)).read_text())
        with pytest.raises(OSError):
            os.kill(pid, 0)
    assert spawn_ranks(3, ["ok", str(tmp_path)], module="misbehave", env_extra=env) == 0
    assert exit_code(-11) == 139 and exit_code(-6) == 134 and exit_code(2) == 2 and exit_code(0) == 0


def test_gpu_count_without_the_hip_runtime(tmp_path, monkeypatch):
    """The launcher parent counts GPUs from the visibility variables / the KFD topology in sysfs, never through HIP."""
    sys.path.insert(0, ROOT)
    from distdiff_amd import launcher as LA
    for v in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "GPU_DEVICE_ORDINAL"):
        monkeypatch.delenv(v, raising=False)
    root = tmp_path / "nodes"
    for i, simd in enumerate([0, 0, 1024, 1024, 1024]):       # two CPU nodes, three GPUs
        d = root / str(i)
        d.mkdir(parents=True)
        (d / "properties").write_text("cpu_cores_count %d\nsimd_count %d\ngfx_target_version %d\n" % (0 if simd else 64, simd, 90500 if simd else 0))
    assert LA.kfd_gpu_count(str(root)) == 3 and LA.kfd_gpu_count(str(tmp_path / "missing")) == 0
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "0,1,2,3")
    assert LA.visible_gpu_count() == 4
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "2,5")
    assert LA.visible_gpu_count() == 2
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert LA.visible_gpu_count() == 0
    env = LA.rank_thread_env(8)
    assert int(env["OMP_NUM_THREADS"]) == max(1, (os.cpu_count() or 1) // 8)
    # ONE source of truth for the PNG writer threads: the launcher's environment, the CLI default and bench.py's output-stage extra
    assert int(env["DD_PNG_THREADS"]) == LA.png_threads(8) and 1 <= LA.png_threads(8) <= LA.PNG_THREADS_MAX
    import bench, inspect
    assert "png_threads(8)" in inspect.getsource(bench.cli_rate)


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
