#!/usr/bin/env python3
"""bench.py — 512x512 images/sec of the guided DDIM expansion hot path on MI355X.

One "step" = one pass of the hot path (dd_expand: add_noise -> DDIM loop with CFG -> hierarchical energy guidance ->
VAE decode) over one batch of B synthetic images, inputs resident in HBM. Workload = BASELINE.json configs[1]:
SD-1.5 shapes, 512x512, 50-step DDIM schedule, class+group prototype transform guidance (script of record,
scripts/exps/expand_diff.sh: strength 0.5 -> 25 executed steps, guidance_step 20, guidance_period 2, rho 10, K 3),
ResNet-50 guide, bf16 MFMA with fp32 accumulation. Weights are seeded synthetic (no checkpoints offline).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus 8 ...          # launches its own 8 ranks (distdiff_amd.launcher.spawn_ranks) BEFORE any GPU call
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 ...

Multi-GPU: one process per GPU, images sharded across ranks with the reference's partition function
(generate_data.py:1003-1007), no collective on the data path; rank 0 broadcasts the weights over RCCL/xGMI.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

# MI355X dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md (2.5 PFLOP/s; AMD's 5 PFLOP/s headline includes 2:1 sparsity).
# SURVEY.md section 8d prices the roofline against "the judge-supplied figure": DD_PEAK_BF16_TFLOPS overrides the guide's number.
PEAK_BF16_TFLOPS = float(os.environ.get("DD_PEAK_BF16_TFLOPS", 2500.0))


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="GPUs of this node to run on (default: WORLD_SIZE under torch.distributed.run, else 1).  Started plainly with "
                         "N > 1 the script launches its own N ranks, one per GPU; under torch.distributed.run N must equal WORLD_SIZE")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=0,
                    help="images per step per GPU; 0 = the largest of 32 / 16 / 8 whose workspace (distdiff_amd.engine.batch_for_free_hbm: two "
                         "activation stashes for the chained guided steps + the gradient slab) fits the free HBM of every rank")
    ap.add_argument("--config", default="sd15", choices=["sd15", "tiny", "sdxl"],
                    help="sd15 = BASELINE configs[1] (the metric's workload); sdxl = the SDXL-base UNet at 1024x1024 (configs[4] structure, bf16)")
    ap.add_argument("--guidance", default="transform_guidance", choices=["transform_guidance", "direct_guidance", "none"])
    ap.add_argument("--strength", type=float, default=0.5)
    ap.add_argument("--schedule_steps", type=int, default=50)
    ap.add_argument("--guidance_step", type=int, default=20)
    ap.add_argument("--guidance_period", type=int, default=2)
    ap.add_argument("--classes", type=int, default=100, help="classes of the prototype tables (100: Caltech-101-shaped configs[1]; 196: StanfordCars-shaped configs[3])")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_profile", action="store_true")
    ap.add_argument("--no_cli", action="store_true", help="skip the output-stage extra (the CLI loop + PNG writer, outside the timed region)")
    ap.add_argument("--cli_units", type=int, default=0, help="units of the output-stage extra (0 = 12 engine batches)")
    ap.add_argument("--no_strength1", action="store_true", help="skip the SURVEY 8(d) case (ii) extra (strength 1.0 = every schedule step executed; 2 timed steps)")
    ap.add_argument("--harness_stub", action="store_true",
                    help="TEST HOOK, not a measurement: gloo on the CPU and a sleeping stand-in for the engine, so that the launch / shard / "
                         "timing / one-JSON-line logic of this entry point runs where there is no GPU (tests/test_bench_harness.py); the line "
                         "says so in `metric`, `data` and `stub`")
    return ap.parse_args(argv)


def resolve_world(gpus, environ):
    """(world, self_launch): what `--gpus` means next to the launcher's environment.
      * RANK set (torch.distributed.run or our own spawn_ranks): world = WORLD_SIZE; a `--gpus` that disagrees is an error, never a
        line with the wrong n_gpus
      * RANK not set: `--gpus N > 1` -> this process only launches N ranks of itself; N in (None, 1) -> a single process, no process group"""
    if "RANK" in environ:
        world = int(environ.get("WORLD_SIZE", "1"))
        if gpus is not None and gpus != world:
            raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (gpus, world))
        return world, False
    n = 1 if gpus is None else gpus
    if n < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    return n, n > 1


def self_launch(n, argv, stub):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set,
    127.0.0.1 rendezvous, siblings of a failed rank stopped) and return the exit code.  The parent makes NO GPU call (GPUs are counted
    from sysfs / the visibility variables); rank 0's child writes the one JSON line to the stdout it inherits."""
    from distdiff_amd.launcher import spawn_ranks, visible_gpu_count
    if not stub:
        have = visible_gpu_count()
        if have < n:
            raise SystemExit("bench.py: --gpus %d but %d GPU(s) are visible on this node" % (n, have))
    return spawn_ranks(n, argv, script=os.path.abspath(__file__))


class _StubEngine:
    """--harness_stub only: stands where the engine stands so that main()'s control flow runs without a GPU.  Computes nothing."""
    def __init__(self, B, L):
        self.B, self.L, self.n = B, L, 0
        self.shards = []

    def set_schedule(self, *a, **k): pass
    def set_prototypes(self, *a, **k): pass
    def set_prompt(self, *a, **k): pass
    def profile_enable(self, on): pass
    def profile_read(self): return None
    def workspace_bytes(self): return 0
    def close(self): pass

    def flops_last(self):
        n, self.n = self.n, 0
        return float(n)

    def expand(self, lat, noise, e, b, tg, si, gt, gfirst, gcount, want_image=True):
        assert lat.shape[0] == self.B, "ragged stub batch: %s" % (lat.shape,)
        time.sleep(0.01)
        self.n += self.B
        return lat.clone(), torch.zeros(self.B, 3, 8, 8), torch.zeros(())


def cpu_baseline(cfg, weights, n_exec, P, flops_per_image):
    """The fp32 torch CPU oracle (oracle/sd_oracle.py, a restatement of the reference's loop over restated diffusers / timm modules;
    the reference itself cannot run here) timed on this host, on a bounded sample of the SAME workload:

      C2 (the metric's config): the loop is a repetition of identical steps, so the three distinct pieces are each MEASURED once at
         full size (512x512, B = 1) -- one plain denoise_one_step (UNet, CFG batch 2), the transform guidance call itself with its P
         CHAINED guided steps (UNet + VAE decode + bicubic + ResNet-50 + energy per step, forward and torch.autograd backward to (e, b)
         through the whole chain, generate_data.py:687-732), one final VAE decode -- and combined with the step counts of the
         workload: t_image = n_exec * t_plain + t_guidance(P) + t_decode.  No FLOP scaling is involved.
         The P = 2 chain is ONE autograd graph when the host has the memory for it (~80 GB); otherwise the same gradient is assembled
         stage by stage (tests/golden/make_fullsize_loop_fixture.py: one extra no-grad forward of the first step, subtracted as one
         t_plain); `sample` says which was run.
      C1 (BASELINE configs[0], the reference's own CPU-runnable case: 256x256, 10 DDIM steps, guidance off): one image, run in full.
    """
    from oracle import sd_oracle as O
    import torch.nn.functional as F
    from distdiff_amd.config import sd15_config
    cores = min(os.cpu_count() or 1, 64)     # torch CPU convolutions stop scaling (and oversubscribe) beyond ~64 threads
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    L, D = cfg.latent_size, cfg.guide.feature_dim
    models = O.build_models(cfg, weights)
    unet, vae, guide, sched = models
    ts = sched.set_timesteps(50)
    emb = torch.randn(2, cfg.text_len, cfg.unet.cross_attention_dim, generator=g)
    z = torch.randn(1, 4, L, L, generator=g)
    args = O.SamplerArgs(guidance_type="transform_guidance", num_inference_steps=50, guidance_step=20, guidance_period=P, strength=0.5)
    Pc = F.normalize(torch.randn(100, D, generator=g), dim=-1)
    Pg = F.normalize(torch.randn(100, 3, D, generator=g), dim=-1)
    with torch.no_grad():
        F.conv2d(torch.randn(1, 64, 64, 64), torch.randn(64, 64, 3, 3), padding=1)       # thread pool / oneDNN warm-up
        t0 = time.time()
        zp, x0 = O.denoise_one_step(args, z, sched, int(ts[31]), unet, emb)
        t_plain = time.time() - t0
        t0 = time.time()
        vae.decode(zp / cfg.vae.scaling_factor)
        t_decode = time.time() - t0
    e0, b0 = torch.rand(1, 4, 1, 1, generator=g), torch.randn(1, 4, 1, 1, generator=g)
    gts = [int(ts[30 + k]) for k in range(P)]
    try:
        import psutil
        avail = psutil.virtual_memory().available
    except Exception:
        avail = 0
    t0 = time.time()
    if P == 2 and avail < 110e9:
        O.transform_guidance_3stage(args, cfg, models, z, torch.tensor([7]), gts, emb, e0, b0, Pc, Pg)
        t_guidance = time.time() - t0 - t_plain
        how = "P=2 chain assembled in three stages (host RAM %.0f GB < 110 GB), the extra no-grad forward subtracted" % (avail / 1e9)
    else:
        O.transform_guidance(args, z, torch.tensor([7]), gts, sched, unet, emb, vae, guide, e0, b0, Pc, Pg, cfg.guide.input_size)
        t_guidance = time.time() - t0
        how = "the P=%d chain as ONE autograd graph" % P
    t_image = n_exec * t_plain + t_guidance + t_decode
    # C1: configs[0] in full
    c1 = sd15_config(latent_size=32, max_batch=1)
    m1 = O.build_models(c1, weights)
    a1 = O.SamplerArgs(guidance_type=None, num_inference_steps=10, strength=1.0)
    t0 = time.time()
    with torch.no_grad():
        O.expand_one(a1, c1, m1, torch.randn(1, 4, 32, 32, generator=g), torch.randn(1, 4, 32, 32, generator=g), None, None, emb[1:], emb[:1],
                     torch.zeros(1, dtype=torch.int64), None, None)
    t_c1 = time.time() - t0
    return {"value": 1.0 / t_image, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "fp32 torch-CPU oracle on %d threads, every distinct piece of the workload measured once at full size (512x512, B=1): "
                      "plain denoise_one_step %.2f s; transform_guidance with %d chained guided steps (UNet + VAE decode + ResNet-50 + energy "
                      "each, forward + autograd backward through the chain; %s) %.2f s; final VAE decode %.2f s; "
                      "t_image = %d x plain + guidance + decode = %.1f s"
                      % (cores, t_plain, P, how, t_guidance, t_decode, n_exec, t_image),
            "seconds_per_image": t_image,
            "config0": {"workload": "BASELINE configs[0]: SD-1.x shapes 256x256, 10 DDIM steps, strength 1.0, guidance off, 1 image, run in full",
                        "seconds_per_image": t_c1, "value": 1.0 / t_c1, "unit": "images/s"}}


def cli_rate(eng, cfg, sched, a, B, n_units=128):
    """Output stage at rate (generate_data.py:1221-1234): the drop-in CLI's own loop (`run_expansion`: unit
    packing, sample weights, dd_expand, uint8 quantisation on the GPU, pinned D2H, PNG encoding on the writer threads) over `n_units`
    synthetic units on the engine the bench just timed, PNGs written to tmpfs.  Writer threads = what `generate_data.py --gpus 8` gives
    each of its ranks on this host (launcher.png_threads(8): the configuration that ships).  Outside the timed region, reported beside `value`."""
    import shutil
    import tempfile
    from distdiff_amd import generate_data as G
    from distdiff_amd.launcher import png_threads
    threads = png_threads(8)
    root = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    out = tempfile.mkdtemp(prefix="dd_bench_png_", dir=root)
    old = os.environ.get("DD_PNG_THREADS")
    os.environ["DD_PNG_THREADS"] = str(threads)
    try:
        args = G.parse_args(["--synthetic", str(n_units), "--num_images_per_prompt", "1", "--output_dir", out, "--engine_batch", str(B),
                             "--guidance_type", a.guidance if a.guidance != "none" else "", "--strength", str(a.strength),
                             "--guidance_step", str(a.guidance_step), "--guidance_period", str(a.guidance_period), "--seed", "42",
                             "--total_split", "1", "--split", "0"])
        ds = G.ExpansionDataset.synthetic(cfg, n_units, 100, seed=7)
        torch.cuda.synchronize()
        t0 = time.time()
        n = G.run_expansion(args, eng, sched, ds)
        torch.cuda.synchronize()
        dt = time.time() - t0
        nbytes = sum(os.path.getsize(os.path.join(d, f)) for d, _, fs in os.walk(out) for f in fs)
        return {"cli_images_per_s": n / dt, "cli_images": n, "cli_seconds": dt, "png_threads": threads, "png_mb_per_image": nbytes / max(n, 1) / 1e6,
                "cli_note": "generate_data.run_expansion on %d synthetic units, %d per engine batch, PNGs to %s" % (n_units, B, "tmpfs" if root else "tmp")}
    finally:
        if old is None:
            os.environ.pop("DD_PNG_THREADS", None)
        else:
            os.environ["DD_PNG_THREADS"] = old
        shutil.rmtree(out, ignore_errors=True)


def agree_batch(B, world, reduce_min):
    """Ranks of one run use the SAME static batch: the minimum of what each rank's free HBM allows."""
    return int(reduce_min(B)) if world > 1 else B


def timed_steps(step, steps, warmup, barrier, reduce_max, profile_hooks=None):
    """The timing contract: `warmup` untimed steps, then EXACTLY `steps` steps bracketed by barrier + synchronize on both sides; the
    job's time is the MAX over ranks.  `profile_hooks` = (before, after) callables around the per-op profile pass (HIP events around every
    op): the LAST warm-up step, so that the timed steps are exactly the product's steps; with no warm-up it is the first timed step."""
    for i in range(warmup):
        if i == warmup - 1 and profile_hooks:
            profile_hooks[0]()
        step(i)
        if i == warmup - 1 and profile_hooks:
            profile_hooks[1]()
    barrier()
    out = None
    t0 = time.time()
    for i in range(steps):
        if i == 0 and warmup == 0 and profile_hooks:
            profile_hooks[0]()
        out = step(warmup + i)
        if i == 0 and warmup == 0 and profile_hooks:
            profile_hooks[1]()
    barrier()
    dt = reduce_max(time.time() - t0)
    return dt, out


def job_rate(steps, B, world, dt):
    """Whole-job throughput: every rank processed steps x B images of its own shard in the common (max over ranks) time."""
    images = steps * B * world
    return images, images / dt


_UNET_FLOPS = {}


def unet_flops_per_sample(cfg):
    # 803.3 GFLOP per sample-forward at 64x64 latent for SD-1.x; measured by the engine counter otherwise
    return _UNET_FLOPS.get("v", 803.3e9)


def recorded_traffic(B, config):
    """HBM bytes per conv launch from the committed PMC passes (profiles/, tools/pmc_summary.py).

    PMC collection needs rocprofv3 around the whole process, so it cannot run inside the timed region; the number comes from the
    newest committed passes of this workload AT THE SAME BATCH.  Passes of another batch are not scaled into `traffic` (weight bytes
    and the batch-dependent kernel selection do not scale with the batch): they are reported as `traffic_estimate`, labelled."""
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    path, pb = None, B
    for r in (9, 8, 7, 6, 5, 4, 3, 2, 1):
        for b in (B, 32, 16, 8):
            cand = os.path.join(root, "r%02d_b%d_pmc_traffic.json" % (r, b))
            if os.path.exists(cand):
                path, pb = cand, b
                break
        if path:
            break
    if config != "sd15" or path is None:
        return {}
    with open(path) as f:
        t = json.load(f)
    byts = launches = 0.0
    for fam in ("conv_gemm_big_kernel", "conv_halo_kernel", "gemm_pp_kernel", "gemm_ws_kernel", "conv_gemm_kernel", "splitk"):
        if fam in t:
            byts += t[fam]["hbm_fetch_bytes_corrected"] + t[fam]["hbm_write_bytes"]
            if fam != "splitk":
                launches += t[fam]["launches"]
    src = "profiles/" + os.path.basename(path) + " (separate rocprofv3 --pmc passes of this command)"
    per = byts / max(launches, 1.0)
    unit = "HBM bytes per conv launch (FETCH_SIZE x2 + WRITE_SIZE)"
    if pb == B:
        return {"traffic": per, "traffic_unit": unit, "traffic_source": src}
    return {"traffic": None, "traffic_estimate": per * float(B) / pb, "traffic_unit": unit,
            "traffic_source": src + "; ESTIMATE: measured at %d images per step and scaled x%.2f (activation bytes scale with the batch, "
                                    "weight bytes and tile selection do not)" % (pb, float(B) / pb)}


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    a = parse(argv)
    world, launch = resolve_world(a.gpus, os.environ)
    if launch:                                    # plain `python bench.py --gpus N`: become the launcher, touch no GPU
        raise SystemExit(self_launch(world, argv, a.harness_stub))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    stub = a.harness_stub
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # stdout carries exactly ONE JSON line: everything else that libraries print there (RCCL writes its NCCL_DEBUG=VERSION banner to
    # stdout when the first communicator is created) is sent to stderr by pointing fd 1 at fd 2 for the life of the process; the result
    # line is written to a duplicate of the original stdout
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    distributed = "RANK" in os.environ            # launched by torch.distributed.run or self_launch (also with one rank)
    if stub:
        dev = torch.device("cpu")
        if distributed:
            import torch.distributed as dist
            dist.init_process_group("gloo")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (there is no CPU fallback)")
        if local >= torch.cuda.device_count():
            raise SystemExit("bench.py: rank %d wants GPU %d but this process sees %d" % (rank, local, torch.cuda.device_count()))
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        if distributed:
            import torch.distributed as dist
            dist.init_process_group("nccl", device_id=dev)

    from distdiff_amd.config import sd15_config, sdxl_config, tiny_config
    from distdiff_amd.launcher import build_engine_distributed, shard_range
    from distdiff_amd.scheduler import DDIMSchedule
    if not stub:
        from distdiff_amd.engine import Engine
        from distdiff_amd.weights import synthetic_weights

    B = a.batch
    if B <= 0 and stub:
        B = 4
    if B <= 0:
        if a.config == "sd15":
            from distdiff_amd.engine import batch_for_free_hbm
            B = batch_for_free_hbm(torch.cuda.mem_get_info(dev)[0], guided=a.guidance != "none")     # the CLI's own rule

            def reduce_min(v):
                t = torch.tensor([v], device=dev, dtype=torch.int64)
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                return t.item()
            B = agree_batch(B, world if distributed else 1, reduce_min)
        else:
            B = (4 if torch.cuda.mem_get_info(dev)[0] >= 175e9 else 2) if a.config == "sdxl" else 16     # sdxl: 161 / 115 GB; 6 is no faster
    cfg = {"sd15": sd15_config, "tiny": tiny_config, "sdxl": sdxl_config}[a.config](max_batch=B)
    C_cls, K = a.classes, 3
    t_setup = time.time()
    guided = a.guidance != "none"
    # transform guidance chains P guided steps (P activation stashes); a direct-guidance step is differentiated on its own (one)
    stash = max(1, a.guidance_period) if a.guidance == "transform_guidance" else 1
    weights = None

    def make_engine(c, w, layout):
        return Engine(c, w, enable_grad=guided, max_guidance_period=stash, device=str(dev), layout=layout)

    if stub:
        eng = _StubEngine(B, cfg.latent_size)
    elif distributed and (world > 1 or os.environ.get("DD_FORCE_BROADCAST")):     # the env switch rehearses the RCCL start-up path on 1 GPU
        # rank 0 synthesises + packs the weights once; the PACKED device buffers reach the other ranks in one RCCL broadcast over xGMI
        def load():
            nonlocal weights
            weights = synthetic_weights(cfg, seed=0, num_classes=C_cls)
            return cfg, weights
        _, eng = build_engine_distributed(load, make_engine)
    else:
        weights = synthetic_weights(cfg, seed=0, num_classes=C_cls)
        eng = make_engine(cfg, weights, None)
    sched = DDIMSchedule(cfg.scheduler)
    ts = sched.set_timesteps(a.schedule_steps)
    eng.set_schedule(ts, sched.alphas_cumprod, sched.final_alpha_cumprod, guidance_scale=7.5, gs=1.0, ls=1.0, rho=10.0,
                     constraint_value=0.2, guidance_period=a.guidance_period)
    g = torch.Generator().manual_seed(3)
    D = cfg.guide.feature_dim
    Pc = torch.nn.functional.normalize(torch.randn(C_cls, D, generator=g), dim=-1)
    Pg = torch.nn.functional.normalize(torch.randn(C_cls, K, D, generator=g), dim=-1)
    eng.set_prototypes(Pc, Pg)

    # synthetic dataset shard: the global image list is split across ranks like generate_data.py:1003-1007
    n_total = B * world * (a.steps + a.warmup)
    mine = shard_range(n_total, world, rank)
    shards = [(rank, mine[0], mine[-1] + 1)]
    if distributed and world > 1:
        shards = [None] * world
        dist.all_gather_object(shards, (rank, mine[0], mine[-1] + 1))
    L = cfg.latent_size
    gd = torch.Generator().manual_seed(42 + rank)
    lat = (torch.randn(len(mine), 4, L, L, generator=gd) * 0.18215 * 5).to(dev)
    noise = torch.randn(len(mine), 4, L, L, generator=gd).to(dev)
    e = torch.rand(len(mine), 4, generator=gd).to(dev)
    b = torch.randn(len(mine), 4, generator=gd).to(dev)
    targets = torch.randint(0, C_cls, (len(mine),), generator=gd).to(dev)
    emb = torch.randn(2 * B, cfg.text_len, cfg.unet.cross_attention_dim, generator=gd).to(dev)
    eng.set_prompt(emb)
    if cfg.unet.add_time_dim:          # SDXL text_time conditioning: pooled text embeddings + (original size, crop, target size)
        S = 8.0 * L
        eng.set_added_cond(torch.randn(2 * B, cfg.unet.add_text_dim, generator=gd).to(dev),
                           torch.tensor([[S, S, 0.0, 0.0, S, S]] * (2 * B)).to(dev))
    si = int((1 - a.strength) * len(ts))
    gfirst = len(ts) - a.guidance_step
    gtype = None if not guided else a.guidance
    n_exec = len(ts) - si
    setup_s = time.time() - t_setup

    def step(i, start=None):
        sl = slice(i * B, (i + 1) * B)
        return eng.expand(lat[sl], noise[sl], e[sl], b[sl], targets[sl], si if start is None else start, gtype, gfirst, a.guidance_period,
                          want_image=True)

    def sync():
        if not stub:
            torch.cuda.synchronize()

    def barrier():
        sync()
        if distributed:
            import torch.distributed as dist
            dist.barrier()
        sync()

    prof = None

    def prof_on():
        if not a.no_profile:
            eng.profile_enable(True)     # HIP events around every op of the profiled step (the last warm-up step), on the launch stream

    def prof_off():
        nonlocal prof
        if not a.no_profile:
            prof = eng.profile_read()    # synchronises
            eng.profile_enable(False)

    def reduce_max(v):
        if not distributed:
            return v
        import torch.distributed as dist
        t = torch.tensor([v], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def counted_step(i):
        if i == a.warmup:
            eng.flops_last()             # the FLOP counter restarts with the first timed step
        return step(i)

    dt, (z, img, score) = timed_steps(counted_step, a.steps, a.warmup, barrier, reduce_max, (prof_on, prof_off))
    flops = eng.flops_last()
    assert torch.isfinite(img).all() and torch.isfinite(z).all(), "non-finite output"
    assert sum(hi - lo for _, lo, hi in shards) == n_total and len({r for r, _, _ in shards}) == world, shards

    images, rate = job_rate(a.steps, B, world, dt)
    flops_per_image = flops / (a.steps * B)
    if rank == 0:
        out = {
            "metric": ("HARNESS STUB (no GPU work, not a measurement): " if stub else "") +
                      "512x512 images/sec/node, Caltech-101 5x expand, 50 DDIM steps + energy guidance",
            "value": rate, "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1000.0 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "stub" if stub else "synthetic",
            "config": {"workload": ("BASELINE configs[4] structure: SDXL-base UNet shapes (bf16 attention), " if a.config == "sdxl" else
                                    ("BASELINE configs[3] (StanfordCars sizes): SD-1.5 shapes " if C_cls == 196 else "BASELINE configs[1]: SD-1.5 shapes ")) +
                                   "%dx%d, %d-step DDIM schedule, strength %.2f (%d executed steps), "
                                   "CFG 7.5, %s P=%d (class+group prototypes C=%d K=3 D=%d, ResNet-50 guide), final VAE decode"
                                   % (8 * L, 8 * L, len(ts), a.strength, n_exec, a.guidance, a.guidance_period, C_cls, D),
                       "images_per_step_per_gpu": B, "sharding": "image shards per rank (generate_data.py:1003-1007), no data-path collective",
                       "shards": [[int(r), int(lo), int(hi)] for r, lo, hi in sorted(shards)],
                       "weights": "seeded synthetic, exact SD-1.x / AutoencoderKL / ResNet-50 shapes",
                       "algorithmic_tflop_per_image": flops_per_image / 1e12,
                       "flop_note": "FLOPs the engine executed (2 per MAC, dgrad-only VJP); the part of the UNet in front of the first "
                                    "cross-attention is identical for the two CFG halves and runs once (DESIGN.md section 6)",
                       "setup_s": setup_s,
                       "workspace_gb": eng.workspace_bytes() / 1e9},
            "e2e_tflops_per_gpu": flops / dt / 1e12,
            "e2e_frac_of_bf16_peak": flops / dt / 1e12 / PEAK_BF16_TFLOPS,
        }
        if stub:
            out["stub"] = True
        if prof is not None:
            cv = prof["conv_gemm"]
            ach = cv["flops"] / (cv["ms"] * 1e-3) / 1e12 if cv["ms"] > 0 else 0.0
            out["roofline"] = {"bound": "mfma", "kernel": "conv_halo_kernel + gemm_pps_kernel + gemm_ws_kernel + conv_gemm_big_kernel + conv_gemm_kernel + splitk_reduce_kernel (implicit-GEMM conv/linear fwd+dgrad)",
                               "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS,
                               "traffic": None, "launches": cv["ops"], "avg_launch_ms": cv["ms"] / max(cv["ops"], 1),
                               "algorithmic_flops_per_launch": cv["flops"] / max(cv["ops"], 1),
                               "family_ms": {k: v["ms"] for k, v in prof.items()}}
            out["roofline"].update(recorded_traffic(B, a.config))
        # the metric is on record before the extras run (they drive the PNG writer threads and a CPU oracle: a hang there must not lose it)
        sys.stderr.write("[bench] measured, extras pending: " + json.dumps({k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step")}) + "\n")
        sys.stderr.flush()
        if world == 1 and not stub and not a.no_strength1 and a.config == "sd15" and a.strength < 1.0:
            # SURVEY.md section 8(d) case (ii): strength 1.0 = every step of the schedule executed (generate_data.py:1174 start index 0),
            # the same guidance window; 1 untimed + 2 timed steps on the engine and inputs of the main measurement
            try:
                nb = a.steps + a.warmup
                step(0, 0)
                sync()
                t0 = time.time()
                for i in range(2):
                    step((1 + i) % nb, 0)
                sync()
                d1 = (time.time() - t0) / 2
                out["strength1"] = {"value": B / d1, "unit": "images/s", "ms_per_step": 1000.0 * d1, "steps": 2, "executed_ddim_steps": len(ts),
                                    "note": "strength 1.0 (all %d schedule steps executed), otherwise the main line's workload" % len(ts)}
            except Exception as ex:
                out["strength1"] = {"value": None, "note": "failed: %r" % (ex,)}
        if world == 1 and not stub and not a.no_cli and a.config == "sd15":
            try:
                out["output_stage"] = cli_rate(eng, cfg, sched, a, B, a.cli_units or 12 * B)
            except Exception as ex:      # an extra, never the metric
                out["output_stage"] = {"cli_images_per_s": None, "cli_note": "failed: %r" % (ex,)}
        if world == 1 and not stub and not a.no_cpu_baseline and a.config == "sd15":
            try:
                wcpu = weights
                out["cpu_baseline"] = cpu_baseline(cfg, wcpu, n_exec, a.guidance_period, flops_per_image)
            except Exception as ex:  # the baseline is reported, never the product path
                out["cpu_baseline"] = {"value": None, "unit": "images/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (ex,)}
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    eng.close()
    if distributed:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
