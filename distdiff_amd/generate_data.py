"""Drop-in CLI for the expansion step of haoweiz23/DistDiff: same flags, defaults, shard function, resume rule and
output layout as /root/reference/generate_data.py (parse_args :164-639, main :815-1249), with the body of the hot
loop (:1130-1236) executed by the MI355X engine (libdistdiff_hip.so) instead of diffusers/timm/autograd.

    python generate_data.py --guidance_type=transform_guidance -a resnet50 -d caltech-101 --output_dir OUT \
        --pretrained_model_name_or_path /path/to/sd-v1-4 --K 3 --train_batch_size 1 \
        --optimize_targets global_prototype-local_prototype --strength 0.5 --num_images_per_prompt 5 \
        --guidance_step 20 --guidance_period 2 --encoder_weight_path ckpt.pth.tar --guidance_scale 7.5 \
        --constraint_value 0.2 --rho 10 --total_split 4 --split 0

Superset over the reference (documented, defaults reproduce it): `--steps` and `--resolution` are honoured
(the reference parses but ignores them); `--synthetic N` runs on N seeded synthetic images with
synthetic weights (no checkpoints / datasets are needed; used by tests and benchmarks); `--gpus N` replaces the reference's
per-GPU shell fan-out (`--split k --total_split N` per process, scripts/exps/expand_diff.sh:19-24): one launcher spawns N ranks,
rank 0 loads and packs the weights once and broadcasts the packed device buffers over RCCL/xGMI (distdiff_amd/launcher.py).
`--center_crop` is accepted and ignored like the reference, which hard-codes `center_crop=False` for SDDataset
(generate_data.py:1001).  `--offset_noise` (:1164-1168) and `--language_enhance` (dataloader.py:769-778, 832-833) are implemented.
Random streams: `e`/`b` of transform_guidance come from the CPU global generator in the reference's order (one draw of each per
reference batch, :692-694); the initial noise comes from a dedicated generator seeded with --seed (the reference draws it from the
CUDA global generator, whose stream no other device reproduces) -- see INTEGRATION.md.
The stage before the loop runs on the engine too: image latents come from the reference's cache
`save/vae_embedding/<dataset>/<model>/image_latents.pt` when it exists and are otherwise produced by the HIP VAE encoder and
written to that path in the same format (dataloader.py:788-811); class prompts go through the Hugging Face tokenizer of the
model directory and the HIP CLIP text encoder (dataloader.py:633-661, 780-786).  `--synthetic N --synthetic_encode` drives
both encoders on seeded synthetic images / token ids.
"""
import argparse
import logging
import math
import os
import sys

import torch

log = logging.getLogger("distdiff_amd")

CUSTOM_TEMPLATE_DEFAULT = "a photo of a {}."     # dataloader.py:58 (caltech-101)

# flags the reference parses (DreamBooth leftovers) and never reads on this path: accepted and ignored
_IGNORED_VALUE_FLAGS = ["--revision", "--variant", "--dataset_name", "--dataset_config_name", "--train_data_dir", "--image_column",
                        "--caption_column", "--tokenizer_name", "--instance_data_dir", "--class_data_dir", "--instance_prompt",
                        "--class_prompt", "--prior_loss_weight", "--num_class_images", "--val_batch_size", "--sample_batch_size",
                        "--num_train_epochs", "--max_train_steps", "--max_train_samples", "--checkpointing_steps",
                        "--checkpoints_total_limit", "--resume_from_checkpoint", "--gradient_accumulation_steps", "--lr_scheduler",
                        "--lr_warmup_steps", "--lr_num_cycles", "--lr_power", "--max_grad_norm", "--logging_dir", "--report_to",
                        "--mixed_precision", "--prior_generation_precision", "--local_rank", "--snr_gamma", "--tokenizer_max_length",
                        "--validation_images", "--class_labels_conditioning", "--validation_scheduler", "--dataloader_num_workers"]
_IGNORED_BOOL_FLAGS = ["--center_crop", "--random_flip", "--with_prior_preservation", "--train_text_encoder", "--gradient_checkpointing",
                       "--scale_lr", "--use_8bit_adam", "--allow_tf32", "--enable_xformers_memory_efficient_attention",
                       "--set_grads_to_none", "--pre_compute_text_embeddings", "--text_encoder_use_attention_mask",
                       "--skip_save_text_encoder"]


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="MI355X-native DistDiff data expansion (drop-in for generate_data.py)")
    p.add_argument("--pretrained_model_name_or_path", type=str, default="CompVis/stable-diffusion-v1-4")
    p.add_argument("--dataset", "-d", type=str, default="caltech-101")
    p.add_argument("--arch", "-a", type=str, default="open_clip_vit_b32")
    p.add_argument("--encoder_weight_path", type=str, default=None)
    p.add_argument("--guidance_type", default=None)
    p.add_argument("--constraint_value", default=0.8, type=float)
    p.add_argument("--steps", default=50, type=int)
    p.add_argument("--K", default=3, type=int)
    p.add_argument("--guidance_step", default=1, type=int)
    p.add_argument("--guidance_period", default=1, type=int)
    p.add_argument("--total_split", default=8, type=int)
    p.add_argument("--split", default=0, type=int)
    p.add_argument("--num_images_per_prompt", default=4, type=int)
    p.add_argument("--first_image_index", default=0, type=int)
    p.add_argument("--optimize_targets", default=None, type=str)
    p.add_argument("--rho", type=float, default=10.0)
    p.add_argument("--gs", type=float, default=1.0)
    p.add_argument("--ls", type=float, default=1.0)
    p.add_argument("--strength", type=float, default=0.9)
    p.add_argument("--cache_dir", type=str, default=None)
    p.add_argument("--resolution", type=int, default=512)
    p.add_argument("--text_to_img", default=False, action="store_true")
    p.add_argument("--output_dir", type=str, default="data_expand")
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--train_batch_size", type=int, default=2)
    p.add_argument("--guidance_scale", type=float, default=7.5)
    p.add_argument("--do_classifier_free_guidance", type=bool, default=True)   # type=bool: any string -> True, as in the reference
    p.add_argument("--offset_noise", action="store_true", default=False)
    p.add_argument("--language_enhance", "-le", action="store_true", default=False)
    for f in _IGNORED_VALUE_FLAGS:
        p.add_argument(f, default=None, help=argparse.SUPPRESS)
    for f in _IGNORED_BOOL_FLAGS:
        p.add_argument(f, action="store_true", default=False, help=argparse.SUPPRESS)
    # superset
    p.add_argument("--synthetic", type=int, default=0, help="run on N seeded synthetic images with synthetic weights")
    p.add_argument("--synthetic_classes", type=int, default=4)
    p.add_argument("--synthetic_encode", action="store_true",
                   help="with --synthetic: produce latents / prompt embeddings with the HIP VAE encoder and CLIP text encoder")
    p.add_argument("--tiny", action="store_true", help="use the tiny test architecture (with --synthetic)")
    p.add_argument("--synthetic_arch", default="sd15", choices=["sd15", "sdxl"],
                   help="with --synthetic: the SD-1.x structure or the SDXL-base one (two text towers, text_time conditioning; BASELINE "
                   "configs[4]).  A real model directory is recognised by its text_encoder_2/ sub-folder")
    p.add_argument("--engine_batch", type=int, default=0, help="images per engine launch (0 = the largest of 32 / 16 / 8 whose workspace fits the free HBM -- engine.batch_for_free_hbm: "
                   "about 5.85 GB per image with transform guidance at 512x512 + 12 GB, i.e. 199 / 106 / 59 GB; 32 is 4 %% faster than 16 and 16 12 %% "
                   "faster than 8; SDXL at 1024x1024: 4 or 2 --, the minimum over the ranks of a run); units (image, expand index) are independent")
    p.add_argument("--attn_fp8", action="store_true", help="BASELINE configs[4]: P.V of the UNet's d = 64 attention heads (SDXL) on the fp8 MFMA "
                   "(dd_config.unet_attn_fp8).  Off by default: the bf16 form is faster on MI355X and closer to fp32 (DESIGN.md Appendix A row 28)")
    p.add_argument("--data_root", type=str, default="data")
    p.add_argument("--gpus", type=int, default=1, help="spawn this many ranks (one per GPU) that shard the images like --total_split; "
                   "weights are loaded once on rank 0 and broadcast over RCCL")
    p.add_argument("--device", type=str, default=None)
    args = p.parse_args(argv)
    env_local_rank = int(os.environ.get("LOCAL_RANK", -1))
    if env_local_rank != -1:
        args.local_rank = env_local_rank
    if args.text_to_img:
        raise SystemExit("--text_to_img is broken in the reference (generate_data.py:1155 uses `generator` before assignment) and is "
                         "out of scope")
    if not args.do_classifier_free_guidance:
        raise SystemExit("the engine always runs classifier-free guidance (the reference's type=bool flag cannot be switched off either)")
    return args


# ------------------------------------------------------------------------------------------------
# dataset side: the reference's SDDataset.__getitem__ dict (dataloader.py:813-849), from caches or synthetic
# ------------------------------------------------------------------------------------------------
class ExpansionDataset:
    """image_latents [N,4,L,L], per-class text embeds [C,T,Dm], uncond embeds [1,T,Dm], targets, class names, image paths; SDXL-style
    models also carry the pooled text embeddings of the second tower (per class [C,P] and of the negative prompt [1,P])."""

    def __init__(self, latents, class_embeds, uncond_embeds, targets, class_names, image_paths, class_pooled=None, uncond_pooled=None):
        self.latents, self.class_embeds, self.uncond = latents, class_embeds, uncond_embeds
        self.targets, self.class_names, self.image_paths = targets, class_names, image_paths
        self.class_pooled, self.uncond_pooled = class_pooled, uncond_pooled

    def __len__(self):
        return self.latents.shape[0]

    @staticmethod
    def synthetic(cfg, n, n_classes, seed):
        g = torch.Generator().manual_seed(seed)
        L = cfg.latent_size
        per = math.ceil(n / n_classes)
        targets = torch.tensor([min(i // per, n_classes - 1) for i in range(n)])     # class-sorted like the reference datasets
        names = ["class %d" % c for c in range(n_classes)]
        ds = ExpansionDataset(torch.randn(n, 4, L, L, generator=g) * 0.9,
                              torch.randn(n_classes, cfg.text_len, cfg.unet.cross_attention_dim, generator=g),
                              torch.randn(1, cfg.text_len, cfg.unet.cross_attention_dim, generator=g),
                              targets, names, ["%s/image_%04d.jpg" % (names[int(t)], i) for i, t in enumerate(targets)])
        if cfg.unet.add_time_dim:
            ds.class_pooled = torch.randn(n_classes, cfg.unet.add_text_dim, generator=g)
            ds.uncond_pooled = torch.randn(1, cfg.unet.add_text_dim, generator=g)
        return ds

    @staticmethod
    def from_dataset(args, cfg, eng):
        """Train listing as StandardDataLoader.load_dataset (dataloader.py:95-130; caltech-101 :272-315, stanford_cars :167-228);
        latents and prompt embeddings as SDDataset.__init__ (dataloader.py:750-796)."""
        from .datasets import load_train_listing
        from .preprocess import class_prompt_embeddings, load_or_encode_latents, load_tokenizer
        paths, targets, names = load_train_listing(args.dataset, args.data_root)
        # image latents: the reference's cache file if present (validated against this listing), else the HIP VAE encoder fills it
        # (dataloader.py:788-796).  center_crop=False: the reference hard-codes it for SDDataset (generate_data.py:1001).
        lat = load_or_encode_latents(eng, args.dataset, args.pretrained_model_name_or_path, paths, args.resolution,
                                     center_crop=False, seed=args.seed or 0)
        lat = torch.cat([x.float().cpu() for x in lat], dim=0)
        # class prompts + the empty prompt through the HIP CLIP text encoder (dataloader.py:766-786)
        tokenizer = load_tokenizer(args.pretrained_model_name_or_path, args.revision)
        if cfg.text2 is not None:
            # SDXL layout: the class prompts go through both towers (tokenizer / tokenizer_2), StableDiffusionXLPipeline.encode_prompt
            from .preprocess import class_prompt_embeddings_sdxl
            if args.language_enhance:
                raise SystemExit("--language_enhance is not built for two-tower (SDXL) models")
            tokenizer_2 = load_tokenizer(args.pretrained_model_name_or_path, args.revision, subfolder="tokenizer_2")
            ce, ue, cp, up = class_prompt_embeddings_sdxl(eng, (tokenizer, tokenizer_2), args.dataset, names)
            return ExpansionDataset(lat, ce, ue, torch.tensor(targets), names, paths, class_pooled=cp, uncond_pooled=up)
        class_embeds, uncond = class_prompt_embeddings(eng, tokenizer, args.dataset, names, language_enhance=args.language_enhance,
                                                       data_root=args.data_root)
        return ExpansionDataset(lat, class_embeds, uncond, torch.tensor(targets), names, paths)

    @staticmethod
    def synthetic_encoded(cfg, eng, n, n_classes, seed):
        """--synthetic N --synthetic_encode: like `synthetic`, but the latents come from seeded synthetic images through the
        HIP VAE encoder and the prompt embeddings from seeded token ids through the HIP text encoder (8f-2 path end to end)."""
        from .preprocess import encode_token_ids
        base = ExpansionDataset.synthetic(cfg, n, n_classes, seed)
        g = torch.Generator().manual_seed(seed + 1)
        S, L, B = 8 * cfg.latent_size, cfg.latent_size, eng.B
        lat = []
        for i in range(0, n, B):
            k = min(B, n - i)
            x = torch.rand(B, 3, S, S, generator=g) * 2 - 1
            noise = torch.randn(B, cfg.vae.latent_channels, L, L, generator=g)
            lat.append(eng.vae_encode(x, noise).cpu()[:k])
        ids = torch.randint(0, cfg.text.vocab_size, (n_classes + 1, cfg.text_len), generator=g).int()
        if cfg.text2 is not None:
            from .preprocess import encode_token_ids_sdxl
            ids2 = torch.randint(0, cfg.text2.vocab_size, (n_classes + 1, cfg.text_len), generator=g).int()
            emb, pooled = encode_token_ids_sdxl(eng, ids, ids2)
            return ExpansionDataset(torch.cat(lat), emb[:-1], emb[-1:], base.targets, base.class_names, base.image_paths,
                                    class_pooled=pooled[:-1], uncond_pooled=pooled[-1:])
        emb = encode_token_ids(eng, ids)
        return ExpansionDataset(torch.cat(lat), emb[:-1], emb[-1:], base.targets, base.class_names, base.image_paths)


def output_path(output_dir, class_name, image_path, image_i):
    """generate_data.py:1134-1135 / 1231-1232."""
    stem = os.path.basename(image_path).split(".")[0]
    return "%s/%s/%s_expand_%d.png" % (output_dir, class_name, stem, image_i)


def save_png(img, path):
    """Writes one image: uint8 HWC array (already quantised on the GPU) or, for the CPU tests, a CHW float tensor in [0,1]
    with torchvision.utils.save_image semantics (mul(255).add_(0.5).clamp_(0,255) -> uint8)."""
    from PIL import Image
    if isinstance(img, torch.Tensor) and img.dtype != torch.uint8:
        img = img.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to("cpu", torch.uint8)
    arr = img.cpu().numpy() if isinstance(img, torch.Tensor) else img
    os.makedirs(os.path.dirname(path), exist_ok=True)
    Image.fromarray(arr).save(path)


class AsyncPNGWriter:
    """Output stage: quantise on the GPU, copy device->pinned host asynchronously, encode PNGs on a thread
    pool.  `submit` never waits for the batch it is given: the uint8 conversion, the device->host copies and an event are enqueued
    behind that batch's kernels, and the host goes on to prepare and enqueue the NEXT batch; what it waits for is the event of the
    batch before -- finished long ago -- whose images then go to the writer threads while the GPU runs the batch just enqueued.
    The host is thus exactly one batch ahead and the GPU never idles between batches.  Pinned host buffers are a small ring, reused
    once the PNG jobs reading them are done (a pinned allocation per batch costs milliseconds of host time)."""
    RING = 3

    def __init__(self, engine, writer=save_png, threads=None):
        if not threads:      # the launcher sets DD_PNG_THREADS for its ranks; a plain single-process run takes the same function's answer
            from .launcher import png_threads
            threads = int(os.environ.get("DD_PNG_THREADS", 0)) or png_threads(int(os.environ.get("WORLD_SIZE", "1")))
        from concurrent.futures import ThreadPoolExecutor
        self.engine, self.writer = engine, writer
        self.pool = ThreadPoolExecutor(max_workers=threads)
        self.pending = None
        self.ring = []          # [pinned buffer, futures of the PNG jobs reading it]
        self.futures = []
        self.written = 0

    def _host_buffer(self, shape):
        """A pinned uint8 buffer of at least `shape` none of whose PNG jobs is still running (oldest first; allocates up to RING)."""
        n = 1
        for d in shape:
            n *= int(d)
        for slot in self.ring:
            if slot[0].numel() >= n and all(f.done() for f in slot[1]) and not slot[2]:
                slot[1], slot[2] = [], True
                return slot
        if len(self.ring) < self.RING:
            slot = [torch.empty(n, dtype=torch.uint8, pin_memory=True), [], True]
            self.ring.append(slot)
            return slot
        slot = next(s for s in self.ring if not s[2] and s[0].numel() >= n)      # all busy: wait for the oldest one's jobs
        for f in slot[1]:
            f.result()
        slot[1], slot[2] = [], True
        return slot

    def submit(self, img_dev, paths, scores_dev=None, after=None):
        """img_dev [n,3,H,W] fp32 on the device (n = len(paths)); scores_dev: a small device tensor read back with the images;
        after(scores_host or None): called on the host once this batch has arrived (one submit later, or at close)."""
        u8 = self.engine.image_to_u8(img_dev)
        slot = self._host_buffer(u8.shape)
        host = slot[0][:u8.numel()].view(u8.shape)
        host.copy_(u8, non_blocking=True)
        sc_host = None
        if scores_dev is not None:
            sc_host = torch.empty(scores_dev.shape, dtype=scores_dev.dtype, pin_memory=True)
            sc_host.copy_(scores_dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        prev, self.pending = self.pending, (ev, slot, host, paths, (u8, img_dev, scores_dev), sc_host, after)
        self._drain(prev)

    def _drain(self, item):
        if item is None:
            return
        ev, slot, host, paths, _keep, sc_host, after = item
        ev.synchronize()
        for k, p in enumerate(paths):
            f = self.pool.submit(self.writer, host[k].numpy(), p)
            slot[1].append(f)
            self.futures.append(f)
            self.written += 1
        slot[2] = False         # in flight on the writer threads only: reusable once they are done
        if after is not None:
            after(sc_host)
        self.futures = [f for f in self.futures if not f.done() or f.exception() is not None]

    def close(self):
        self._drain(self.pending)
        self.pending = None
        for f in self.futures:
            f.result()
        self.pool.shutdown()
        return self.written


def class_embedding(ds, target):
    """`example["instance_prompt_ids"]` of SDDataset.__getitem__ (dataloader.py:832-835): the class prompt embedding, or with
    --language_enhance `random.choice` among the class's sentences (python `random`, seeded by --seed like accelerate.set_seed)."""
    ce = ds.class_embeds[int(target)]
    if ce.dim() == 3:                       # [n_sentences, T, D]
        import random
        return ce[random.randrange(ce.shape[0])]
    return ce


def run_expansion(args, engine, sched, ds, writer=save_png, rng_device="cpu"):
    """The reference main loop, generate_data.py:1001-1009 (shard) and :1130-1236 (batches x expand index)."""
    from .launcher import shard_range
    from .scheduler import guide_window, start_index
    idx = [i for i in shard_range(len(ds), args.total_split, args.split) if i < len(ds)]
    ts = sched.timesteps
    n = len(ts)
    si = start_index(args.strength, n)
    gfirst, gcount = (0, 0)
    if args.guidance_type:
        gfirst, gcount = guide_window(n, args.guidance_step, args.guidance_period)
        log.info("Guidance timesteps: %s", ", ".join(str(t) for t in ts[gfirst:gfirst + gcount]))
    written = 0
    dev = engine.device if engine is not None else torch.device("cpu")
    async_writer = AsyncPNGWriter(engine, writer) if hasattr(engine, "image_to_u8") and dev.type == "cuda" else None
    # Units of work = (train image i, expand index j), enumerated in the reference's order and filtered by its resume rule
    # (a (batch, image_i) group is skipped when all of its PNGs exist, :1132-1143).  The units are independent, so they are
    # re-packed into engine batches of EB = engine.B images (>= train_batch_size): --train_batch_size 1 of the script of record
    # would leave an MI355X mostly idle.  What is NOT independent of the reference batch is the energy: it is a `.mean()` over the
    # train_batch_size images of a group (:709-719, :750-760), so every unit carries w = 1/|its group| into the engine
    # (dd_set_sample_weights) and the logged score is the mean over the group, not over the engine batch.
    B = args.train_batch_size
    EB = engine.B
    transform = args.guidance_type == "transform_guidance"
    noise_gen = torch.Generator(device=rng_device)
    if args.seed is None:
        # no seed (programmatic callers; the CLI's default is the reference's 42, generate_data.py:370): fresh noise per run, like the
        # reference's unseeded global generator, and logged so that the run can be repeated
        log.info("initial-noise generator seed: %d", noise_gen.seed())
    else:
        # the reference seeds its global generators once per process with --seed (set_seed, :859-861), so every invocation -- a resumed
        # run, a later --first_image_index -- replays the same noise stream from its first unit; kept as is
        noise_gen.manual_seed(int(args.seed))
    Cl = ds.latents.shape[1]
    units = []      # (dataset index, path, group id, group size, e[4], b[4], prompt embedding, noise offset[C] or None)
    n_groups = 0
    for s0 in range(0, len(idx), B):
        bidx = idx[s0:s0 + B]
        prompts = [class_embedding(ds, ds.targets[i]) for i in bidx]      # one __getitem__ per image per batch (:1130)
        for image_i in range(args.first_image_index, args.num_images_per_prompt):
            paths = [output_path(args.output_dir, ds.class_names[int(ds.targets[i])], ds.image_paths[i], image_i) for i in bidx]
            if all(os.path.exists(p) for p in paths):
                for p in paths:
                    print("File %s exists, so skipped." % p)
                continue
            nb = len(bidx)
            # :1164-1168: noise += 0.1 * randn(b, c, 1, 1)
            off = 0.1 * torch.randn(nb, Cl, 1, 1, generator=noise_gen, device=rng_device).cpu() if args.offset_noise else None
            if transform:
                e = torch.rand([nb, 4, 1, 1])                                 # :692 CPU global RNG, one draw per reference batch
                b = torch.zeros([nb, 4, 1, 1]).normal_(0, 1)                  # :694
            else:
                e = b = torch.zeros([nb, 4, 1, 1])
            for k, (i, p) in enumerate(zip(bidx, paths)):
                units.append((i, p, n_groups, nb, e[k], b[k], prompts[k], off[k] if off is not None else None))
            n_groups += 1
    group_scores = {}
    for u0 in range(0, len(units), EB):
        chunk = units[u0:u0 + EB]
        nb = len(chunk)
        chunk_p = chunk + [chunk[-1]] * (EB - nb)                           # ragged last batch: pad to the static batch
        pad = [u[0] for u in chunk_p]
        paths = [u[1] for u in chunk]
        lat = ds.latents[pad]
        tg = ds.targets[pad]
        noise = torch.randn(lat.shape, generator=noise_gen, device=rng_device).to(lat.dtype).cpu()      # :1170
        if args.offset_noise:
            noise = noise + torch.stack([u[7] for u in chunk_p])
        e = torch.stack([u[4] for u in chunk_p])
        b = torch.stack([u[5] for u in chunk_p])
        emb = torch.cat([ds.uncond.expand(EB, -1, -1), torch.stack([u[6] for u in chunk_p])])   # cat[negative, prompt], :1184
        engine.set_prompt(emb.to(dev))
        if ds.class_pooled is not None:
            # SDXL added_cond_kwargs (StableDiffusionXLPipeline.__call__): pooled text embeddings cat[negative, prompt] and
            # add_time_ids = (original size, crop top-left, target size), the same for both halves
            S = float(8 * ds.latents.shape[-1])
            pooled = torch.cat([ds.uncond_pooled.expand(EB, -1), ds.class_pooled[tg]])
            engine.set_added_cond(pooled.to(dev), torch.tensor([[S, S, 0.0, 0.0, S, S]]).expand(2 * EB, -1).to(dev))
        if args.guidance_type and hasattr(engine, "set_sample_weights"):
            engine.set_sample_weights([1.0 / u[3] for u in chunk] + [0.0] * (EB - nb))
        z, img, score = engine.expand(lat, noise, e, b, tg, si, args.guidance_type or None, gfirst, gcount, want_image=True)

        def log_scores(per_image, chunk=chunk):
            for u, sc in zip(chunk, per_image):
                acc = group_scores.setdefault(u[2], [])
                acc.append(sc)
                if len(acc) == u[3]:                                          # the reference's log line (:1208-1216), one per batch
                    log.info("%s at t=%d for %d steps, score: %.4f", args.guidance_type, ts[gfirst], gcount, sum(acc) / len(acc))
                    del group_scores[u[2]]

        if async_writer is not None:
            # nothing here waits for this batch: images and scores are read back behind its kernels, the log lines and the PNG jobs
            # follow one batch later (AsyncPNGWriter), and the host goes on to enqueue the next batch
            sc_dev = engine.image_scores() if args.guidance_type and hasattr(engine, "image_scores") else None
            async_writer.submit(img[:nb], paths, sc_dev, (lambda h: log_scores(h.tolist())) if sc_dev is not None else None)
        else:
            if args.guidance_type:
                log_scores(engine.image_scores().cpu().tolist() if hasattr(engine, "image_scores") else [float(score)] * EB)
            for k in range(nb):
                writer(img[k], paths[k])
                written += 1
    if async_writer is not None:
        written += async_writer.close()
    return written


def load_config_and_weights(args, B):
    """The model objects the reference builds at generate_data.py:863-922 and :1100-1104, as (EngineConfig, state dicts)."""
    from .config import GUIDE_ARCHS, from_model_dir, guide_config, sd15_config, sdxl_config, tiny_config, tiny_sdxl_config
    from .model_utils import SUPPORTED, create_model
    from .weights import load_safetensors_dir, synthetic_weights
    latent = args.resolution // 8
    if args.synthetic:
        if args.synthetic_arch == "sdxl":
            cfg = tiny_sdxl_config(max_batch=B) if args.tiny else sdxl_config(latent, B)
        else:
            cfg = tiny_config(max_batch=B) if args.tiny else sd15_config(latent, B)
        if args.arch in GUIDE_ARCHS:                 # -a resnext50 / wideresnet50 / open_clip_vit_b32 (full-size guide only)
            if args.tiny and GUIDE_ARCHS[args.arch].get("kind") == "vit":
                cfg.guide.kind, cfg.guide.input_size = "vit", 64
                cfg.guide.vit_width, cfg.guide.vit_layers, cfg.guide.vit_heads, cfg.guide.vit_mlp, cfg.guide.vit_out = 64, 2, 2, 128, 32
                cfg.guide.vit_patch = 16
            else:
                for k, v in GUIDE_ARCHS[args.arch].items():
                    setattr(cfg.guide, k, v)
            cfg.guide.arch = args.arch
        return cfg, synthetic_weights(cfg, seed=0, num_classes=args.synthetic_classes, encoders=args.synthetic_encode)
    path = args.pretrained_model_name_or_path
    if not os.path.isdir(path):
        raise SystemExit("%s is not a local model directory (no network access here); pass a local Hugging Face layout with unet/, vae/, "
                         "text_encoder/, tokenizer/, scheduler/, or use --synthetic N" % path)
    cfg = from_model_dir(path, latent, B)
    arch = args.arch if (args.guidance_type or args.arch in SUPPORTED) else "resnet50"   # unguided runs never evaluate the guide
    cfg.guide = guide_config(arch)
    guide = create_model(arch, pretrained=False, num_classes=1, weight_path=args.encoder_weight_path if args.guidance_type else None)
    gsd = guide.state_dict()
    if cfg.guide.kind == "vit":                   # the CLIP checkpoint also carries the text tower: only the image tower is on this path
        gsd = {k: v for k, v in gsd.items() if k.startswith("visual.")}
    weights = {"unet": load_safetensors_dir(path, "unet"), "vae": load_safetensors_dir(path, "vae"), "guide": gsd,
               "text": {k: v for k, v in load_safetensors_dir(path, "text_encoder").items() if "position_ids" not in k}}
    if cfg.text2 is not None:
        weights["text2"] = {k: v for k, v in load_safetensors_dir(path, "text_encoder_2").items() if "position_ids" not in k}
    return cfg, weights


def auto_engine_batch(args, dev, distributed=False):
    """Static engine batch when --engine_batch is not given: 8 for --tiny; at 512x512 the largest of 32 / 16 / 8 whose workspace fits the
    free HBM (engine.batch_for_free_hbm: the two activation stashes of the chained guided steps + the liveness-packed gradient slab,
    186.7 GB measured at 32 images; 32 images per launch are 4 % faster than 16 on an MI355X); 16 otherwise.  Ranks of one run agree on the minimum."""
    if args.tiny:
        return 8
    on_gpu = torch.device(dev).type == "cuda" and torch.cuda.is_available()
    if args.synthetic_arch == "sdxl" or os.path.isdir(os.path.join(args.pretrained_model_name_or_path, "text_encoder_2")):
        # SDXL-base at 1024 x 1024: 161 GB at 4 images, 115 GB at 2 (bench.py --config sdxl)
        B = (4 if torch.cuda.mem_get_info(torch.device(dev))[0] >= 175e9 else 2) if on_gpu else 2
    elif args.resolution == 512 and on_gpu:
        from .engine import batch_for_free_hbm
        B = batch_for_free_hbm(torch.cuda.mem_get_info(torch.device(dev))[0], guided=bool(args.guidance_type))
    else:
        return 16
    if distributed and on_gpu:
        # rank 0's configuration (max_batch included) is what every rank builds its engine from (build_engine_distributed): the batch
        # must fit the rank with the LEAST free HBM, and every rank's loop must pack the same EB
        import torch.distributed as dist
        t = torch.tensor([B], dtype=torch.int64, device=torch.device(dev))
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        B = int(t.item())
    return B


def build_engine(args, device=None, distributed=False):
    from .engine import Engine
    from .scheduler import DDIMSchedule
    dev = device or args.device or "cuda:0"
    B = args.engine_batch or max(args.train_batch_size, auto_engine_batch(args, dev, distributed))
    guided = bool(args.guidance_type)
    # transform_guidance differentiates through P chained steps (P activation stashes); direct_guidance one step at a time
    stash = max(1, args.guidance_period) if args.guidance_type == "transform_guidance" else 1

    def make_engine(cfg, weights, layout):
        return Engine(cfg, weights, enable_grad=guided, max_guidance_period=stash, device=dev, layout=layout, attn_fp8=args.attn_fp8)

    if distributed:
        # rank 0 loads + packs once; the packed device buffers go to the other ranks in one RCCL broadcast (launcher.py)
        from .launcher import build_engine_distributed
        cfg, eng = build_engine_distributed(lambda: load_config_and_weights(args, B), make_engine)
    else:
        cfg, weights = load_config_and_weights(args, B)
        eng = make_engine(cfg, weights, None)
    sched = DDIMSchedule(cfg.scheduler)
    ts = sched.set_timesteps(args.steps)
    targets = (args.optimize_targets or "").split("-")
    eng.set_schedule(ts, sched.alphas_cumprod, sched.final_alpha_cumprod, guidance_scale=args.guidance_scale, gs=args.gs, ls=args.ls,
                     rho=args.rho, constraint_value=args.constraint_value, use_global="global_prototype" in targets,
                     use_local="local_prototype" in targets, guidance_period=args.guidance_period)
    return cfg, eng, sched


def main(argv=None):
    import time
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    logging.basicConfig(format="%(asctime)s - %(levelname)s - %(name)s - %(message)s", datefmt="%m/%d/%Y %H:%M:%S", level=logging.INFO)
    if args.guidance_type not in (None, "transform_guidance", "direct_guidance"):
        raise SystemExit("unknown --guidance_type %r" % args.guidance_type)
    from .model_utils import SUPPORTED
    if args.guidance_type and args.arch not in SUPPORTED and not args.synthetic:
        raise SystemExit("guide arch %r is not built (built: %s; SURVEY.md section 8f-4)" % (args.arch, ", ".join(SUPPORTED)))
    if args.gpus > 1 and "RANK" not in os.environ:
        # the in-process fan-out that replaces single_exp.sh / expand_diff.sh's one shell per GPU: start the ranks before any GPU call
        from .launcher import spawn_ranks, visible_gpu_count
        have = visible_gpu_count()                # sysfs / visibility variables: the parent never loads the HIP runtime
        if 0 < have < args.gpus:
            raise SystemExit("--gpus %d but only %d GPU(s) are visible" % (args.gpus, have))
        return spawn_ranks(args.gpus, argv)
    distributed = "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1
    rank, world, device = 0, 1, None
    if distributed:
        from .launcher import init_distributed
        lr = int(os.environ.get("LOCAL_RANK", "0"))
        if lr >= torch.cuda.device_count():
            raise SystemExit("rank %s: LOCAL_RANK %d but only %d GPU(s) are visible" % (os.environ.get("RANK"), lr, torch.cuda.device_count()))
        device = "cuda:%d" % lr
        torch.cuda.set_device(torch.device(device))
        rank, world = init_distributed(device)
        args.split, args.total_split = rank, world          # the reference's --split / --total_split, one per rank
    if os.environ.get("OMP_NUM_THREADS"):
        torch.set_num_threads(max(1, int(os.environ["OMP_NUM_THREADS"])))       # the launcher's per-rank share of the host cores
    if args.seed is not None:
        import random
        import numpy as np
        random.seed(args.seed); np.random.seed(args.seed); torch.manual_seed(args.seed)   # accelerate.set_seed, :861
    cfg, eng, sched = build_engine(args, device=device, distributed=distributed)
    dev = eng.device
    if args.synthetic:
        if args.synthetic_encode:
            ds = ExpansionDataset.synthetic_encoded(cfg, eng, args.synthetic, args.synthetic_classes, seed=args.seed or 0)
        else:
            ds = ExpansionDataset.synthetic(cfg, args.synthetic, args.synthetic_classes, seed=args.seed or 0)
        if args.guidance_type:
            g = torch.Generator().manual_seed(3)
            D = cfg.guide.feature_dim
            Pc = torch.randn(args.synthetic_classes, D, generator=g)
            Pg = torch.randn(args.synthetic_classes, args.K, D, generator=g)
    else:
        if distributed and rank != 0:
            import torch.distributed as dist
            dist.barrier()                                  # rank 0 fills the latent cache first, the others then read it
        ds = ExpansionDataset.from_dataset(args, cfg, eng)
        if distributed and rank == 0:
            import torch.distributed as dist
            dist.barrier()
        if args.guidance_type:
            from .prototypes import extract_prototypes_with_encoder
            assert args.encoder_weight_path and os.path.exists(args.encoder_weight_path)       # :1108
            Pc, Pg = extract_prototypes_with_encoder(args, eng, ds, rank=rank, world=world)
    if args.guidance_type:
        Pc = Pc / Pc.norm(dim=-1, keepdim=True)                            # re-normalisation, :1115-1116, 1121-1122
        Pg = Pg / Pg.norm(dim=-1, keepdim=True)
        if rank == 0:
            print("optimize strategy: %s, target: %s, learning rate: %s" % (args.guidance_type, args.optimize_targets, args.rho))
        eng.set_prototypes(Pc, Pg)
    t0 = time.time()
    n = run_expansion(args, eng, sched, ds)
    torch.cuda.synchronize()
    dt = time.time() - t0
    log.info("rank %d/%d wrote %d images under %s in %.1f s", rank, world, n, args.output_dir, dt)
    if distributed:
        from .launcher import reduce_run_stats
        import torch.distributed as dist
        total, tmax = reduce_run_stats(n, dt, device=dev)
        if rank == 0:
            log.info("node total: %d images in %.1f s = %.2f images/s on %d GPUs", total, tmax, total / max(tmax, 1e-9), world)
        eng.close()
        dist.destroy_process_group()
    else:
        eng.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
