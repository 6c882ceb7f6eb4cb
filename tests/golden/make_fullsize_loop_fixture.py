"""Generates tests/golden/fullsize_loop_fixture.pt -- the fp32 CPU oracle's WHOLE expansion loop (generate_data.py:1161-1234: add_noise ->
every executed DDIM step incl. the guided ones -> final decode -> denormalise -> uint8) at BASELINE.json's FULL sizes (SD-1.5 UNet /
AutoencoderKL / ResNet-50 widths, 512x512, train_batch_size 1), for two independent input rows ("a", "b") each:

  configs[1]  the script of record (expand_diff.sh:3-15): strength 0.5 (25 executed steps of a 50-step schedule), transform guidance at
              t = 381 with guidance_period 2 (two chained guided steps differentiated end to end, then the re-step of t = 381,
              generate_data.py:1203-1207), rho 10, constraint 0.2, C = 100, K = 3, e ~ U[0,1), b ~ N(0,1) (:692-695)
  configs[3]  StanfordCars sizes (C = 196, K = 3), direct guidance on each of the last 10 steps (guidance_step = guidance_period = 10,
              :1210-1216), shortened to strength 0.3 = 15 executed steps (5 plain + 10 guided)

Unlike the per-step fixtures nothing is forced to a common point here: the guide's masks are drawn at the oracle's OWN images, so the
comparison with the engine's loop measures what a user of the PNGs sees (error accumulated over all steps + the clamp decisions).

Stored per row: the latents after add_noise and after every executed step ("traj"), the latents right after the transform update,
the guidance scores, the final latents (traj in fp16: a diagnostic; the end points in fp32), the final image (fp16 and the uint8 bytes of torchvision's save_image quantisation, :1232).
Inputs and weights are regenerated from seeds by the test (`loop_inputs`, synthetic_weights(cfg, seed=0)).

The chained P = 2 autograd graph does not fit this container's 64 GB; the gradient is assembled by the chain rule in three stages
exactly as in make_fullsize_p2_fixture.py (checked there against the one-graph gradient at the tiny config).

About 15 min on 8 cores (plain step 3-5 s, guided step 30-70 s), ~40 GB peak:   python tests/golden/make_fullsize_loop_fixture.py [c1|c3|all]
"""
import gc
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fullsize_loop_fixture.pt")

SEEDS = {"a": 7001, "b": 7002}
C3_STRENGTH = 0.3


def loop_inputs(cfg, tag):
    """Seeded inputs of one row, shared with tests/test_fullsize_loop_gpu.py.  VAE latents of natural images have a standard
    deviation near 0.18215 * 5; e / b follow the reference's draws (generate_data.py:692-695)."""
    g = torch.Generator().manual_seed(SEEDS[tag])
    L = cfg.latent_size
    return {
        "latents": torch.randn(1, 4, L, L, generator=g) * (0.18215 * 5),
        "noise": torch.randn(1, 4, L, L, generator=g),
        "e": torch.rand(1, 4, 1, 1, generator=g),
        "b": torch.randn(1, 4, 1, 1, generator=g),
        "neg": torch.randn(1, cfg.text_len, cfg.unet.cross_attention_dim, generator=g),
        "pos": torch.randn(1, cfg.text_len, cfg.unet.cross_attention_dim, generator=g),
        "t100": torch.randint(0, 100, (1,), generator=g),
        "t196": torch.randint(0, 196, (1,), generator=g),
    }


def to_u8(img01):
    """torchvision.utils.save_image's quantisation (generate_data.py:1232): mul(255).add_(0.5).clamp_(0, 255).to(uint8), HWC."""
    return img01.detach().clone().mul(255).add_(0.5).clamp_(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()


def run_loop(O, args, cfg, models, d, targets, Pc, Pg, tag):
    """oracle.expand_one (generate_data.py:1161-1228) with the trajectory recorded."""
    unet, vae, guide, sched = models
    T0 = time.time()
    ts = sched.set_timesteps(args.num_inference_steps)
    si = O.start_index(args.strength, len(ts))
    with torch.no_grad():
        z = sched.add_noise(d["latents"], d["noise"], ts[si])             # :1176
    gts = O.guide_timesteps(ts, args.guidance_step, args.guidance_period)
    emb = torch.cat([d["neg"], d["pos"]])
    traj, scores, fx = [z.clone()], [], {}
    gsz = cfg.guide.input_size
    for t in ts[si:]:
        t = int(t)
        if t == gts[0] and args.guidance_type == "transform_guidance":
            z, score, (ge, gb) = O.transform_guidance_3stage(args, cfg, models, z, targets, gts, emb, d["e"], d["b"], Pc, Pg)
            fx.update({"z_guided": z.clone(), "ge": ge.clone(), "gb": gb.clone()})
            scores.append(float(score))
            with torch.no_grad():
                z, _ = O.denoise_one_step(args, z, sched, t, unet, emb)   # :1207
        elif t in gts and args.guidance_type == "direct_guidance":
            z, _, score, _ = O.direct_guidance(args, z, targets, t, sched, unet, emb, vae, guide, Pc, Pg, gsz)
            scores.append(float(score))
            gc.collect()
        else:
            with torch.no_grad():
                z, _ = O.denoise_one_step(args, z, sched, t, unet, emb)
        traj.append(z.detach().clone())
        print("[%s] t=%d done, %.0f s%s" % (tag, t, time.time() - T0, (" score %.5f" % scores[-1]) if scores and t in gts and (t == gts[0] or args.guidance_type == "direct_guidance") else ""),
              flush=True)
    with torch.no_grad():
        img = vae.decode(z / cfg.vae.scaling_factor)[0]                   # :1223
        img = (img / 2 + 0.5).clamp(0, 1)                                 # :1227 do_denormalize
    fx.update({"start_index": si, "guide_timesteps": gts, "traj": torch.stack([x[0] for x in traj]).half(), "scores": torch.tensor(scores),
               "z_final": z.detach().clone(), "image16": img.half(), "image_u8": to_u8(img)})
    return fx


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    from make_fullsize_fixture import inputs
    from distdiff_amd.config import sd15_config
    from distdiff_amd.weights import synthetic_weights
    from oracle import sd_oracle as O
    torch.set_num_threads(os.cpu_count() or 8)
    cfg = sd15_config(latent_size=64, max_batch=1)
    w = synthetic_weights(cfg, seed=0, num_classes=100)
    models = O.build_models(cfg, w)
    proto = inputs(cfg)                       # prototypes (Pc100 / Pg100 / Pc196 / Pg196) of the per-step fixtures
    fx = torch.load(OUT, weights_only=False) if os.path.exists(OUT) else {}
    fx["weights_checksum"] = float(sum(v.double().sum() for v in w["unet"].values()))
    fx["c3_strength"] = C3_STRENGTH
    a1 = O.SamplerArgs(guidance_type="transform_guidance", num_inference_steps=50, guidance_step=20, guidance_period=2, strength=0.5,
                       rho=10.0, constraint_value=0.2)
    a3 = O.SamplerArgs(guidance_type="direct_guidance", num_inference_steps=50, guidance_step=10, guidance_period=10,
                       strength=C3_STRENGTH, rho=10.0, constraint_value=0.2)
    for tag in ("a", "b"):
        d = loop_inputs(cfg, tag)
        if which in ("c1", "all"):
            fx["c1_" + tag] = run_loop(O, a1, cfg, models, d, d["t100"], proto["Pc100"], proto["Pg100"], "c1/" + tag)
            torch.save(fx, OUT)
        if which in ("c3", "all"):
            fx["c3_" + tag] = run_loop(O, a3, cfg, models, d, d["t196"], proto["Pc196"], proto["Pg196"], "c3/" + tag)
            torch.save(fx, OUT)
    print("wrote", OUT, "%.1f MB" % (os.path.getsize(OUT) / 1e6))


if __name__ == "__main__":
    main()
