"""Generates tests/golden/fullsize_fixture.pt -- the fp32 CPU oracle at BASELINE.json's FULL sizes (SD-1.5 UNet / AutoencoderKL /
ResNet-50 widths, 512x512, B = 1), one guided step differentiated end to end with torch.autograd:

  configs[1]  transform_guidance, one chained step (P = 1) at t = 381, C = 100 classes, K = 3 group prototypes: x0, z_next, the
              decoded image, guide features, score, dE/dz0, (ge, gb), the updated latents           (generate_data.py:687-732)
  configs[3]  direct_guidance from the same latents at the same step with the StanfordCars sizes (C = 196, K = 3): score, dE/dz,
              z_next                                                                                 (generate_data.py:735-767)
  module VJPs with random cotangents through the SAME forward graph: UNet (J^T g_eps2 -> g_z), decoder (J^T g_image -> g_x0)

Run in the build container (about 10 min on 8 cores, ~40 GB of autograd stash); the test regenerates the seeded inputs and weights
(synthetic_weights(cfg, seed=0)) and only the oracle's OUTPUTS travel, as a small fixture.  The decoded image is stored in fp32: the
same-image gradient test feeds it to the engine so that both sides evaluate the guide's ReLU masks at the same point.

    python tests/golden/make_fullsize_fixture.py
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fullsize_fixture.pt")

STEP_INDEX = 30          # timesteps[30] = 381 = guide_timesteps[0] of the script of record (guidance_step 20)


def inputs(cfg):
    """Seeded inputs shared by this script and tests/test_fullsize_gpu.py."""
    g = torch.Generator().manual_seed(1234)
    L, D = cfg.latent_size, cfg.guide.feature_dim
    d = {
        "z": torch.randn(1, 4, L, L, generator=g),
        "e": torch.rand(1, 4, 1, 1, generator=g),
        "b": torch.randn(1, 4, 1, 1, generator=g) * 0.3,
        "neg": torch.randn(1, cfg.text_len, cfg.unet.cross_attention_dim, generator=g),
        "pos": torch.randn(1, cfg.text_len, cfg.unet.cross_attention_dim, generator=g),
        "Pc100": F.normalize(torch.randn(100, D, generator=g), dim=-1),
        "Pg100": F.normalize(torch.randn(100, 3, D, generator=g), dim=-1),
        "Pc196": F.normalize(torch.randn(196, D, generator=g), dim=-1),
        "Pg196": F.normalize(torch.randn(196, 3, D, generator=g), dim=-1),
        "t100": torch.tensor([7]), "t196": torch.tensor([150]),
        "gg": torch.randn(2, 4, L, L, generator=g),
        "gimg": torch.randn(1, 3, 8 * L, 8 * L, generator=g),
    }
    return d


def main():
    from distdiff_amd.config import sd15_config
    from distdiff_amd.weights import synthetic_weights
    from oracle import sd_oracle as O
    torch.set_num_threads(os.cpu_count() or 8)
    cfg = sd15_config(latent_size=64, max_batch=1)
    w = synthetic_weights(cfg, seed=0, num_classes=100)
    unet, vae, guide, sched = O.build_models(cfg, w)
    ts = sched.set_timesteps(50)
    t = int(ts[STEP_INDEX])
    d = inputs(cfg)
    emb = torch.cat([d["neg"], d["pos"]])
    args = O.SamplerArgs(guidance_type="transform_guidance", num_inference_steps=50, guidance_step=20, guidance_period=1, strength=0.5,
                         rho=10.0, constraint_value=0.2)
    t0 = time.time()
    e = d["e"].clone().requires_grad_(True)
    b = d["b"].clone().requires_grad_(True)
    z0 = d["z"] * (1 + e) + b                                            # :696
    z0.retain_grad()
    # denoise_one_step (:109-121) with the UNet output kept for the module VJP
    x2 = torch.cat([z0] * 2)
    eps2 = unet(x2, t, emb)[0]
    u, c = eps2.chunk(2)
    eps = u + args.guidance_scale * (c - u)
    out = sched.step(eps, t, z0, return_dict=True)
    z_next, x0 = out["prev_sample"], out["pred_original_sample"]
    img = vae.decode(x0 / cfg.vae.scaling_factor)[0]                      # :701
    gi = F.interpolate(img, size=(cfg.guide.input_size,) * 2, mode="bicubic")   # :704
    feats = guide.encode_image(gi).float()                                # :705
    print("forward %.0f s" % (time.time() - t0), flush=True)
    fx = {"step_index": STEP_INDEX, "t": t, "x0": x0.detach(), "z_next": z_next.detach(), "eps2": eps2.detach(), "image": img.detach(),
          "feats": feats.detach(), "weights_checksum": float(sum(v.double().sum() for v in w["unet"].values()))}
    # configs[1]: transform guidance, P = 1 (:707-731)
    score = O.energy(args, feats, d["t100"], d["Pc100"], d["Pg100"]) / args.guidance_period
    ge, gb, gz0 = torch.autograd.grad(score, [e, b, z0], retain_graph=True)
    e2, b2 = e.detach() - args.rho * ge, b.detach() - args.rho * gb
    new = d["z"] * (1 + e2) + b2
    lo, hi = d["z"] - args.constraint_value, d["z"] + args.constraint_value
    new = torch.where(new < lo, lo, new)
    new = torch.where(new > hi, hi, new)
    fx.update({"transform_score": score.detach(), "transform_ge": ge, "transform_gb": gb, "transform_gz0": gz0, "transform_z": new})
    print("transform backward %.0f s, score %.5f" % (time.time() - t0, float(score)), flush=True)
    # configs[3]: direct guidance from z0 (:747-762), StanfordCars sizes
    fh = feats / feats.norm(dim=-1, keepdim=True)
    sd = O.energy(args, fh, d["t196"], d["Pc196"], d["Pg196"])
    (gz,) = torch.autograd.grad(sd, z0, retain_graph=True)
    fx.update({"direct_score": sd.detach(), "direct_gz": gz, "direct_z_next": (z_next - args.rho * gz).detach()})
    print("direct backward %.0f s, score %.5f" % (time.time() - t0, float(sd)), flush=True)
    # module VJPs through the same graph
    (gzu,) = torch.autograd.grad(eps2, z0, d["gg"], retain_graph=True)
    (gx0,) = torch.autograd.grad(img, x0, d["gimg"], retain_graph=False)
    fx.update({"unet_vjp": gzu, "decode_vjp": gx0})
    print("module VJPs %.0f s" % (time.time() - t0), flush=True)
    fx = {k: (v.detach().clone() if isinstance(v, torch.Tensor) else v) for k, v in fx.items()}
    torch.save(fx, OUT)
    print("wrote", OUT, "%.1f MB" % (os.path.getsize(OUT) / 1e6))


if __name__ == "__main__":
    main()
