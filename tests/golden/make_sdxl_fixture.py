"""Generates tests/golden/sdxl_fixture.pt -- the fp32 CPU oracle at BASELINE.json configs[4]'s FULL size: the SDXL-base UNet
(block widths 320 / 640 / 1280, transformer depths 1 / 2 / 10, heads 5 / 10 / 20, cross_attention_dim 2048, text_time conditioning)
at 128x128 latents (1024x1024 images), B = 1 (CFG batch 2), one denoise step at t = 381:

  eps2 (both CFG halves), z_next, x0 of `denoise_one_step` (generate_data.py:109-121 with added_cond_kwargs)
  the UNet VJP with a random cotangent through the same graph (torch.autograd): g_z = J^T g_eps2
  the AutoencoderKL decode of x0 at 1024x1024 (forward only: its autograd stash would be ~4x the 512x512 one)

The reference cannot run this model at all (single text encoder, no added_cond_kwargs: SURVEY.md 8d C5); the oracle restates the
published diffusers UNet2DConditionModel (oracle/sd_oracle.py).  Weights and inputs are regenerated from seeds by the test; only the
oracle's OUTPUTS travel.  About 10 min on 8 cores, ~45 GB peak.

    python tests/golden/make_sdxl_fixture.py
"""
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sdxl_fixture.pt")
STEP_INDEX = 30


def inputs(cfg):
    g = torch.Generator().manual_seed(777)
    L = cfg.latent_size
    return {
        "z": torch.randn(1, 4, L, L, generator=g),
        "neg": torch.randn(1, cfg.text_len, cfg.unet.cross_attention_dim, generator=g),
        "pos": torch.randn(1, cfg.text_len, cfg.unet.cross_attention_dim, generator=g),
        "te": torch.randn(2, cfg.unet.add_text_dim, generator=g),
        "ti": torch.tensor([[1024.0, 1024.0, 0.0, 0.0, 1024.0, 1024.0], [1024.0, 1024.0, 0.0, 0.0, 1024.0, 1024.0]]),
        "gg": torch.randn(2, 4, L, L, generator=g),
    }


def main():
    from distdiff_amd.config import sdxl_config
    from distdiff_amd.weights import synthetic_weights
    from oracle import sd_oracle as O
    torch.set_num_threads(os.cpu_count() or 8)
    t0 = time.time()
    cfg = sdxl_config(latent_size=128, max_batch=1)
    w = synthetic_weights(cfg, seed=0, num_classes=100)
    print("weights %.0f s" % (time.time() - t0), flush=True)
    unet, vae, guide, sched = O.build_models(cfg, w)
    ts = sched.set_timesteps(50)
    t = int(ts[STEP_INDEX])
    d = inputs(cfg)
    unet.added_cond = {"text_embeds": d["te"], "time_ids": d["ti"]}
    emb = torch.cat([d["neg"], d["pos"]])
    args = O.SamplerArgs()
    z = d["z"].clone().requires_grad_(True)
    eps2 = unet(torch.cat([z] * 2), t, emb)[0]
    u, c = eps2.chunk(2)
    eps = u + args.guidance_scale * (c - u)
    out = sched.step(eps, t, z, return_dict=True)
    print("forward %.0f s" % (time.time() - t0), flush=True)
    (gz,) = torch.autograd.grad(eps2, z, d["gg"])
    print("vjp %.0f s" % (time.time() - t0), flush=True)
    fx = {"step_index": STEP_INDEX, "t": t, "eps2": eps2.detach().clone(), "z_next": out["prev_sample"].detach().clone(),
          "x0": out["pred_original_sample"].detach().clone(), "unet_vjp": gz.clone(),
          "weights_checksum": float(sum(v.double().sum() for v in w["unet"].values()))}
    del eps2, eps, out, u, c, gz
    with torch.no_grad():
        img = vae.decode(fx["x0"] / cfg.vae.scaling_factor)[0]
    fx["image_f16"] = img.half()
    print("decode %.0f s" % (time.time() - t0), flush=True)
    torch.save(fx, OUT)
    print("wrote", OUT, "%.1f MB" % (os.path.getsize(OUT) / 1e6))


if __name__ == "__main__":
    main()
