"""Generates tests/golden/sdxl_loop_fixture.pt -- the fp32 CPU oracle's GUIDED loop at BASELINE.json configs[4]'s full size: the SDXL-base
UNet (text_time conditioning) at 128x128 latents, the AutoencoderKL decoder at 1024x1024, ResNet-50 guide, transform guidance with two
chained guided steps (generate_data.py:687-732) inside a short expansion loop (generate_data.py:1161-1228): a 10-step DDIM schedule
(timesteps 901 ... 1), strength 0.5 = 5 executed steps (401, 301, 201, 101, 1), guidance_step 4 / guidance_period 2 -> guided at
t = 301 (chain 301, 201) + the re-step of 301, C = 100, K = 3, rho 10, constraint 0.2, final decode + denormalise + uint8.
(The reference cannot run this model at all; the oracle restates the published diffusers modules.)

Neither the two-step chain nor even ONE guided step's autograd graph (SDXL UNet + 1024x1024 decoder) fits this container's 64 GB, so
the gradient is assembled by the chain rule from graphs of one module at a time -- mathematically torch.autograd.grad of the chained
score, as in make_fullsize_loop_fixture.py:

    no grad      z0 -> (z1, x0_1) -> (z2, x0_2)
    decoder      g_x0_k = d E_k / d x0_k          (decode 1024x1024 -> bicubic 224 -> ResNet-50 -> energy), k = 1, 2
    UNet step 2  g_z1 = d <g_x0_2, x0_2(z1)> / d z1
    UNet step 1  (ge, gb) = d [ (<g_x0_1, x0_1> + <g_z1, z1>) / P ] / d (e, b)

Stored: latents after add_noise and after every executed step, the latents after the transform update, (ge, gb), the score, the final
latents and the final image as the uint8 bytes of save_image's quantisation.  About 7 min on 8 cores, ~45 GB peak.

    python tests/golden/make_sdxl_loop_fixture.py
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
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sdxl_loop_fixture.pt")
N_STEPS, STRENGTH, GUIDE_STEP, P = 10, 0.5, 4, 2


def loop_inputs(cfg):
    g = torch.Generator().manual_seed(8801)
    L = cfg.latent_size
    return {
        "latents": torch.randn(1, 4, L, L, generator=g) * (0.13025 * 7),
        "noise": torch.randn(1, 4, L, L, generator=g),
        "e": torch.rand(1, 4, 1, 1, generator=g),
        "b": torch.randn(1, 4, 1, 1, generator=g),
        "neg": torch.randn(1, cfg.text_len, cfg.unet.cross_attention_dim, generator=g),
        "pos": torch.randn(1, cfg.text_len, cfg.unet.cross_attention_dim, generator=g),
        "te": torch.randn(2, cfg.unet.add_text_dim, generator=g),
        "ti": torch.tensor([[1024.0, 1024.0, 0.0, 0.0, 1024.0, 1024.0], [1024.0, 1024.0, 0.0, 0.0, 1024.0, 1024.0]]),
        "target": torch.randint(0, 100, (1,), generator=g),
    }


def main():
    from make_fullsize_fixture import inputs as proto_inputs
    from make_fullsize_loop_fixture import to_u8
    from distdiff_amd.config import sd15_config, sdxl_config
    from distdiff_amd.weights import synthetic_weights
    from oracle import sd_oracle as O
    torch.set_num_threads(os.cpu_count() or 8)
    T0 = time.time()
    cfg = sdxl_config(latent_size=128, max_batch=1)
    w = synthetic_weights(cfg, seed=0, num_classes=100)
    unet, vae, guide, sched = O.build_models(cfg, w)
    ts = sched.set_timesteps(N_STEPS)
    d = loop_inputs(cfg)
    proto = proto_inputs(sd15_config(latent_size=64, max_batch=1))          # Pc100 / Pg100 of the SD-1.5 fixtures (D = 2048)
    Pc, Pg = proto["Pc100"], proto["Pg100"]
    unet.added_cond = {"text_embeds": d["te"], "time_ids": d["ti"]}
    emb = torch.cat([d["neg"], d["pos"]])
    args = O.SamplerArgs(guidance_type="transform_guidance", num_inference_steps=N_STEPS, guidance_step=GUIDE_STEP, guidance_period=P,
                         strength=STRENGTH, rho=10.0, constraint_value=0.2)
    si = O.start_index(STRENGTH, len(ts))
    gts = O.guide_timesteps(ts, GUIDE_STEP, P)
    gsz = cfg.guide.input_size
    say = lambda m: print("[%5.0f s] %s" % (time.time() - T0, m), flush=True)
    say("models built; executed timesteps %s, guided %s" % ([int(t) for t in ts[si:]], gts))

    def plain(z, t):
        with torch.no_grad():
            return O.denoise_one_step(args, z, sched, t, unet, emb)

    def energy_grad(x0):                       # decoder + guide graph only
        xr = x0.detach().clone().requires_grad_(True)
        img = vae.decode(xr / cfg.vae.scaling_factor)[0]
        gi = F.interpolate(img, size=(gsz, gsz), mode="bicubic")
        E = O.energy(args, guide.encode_image(gi).float(), d["target"], Pc, Pg)
        (g,) = torch.autograd.grad(E, xr)
        E = E.detach()
        del img, gi, xr
        gc.collect()
        return E, g

    with torch.no_grad():
        z = sched.add_noise(d["latents"], d["noise"], ts[si])
    traj, fx = [z.clone()], {}
    for t in ts[si:]:
        t = int(t)
        if t == gts[0]:
            t0, t1 = gts
            e0, b0 = d["e"], d["b"]
            z0 = z * (1 + e0) + b0
            z1, x0_1 = plain(z0, t0)
            z2, x0_2 = plain(z1, t1)
            say("chain forward done")
            E1, g_x0_1 = energy_grad(x0_1)
            E2, g_x0_2 = energy_grad(x0_2)
            say("decoder / guide gradients done: E1 %.5f E2 %.5f" % (float(E1), float(E2)))
            z1r = z1.detach().clone().requires_grad_(True)
            _, x0_2g = O.denoise_one_step(args, z1r, sched, t1, unet, emb)
            (g_z1,) = torch.autograd.grad((g_x0_2 * x0_2g).sum(), z1r)
            del x0_2g, z1r
            gc.collect()
            say("UNet step 2 VJP done")
            e = e0.clone().requires_grad_(True)
            b = b0.clone().requires_grad_(True)
            z0g = z * (1 + e) + b
            z1g, x0_1g = O.denoise_one_step(args, z0g, sched, t0, unet, emb)
            ge, gb = torch.autograd.grad(((g_x0_1 * x0_1g).sum() + (g_z1 * z1g).sum()) / P, [e, b])
            del z1g, x0_1g, z0g
            gc.collect()
            say("UNet step 1 VJP done")
            score = (E1 + E2) / P
            e2, b2 = e0 - args.rho * ge, b0 - args.rho * gb
            new = z * (1 + e2) + b2
            lo, hi = z - args.constraint_value, z + args.constraint_value
            new = torch.where(new < lo, lo, new)
            new = torch.where(new > hi, hi, new)
            z = new.detach()
            fx.update({"z_guided": z.clone(), "ge": ge.clone(), "gb": gb.clone(), "score": score.clone(), "E": torch.stack([E1, E2]),
                       "chain_x0": torch.cat([x0_1, x0_2]).clone()})
            z, _ = plain(z, t)
        else:
            z, _ = plain(z, t)
        traj.append(z.clone())
        say("t=%d done" % t)
    with torch.no_grad():
        img = vae.decode(z / cfg.vae.scaling_factor)[0]
        img = (img / 2 + 0.5).clamp(0, 1)
    fx.update({"n_steps": N_STEPS, "strength": STRENGTH, "guidance_step": GUIDE_STEP, "guidance_period": P, "start_index": si,
               "guide_timesteps": gts, "traj": torch.stack([x[0] for x in traj]).half(), "z_final": z.clone(), "image_u8": to_u8(img),
               "weights_checksum": float(sum(v.double().sum() for v in w["unet"].values()))})
    torch.save(fx, OUT)
    say("wrote %s %.1f MB" % (OUT, os.path.getsize(OUT) / 1e6))


if __name__ == "__main__":
    main()
