"""Generates tests/golden/fullsize_p2_fixture.pt -- the fp32 CPU oracle at BASELINE.json's FULL sizes (SD-1.5 UNet / AutoencoderKL /
ResNet-50 widths, 512x512, B = 1) for the script of record's CHAINED transform guidance (guidance_period = 2: expand_diff.sh:6,
generate_data.py:699-719): two guided denoise steps at t = 381, 361, the energy of both decoded images, differentiated end to end.

Two independent input rows -- "a": the inputs of tests/golden/fullsize_fixture.pt, "b": its own latents / e / b / prompt / class -- so
that the batched parity tests (tests/test_fullsize_batch_gpu.py) can fill an engine batch with rows that differ.

The two-step autograd graph does not fit this container's 64 GB, so the chain rule over the two steps is applied by hand on top of
torch.autograd (mathematically the same gradient as `torch.autograd.grad(score, [e, b])` through the whole chain):

    stage 1 (no grad)   z0 -> step(t0) -> (z1, x0_1)
    stage 2 (autograd)  z1 -> step(t1) -> x0_2 -> decode -> bicubic -> guide -> E2 ;  g_z1 = dE2/dz1
    stage 3 (autograd)  (e, b) -> z0 -> step(t0) -> (z1, x0_1) -> ... -> E1 ;  d[(E1 + <g_z1, z1>) / P] / d(e, b, z0)

Stage 3 also yields the P = 1 gradients of the row (dE1/d(e, b, z0)).  About 4 min on 8 cores, ~40 GB peak.

    python tests/golden/make_fullsize_p2_fixture.py
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
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fullsize_p2_fixture.pt")

STEP_INDEX = 30          # timesteps[30], timesteps[31] = 381, 361 = guide_timesteps of the script of record


def inputs2(cfg):
    """Seeded inputs of the second row (prototypes are those of make_fullsize_fixture.inputs)."""
    g = torch.Generator().manual_seed(4321)
    L = cfg.latent_size
    return {
        "z": torch.randn(1, 4, L, L, generator=g),
        "e": torch.rand(1, 4, 1, 1, generator=g),
        "b": torch.randn(1, 4, 1, 1, generator=g) * 0.3,
        "neg": torch.randn(1, cfg.text_len, cfg.unet.cross_attention_dim, generator=g),
        "pos": torch.randn(1, cfg.text_len, cfg.unet.cross_attention_dim, generator=g),
        "t100": torch.tensor([61]),
    }


def clamp_update(args, z, e, b, ge, gb):
    e2, b2 = e - args.rho * ge, b - args.rho * gb                        # :723-724
    new = z * (1 + e2) + b2
    lo, hi = z - args.constraint_value, z + args.constraint_value
    new = torch.where(new < lo, lo, new)                                  # tensor_clamp: lower bound first (:129-132)
    return torch.where(new > hi, hi, new)


def run_row(tag, d, models, cfg, Pc, Pg, args, t0, t1, P):
    """One input row through the two chained guided steps; returns its slice of the fixture."""
    from oracle import sd_oracle as O
    unet, vae, guide, sched = models
    emb = torch.cat([d["neg"], d["pos"]])
    gsz = cfg.guide.input_size
    T0 = time.time()

    def feats_of(x0):
        img = vae.decode(x0 / cfg.vae.scaling_factor)[0]                  # :701
        # The guide is evaluated AT the fp16 rounding of the decoded image (gradients flow through the decoder unchanged; the oracle's
        # own test hook, sd_oracle._guide_features image_at): the fixture then carries the image in 2 bytes per value, and the engine is
        # given exactly this image (dd_debug_set_images), so both sides draw the guide's ReLU / max-pool masks at the same point.
        img16 = img.detach().half()
        img = img + (img16.float() - img).detach()
        gi = F.interpolate(img, size=(gsz, gsz), mode="bicubic")          # :704
        return img16, guide.encode_image(gi).float()                      # :705

    # stage 1: the first chained step without a graph
    with torch.no_grad():
        z0 = d["z"] * (1 + d["e"]) + d["b"]                               # :696
        eps2 = unet(torch.cat([z0] * 2), t0, emb)[0]
        z1, x0_1 = O.denoise_one_step(args, z0, sched, t0, unet, emb)
    fx = {"eps2": eps2.clone(), "x0_1": x0_1.clone(), "z1": z1.clone()}
    print("[%s] stage 1 (plain step) %.0f s" % (tag, time.time() - T0), flush=True)

    # stage 2: second chained step, dE2/dz1
    z1r = z1.detach().clone().requires_grad_(True)
    z2, x0_2 = O.denoise_one_step(args, z1r, sched, t1, unet, emb)
    img2, f2 = feats_of(x0_2)
    E2 = O.energy(args, f2, d["t100"], Pc, Pg)
    (g_z1,) = torch.autograd.grad(E2, z1r)
    fx.update({"x0_2": x0_2.detach().clone(), "z2": z2.detach().clone(), "image_2": img2.clone(), "feats_2": f2.detach().clone(),
               "E2": E2.detach().clone(), "g_z1_E2": g_z1.clone()})
    del z2, x0_2, img2, f2, E2, z1r
    gc.collect()
    print("[%s] stage 2 (second step forward + backward) %.0f s" % (tag, time.time() - T0), flush=True)

    # stage 3: first chained step with a graph; chain rule through z1
    e = d["e"].clone().requires_grad_(True)
    b = d["b"].clone().requires_grad_(True)
    z0 = d["z"] * (1 + e) + b
    z1g, x0_1g = O.denoise_one_step(args, z0, sched, t0, unet, emb)
    img1, f1 = feats_of(x0_1g)
    E1 = O.energy(args, f1, d["t100"], Pc, Pg)
    fx.update({"image_1": img1.clone(), "feats_1": f1.detach().clone(), "E1": E1.detach().clone()})
    # P = 1 (guidance_period 1): score = E1
    ge1, gb1, gz01 = torch.autograd.grad(E1, [e, b, z0], retain_graph=True)
    fx.update({"p1_score": E1.detach().clone(), "p1_ge": ge1, "p1_gb": gb1, "p1_gz0": gz01,
               "p1_z": clamp_update(args, d["z"], d["e"], d["b"], ge1, gb1)})
    # P = 2: score = (E1 + E2) / P  (:719);  dE2/d. = <g_z1, dz1/d.>
    total = (E1 + (g_z1 * z1g).sum()) / P
    ge, gb, gz0 = torch.autograd.grad(total, [e, b, z0])
    score = (fx["E1"] + fx["E2"]) / P
    fx.update({"p2_score": score, "p2_ge": ge, "p2_gb": gb, "p2_gz0": gz0, "p2_z": clamp_update(args, d["z"], d["e"], d["b"], ge, gb)})
    print("[%s] stage 3 (first step forward, P=1 and P=2 backward) %.0f s; E1 %.5f E2 %.5f" % (tag, time.time() - T0, float(fx["E1"]), float(fx["E2"])),
          flush=True)
    del z1g, x0_1g, img1, f1, E1, total
    gc.collect()
    return {k: (v.detach().clone() if isinstance(v, torch.Tensor) else v) for k, v in fx.items()}


def main():
    from make_fullsize_fixture import inputs
    from distdiff_amd.config import sd15_config
    from distdiff_amd.weights import synthetic_weights
    from oracle import sd_oracle as O
    torch.set_num_threads(os.cpu_count() or 8)
    cfg = sd15_config(latent_size=64, max_batch=1)
    w = synthetic_weights(cfg, seed=0, num_classes=100)
    models = O.build_models(cfg, w)
    ts = models[3].set_timesteps(50)
    t0, t1 = int(ts[STEP_INDEX]), int(ts[STEP_INDEX + 1])
    proto = inputs(cfg)
    P = 2
    args = O.SamplerArgs(guidance_type="transform_guidance", num_inference_steps=50, guidance_step=20, guidance_period=P, strength=0.5,
                         rho=10.0, constraint_value=0.2)
    fx = {"step_index": STEP_INDEX, "t": [t0, t1], "weights_checksum": float(sum(v.double().sum() for v in w["unet"].values()))}
    # row "a": the inputs of fullsize_fixture.pt; row "b": the second seeded row
    fx["a"] = run_row("a", proto, models, cfg, proto["Pc100"], proto["Pg100"], args, t0, t1, P)
    fx["b"] = run_row("b", inputs2(cfg), models, cfg, proto["Pc100"], proto["Pg100"], args, t0, t1, P)
    torch.save(fx, OUT)
    print("wrote", OUT, "%.1f MB" % (os.path.getsize(OUT) / 1e6))


if __name__ == "__main__":
    main()
