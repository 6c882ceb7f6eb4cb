"""Engine vs CPU oracle on the tiny config: prints relative errors per stage (debug helper)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import torch.nn.functional as F
from distdiff_amd.config import tiny_config
from distdiff_amd.weights import synthetic_weights
from distdiff_amd.engine import Engine
from oracle import sd_oracle as O


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item(), (a - b).abs().max().item(), b.abs().max().item()


def main():
    B = 2
    cfg = tiny_config(max_batch=B)
    w = synthetic_weights(cfg, seed=0, num_classes=5)
    models = O.build_models(cfg, w)
    unet, vae, guide, sched = models
    L = cfg.latent_size
    g = torch.Generator().manual_seed(1)
    lat = torch.randn(B, 4, L, L, generator=g) * 0.9
    noise = torch.randn(B, 4, L, L, generator=g)
    e = torch.rand(B, 4, 1, 1, generator=g); b = torch.randn(B, 4, 1, 1, generator=g)
    pe = torch.randn(B, cfg.text_len, cfg.unet.cross_attention_dim, generator=g)
    ne = torch.randn(1, cfg.text_len, cfg.unet.cross_attention_dim, generator=g).expand(B, -1, -1)
    D = cfg.guide.feature_dim
    Pc = F.normalize(torch.randn(5, D, generator=g), dim=-1); Pg = F.normalize(torch.randn(5, 3, D, generator=g), dim=-1)
    tg = torch.tensor([1, 3])
    n_steps = 10
    args = O.SamplerArgs(guidance_type="transform_guidance", num_inference_steps=n_steps, guidance_step=4, guidance_period=2, strength=0.5)
    ts = sched.set_timesteps(n_steps)
    t0 = time.time()
    eng = Engine(cfg, w, enable_grad=True, max_guidance_period=2)
    eng.set_schedule(ts.tolist(), sched.alphas_cumprod.numpy(), float(sched.final_alpha_cumprod), guidance_scale=args.guidance_scale,
                     gs=args.gs, ls=args.ls, rho=args.rho, constraint_value=args.constraint_value, guidance_period=args.guidance_period)
    eng.set_prototypes(Pc, Pg)
    embeds = torch.cat([ne, pe])
    eng.set_prompt(embeds.cuda())
    print("engine built in %.1fs, workspace %.1f MB" % (time.time() - t0, eng.workspace_bytes() / 1e6))
    z = sched.add_noise(lat, noise, ts[5])
    print("add_noise", rel(eng.add_noise(lat, noise, 5), z))
    with torch.no_grad():
        eps_ref = unet(torch.cat([z, z]), int(ts[5]), embeds)[0]
    print("unet eps2", rel(eng.unet_forward(z, 5), eps_ref))
    with torch.no_grad():
        zp_ref, x0_ref = O.denoise_one_step(args, z, sched, int(ts[5]), unet, embeds)
    zp, x0 = eng.denoise_step(z, 5)
    print("denoise z_prev", rel(zp, zp_ref), "x0", rel(x0, x0_ref))
    with torch.no_grad():
        img_ref = vae.decode(x0_ref / cfg.vae.scaling_factor)[0]
    print("decode raw", rel(eng.decode(x0_ref, denormalize=False), img_ref))
    print("decode denorm", rel(eng.decode(x0_ref, denormalize=True), (img_ref / 2 + 0.5).clamp(0, 1)))
    gi = F.interpolate(img_ref, size=(cfg.guide.input_size,) * 2, mode="bicubic")
    with torch.no_grad():
        f_ref = guide.encode_image(gi)
    print("guide feats", rel(eng.guide_encode(gi), f_ref))
    # per-module VJPs with random cotangents
    gg = torch.randn(2 * B, 4, L, L, generator=g)
    zr = z.clone().requires_grad_(True)
    (gz_u,) = torch.autograd.grad(unet(torch.cat([zr, zr]), int(ts[5]), embeds)[0], zr, gg)
    print("unet vjp", rel(eng.unet_vjp(z, 5, gg), gz_u))
    gim = torch.randn(B, 3, 8 * L, 8 * L, generator=g)
    xr = x0_ref.clone().requires_grad_(True)
    (gz_v,) = torch.autograd.grad(vae.decode(xr / cfg.vae.scaling_factor)[0], xr, gim)
    print("decode vjp", rel(eng.decode_vjp(x0_ref, gim), gz_v))
    gf = torch.randn(B, D, generator=g)
    gir = gi.clone().requires_grad_(True)
    (gi_g,) = torch.autograd.grad(guide.encode_image(gir), gir, gf)
    print("guide vjp", rel(eng.guide_vjp(gi, gf), gi_g))
    fr = guide.encode_image(gi).detach()
    gfr = fr / fr.norm(dim=-1, keepdim=True)   # radial cotangent
    (gi_r,) = torch.autograd.grad(guide.encode_image(gir), gir, gfr)
    print("guide vjp radial", rel(eng.guide_vjp(gi, gfr), gi_r))
    gft = gf - gfr * (gfr * gf).sum(-1, keepdim=True)   # tangential cotangent
    (gi_t,) = torch.autograd.grad(guide.encode_image(gir), gir, gft)
    print("guide vjp tangential", rel(eng.guide_vjp(gi, gft), gi_t))
    gts = O.guide_timesteps(ts, args.guidance_step, args.guidance_period)
    first = [int(t) for t in ts].index(gts[0])
    znew_ref, score_ref, (ge_ref, gb_ref) = O.transform_guidance(args, z, tg, gts, sched, unet, embeds, vae, guide, e, b, Pc, Pg, cfg.guide.input_size)
    znew, score, gz0 = eng.transform_guidance(z, tg, e, b, first, 2)
    ge = (gz0.cpu() * z).sum((2, 3), keepdim=True); gb = gz0.cpu().sum((2, 3), keepdim=True)
    print("transform score", score.item(), score_ref.item())
    print("transform ge", rel(ge, ge_ref), "gb", rel(gb, gb_ref))
    print("transform z_new", rel(znew, znew_ref))
    zn_ref, x0d_ref, sc_ref, gz_ref = O.direct_guidance(args, z, tg, gts[0], sched, unet, embeds, vae, guide, Pc, Pg, cfg.guide.input_size)
    zn, x0d, sc, gz = eng.direct_guidance(z, tg, first)
    print("direct score", sc.item(), sc_ref.item())
    print("direct g_z", rel(gz, gz_ref), "z_next", rel(zn, zn_ref))
    for gt in ["transform_guidance", "direct_guidance", None]:
        args.guidance_type = gt
        zf_ref, img_ref2, s_ref = O.expand_one(args, cfg, models, lat, noise, e, b, pe, ne, tg, Pc, Pg)
        si = O.start_index(args.strength, n_steps)
        zf, img, s = eng.expand(lat, noise, e, b, tg, si, gt, first, 2)
        print("expand", gt, "z", rel(zf, zf_ref), "img", rel(img, img_ref2), "score", s.item(), s_ref)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
