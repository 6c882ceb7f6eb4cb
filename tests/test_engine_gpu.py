"""Parity of the HIP engine (through the production C ABI, include/distdiff_hip.h) against the CPU oracle and the
committed golden fixtures (produced by the reference's own sampler functions, tests/golden/make_fixtures.py).

Tolerance statement (UNet / VAE: bf16 storage + bf16 MFMA inputs, fp32 accumulation; guide network, bicubic resize, energy: exact
fp32; all vs the fp32 oracle), relative L2 error:
  forward tensors (eps, z_prev, x0, decoded image)                  <= 3 %   (measured 1.3-2.3 %)
  guide features / guide VJP (fp32 v_mfma_f32_32x32x2_f32)          <= 1e-4 / 2e-3   (measured 2e-7 / 4e-7)
  VJP of UNet / VAE decoder with random cotangents                  <= 5 % / 3 %   (measured 2.1 % / 0.9 %)
  guidance scores                                                   <= 0.5 %
  energy gradient, masks at the SAME image (test_energy_gradient_at_the_same_image): g_z of direct guidance <= 5 %, (e, b) gradients of
      transform guidance (two chained steps)                        <= 8 %   (measured 6.6 % / 2.4 %; full size: tests/test_fullsize_gpu.py)
  energy gradient against the oracle's own forward point: NOT a parity statement at the 5 % level for any bf16 UNet -- the guide's
      input-gradient is piecewise constant in the image (ReLU / max-pool masks), and in the fp32 oracle itself a 1 % perturbation
      of the image moves it by 14-20 % (tests/test_oracle.py::test_guide_gradient_conditioning).  With the engine's x0 within
      2.3 % of the oracle's: (e, b) gradients within max(20 %, 1.5 x the oracle's own response to a 2 % perturbation of its input
      latents, measured in the test: 18-42 %); measured 5-25 % over four builds; per-pixel g_z of direct guidance measured 36 %
  latents after transform guidance                                  <= 7 % vs the reference (measured 3.7-5.4 %), and == the update
      rule applied to the engine's own gradient to 2e-4
  latents after direct guidance / after the whole loop              <= 3 %, decoded image max abs error <= 0.08 (of [0,1])
"""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
FIX = os.path.join(os.path.dirname(__file__), "golden", "tiny_fixture.pt")


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    assert torch.isfinite(a).all()
    return ((a - b).norm() / (b.norm() + 1e-20)).item()


@pytest.fixture(scope="module")
def fx():
    return torch.load(FIX, weights_only=False)


@pytest.fixture(scope="module")
def setup(hip_lib, fx):
    from distdiff_amd.config import tiny_config
    from distdiff_amd.engine import Engine
    from distdiff_amd.scheduler import DDIMSchedule
    from distdiff_amd.weights import synthetic_weights
    from oracle import sd_oracle as O
    cfg = tiny_config(max_batch=2)
    w = synthetic_weights(cfg, seed=0, num_classes=5)
    eng = Engine(cfg, w, enable_grad=True, max_guidance_period=2)
    sched = DDIMSchedule(cfg.scheduler)
    ts = sched.set_timesteps(fx["n_steps"])
    assert ts == fx["timesteps"].tolist()
    a = fx["args"]
    eng.set_schedule(ts, sched.alphas_cumprod, sched.final_alpha_cumprod, guidance_scale=a["guidance_scale"], gs=a["gs"], ls=a["ls"],
                     rho=a["rho"], constraint_value=a["constraint_value"], guidance_period=a["guidance_period"])
    eng.set_prototypes(fx["Pc"], fx["Pg"])
    eng.set_prompt(torch.cat([fx["negative_embeds"], fx["prompt_embeds"]]).cuda())
    models = O.build_models(cfg, w)
    models[3].set_timesteps(fx["n_steps"])
    yield cfg, eng, models, O
    eng.close()


def test_denoise_step_vs_reference_fixture(setup, fx):
    cfg, eng, models, O = setup
    first = fx["timesteps"].tolist().index(fx["guide_timesteps"][0])
    zp, x0 = eng.denoise_step(fx["z"], first)
    assert rel(zp, fx["ref_denoise_z_prev"]) < 0.03
    assert rel(x0, fx["ref_denoise_x0"]) < 0.04


def test_unet_decode_guide_forward(setup, fx):
    cfg, eng, (unet, vae, guide, sched), O = setup
    emb = torch.cat([fx["negative_embeds"], fx["prompt_embeds"]])
    z = fx["z"]
    with torch.no_grad():
        eps = unet(torch.cat([z, z]), fx["timesteps"][5].item(), emb)[0]
        img = vae.decode(fx["ref_denoise_x0"] / cfg.vae.scaling_factor)[0]
        gi = F.interpolate(img, size=(cfg.guide.input_size,) * 2, mode="bicubic")
        f = guide.encode_image(gi)
    assert rel(eng.unet_forward(z, 5), eps) < 0.03
    assert rel(eng.decode(fx["ref_denoise_x0"], denormalize=False), img) < 0.02
    d = eng.decode(fx["ref_denoise_x0"], denormalize=True)
    assert float(d.min()) >= 0.0 and float(d.max()) <= 1.0
    assert (d.cpu() - (img / 2 + 0.5).clamp(0, 1)).abs().max() < 0.06
    assert rel(eng.guide_encode(gi), f) < 1e-4


def test_module_vjps_vs_autograd(setup, fx):
    cfg, eng, (unet, vae, guide, sched), O = setup
    g = torch.Generator().manual_seed(11)
    emb = torch.cat([fx["negative_embeds"], fx["prompt_embeds"]])
    B, L = 2, cfg.latent_size
    z = fx["z"]
    gg = torch.randn(2 * B, 4, L, L, generator=g)
    zr = z.clone().requires_grad_(True)
    (gz,) = torch.autograd.grad(unet(torch.cat([zr, zr]), fx["timesteps"][5].item(), emb)[0], zr, gg)
    assert rel(eng.unet_vjp(z, 5, gg), gz) < 0.05
    x0 = fx["ref_denoise_x0"]
    gim = torch.randn(B, 3, 8 * L, 8 * L, generator=g)
    xr = x0.clone().requires_grad_(True)
    (gv,) = torch.autograd.grad(vae.decode(xr / cfg.vae.scaling_factor)[0], xr, gim)
    assert rel(eng.decode_vjp(x0, gim), gv) < 0.03
    gi = torch.randn(B, 3, cfg.guide.input_size, cfg.guide.input_size, generator=g) * 0.5
    gf = torch.randn(B, cfg.guide.feature_dim, generator=g)
    gir = gi.clone().requires_grad_(True)
    (gr,) = torch.autograd.grad(guide.encode_image(gir), gir, gf)
    assert rel(eng.guide_vjp(gi, gf), gr) < 2e-3       # exact fp32 guide (guide_f32.hip): summation order only
    # linearity of the hand-derived VJP in the cotangent (size-independent property)
    a = eng.unet_vjp(z, 5, gg)
    b = eng.unet_vjp(z, 5, 2.0 * gg)
    assert rel(b, 2.0 * a) < 0.01


def test_transform_guidance_vs_reference_fixture(setup, fx):
    cfg, eng, models, O = setup
    first = fx["timesteps"].tolist().index(fx["guide_timesteps"][0])
    z, score, gz0 = eng.transform_guidance(fx["z"], fx["targets"], fx["e"], fx["b"], first, 2)
    assert abs(score.item() - float(fx["ref_transform_score"])) < 0.005 * abs(float(fx["ref_transform_score"]))
    assert float((z.cpu() - fx["z"]).abs().max()) <= fx["args"]["constraint_value"] + 1e-5   # L-inf ball (generate_data.py:124-137)
    # the update rule itself (:696, :721-728), exactly: e -= rho*ge, b -= rho*gb, re-affine, clamp (lower bound first) -- evaluated
    # with the engine's own gradient, so this part is free of the gradient noise discussed below
    a, z0, g0 = fx["args"], fx["z"], gz0.cpu()
    e2 = fx["e"].reshape(-1, 4, 1, 1) - a["rho"] * (g0 * z0).sum((2, 3), keepdim=True)
    b2 = fx["b"].reshape(-1, 4, 1, 1) - a["rho"] * g0.sum((2, 3), keepdim=True)
    new = z0 * (1 + e2) + b2
    lo, hi = z0 - a["constraint_value"], z0 + a["constraint_value"]
    new = torch.where(new < lo, lo, new)
    new = torch.where(new > hi, hi, new)
    assert (z.cpu() - new).abs().max().item() < 2e-4
    # against the reference's own output (measured 0.037-0.054; the forward x0 error of the bf16 UNet re-draws some of the guide's masks)
    assert rel(z, fx["ref_transform_z"]) < 0.07
    # gradient wrt (e, b) against the oracle's autograd
    args = O.SamplerArgs(**fx["args"])
    unet, vae, guide, sched = models
    emb = torch.cat([fx["negative_embeds"], fx["prompt_embeds"]])
    _, _, (ge, gb) = O.transform_guidance(args, fx["z"], fx["targets"], fx["guide_timesteps"], sched, unet, emb, vae, guide, fx["e"], fx["b"],
                                          fx["Pc"], fx["Pg"], cfg.guide.input_size)
    ge_h = (gz0.cpu() * fx["z"]).sum((2, 3), keepdim=True)
    gb_h = gz0.cpu().sum((2, 3), keepdim=True)
    # against the oracle's OWN forward point (its masks, not the engine's): bounded by the conditioning of the guide's gradient, see
    # the module docstring.  The parity statement proper is test_energy_gradient_at_the_same_image; what is asserted here is that the
    # engine is no further from the oracle than the oracle is from ITSELF under a perturbation of its input latents of the size of the
    # engine's forward error (x0 within 2.3 %: test_denoise_step_vs_reference_fixture) -- measured in this test, two draws, so that the
    # bound follows the conditioning of this (weights, inputs) pair instead of a constant that every change of rounding inside the
    # engine re-draws.  Measured, same box: engine (ge, gb) 0.168 / 0.048 with the rcp + exp2 GELU in the GEGLU epilogues, 0.250 / 0.078
    # with the polynomial GELU (1e-5 accurate, tests/test_oracle.py::test_gelu_polynomial) -- and the oracle against itself under the
    # two perturbation draws 0.417 / 0.146 and 0.184 / 0.068
    own = []
    for seed in (1, 2):
        gp = torch.Generator().manual_seed(seed)
        zp = fx["z"] * (1 + 0.02 * torch.randn(fx["z"].shape, generator=gp))
        _, _, (ge_p, gb_p) = O.transform_guidance(args, zp, fx["targets"], fx["guide_timesteps"], sched, unet, emb, vae, guide, fx["e"], fx["b"],
                                                  fx["Pc"], fx["Pg"], cfg.guide.input_size)
        # d(ge)/dz itself contributes (ge is a sum over g_z * z): first order in the 2 % perturbation, far below the mask effect
        own.append((rel(ge_p, ge), rel(gb_p, gb)))
    c_ge, c_gb = max(o[0] for o in own), max(o[1] for o in own)
    print("transform guidance, own forward point: engine vs oracle (ge, gb) rel %.3f %.3f | oracle vs itself under a 2 %% perturbation of z: %s"
          % (rel(ge_h, ge), rel(gb_h, gb), " ".join("(%.3f %.3f)" % o for o in own)))
    assert rel(ge_h, ge) < max(0.20, 1.5 * c_ge) and rel(gb_h, gb) < max(0.20, 1.5 * c_gb)


def test_energy_gradient_at_the_same_image(setup, fx):
    """The hand-derived energy gradient (A9: replaces torch.autograd.grad at generate_data.py:721 / :761) against the oracle's autograd
    with the guide evaluated AT the engine's decoded image (straight-through hook of the oracle: same ReLU / max-pool masks on both
    sides, gradient through the oracle's own decoder, DDIM/CFG algebra and UNet)."""
    cfg, eng, models, O = setup
    unet, vae, guide, sched = models
    args = O.SamplerArgs(**fx["args"])
    emb = torch.cat([fx["negative_embeds"], fx["prompt_embeds"]])
    first = fx["timesteps"].tolist().index(fx["guide_timesteps"][0])
    # direct guidance: per-pixel g_z
    zn, x0, score, gz = eng.direct_guidance(fx["z"], fx["targets"], first)
    img = eng.guided_image(0)
    _, _, sc_ref, g_ref = O.direct_guidance(args, fx["z"], fx["targets"], fx["guide_timesteps"][0], sched, unet, emb, vae, guide, fx["Pc"],
                                            fx["Pg"], cfg.guide.input_size, image_at=img)
    assert abs(score.item() - float(sc_ref)) < 1e-4 * abs(float(sc_ref))       # same image, fp32 guide: the energies agree to fp32
    assert rel(gz, g_ref) < 0.05, rel(gz, g_ref)
    # transform guidance: two chained steps, (e, b) gradients
    z, score, gz0 = eng.transform_guidance(fx["z"], fx["targets"], fx["e"], fx["b"], first, 2)
    imgs = [eng.guided_image(0), eng.guided_image(1)]
    _, s_ref, (ge, gb) = O.transform_guidance(args, fx["z"], fx["targets"], fx["guide_timesteps"], sched, unet, emb, vae, guide, fx["e"],
                                              fx["b"], fx["Pc"], fx["Pg"], cfg.guide.input_size, images_at=imgs)
    assert abs(score.item() - float(s_ref)) < 1e-4 * abs(float(s_ref))
    ge_h = (gz0.cpu() * fx["z"]).sum((2, 3), keepdim=True)
    gb_h = gz0.cpu().sum((2, 3), keepdim=True)
    assert rel(ge_h, ge) < 0.08 and rel(gb_h, gb) < 0.08, (rel(ge_h, ge), rel(gb_h, gb))


def test_energy_mean_runs_over_the_reference_batch(hip_lib, fx):
    """The reference's energy is a `.mean()` over ITS batch (train_batch_size images, generate_data.py:709-719, 750-760).  When the CLI
    packs 2 reference batches of 1 image into one engine batch of 2 (dd_set_sample_weights w = [1, 1]) each image must get the
    gradient a B = 1 engine gives it -- not half of it."""
    from distdiff_amd.config import tiny_config
    from distdiff_amd.engine import Engine
    from distdiff_amd.scheduler import DDIMSchedule
    from distdiff_amd.weights import synthetic_weights
    a = fx["args"]
    first = fx["timesteps"].tolist().index(fx["guide_timesteps"][0])
    emb_n, emb_p = fx["negative_embeds"], fx["prompt_embeds"]

    def make(B):
        cfg = tiny_config(max_batch=B)
        eng = Engine(cfg, synthetic_weights(cfg, seed=0, num_classes=5), enable_grad=True, max_guidance_period=2)
        sched = DDIMSchedule(cfg.scheduler)
        eng.set_schedule(sched.set_timesteps(fx["n_steps"]), sched.alphas_cumprod, sched.final_alpha_cumprod, guidance_scale=a["guidance_scale"],
                         gs=a["gs"], ls=a["ls"], rho=a["rho"], constraint_value=a["constraint_value"], guidance_period=a["guidance_period"])
        eng.set_prototypes(fx["Pc"], fx["Pg"])
        return eng

    e2 = make(2)
    e2.set_prompt(torch.cat([emb_n, emb_p]).cuda())
    _, s_mean, g_mean = e2.transform_guidance(fx["z"], fx["targets"], fx["e"], fx["b"], first, 2)      # default: mean over the engine batch
    e2.set_sample_weights([1.0, 1.0])
    _, s_sum, g_w = e2.transform_guidance(fx["z"], fx["targets"], fx["e"], fx["b"], first, 2)
    per_image = e2.image_scores().cpu()
    assert torch.allclose(g_w, 2.0 * g_mean, rtol=1e-3, atol=1e-7)
    assert abs(per_image.mean().item() - s_mean.item()) < 1e-4 * s_mean.item() and abs(per_image.sum().item() - s_sum.item()) < 1e-4 * s_sum.item()
    e2.set_sample_weights([1.0, 0.0])                      # a padding row neither contributes to the score nor receives a gradient
    _, s_pad, g_pad = e2.transform_guidance(fx["z"], fx["targets"], fx["e"], fx["b"], first, 2)
    assert float(g_pad[1].abs().max()) == 0.0 and abs(s_pad.item() - per_image[0].item()) < 1e-4 * s_pad.item()
    e2.close()
    e1 = make(1)
    for i in range(2):
        e1.set_prompt(torch.cat([emb_n[i:i + 1], emb_p[i:i + 1]]).cuda())
        _, s1, g1 = e1.transform_guidance(fx["z"][i:i + 1], fx["targets"][i:i + 1], fx["e"][i:i + 1], fx["b"][i:i + 1], first, 2)
        assert abs(s1.item() - per_image[i].item()) < 2e-3 * s1.item()
        assert rel(g_w[i:i + 1], g1) < 0.05, rel(g_w[i:i + 1], g1)       # different batch -> different tile paths / rounding patterns
    e1.close()


def test_direct_guidance_vs_reference_fixture(setup, fx):
    cfg, eng, models, O = setup
    first = fx["timesteps"].tolist().index(fx["guide_timesteps"][0])
    zn, x0, score, gz = eng.direct_guidance(fx["z"], fx["targets"], first)
    assert abs(score.item() - float(fx["ref_direct_score"])) < 0.01 * abs(float(fx["ref_direct_score"]))
    assert rel(zn, fx["ref_direct_z_next"]) < 0.03
    assert rel(x0, fx["ref_direct_x0"]) < 0.04
    assert abs(score.item() - float(fx["ref_direct_score"])) < 0.002 * abs(float(fx["ref_direct_score"]))


@pytest.mark.parametrize("gt", ["transform_guidance", "direct_guidance", None])
def test_expand_loop_vs_golden(setup, fx, gt):
    cfg, eng, models, O = setup
    first = fx["timesteps"].tolist().index(fx["guide_timesteps"][0])
    key = gt or "none"
    z, img, score = eng.expand(fx["lat"], fx["noise"], fx["e"], fx["b"], fx["targets"], fx["start_index"], gt, first, 2)
    assert rel(z, fx["expand_%s_z" % key]) < 0.03
    ref = fx["expand_%s_img_u8" % key].float() / 255.0
    err = (img.cpu() - ref).abs()
    assert float(err.max()) < 0.08
    # output stage (f-3, generate_data.py:1227-1234): dd_image_to_u8 bytes == save_image's mul(255).add_(0.5).clamp_(0,255).to(uint8)
    # of the engine's own fp32 image, exactly; and within the forward tolerance of the golden bytes
    u8 = eng.image_to_u8(img).cpu()
    mine = img.cpu().mul(255).add_(0.5).clamp_(0, 255).to(torch.uint8).permute(0, 2, 3, 1)
    assert torch.equal(u8, mine)
    gold = fx["expand_%s_img_u8" % key]
    gold = gold.permute(0, 2, 3, 1) if gold.shape[1] == 3 else gold
    d = (u8.int() - gold.int()).abs()
    assert int(d.max()) <= 21 and float((d <= 4).float().mean()) > 0.9, (int(d.max()), float((d <= 4).float().mean()))
    mse = float((err ** 2).mean())
    psnr = 10 * torch.log10(torch.tensor(1.0 / mse)).item()
    assert psnr > 30.0, psnr
    if gt:
        s_ref = float(fx["expand_%s_score" % key])
        assert abs(score.item() - s_ref) < 0.005 * abs(s_ref)
    # determinism: no float atomics on the data path -> bitwise identical on a second run
    z2, img2, _ = eng.expand(fx["lat"], fx["noise"], fx["e"], fx["b"], fx["targets"], fx["start_index"], gt, first, 2)
    assert torch.equal(z, z2) and torch.equal(img, img2)


def test_full_size_sd15_properties(hip_lib):
    """BASELINE.json full sizes (SD-1.5 shapes, 512x512): size-independent properties of the guided step."""
    from distdiff_amd.config import sd15_config
    from distdiff_amd.engine import Engine
    from distdiff_amd.scheduler import DDIMSchedule
    from distdiff_amd.weights import synthetic_weights
    cfg = sd15_config(max_batch=1)
    eng = Engine(cfg, synthetic_weights(cfg, seed=0, num_classes=100), enable_grad=True, max_guidance_period=2)
    sched = DDIMSchedule(cfg.scheduler)
    ts = sched.set_timesteps(50)
    eng.set_schedule(ts, sched.alphas_cumprod, sched.final_alpha_cumprod)
    g = torch.Generator().manual_seed(0)
    D = cfg.guide.feature_dim
    eng.set_prototypes(F.normalize(torch.randn(100, D, generator=g), dim=-1), F.normalize(torch.randn(100, 3, D, generator=g), dim=-1))
    eng.set_prompt(torch.randn(2, 77, 768, generator=g).cuda())
    z = torch.randn(1, 4, 64, 64, generator=g)
    tg = torch.tensor([7])
    e, b = torch.rand(1, 4, generator=g), torch.randn(1, 4, generator=g)
    z1, s1, g1 = eng.transform_guidance(z, tg, e, b, 30, 2)
    z2, s2, g2 = eng.transform_guidance(z, tg, e, b, 30, 2)
    assert torch.isfinite(z1).all() and torch.isfinite(g1).all() and torch.isfinite(s1).all()
    assert torch.equal(z1, z2) and torch.equal(g1, g2) and torch.equal(s1, s2)          # idempotent / deterministic
    assert float((z1.cpu() - z).abs().max()) <= 0.2 + 1e-5                              # L-inf projection
    assert float(g1.abs().max()) > 0                                                    # a gradient actually flowed
    zp, x0 = eng.denoise_step(z, 30)
    a, ap = float(sched.alphas_cumprod[ts[30]]), float(sched.alphas_cumprod[ts[31]])
    eps = (z.cuda() - a ** 0.5 * x0) / (1 - a) ** 0.5                                   # invert the DDIM algebra
    assert torch.allclose(zp, ap ** 0.5 * x0 + (1 - ap) ** 0.5 * eps, rtol=1e-3, atol=1e-3)
    img = eng.decode(zp)
    assert img.shape == (1, 3, 512, 512) and float(img.min()) >= 0 and float(img.max()) <= 1
    # VJP linearity at full size
    gg = torch.randn(2, 4, 64, 64, generator=g)
    # linear in the cotangent up to the bf16 rounding-pattern noise of the reverse pass (measured by bisecting the reverse program op by op: MFMA accumulation is
    # homogeneous only to ~1e-7, which re-draws the bf16 roundings downstream; the two runs then differ like two noise draws)
    v1, v2 = eng.unet_vjp(z, 30, gg), eng.unet_vjp(z, 30, -4.0 * gg)
    assert rel(v2, -4.0 * v1) < 0.03
    eng.close()


def test_config1_sd15_shapes_256px_guidance_off_vs_oracle(hip_lib):
    """BASELINE.json configs[0]: real SD-1.x / AutoencoderKL shapes, 256x256 (latent 32), 10-step DDIM schedule, strength 1.0,
    energy guidance OFF — the engine against the fp32 CPU oracle on identical seeded inputs (full architecture, not the tiny one)."""
    from distdiff_amd.config import sd15_config
    from distdiff_amd.engine import Engine
    from distdiff_amd.scheduler import DDIMSchedule
    from distdiff_amd.weights import synthetic_weights
    from oracle import sd_oracle as O
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    B = 1
    cfg = sd15_config(latent_size=32, max_batch=B)
    w = synthetic_weights(cfg, seed=0, num_classes=10)
    eng = Engine(cfg, w, enable_grad=False, max_guidance_period=1)
    sched = DDIMSchedule(cfg.scheduler)
    ts = sched.set_timesteps(10)
    eng.set_schedule(ts, sched.alphas_cumprod, sched.final_alpha_cumprod)
    g = torch.Generator().manual_seed(1)
    lat = torch.randn(B, 4, 32, 32, generator=g) * 0.18215 * 5
    noise = torch.randn(B, 4, 32, 32, generator=g)
    pe = torch.randn(B, 77, 768, generator=g)
    ne = torch.randn(B, 77, 768, generator=g)
    eng.set_prompt(torch.cat([ne, pe]).cuda())
    z, img, _ = eng.expand(lat, noise, None, None, torch.zeros(B, dtype=torch.int64), 0, None, 0, 0)
    args = O.SamplerArgs(guidance_type=None, num_inference_steps=10, strength=1.0)
    zr, imr, _ = O.expand_one(args, cfg, O.build_models(cfg, w), lat, noise, None, None, pe, ne, torch.zeros(B, dtype=torch.int64), None, None)
    assert rel(z, zr) < 0.05, rel(z, zr)
    err = (img.cpu() - imr).abs()
    psnr = 10 * torch.log10(1.0 / (err ** 2).mean()).item()
    assert float(err.max()) < 0.2 and psnr > 28.0, (float(err.max()), psnr)
    eng.close()


def test_packed_weights_export_import(hip_lib, fx):
    """Multi-GPU start-up path (launcher.build_engine_distributed) on one device: an engine built from the tensor SHAPES alone
    (dd_declare_tensor) receives the packed weight buffers of a loaded engine (dd_export_packed -> dd_import_packed, in ragged buckets)
    and then produces bitwise identical results, UNet / VAE / guide / text encoder / time-embedding tables included."""
    from distdiff_amd.config import tiny_config
    from distdiff_amd.engine import Engine
    from distdiff_amd.scheduler import DDIMSchedule
    from distdiff_amd.weights import synthetic_weights
    cfg = tiny_config(max_batch=2)
    w = synthetic_weights(cfg, seed=0, num_classes=5, encoders=True)
    a = Engine(cfg, w, enable_grad=True, max_guidance_period=2)
    b = Engine(cfg, None, enable_grad=True, max_guidance_period=2, layout=a.weight_layout())
    total = a.packed_bytes()
    assert total == b.packed_bytes() and total > 1 << 20
    off, step = 0, 1000003                      # deliberately not aligned to any buffer boundary
    while off < total:
        n = min(step, total - off)
        buf = torch.empty(n, dtype=torch.uint8, device="cuda")
        a.export_packed(buf, off)
        b.import_packed(buf, off)
        off += n
    torch.cuda.synchronize()
    sched = DDIMSchedule(cfg.scheduler)
    ts = sched.set_timesteps(fx["n_steps"])
    first = fx["timesteps"].tolist().index(fx["guide_timesteps"][0])
    outs = []
    for eng in (a, b):
        eng.set_schedule(ts, sched.alphas_cumprod, sched.final_alpha_cumprod, guidance_period=2)
        eng.set_prototypes(fx["Pc"], fx["Pg"])
        eng.set_prompt(torch.cat([fx["negative_embeds"], fx["prompt_embeds"]]).cuda())
        z, score, gz0 = eng.transform_guidance(fx["z"], fx["targets"], fx["e"], fx["b"], first, 2)
        ids = torch.arange(2 * cfg.text_len).reshape(2, cfg.text_len) % cfg.text.vocab_size
        img = torch.linspace(-1, 1, 2 * 3 * (8 * cfg.latent_size) ** 2).reshape(2, 3, 8 * cfg.latent_size, 8 * cfg.latent_size)
        outs.append((z, score, gz0, eng.text_encode(ids), eng.vae_encode(img), eng.decode(z)))
    for x, y in zip(*outs):
        assert torch.equal(x, y)
    a.close(); b.close()


def test_graph_replay_of_the_plain_step_is_bitwise_identical(hip_lib, fx):
    """DD_GRAPH=1: the plain denoise step (UNet forward + CFG + DDIM, ~700 launches) is captured into a hipGraph on its second
    sighting and replayed afterwards; results are bitwise those of the plain launches (off by default: no gain measured)."""
    import subprocess
    import sys
    code = (
        "import torch, sys\n"
        "sys.path.insert(0, %r)\n"
        "from distdiff_amd.config import tiny_config\n"
        "from distdiff_amd.engine import Engine\n"
        "from distdiff_amd.scheduler import DDIMSchedule\n"
        "from distdiff_amd.weights import synthetic_weights\n"
        "cfg = tiny_config(max_batch=2)\n"
        "eng = Engine(cfg, synthetic_weights(cfg, seed=0, num_classes=5), enable_grad=False, max_guidance_period=1)\n"
        "s = DDIMSchedule(cfg.scheduler); ts = s.set_timesteps(10)\n"
        "eng.set_schedule(ts, s.alphas_cumprod, s.final_alpha_cumprod)\n"
        "g = torch.Generator().manual_seed(0)\n"
        "eng.set_prompt(torch.randn(4, cfg.text_len, cfg.unet.cross_attention_dim, generator=g).cuda())\n"
        "lat = torch.randn(2, 4, cfg.latent_size, cfg.latent_size, generator=g); nz = torch.randn(lat.shape, generator=g)\n"
        "outs = [eng.expand(lat, nz, None, None, torch.zeros(2, dtype=torch.int64), 0, None, 0, 0)[0].cpu() for _ in range(4)]\n"
        "assert all(torch.equal(outs[0], o) for o in outs[1:])\n"
        "print('SUM %%.6f' %% float(outs[0].double().abs().sum()))\n" % os.path.join(os.path.dirname(__file__), ".."))
    res = {}
    for flag in ("0", "1"):
        env = dict(os.environ, DD_GRAPH=flag)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[flag] = [l for l in r.stdout.splitlines() if l.startswith("SUM")][0]
    assert res["0"] == res["1"], res


@pytest.mark.parametrize("which", ["tiny", "sd15_256px", "tiny_sdxl", "tiny_vit_guide"])
def test_gradient_slab_by_liveness_is_bitwise_identical(hip_lib, which):
    """plan_grad_memory (engine_graph.cpp): gradient buffers share bytes when their live intervals in the reverse run are disjoint.  The
    guided results -- updated latents, scores, dE/dz0 of the chained transform guidance and of direct guidance -- must be BITWISE those of
    one private range per tensor (DD_NO_GRAD_REUSE=1), twice in a row (stale bytes of a previous run in a shared range must not leak), and
    the workspace must shrink.  tiny config and SD-1.5 widths at 256x256 (every op kind of the three programs), the SDXL structure (per-level
    transformer depth, nn.Linear proj_in / proj_out, text_time conditioning) and the ViT guide (OP_PATCHIFY / OP_VITEMBED / OP_SELECT /
    OP_ACT, fp32 image gradient).  tests/conftest.py starts the whole suite with DD_GRAD_CHECK=1: every gradient access of the reverse
    programs is checked against the interval the slab was packed by (a violation raises out of the guided call)."""
    from distdiff_amd.config import GuideConfig, sd15_config, tiny_config, tiny_sdxl_config
    from distdiff_amd.engine import Engine
    from distdiff_amd.scheduler import DDIMSchedule
    from distdiff_amd.weights import synthetic_weights
    cfg = {"tiny": lambda: tiny_config(max_batch=2), "sd15_256px": lambda: sd15_config(latent_size=32, max_batch=2),
           "tiny_sdxl": lambda: tiny_sdxl_config(max_batch=2), "tiny_vit_guide": lambda: tiny_config(max_batch=2)}[which]()
    if which == "tiny_vit_guide":
        cfg.guide = GuideConfig(kind="vit", input_size=64, vit_width=64, vit_layers=2, vit_heads=2, vit_patch=16, vit_mlp=128, vit_out=32)
    ncls = 100 if which == "sd15_256px" else 5
    w = synthetic_weights(cfg, seed=0, num_classes=ncls)
    g = torch.Generator().manual_seed(9)
    L, D = cfg.latent_size, cfg.guide.feature_dim
    z = torch.randn(2, 4, L, L, generator=g)
    e, b = torch.rand(2, 4, 1, 1, generator=g), torch.randn(2, 4, 1, 1, generator=g) * 0.3
    emb = torch.randn(4, cfg.text_len, cfg.unet.cross_attention_dim, generator=g)
    Pc = torch.nn.functional.normalize(torch.randn(ncls, D, generator=g), dim=-1)
    Pg = torch.nn.functional.normalize(torch.randn(ncls, 3, D, generator=g), dim=-1)
    tg = torch.tensor([1, 3])
    outs, sizes = [], []
    old = os.environ.pop("DD_NO_GRAD_REUSE", None)
    try:
        for flag in (None, "1"):
            if flag:
                os.environ["DD_NO_GRAD_REUSE"] = flag
            eng = Engine(cfg, w, enable_grad=True, max_guidance_period=2)
            os.environ.pop("DD_NO_GRAD_REUSE", None)
            sched = DDIMSchedule(cfg.scheduler)
            ts = sched.set_timesteps(10)
            eng.set_schedule(ts, sched.alphas_cumprod, sched.final_alpha_cumprod, guidance_period=2)
            eng.set_prototypes(Pc, Pg)
            eng.set_prompt(emb.cuda())
            if cfg.unet.add_time_dim:
                eng.set_added_cond(torch.randn(4, cfg.unet.add_text_dim, generator=torch.Generator().manual_seed(2)),
                                   torch.tensor([[128.0, 96.0, 0.0, 8.0, 128.0, 128.0], [64.0, 64.0, 4.0, 0.0, 128.0, 128.0]] * 2))
            eng.set_sample_weights([1.0, 1.0])
            res = []
            for _ in range(2):
                zt, st, gt = eng.transform_guidance(z, tg, e, b, 5, 2)
                zd, x0, sd, gd = eng.direct_guidance(z, tg, 7)
                res += [zt.cpu(), st.cpu(), gt.cpu(), zd.cpu(), sd.cpu(), gd.cpu()]
            outs.append(res)
            sizes.append(eng.workspace_bytes())
            eng.close()
    finally:
        if old is not None:
            os.environ["DD_NO_GRAD_REUSE"] = old
    for x, y in zip(*outs):
        assert torch.isfinite(x).all() and torch.equal(x, y)
    for x, y in zip(outs[0][:6], outs[0][6:]):
        assert torch.equal(x, y)                          # run 2 == run 1
    assert sizes[0] < sizes[1], sizes
    print("%s: workspace %.3f GB with liveness-packed gradients, %.3f GB with one range per tensor" % (which, sizes[0] / 1e9, sizes[1] / 1e9))
