"""SDXL-style UNet (BASELINE.json configs[4]: SDXL-base UNet; beyond what the reference can run) through the same engine:
3 levels, `transformer_layers_per_block` > 1, per-level head counts, nn.Linear proj_in / proj_out, `text_time` additional conditioning
(per-image time-embedding bias).  Parity against the oracle's restatement of the diffusers UNet2DConditionModel at a test-size config;
tolerances as for the SD-1.x UNet (bf16 storage / MFMA, fp32 accumulation): forward 3 %, VJP 5 %."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    assert torch.isfinite(a).all()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


@pytest.fixture(scope="module")
def world(hip_lib):
    from distdiff_amd.config import tiny_sdxl_config
    from distdiff_amd.engine import Engine
    from distdiff_amd.scheduler import DDIMSchedule
    from distdiff_amd.weights import synthetic_weights
    from oracle import sd_oracle as O
    B = 2
    cfg = tiny_sdxl_config(max_batch=B)
    w = synthetic_weights(cfg, seed=0, num_classes=5)
    eng = Engine(cfg, w, enable_grad=True, max_guidance_period=2)
    sched = DDIMSchedule(cfg.scheduler)
    ts = sched.set_timesteps(10)
    eng.set_schedule(ts, sched.alphas_cumprod, sched.final_alpha_cumprod, guidance_period=2)
    g = torch.Generator().manual_seed(4)
    L, D = cfg.latent_size, cfg.guide.feature_dim
    d = {"z": torch.randn(B, 4, L, L, generator=g),
         "pe": torch.randn(B, cfg.text_len, cfg.unet.cross_attention_dim, generator=g),
         "ne": torch.randn(B, cfg.text_len, cfg.unet.cross_attention_dim, generator=g),
         "te": torch.randn(2 * B, cfg.unet.add_text_dim, generator=g),
         "ti": torch.tensor([[128.0, 96.0, 0.0, 8.0, 128.0, 128.0], [64.0, 64.0, 4.0, 0.0, 128.0, 128.0]] * B),
         "e": torch.rand(B, 4, 1, 1, generator=g), "b": torch.randn(B, 4, 1, 1, generator=g) * 0.3,
         "Pc": F.normalize(torch.randn(5, D, generator=g), dim=-1), "Pg": F.normalize(torch.randn(5, 3, D, generator=g), dim=-1),
         "tg": torch.tensor([1, 3]), "gg": torch.randn(2 * B, 4, L, L, generator=g)}
    eng.set_prompt(torch.cat([d["ne"], d["pe"]]).cuda())
    eng.set_added_cond(d["te"], d["ti"])
    eng.set_prototypes(d["Pc"], d["Pg"])
    unet, vae, guide, so = O.build_models(cfg, w)
    unet.added_cond = {"text_embeds": d["te"], "time_ids": d["ti"]}
    tso = so.set_timesteps(10)
    yield cfg, eng, d, (unet, vae, guide, so), [int(t) for t in tso], O
    eng.close()


def test_sdxl_unet_forward_and_vjp(world):
    cfg, eng, d, (unet, vae, guide, so), ts, O = world
    emb = torch.cat([d["ne"], d["pe"]])
    zr = d["z"].clone().requires_grad_(True)
    eps = unet(torch.cat([zr, zr]), ts[5], emb)[0]
    (gz,) = torch.autograd.grad(eps, zr, d["gg"])
    assert rel(eng.unet_forward(d["z"], 5), eps.detach()) < 0.03
    assert rel(eng.unet_vjp(d["z"], 5, d["gg"]), gz) < 0.05
    # the conditioning is per image and per step: another step index / other time ids must change the output
    assert rel(eng.unet_forward(d["z"], 6), eps.detach()) > 0.005
    eng.set_added_cond(d["te"], d["ti"].flip(0))
    assert rel(eng.unet_forward(d["z"], 5), eps.detach()) > 0.005
    eng.set_added_cond(d["te"], d["ti"])


def test_sdxl_guided_step_and_loop(world):
    cfg, eng, d, (unet, vae, guide, so), ts, O = world
    emb = torch.cat([d["ne"], d["pe"]])
    args = O.SamplerArgs(guidance_type="transform_guidance", num_inference_steps=10, guidance_step=5, guidance_period=2, strength=0.5)
    zn, score, gz0 = eng.transform_guidance(d["z"], d["tg"], d["e"], d["b"], 5, 2)
    zr, sr, (ge, gb) = O.transform_guidance(args, d["z"], d["tg"], ts[5:7], so, unet, emb, vae, guide, d["e"], d["b"], d["Pc"], d["Pg"],
                                            cfg.guide.input_size, images_at=None)
    assert abs(score.item() - float(sr)) < 0.01 * abs(float(sr))
    assert float((zn.cpu() - d["z"]).abs().max()) <= 0.2 + 1e-5
    imgs = [eng.guided_image(0), eng.guided_image(1)]
    _, s2, (ge2, gb2) = O.transform_guidance(args, d["z"], d["tg"], ts[5:7], so, unet, emb, vae, guide, d["e"], d["b"], d["Pc"], d["Pg"],
                                             cfg.guide.input_size, images_at=imgs)
    ge_h = (gz0.cpu() * d["z"]).sum((2, 3), keepdim=True)
    gb_h = gz0.cpu().sum((2, 3), keepdim=True)
    assert rel(ge_h, ge2) < 0.08 and rel(gb_h, gb2) < 0.08, (rel(ge_h, ge2), rel(gb_h, gb2))
    # whole loop against the oracle, guidance off and on
    lat = d["z"] * 0.5
    noise = d["gg"][:2]
    for gt in (None, "transform_guidance"):
        a = O.SamplerArgs(guidance_type=gt, num_inference_steps=10, guidance_step=5, guidance_period=2, strength=0.5)
        z, img, s = eng.expand(lat, noise, d["e"], d["b"], d["tg"], 5, gt, 5, 2)
        zo, imo, _ = O.expand_one(a, cfg, (unet, vae, guide, so), lat, noise, d["e"], d["b"], d["pe"], d["ne"], d["tg"], d["Pc"], d["Pg"])
        # guided: the oracle's masks are drawn at ITS forward point (conditioning of the guide's gradient, tests/test_engine_gpu.py): 6.7 %
        # with the round-5 binary, 12.1 % with round 6's (another equally accurate rounding of the GEGLU epilogue re-draws them)
        assert rel(z, zo) < (0.15 if gt else 0.04), (gt, rel(z, zo))
        imax = float((img.cpu() - imo).abs().max())
        print("tiny SDXL loop (%s): latents rel %.4f, image rel %.4f, image max abs %.4f" % (gt, rel(z, zo), rel(img, imo), imax))
        # (guided, worst single pixel of the tiny random-weight model: 0.17 with the round-5 binary, 0.32 with round 6's)
        assert imax < (0.4 if gt else 0.1)
