"""CPU tests of the oracle (test infrastructure) — the restated sampler vs fixtures produced by the REFERENCE's own
functions (tests/golden/make_fixtures.py executes generate_data.py:109-137, 687-767 from /root/reference), plus the
schedule / shard known answers of SURVEY.md section 8c."""
import os

import pytest
import torch
import torch.nn.functional as F

from distdiff_amd.config import tiny_config
from distdiff_amd.weights import synthetic_weights
from oracle import sd_oracle as O

FIX = os.path.join(os.path.dirname(__file__), "golden", "tiny_fixture.pt")


@pytest.fixture(scope="module")
def fx():
    return torch.load(FIX, weights_only=False)


@pytest.fixture(scope="module")
def models():
    cfg = tiny_config(max_batch=2)
    w = synthetic_weights(cfg, seed=0, num_classes=5)
    return cfg, w, O.build_models(cfg, w)


def test_schedule_known_answers():
    from distdiff_amd.config import sd15_config
    s = O.DDIMSchedulerOracle(sd15_config())
    ts = s.set_timesteps(50)
    assert ts.tolist() == [981 - 20 * i for i in range(50)]
    assert [O.start_index(x, 50) for x in (0.5, 0.9, 1.0)] == [25, 4, 0]
    assert O.guide_timesteps(ts, 20, 2) == [381, 361]
    a = s.alphas_cumprod
    assert abs(float(a[0]) - 0.99915) < 1e-5 and abs(float(a[999]) - 0.0046602) < 1e-6
    # prev of the last step (t=1) is final_alpha_cumprod = alphas_cumprod[0] (set_alpha_to_one=False)
    at, ap = s.coefficients(1)
    assert float(ap) == float(a[0])


def test_product_schedule_matches_oracle():
    from distdiff_amd.config import sd15_config
    from distdiff_amd.scheduler import DDIMSchedule, guide_window, start_index
    p = DDIMSchedule(sd15_config().scheduler)
    o = O.DDIMSchedulerOracle(sd15_config())
    assert p.set_timesteps(50) == o.set_timesteps(50).tolist()
    assert torch.allclose(torch.from_numpy(p.alphas_cumprod), o.alphas_cumprod, rtol=1e-6, atol=0)
    assert guide_window(50, 20, 2) == (30, 2) and p.timesteps[30:32] == [381, 361]
    assert start_index(0.5, 50) == 25


def test_shard_ranges_match_reference_formula():
    from distdiff_amd.launcher import shard_range
    for total, nsplit in [(9144, 4), (9144, 8), (8, 8), (10, 4), (1, 1)]:
        per = -(-total // nsplit)
        got = [shard_range(total, nsplit, s) for s in range(nsplit)]
        assert got == [O.shard_range(total, nsplit, s) for s in range(nsplit)]
        flat = [i for r in got for i in r if i < total]
        assert sorted(set(flat)) == list(range(total))
        assert all(len(r) <= per for r in got)


def test_reference_denoise_one_step(fx, models):
    cfg, w, (unet, vae, guide, sched) = models
    sched.set_timesteps(fx["n_steps"])
    args = O.SamplerArgs(**fx["args"])
    emb = torch.cat([fx["negative_embeds"], fx["prompt_embeds"]])
    with torch.no_grad():
        zp, x0 = O.denoise_one_step(args, fx["z"], sched, fx["guide_timesteps"][0], unet, emb)
    assert torch.allclose(zp, fx["ref_denoise_z_prev"], rtol=1e-5, atol=1e-5)
    assert torch.allclose(x0, fx["ref_denoise_x0"], rtol=1e-5, atol=1e-5)


def test_reference_transform_guidance(fx, models):
    cfg, w, (unet, vae, guide, sched) = models
    sched.set_timesteps(fx["n_steps"])
    args = O.SamplerArgs(**fx["args"])
    emb = torch.cat([fx["negative_embeds"], fx["prompt_embeds"]])
    z, score, _ = O.transform_guidance(args, fx["z"], fx["targets"], fx["guide_timesteps"], sched, unet, emb, vae, guide, fx["e"], fx["b"],
                                       fx["Pc"], fx["Pg"], cfg.guide.input_size)
    assert abs(float(score) - float(fx["ref_transform_score"])) < 1e-4
    assert torch.allclose(z, fx["ref_transform_z"], rtol=1e-4, atol=1e-4)
    # quirk 7: the result sits inside the L-inf ball around the input latent
    assert float((z - fx["z"]).abs().max()) <= args.constraint_value + 1e-6


def test_reference_direct_guidance(fx, models):
    cfg, w, (unet, vae, guide, sched) = models
    sched.set_timesteps(fx["n_steps"])
    args = O.SamplerArgs(**fx["args"])
    emb = torch.cat([fx["negative_embeds"], fx["prompt_embeds"]])
    zn, x0, score, _ = O.direct_guidance(args, fx["z"], fx["targets"], fx["guide_timesteps"][0], sched, unet, emb, vae, guide, fx["Pc"],
                                         fx["Pg"], cfg.guide.input_size)
    assert abs(float(score) - float(fx["ref_direct_score"])) < 1e-5
    assert torch.allclose(zn, fx["ref_direct_z_next"], rtol=1e-4, atol=1e-5)
    assert torch.allclose(x0, fx["ref_direct_x0"], rtol=1e-5, atol=1e-5)


def test_reference_linfball_clamp_order(fx):
    c, t = fx["clamp_center"], fx["clamp_t"]
    lo, hi = c - 0.2, c + 0.2
    out = torch.where(t < lo, lo, t)
    out = torch.where(out > hi, hi, out)
    assert torch.equal(out, fx["ref_clamp_out"])


def test_expand_loop_call_trace_and_vectors(fx, models):
    cfg, w, ms = models
    args = O.SamplerArgs(**fx["args"])
    for gt in ("transform_guidance", "direct_guidance", None):
        key = gt or "none"
        a = O.SamplerArgs(**{**fx["args"], "guidance_type": gt})
        tr = []
        z, img, s = O.expand_one(a, cfg, ms, fx["lat"], fx["noise"], fx["e"], fx["b"], fx["prompt_embeds"], fx["negative_embeds"],
                                 fx["targets"], fx["Pc"], fx["Pg"], trace=tr)
        assert tr == fx["expand_%s_trace" % key]
        assert torch.allclose(z, fx["expand_%s_z" % key], rtol=1e-4, atol=1e-4)
        u8 = (img * 255 + 0.5).clamp(0, 255).to(torch.uint8)
        assert (u8.int() - fx["expand_%s_img_u8" % key].int()).abs().max() <= 1
    # per-image call sequence of the script of record: one transform_guidance + re-run of that step, plain steps elsewhere
    tr = fx["expand_transform_guidance_trace"]
    assert [k for k, _ in tr].count("transform_guidance") == 1
    i = [k for k, _ in tr].index("transform_guidance")
    assert tr[i + 1] == ("denoise", tr[i][1])


def test_primitives_against_torch_functional(models):
    """Known-answer checks of the few primitives the oracle does not delegate to torch.nn.functional."""
    t = torch.tensor([1.0, 981.0])
    emb = O.timestep_embedding(t, 8, True, 0.0)
    half = 4
    freq = torch.exp(-torch.log(torch.tensor(10000.0)) * torch.arange(half) / half)
    ref = torch.cat([torch.cos(t[:, None] * freq), torch.sin(t[:, None] * freq)], -1)
    assert torch.allclose(emb, ref, atol=1e-6)
    # DDIM step algebra (eta = 0): z_prev = sqrt(a_p) x0 + sqrt(1-a_p) eps, x0 = (z - sqrt(1-a) eps)/sqrt(a)
    cfg = models[0]
    s = O.DDIMSchedulerOracle(cfg)
    s.set_timesteps(50)
    z, eps = torch.randn(2, 4, 4, 4), torch.randn(2, 4, 4, 4)
    out = s.step(eps, 381, z)
    a, ap = s.alphas_cumprod[381], s.alphas_cumprod[361]
    x0 = (z - (1 - a).sqrt() * eps) / a.sqrt()
    assert torch.allclose(out["pred_original_sample"], x0) and torch.allclose(out["prev_sample"], ap.sqrt() * x0 + (1 - ap).sqrt() * eps)
    # energy: Euclidean distance to class prototype + nearest (by inner product) group prototype
    f = torch.randn(3, 16)
    Pc = F.normalize(torch.randn(4, 16), dim=-1)
    Pg = F.normalize(torch.randn(4, 3, 16), dim=-1)
    y = torch.tensor([0, 2, 3])
    args = O.SamplerArgs(gs=1.0, ls=0.5)
    e = O.energy(args, f, y, Pc, Pg)
    ref = sum((f[i] - Pc[y[i]]).norm() for i in range(3)) / 3
    ref = ref + 0.5 * sum((f[i] - Pg[y[i], int((Pg[y[i]] @ f[i]).argmax())]).norm() for i in range(3)) / 3
    assert abs(float(e) - float(ref)) < 1e-5


def test_clip_text_oracle_matches_transformers_fixture():
    """oracle clip_text_encode vs the vectors recorded from transformers' own CLIPTextModel (tests/golden/make_clip_fixture.py)."""
    import os
    from distdiff_amd.config import tiny_config
    from distdiff_amd.weights import synthetic_text_encoder
    fx = torch.load(os.path.join(os.path.dirname(__file__), "golden", "clip_fixture.pt"), weights_only=False)
    assert len(fx["cases"]) == 2
    for c in fx["cases"]:
        cfg = tiny_config()
        cfg.text.hidden_act = c["hidden_act"]
        out = O.clip_text_encode(cfg, synthetic_text_encoder(cfg, 0), c["input_ids"])
        assert (out - c["last_hidden_state"]).abs().max().item() < 1e-4
        # causality: changing a later token never changes an earlier position
        ids2 = c["input_ids"].clone()
        ids2[:, 7] = (ids2[:, 7] + 1) % cfg.text.vocab_size
        out2 = O.clip_text_encode(cfg, synthetic_text_encoder(cfg, 0), ids2)
        assert torch.equal(out2[:, :7], out[:, :7]) and not torch.equal(out2[:, 7:], out[:, 7:])


def test_clip_text_oracle_sdxl_towers_match_transformers_fixture():
    """SDXL's text side (StableDiffusionXLPipeline.encode_prompt): oracle hidden_states[-2] of both towers and text_embeds of the second
    vs the vectors recorded from transformers' CLIPTextModel / CLIPTextModelWithProjection (tests/golden/make_clip_fixture.py)."""
    import os
    from distdiff_amd.config import tiny_sdxl_config
    from distdiff_amd.weights import synthetic_text_encoder
    fx = torch.load(os.path.join(os.path.dirname(__file__), "golden", "clip_fixture_sdxl.pt"), weights_only=False)
    cfg = tiny_sdxl_config()
    assert [t["which"] for t in fx["towers"]] == [0, 1]
    hid = []
    for t in fx["towers"]:
        sd = synthetic_text_encoder(cfg, 0, which=t["which"])
        tc = cfg.text2 if t["which"] else cfg.text
        assert t["n_hidden_states"] == tc.num_hidden_layers + 1
        if t["which"]:
            h, pooled = O.clip_text_encode(cfg, sd, t["input_ids"], which=1, hidden_layer=-2, pooled=True)
            assert pooled.shape == (4, tc.projection_dim)
            assert (pooled - t["text_embeds"]).abs().max().item() < 1e-4
            # the pooled token is the FIRST position of the largest id: padding behind it (id 0 here) must not matter
            ids2 = t["input_ids"].clone()
            ids2[0, 7:] = 5
            assert torch.allclose(O.clip_text_encode(cfg, sd, ids2, which=1, hidden_layer=-2, pooled=True)[1][0], pooled[0], atol=1e-6)
        else:
            h = O.clip_text_encode(cfg, sd, t["input_ids"], which=0, hidden_layer=-2)
        assert (h - t["hidden_m2"]).abs().max().item() < 1e-4
        # hidden_states[-2] is not the final output
        assert not torch.allclose(h, O.clip_text_encode(cfg, sd, t["input_ids"], which=t["which"]), atol=1e-3)
        hid.append(h)
    emb, pooled = O.sdxl_encode_prompt(cfg, synthetic_text_encoder(cfg, 0, 0), synthetic_text_encoder(cfg, 0, 1),
                                       fx["towers"][0]["input_ids"], fx["towers"][1]["input_ids"])
    assert emb.shape[-1] == cfg.unet.cross_attention_dim and torch.equal(emb, torch.cat(hid, -1))
    assert pooled.shape[-1] == cfg.unet.add_text_dim


def test_vae_encode_oracle_primitives():
    """the encoder downsample is F.pad(0,1,0,1) + stride-2 conv; the sample is mean + exp(logvar/2) * noise, times scaling_factor."""
    from distdiff_amd.config import tiny_config
    from distdiff_amd.weights import synthetic_vae_decoder, synthetic_vae_encoder
    cfg = tiny_config()
    sd = synthetic_vae_decoder(cfg, 0)
    sd.update(synthetic_vae_encoder(cfg, 0))
    g = torch.Generator().manual_seed(3)
    S = 8 * cfg.latent_size
    x = torch.rand(2, 3, S, S, generator=g) * 2 - 1
    n = torch.randn(2, 4, cfg.latent_size, cfg.latent_size, generator=g)
    with torch.no_grad():
        z, mom = O.vae_encode(cfg, sd, x, n)
        zm, mom2 = O.vae_encode(cfg, sd, x, None)
    assert z.shape == (2, 4, cfg.latent_size, cfg.latent_size) and mom.shape == (2, 8, cfg.latent_size, cfg.latent_size)
    assert torch.equal(mom, mom2)
    assert torch.allclose(zm, mom[:, :4] * cfg.vae.scaling_factor)
    assert torch.allclose(z, (mom[:, :4] + torch.exp(0.5 * mom[:, 4:]) * n) * cfg.vae.scaling_factor)
    assert float(mom[:, 4:].min()) >= -30.0 and float(mom[:, 4:].max()) <= 20.0
    # asymmetric padding: the last output row/column sees one zero row/column, the first sees none
    w = torch.randn(5, 3, 3, 3, generator=g)
    a = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, stride=2)
    b = F.conv2d(F.pad(x, (1, 1, 1, 1)), w, stride=2)
    assert a.shape[-1] == S // 2 and not torch.allclose(a, b[..., : S // 2, : S // 2])


def test_guide_gradient_conditioning():
    """Why the energy-gradient parity tests compare at the SAME image (tests/test_engine_gpu.py::test_energy_gradient_at_the_same_image):
    the input-gradient of the ReLU / max-pool guide (encode_image, model_utils.py:29-41; differentiated by torch.autograd.grad at
    generate_data.py:721 / :761) is piecewise constant in the image.  In the fp32 oracle itself a relative perturbation of the image of
    1e-3 already moves it by several per cent and one bf16 rounding of the image by ~5 %, while the features move by < 0.1 %: a
    tolerance on this gradient across two forward paths that differ by a bf16 UNet (x0 within ~2 %) says nothing about the VJP code."""
    from distdiff_amd.config import tiny_config
    from distdiff_amd.weights import synthetic_weights
    cfg = tiny_config(max_batch=2)
    w = synthetic_weights(cfg, seed=0, num_classes=5)
    guide = O.GuideOracle(cfg, w["guide"])
    S = cfg.guide.input_size
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 3, S, S, generator=g) * 0.5
    gf = torch.randn(2, cfg.guide.feature_dim, generator=g)

    def vjp(xx):
        xr = xx.clone().requires_grad_(True)
        f = guide.encode_image(xr)
        (gr,) = torch.autograd.grad(f, xr, gf)
        return f.detach(), gr

    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    f0, g0 = vjp(x)
    n = torch.randn(x.shape, generator=g)
    f1, g1 = vjp(x + 1e-3 * n * x.norm() / n.norm())
    assert rel(f1, f0) < 1e-3 and rel(g1, g0) > 0.02
    f2, g2 = vjp(x + 1e-2 * n * x.norm() / n.norm())
    assert rel(f2, f0) < 1e-2 and rel(g2, g0) > 0.08
    f3, g3 = vjp(x.bfloat16().float())
    assert rel(g3, g0) > 0.02


def test_reduced_precision_floor_of_the_oracle_loop():
    """What the REFERENCE's own dtype costs: the reference runs UNet / VAE / guide in fp16 (generate_data.py:867, :1039), the engine in
    bf16 (the north_star's dtype: 3 fewer mantissa bits).  The oracle's whole loop (transform guidance P = 2, generate_data.py:1161-1234)
    at the tiny config with every weight AND activation rounded to fp16, and to bf16, against the fp32 run on the same inputs: the
    distance between two reduced-precision executions of the same loop is the floor any fp16 / bf16 engine sits on, and puts the
    engine's full-size numbers (tests/test_fullsize_loop_gpu.py: 44-47 dB, ~70 % of the uint8 bytes differ by one level) in context.
    Measured: fp16 latents 0.4 %, PSNR 59 dB, 18 % of the bytes differ (0.02 % by more than 2 levels); bf16 latents 3.9 %, PSNR 40 dB,
    66 % of the bytes differ (27 % by more than 2) -- the engine (bf16 MFMA inputs, fp32 accumulation, fp32 latents / guide / energy)
    is well inside the all-bf16 execution and between the two."""
    import math
    from distdiff_amd.config import tiny_config
    from distdiff_amd.weights import synthetic_weights
    cfg = tiny_config(max_batch=1)
    w = synthetic_weights(cfg, seed=0, num_classes=5)
    g = torch.Generator().manual_seed(11)
    L, D = cfg.latent_size, cfg.guide.feature_dim
    lat, noise = torch.randn(1, 4, L, L, generator=g) * 0.9, torch.randn(1, 4, L, L, generator=g)
    e, b = torch.rand(1, 4, 1, 1, generator=g), torch.randn(1, 4, 1, 1, generator=g)
    pos = torch.randn(1, cfg.text_len, cfg.unet.cross_attention_dim, generator=g)
    neg = torch.randn(1, cfg.text_len, cfg.unet.cross_attention_dim, generator=g)
    Pc = F.normalize(torch.randn(5, D, generator=g), dim=-1)
    Pg = F.normalize(torch.randn(5, 3, D, generator=g), dim=-1)
    args = O.SamplerArgs(guidance_type="transform_guidance", num_inference_steps=10, guidance_step=4, guidance_period=2, strength=0.5,
                         rho=10.0, constraint_value=0.2)

    def run(dt):
        ww = {k: {n: (t.to(dt) if t.is_floating_point() else t) for n, t in v.items()} for k, v in w.items()}
        return O.expand_one(args, cfg, O.build_models(cfg, ww), lat.to(dt), noise.to(dt), e.to(dt), b.to(dt), pos.to(dt), neg.to(dt),
                            torch.tensor([2]), Pc, Pg)

    z32, i32, s32 = run(torch.float32)
    u8 = lambda im: im.float().mul(255).add(0.5).clamp(0, 255).to(torch.uint8).int()
    out = {}
    for name, dt in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
        z, im, s = run(dt)
        du = (u8(im) - u8(i32)).abs()
        out[name] = (float((z.float() - z32).norm() / z32.norm()), 10 * math.log10(1.0 / float(((im.float() - i32) ** 2).mean())),
                     float((du > 0).float().mean()), float((du > 2).float().mean()), abs(float(s) - float(s32)) / abs(float(s32)))
        print("%s oracle loop vs fp32: latents %.4f, PSNR %.2f dB, u8 bytes differing %.3f (by > 2 levels %.4f), score rel %.2e" % ((name,) + out[name]))
    assert out["fp16"][0] < 0.01 and out["fp16"][1] > 52.0 and out["fp16"][4] < 1e-4
    assert out["bf16"][0] < 0.08 and out["bf16"][1] > 35.0 and out["bf16"][4] < 2e-3
    assert out["fp16"][0] < out["bf16"][0] and out["fp16"][2] < out["bf16"][2]


def test_gelu_polynomial():
    """The transcendental-free GELU of the GEGLU epilogues (distdiff_amd/csrc/common.h: gelu_poly_f): the literals are read from the
    header and evaluated here exactly as the kernel does (fp32 Horner with fused multiply-adds, clamp at +-DD_GELU_CLAMP) against
    x * Phi(x) in float64 -- diffusers' GEGLU is `hidden * F.gelu(gate)` with the erf form.  Stated accuracy:
    absolute <= 5e-5 everywhere, relative <= 1.2e-5 for x > 0, <= 1e-4 for x > -2, Phi~ stays inside [0, 1] to 2e-6."""
    import re
    import numpy as np
    from scipy.special import ndtr
    src = open(os.path.join(os.path.dirname(__file__), "..", "distdiff_amd", "csrc", "common.h")).read()
    clamp = np.float32(re.search(r"#define\s+DD_GELU_CLAMP\s+([0-9.]+)f", src).group(1))
    body = re.search(r"#define\s+DD_GELU_POLY\s*\{(.*?)\}", src, re.S).group(1).replace("\\", " ")
    c = np.array([float(v.strip().rstrip("f")) for v in body.split(",")], dtype=np.float32)
    assert len(c) == 10 and clamp == np.float32(4.5)
    x = np.linspace(-12, 12, 2_400_001).astype(np.float32)
    t = np.clip(x, -clamp, clamp)
    u = (t * t).astype(np.float32)
    p = np.full_like(u, c[9])
    for k in range(8, -1, -1):
        p = (p.astype(np.float64) * u + c[k]).astype(np.float32)              # one rounding per step: fma
    phi = (t.astype(np.float64) * p + 0.5).astype(np.float32)
    g = (x * phi).astype(np.float32)
    xd = x.astype(np.float64)
    ref = xd * ndtr(xd)
    err = np.abs(g - ref)
    rel = err / np.maximum(np.abs(ref), 1e-300)
    assert err.max() <= 5e-5, err.max()
    assert rel[x > 1e-3].max() <= 1.2e-5 and rel[(x > -2) & (np.abs(x) > 1e-3)].max() <= 1e-4
    assert phi.min() >= -2e-6 and phi.max() <= 1 + 2e-6
    assert np.all(phi[x >= clamp] == phi[x >= clamp][0]) and abs(float(phi[x >= clamp][0]) - 1.0) <= 2e-6      # exact ends behind the clamp
