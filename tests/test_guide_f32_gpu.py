"""The guide network in exact fp32 (distdiff_amd/csrc/guide_f32.hip) against plain torch fp32 on the CPU.

The energy gradient of the reference (torch.autograd.grad, generate_data.py:721 / :761) goes through the ReLU / max-pool masks of
image_encoder.encode_image (model_utils.py:29-41).  The input-gradient of such a network is piecewise constant in its input: in the
fp32 oracle itself a relative input perturbation of 1e-3 moves it by ~6 % and one bf16 rounding of the input by 5-8 %
(test_oracle_guide_gradient_conditioning, CPU).  So the guide forward, its masks and its VJP run in fp32 on
v_mfma_f32_32x32x2_f32 (an exact k-ordered fmaf chain): parity with the fp32 oracle is then a matter of summation order only.

Tolerances (fp32 vs fp32, different summation order): conv / dgrad outputs 2e-5 of max|ref|; guide features 1e-4 relative L2;
guide VJP 2e-3 relative L2 (a handful of activations sit within fp32 rounding of zero).
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    assert torch.isfinite(a).all()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def close(got, ref, tol, what):
    got, ref = got.float().cpu(), ref.float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), what
    err = (got - ref).abs().max().item()
    assert err <= tol * max(ref.abs().max().item(), 1e-6), "%s: max err %.3g (ref max %.3g)" % (what, err, ref.abs().max().item())


def rows(x):      # NCHW -> NHWC rows [B*H*W, ld] with ld = roundup(C, 4)
    B, C, H, W = x.shape
    ld = (C + 3) // 4 * 4
    r = torch.zeros(B * H * W, ld)
    r[:, :C] = x.permute(0, 2, 3, 1).reshape(-1, C)
    return r.cuda()


def unrows(r, B, C, H, W):
    return r[:, :C].reshape(B, H, W, C).permute(0, 3, 1, 2)


CASES = [
    # name, B, Cin, Cout, H, W, k, stride, pad, groups
    ("stem7x7", 2, 3, 64, 32, 32, 7, 2, 3, 1),
    ("stem7x7_narrow_dgrad", 2, 3, 64, 96, 96, 7, 2, 3, 1),     # M = 18432 image pixels x 3 channels: the one-thread-per-pixel kernel
    ("1x1", 2, 64, 256, 14, 14, 1, 1, 0, 1),
    ("1x1_ragged", 1, 20, 36, 7, 5, 1, 1, 0, 1),
    ("3x3", 2, 64, 64, 14, 14, 3, 1, 1, 1),
    ("3x3_stride2", 2, 128, 128, 14, 14, 3, 2, 1, 1),
    ("1x1_stride2", 2, 256, 512, 14, 14, 1, 2, 0, 1),
    ("3x3_groups32", 2, 128, 128, 14, 14, 3, 1, 1, 32),
    ("3x3_groups32_stride2", 1, 256, 256, 14, 14, 3, 2, 1, 32),
    ("3x3_groups4_wide", 1, 64, 192, 9, 9, 3, 1, 1, 4),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_f32_forward_and_dgrad(hip_lib, case):
    from distdiff_amd import ops as O
    name, B, Cin, Cout, H, W, k, stride, pad, groups = case
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin // groups, k, k, generator=g) / (Cin // groups * k * k) ** 0.5
    bias = torch.randn(Cout, generator=g)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = torch.randn(B, Cout, Ho, Wo, generator=g)
    xr = x.clone().requires_grad_(True)
    ref = F.relu(F.conv2d(xr, w, bias, stride=stride, padding=pad, groups=groups) + res)
    pk = O.PackedConvF32(w, pad, mode=0, groups=groups, bias=bias)
    y = O.conv_f32(rows(x), pk, B, H, W, Ho, Wo, stride=stride, res=rows(res), relu=True)
    close(unrows(y, B, Cout, Ho, Wo), ref.detach(), 2e-5, name + " fwd")
    # input-gradient: ReLU mask from the forward output, then the transposed (+ flipped, + dilated for stride 2) GEMM
    gy = torch.randn(B, Cout, Ho, Wo, generator=g)
    (gx_ref,) = torch.autograd.grad(ref, xr, gy)
    gm = gy * (ref.detach() > 0)
    pkb = O.PackedConvF32(w, pad, mode=1, groups=groups)
    acc = torch.randn(B, Cin, H, W, generator=g)
    gx = O.conv_f32(rows(gm), pkb, B, Ho, Wo, H, W, stride=1, shift=1 if stride == 2 else 0, parity=1 if stride == 2 else 0, res=rows(acc))
    close(unrows(gx, B, Cin, H, W), gx_ref + acc, 2e-5, name + " dgrad")


def _engine(cfg, w, B):
    from distdiff_amd.engine import Engine
    from distdiff_amd.scheduler import DDIMSchedule
    eng = Engine(cfg, w, enable_grad=True, max_guidance_period=1)
    sched = DDIMSchedule(cfg.scheduler)
    eng.set_schedule(sched.set_timesteps(10), sched.alphas_cumprod, sched.final_alpha_cumprod)
    return eng


@pytest.mark.parametrize("which", ["tiny", "tiny_grouped", "tiny_mbv2", "resnet50", "resnext50", "wideresnet50", "mobilenetv2"])
def test_guide_forward_and_vjp_vs_oracle(hip_lib, which):
    """encode_image and its input-gradient, through the production C ABI, at the tiny widths and at the real widths of the three
    fp32 guides of the reference at 224x224: resnet50 (64-256-512-1024-2048), resnext50 = resnext50_32x4d (32 groups in
    every 3x3), wideresnet50 = wide_resnet50_2 and mobilenetv2 = mobilenetv2_100 (depthwise 3x3, ReLU6) (model_utils.py:47-79)."""
    from distdiff_amd.config import GuideConfig, guide_config, sd15_config, tiny_config
    from distdiff_amd.weights import synthetic_weights
    from oracle import sd_oracle as O
    B = 2
    if which == "tiny":
        cfg = tiny_config(max_batch=B)
    elif which == "tiny_grouped":
        cfg = tiny_config(max_batch=B)
        cfg.guide = GuideConfig(stem_channels=16, planes=(16, 32, 32, 64), blocks=(1, 2, 1, 1), input_size=56, cardinality=4, base_width=16)
    elif which == "tiny_mbv2":
        cfg = tiny_config(max_batch=B)
        cfg.guide = GuideConfig(kind="mbv2", input_size=64, mb_stem=16, mb_channels=(16, 16, 32), mb_repeats=(1, 2, 2), mb_strides=(1, 2, 2),
                                mb_expand=2, mb_head=64)
    else:
        cfg = sd15_config(latent_size=8, max_batch=B)     # SD widths at an 8x8 latent keep the UNet/VAE slabs small; the guide is full size
        cfg.guide = guide_config(which)
    w = synthetic_weights(cfg, seed=0, num_classes=5)
    eng = _engine(cfg, w, B)
    guide = O.build_models(cfg, w)[2]
    S = cfg.guide.input_size
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, 3, S, S, generator=g) * 0.5
    gf = torch.randn(B, cfg.guide.feature_dim, generator=g)
    xr = x.clone().requires_grad_(True)
    f_ref = guide.encode_image(xr)
    (g_ref,) = torch.autograd.grad(f_ref, xr, gf)
    assert rel(eng.guide_encode(x), f_ref.detach()) < 1e-4
    with torch.no_grad():                                       # encode_image(x, pooling='max'), model_utils.py:34-35
        assert rel(eng.guide_encode(x, pooling="max"), guide.encode_image(x, pooling="max")) < 1e-4
    # fp32 vs fp32: summation order only, plus a handful of activations within fp32 rounding of a ReLU / ReLU6 kink (measured 4e-7
    # on the ResNets, 2.7e-3 on the 53-layer two-sided-mask MobileNetV2)
    assert rel(eng.guide_vjp(x, gf), g_ref) < (5e-3 if cfg.guide.kind == "mbv2" else 2e-3)
    eng.close()


def test_bicubic_maxpool_gap_f32(hip_lib):
    """fp32 side kernels through the engine: decode -> bicubic -> guide features equals the oracle evaluated on the engine's OWN
    decoded image (fp32 hand-off, no bf16 rounding between the decoder and the guide's ReLU masks)."""
    from distdiff_amd.config import tiny_config
    from distdiff_amd.weights import synthetic_weights
    from oracle import sd_oracle as O
    B = 2
    cfg = tiny_config(max_batch=B)
    w = synthetic_weights(cfg, seed=0, num_classes=5)
    eng = _engine(cfg, w, B)
    guide = O.GuideOracle(cfg, w["guide"])
    g = torch.Generator().manual_seed(3)
    x0 = torch.randn(B, 4, cfg.latent_size, cfg.latent_size, generator=g) * 0.2
    img = eng.decode(x0, denormalize=False).cpu()
    gi = F.interpolate(img, size=(cfg.guide.input_size,) * 2, mode="bicubic")
    assert rel(eng.guide_encode(gi), guide.encode_image(gi)) < 1e-4
    eng.close()


@pytest.mark.parametrize("which", ["tiny_vit", "open_clip_vit_b32"])
def test_clip_vit_guide_forward_and_vjp_vs_oracle(hip_lib, which):
    """The reference's default guide, `--arch open_clip_vit_b32` (model_utils.py:80-87): encode_image = the open_clip image tower.  No ReLU
    masks: a bf16 program like the UNet (LayerNorm / fused-QKV attention / GELU MLP kernels), tolerance 3 % forward, 5 % VJP."""
    from distdiff_amd.config import GuideConfig, guide_config, sd15_config, tiny_config
    from distdiff_amd.weights import synthetic_weights
    from oracle import sd_oracle as O
    B = 2
    if which == "tiny_vit":
        cfg = tiny_config(max_batch=B)
        cfg.guide = GuideConfig(kind="vit", input_size=64, vit_width=64, vit_layers=2, vit_heads=2, vit_patch=16, vit_mlp=128, vit_out=32)
    else:
        cfg = sd15_config(latent_size=8, max_batch=B)
        cfg.guide = guide_config(which)
    w = synthetic_weights(cfg, seed=0, num_classes=5)
    eng = _engine(cfg, w, B)
    guide = O.GuideOracleViT(cfg, w["guide"])
    S = cfg.guide.input_size
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, 3, S, S, generator=g) * 0.5
    gf = torch.randn(B, cfg.guide.feature_dim, generator=g)
    xr = x.clone().requires_grad_(True)
    f_ref = guide.encode_image(xr)
    (g_ref,) = torch.autograd.grad(f_ref, xr, gf)
    assert rel(eng.guide_encode(x), f_ref.detach()) < 0.03
    assert rel(eng.guide_vjp(x, gf), g_ref) < 0.05
    eng.close()


def test_clip_vit_guided_step_vs_oracle(hip_lib):
    """transform_guidance through the ViT guide on the tiny UNet / VAE: score and (e, b) gradients against the oracle (the ViT has no
    masks, so the comparison at the oracle's own forward point is meaningful: <= 8 %)."""
    from distdiff_amd.config import GuideConfig, tiny_config
    from distdiff_amd.engine import Engine
    from distdiff_amd.scheduler import DDIMSchedule
    from distdiff_amd.weights import synthetic_weights
    from oracle import sd_oracle as O
    B = 2
    cfg = tiny_config(max_batch=B)
    cfg.guide = GuideConfig(kind="vit", input_size=64, vit_width=64, vit_layers=2, vit_heads=2, vit_patch=16, vit_mlp=128, vit_out=32)
    w = synthetic_weights(cfg, seed=0, num_classes=5)
    eng = Engine(cfg, w, enable_grad=True, max_guidance_period=2)
    sched = DDIMSchedule(cfg.scheduler)
    ts = sched.set_timesteps(10)
    eng.set_schedule(ts, sched.alphas_cumprod, sched.final_alpha_cumprod, guidance_period=2)
    g = torch.Generator().manual_seed(2)
    L, D = cfg.latent_size, cfg.guide.feature_dim
    z = torch.randn(B, 4, L, L, generator=g)
    e, b = torch.rand(B, 4, 1, 1, generator=g), torch.randn(B, 4, 1, 1, generator=g) * 0.3
    pe = torch.randn(B, cfg.text_len, cfg.unet.cross_attention_dim, generator=g)
    ne = torch.randn(1, cfg.text_len, cfg.unet.cross_attention_dim, generator=g).expand(B, -1, -1)
    Pc = F.normalize(torch.randn(5, D, generator=g), dim=-1)
    Pg = F.normalize(torch.randn(5, 3, D, generator=g), dim=-1)
    tg = torch.tensor([1, 3])
    eng.set_prototypes(Pc, Pg)
    eng.set_prompt(torch.cat([ne, pe]).cuda())
    zn, score, gz0 = eng.transform_guidance(z, tg, e, b, 5, 2)
    unet, vae, guide, so = O.build_models(cfg, w)
    tso = so.set_timesteps(10)
    args = O.SamplerArgs(guidance_type="transform_guidance", num_inference_steps=10, guidance_step=5, guidance_period=2)
    zr, sr, (ge, gb) = O.transform_guidance(args, z, tg, [int(tso[5]), int(tso[6])], so, unet, torch.cat([ne, pe]), vae, guide, e, b, Pc, Pg,
                                            cfg.guide.input_size)
    assert abs(score.item() - float(sr)) < 0.01 * abs(float(sr))
    ge_h = (gz0.cpu() * z).sum((2, 3), keepdim=True)
    gb_h = gz0.cpu().sum((2, 3), keepdim=True)
    assert rel(ge_h, ge) < 0.08 and rel(gb_h, gb) < 0.08, (rel(ge_h, ge), rel(gb_h, gb))
    assert rel(zn, zr) < 0.05
    eng.close()
