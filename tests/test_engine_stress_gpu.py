"""Engine-level runs with NON-FLAT statistics.  Every other engine-level parity input uses fan_in^-1/2 synthetic weights: attention
logits are then near-uniform and GroupNorm inputs zero-mean, so the peaky / far-from-zero code paths -- the lazy-reference attention
forward on scores with a standard deviation of ~16 (4096-key softmax close to one-hot), the (mean, M2) GroupNorm partials with
|mean| >> std (`CF_STATS`: sq - sa * mean) -- were only exercised by op-level tests.  Here the WHOLE UNet / guided step runs on stressed
weights against the fp32 CPU oracle on the same weights:

  * q/k x4     `attn1.to_q.weight`, `attn1.to_k.weight` x 4: self-attention scores x 16 (standard deviation ~16: a 4096-key softmax
               close to one-hot).  In the FIRST transformer block only ("first"): its inputs are still nearly exact, so the comparison
               with fp32 measures the kernel.  In EVERY block ("all") the network itself is chaotic -- an upstream relative error r moves
               a score by ~16 r, i.e. the 1-2 % of any bf16 (or fp16: the reference's dtype) forward re-ranks the top keys of the later
               blocks, and eps differs from the fp32 run by 40-70 % whatever the kernels do; that case is only required to stay finite
               (no overflow in the lazy softmax, whose reference is the first key tile's maximum) and is reported.
  * GN shift   `conv1.bias` of every ResnetBlock2D + s x (+-1 per GroupNorm group): the input of norm2 sits s standard deviations from
               zero (s = 10, 50).  bf16 STORAGE of such a tensor has a quantum of |mean| / 256, i.e. a noise floor of s / 256 / sqrt(12)
               standard deviations per element (1 % at s = 10, 6 % at s = 50) that no kernel can avoid; the bounds below are that floor's
               effect on eps (measured), an order of magnitude below what a wrong variance (e.g. a cancelled sq - sa * mean) produces.

Tolerances are relative L2 of eps vs the fp32 oracle and are stated per case (measured value in the comment, bound ~1.5x).
"""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    assert torch.isfinite(a).all()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def stress(w, qk_gain=1.0, gn_shift=0.0, groups=32, where="first"):
    u = w["unet"]
    n_q = n_b = 0
    for k in list(u.keys()):
        if qk_gain != 1.0 and (k.endswith("attn1.to_q.weight") or k.endswith("attn1.to_k.weight")) and \
                (where == "all" or k.startswith("down_blocks.0.attentions.0.")):
            u[k] = u[k] * qk_gain
            n_q += 1
        if gn_shift and ".resnets." in k and k.endswith("conv1.bias"):
            C = u[k].numel()
            u[k] = u[k] + gn_shift * (torch.arange(C) // (C // groups) % 2 * 2 - 1).float()
            n_b += 1
    assert (qk_gain == 1.0 or n_q > 0) and (not gn_shift or n_b > 0)
    return w


# (qk gain, where, GroupNorm shift, bound on eps rel-L2 or None = finite only): measured on MI355X, see the print of each case
# measured: 0.0135 | 0.0184 | 0.389 (the oracle itself moves 0.286 under one bf16 rounding of z) | 0.0276 | 0.0976 | 0.0544
TINY_CASES = [(1.0, "first", 0.0, 0.03), (4.0, "first", 0.0, 0.04), (4.0, "all", 0.0, None), (1.0, "first", 10.0, 0.045), (1.0, "first", 50.0, 0.15),
              (4.0, "first", 10.0, 0.08)]


@pytest.mark.parametrize("case", TINY_CASES, ids=lambda c: "qk%g%s_gn%g" % (c[0], c[1], c[2]))
def test_tiny_unet_and_guided_step_on_stressed_weights(hip_lib, case):
    from distdiff_amd.config import tiny_config
    from distdiff_amd.engine import Engine
    from distdiff_amd.scheduler import DDIMSchedule
    from distdiff_amd.weights import synthetic_weights
    from oracle import sd_oracle as O
    gain, where, shift, bound = case
    cfg = tiny_config(max_batch=2)
    w = stress(synthetic_weights(cfg, seed=0, num_classes=5), gain, shift, groups=cfg.unet.norm_num_groups, where=where)
    eng = Engine(cfg, w, enable_grad=True, max_guidance_period=2)
    sched = DDIMSchedule(cfg.scheduler)
    ts = sched.set_timesteps(10)
    eng.set_schedule(ts, sched.alphas_cumprod, sched.final_alpha_cumprod, guidance_period=1)
    g = torch.Generator().manual_seed(5)
    L, D = cfg.latent_size, cfg.guide.feature_dim
    z = torch.randn(2, 4, L, L, generator=g)
    emb = torch.randn(4, cfg.text_len, cfg.unet.cross_attention_dim, generator=g)
    eng.set_prompt(emb.cuda())
    unet, vae, guide, osched = O.build_models(cfg, w)
    osched.set_timesteps(10)
    with torch.no_grad():
        ref = unet(torch.cat([z, z]), ts[4], emb)[0]
    eps = eng.unet_forward(z, 4)
    err = rel(eps, ref)
    if bound is None:
        # the conditioning of the stressed network, measured on the fp32 oracle itself: one bf16 rounding of the INPUT latents (a relative
        # perturbation of 2^-9) moves its eps by `cond` -- when that is already tens of times the perturbation, a comparison of any
        # reduced-precision forward with the fp32 run says nothing about kernels
        with torch.no_grad():
            zq = z.to(torch.bfloat16).float()
            cond = rel(unet(torch.cat([zq, zq]), ts[4], emb)[0], ref)
        print("tiny qk x%g (all): the fp32 oracle's eps moves by %.4f under one bf16 rounding of z (input perturbation %.5f)" % (gain, cond, rel(zq, z)))
        assert cond > 10 * rel(zq, z)
    # one guided step on the same weights: the reverse programs read the stressed statistics too (GroupNorm / attention backward)
    Pc = torch.nn.functional.normalize(torch.randn(5, D, generator=g), dim=-1)
    Pg = torch.nn.functional.normalize(torch.randn(5, 3, D, generator=g), dim=-1)
    eng.set_prototypes(Pc, Pg)
    eng.set_sample_weights([1.0, 1.0])
    e, b = torch.rand(2, 4, 1, 1, generator=g), torch.randn(2, 4, 1, 1, generator=g) * 0.3
    tg = torch.tensor([1, 3])
    args = O.SamplerArgs(guidance_type="transform_guidance", num_inference_steps=10, guidance_step=6, guidance_period=1)
    zs, scs = [], []
    for i in range(2):       # the oracle's energy is a mean over its batch: one image at a time = sample weights 1 (generate_data.py:709)
        zo, so, _ = O.transform_guidance(args, z[i:i + 1], tg[i:i + 1], [ts[4]], osched, unet, torch.cat([emb[i:i + 1], emb[2 + i:3 + i]]), vae, guide,
                                         e[i:i + 1], b[i:i + 1], Pc, Pg, cfg.guide.input_size)
        zs.append(zo)
        scs.append(float(so))
    zg, _, _ = eng.transform_guidance(z, tg, e, b, 4, 1)
    sc = eng.image_scores().cpu()
    zerr = rel(zg, torch.cat(zs))
    serr = max(abs(float(sc[i]) - scs[i]) / abs(scs[i]) for i in range(2))
    print("tiny qk x%g (%s), GroupNorm shift %g: eps rel-L2 %.4f (bound %s) | guided latents %.4f, score rel %.5f" % (gain, where, shift, err, bound, zerr, serr))
    eng.close()
    if bound is None:
        return               # finite (rel() asserts it) and reported
    assert err < bound, err
    assert zerr < max(3 * bound, 0.10) and serr < 0.05, (zerr, serr)


# SD-1.5 widths at 512x512 (4096-token self-attention at d = 40, 8 heads): one UNet forward, B = 1 (CFG batch 2)
# measured: 0.0227 | 0.729 | 0.0284
FULL_CASES = [(4.0, "first", 0.0, 0.04), (4.0, "all", 0.0, None), (1.0, "first", 10.0, 0.045)]


@pytest.mark.parametrize("case", FULL_CASES, ids=lambda c: "qk%g%s_gn%g" % (c[0], c[1], c[2]))
def test_fullsize_unet_forward_on_stressed_weights(hip_lib, case):
    from distdiff_amd.config import sd15_config
    from distdiff_amd.engine import Engine
    from distdiff_amd.scheduler import DDIMSchedule
    from distdiff_amd.weights import synthetic_weights
    from oracle import sd_oracle as O
    gain, where, shift, bound = case
    torch.set_num_threads(min(os.cpu_count() or 8, 64))
    cfg = sd15_config(latent_size=64, max_batch=1)
    w = stress(synthetic_weights(cfg, seed=0, num_classes=100), gain, shift, where=where)
    eng = Engine(cfg, w, enable_grad=False, max_guidance_period=1)
    sched = DDIMSchedule(cfg.scheduler)
    ts = sched.set_timesteps(50)
    eng.set_schedule(ts, sched.alphas_cumprod, sched.final_alpha_cumprod)
    g = torch.Generator().manual_seed(6)
    z = torch.randn(1, 4, 64, 64, generator=g)
    emb = torch.randn(2, cfg.text_len, cfg.unet.cross_attention_dim, generator=g)
    eng.set_prompt(emb.cuda())
    eps = eng.unet_forward(z, 30)
    unet = O.UNetOracle(cfg, w["unet"])
    with torch.no_grad():
        ref = unet(torch.cat([z, z]), ts[30], emb)[0]
    err = rel(eps, ref)
    print("SD-1.5 widths, qk x%g (%s), GroupNorm shift %g: eps rel-L2 %.4f (bound %s)" % (gain, where, shift, err, bound))
    eng.close()
    assert bound is None or err < bound, err
