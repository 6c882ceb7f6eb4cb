import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from distdiff_amd.config import tiny_config, sd15_config
from distdiff_amd.weights import synthetic_weights
from distdiff_amd.engine import Engine
from distdiff_amd.scheduler import DDIMSchedule

def rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-20)).item()

for name, cfg in [("tiny", tiny_config(max_batch=2)), ("sd15", sd15_config(max_batch=1))]:
    eng = Engine(cfg, synthetic_weights(cfg, 0, 5), enable_grad=True, max_guidance_period=1)
    s = DDIMSchedule(cfg.scheduler); ts = s.set_timesteps(10)
    eng.set_schedule(ts, s.alphas_cumprod, s.final_alpha_cumprod)
    g = torch.Generator().manual_seed(0)
    B, L = cfg.max_batch, cfg.latent_size
    eng.set_prompt(torch.randn(2 * B, cfg.text_len, cfg.unet.cross_attention_dim, generator=g).cuda())
    z = torch.randn(B, 4, L, L, generator=g)
    gg = torch.randn(2 * B, 4, L, L, generator=g)
    a1 = eng.unet_vjp(z, 5, gg).clone()
    a2 = eng.unet_vjp(z, 5, gg).clone()
    b = eng.unet_vjp(z, 5, -4.0 * gg).clone()
    a3 = eng.unet_vjp(z, 5, gg).clone()
    print(name, "unet repeat", rel(a2, a1), "after-other", rel(a3, a1), "lin", rel(b, -4 * a1))
    gim = torch.randn(B, 3, 8 * L, 8 * L, generator=g)
    x0 = torch.randn(B, 4, L, L, generator=g)
    v1 = eng.decode_vjp(x0, gim).clone(); v2 = eng.decode_vjp(x0, -4 * gim).clone(); v3 = eng.decode_vjp(x0, gim).clone()
    print(name, "vae after-other", rel(v3, v1), "lin", rel(v2, -4 * v1))
    S = cfg.guide.input_size
    gi = torch.randn(B, 3, S, S, generator=g); gf = torch.randn(B, cfg.guide.feature_dim, generator=g)
    w1 = eng.guide_vjp(gi, gf).clone(); w2 = eng.guide_vjp(gi, -4 * gf).clone(); w3 = eng.guide_vjp(gi, gf).clone()
    print(name, "guide after-other", rel(w3, w1), "lin", rel(w2, -4 * w1))
    eng.close()
