"""Diagnostic: run one plain denoise step of the tiny engine and list the UNet tensors that hold non-finite values (first ones in
tensor order), with their shapes.  python tools/find_nan.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from distdiff_amd.config import tiny_config
from distdiff_amd.engine import Engine
from distdiff_amd.scheduler import DDIMSchedule
from distdiff_amd.weights import synthetic_weights

cfg = tiny_config(max_batch=2)
w = synthetic_weights(cfg, seed=0, num_classes=5)
eng = Engine(cfg, w, enable_grad=True, max_guidance_period=2)
sched = DDIMSchedule(cfg.scheduler)
ts = sched.set_timesteps(10)
eng.set_schedule(ts, sched.alphas_cumprod, sched.final_alpha_cumprod)
g = torch.Generator().manual_seed(0)
eng.set_prompt(torch.randn(4, cfg.text_len, cfg.unet.cross_attention_dim, generator=g).cuda())
z = torch.randn(2, 4, cfg.latent_size, cfg.latent_size, generator=g)
eps = eng.unet_forward(z, 3)
print("eps finite:", bool(torch.isfinite(eps).all()))
n = eng.debug_num_tensors(0)
bad = 0
for i in range(n):
    try:
        t = eng.debug_tensor(0, i)
    except RuntimeError as e:
        continue
    f = torch.isfinite(t)
    if not f.all():
        rows = (~f).any(1).nonzero().flatten()
        cols = (~f).any(0).nonzero().flatten()
        print("tensor %d shape %s: %d non-finite; rows %s.. cols %s.." % (i, tuple(t.shape), int((~f).sum()), rows[:6].tolist(), cols[:6].tolist()))
        bad += 1
        if bad > 12:
            break
eng.close()
