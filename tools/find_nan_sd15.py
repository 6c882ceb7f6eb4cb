"""Diagnostic: one plain denoise step of the SD-1.5-shaped engine (batch from argv, default 4: M = 32768 at the 64x64 level) and the first
UNet tensors that hold non-finite values.  python tools/find_nan_sd15.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from distdiff_amd.config import sd15_config
from distdiff_amd.engine import Engine
from distdiff_amd.scheduler import DDIMSchedule
from distdiff_amd.weights import synthetic_weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = sd15_config(latent_size=64, max_batch=B)
w = synthetic_weights(cfg, seed=0, num_classes=100)
eng = Engine(cfg, w, enable_grad=False)
sched = DDIMSchedule(cfg.scheduler)
ts = sched.set_timesteps(50)
eng.set_schedule(ts, sched.alphas_cumprod, sched.final_alpha_cumprod)
g = torch.Generator().manual_seed(0)
eng.set_prompt(torch.randn(2 * B, cfg.text_len, cfg.unet.cross_attention_dim, generator=g).cuda())
z = torch.randn(B, 4, cfg.latent_size, cfg.latent_size, generator=g)
eps = eng.unet_forward(z, 3)
print("mask", os.environ.get("DD_GEMM_WS_MASK"), "eps finite:", bool(torch.isfinite(eps).all()), "abs mean %.4f" % float(eps[torch.isfinite(eps)].abs().mean()))
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
        if bad > 4:
            break
eng.close()
