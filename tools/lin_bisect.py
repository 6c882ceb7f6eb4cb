import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from distdiff_amd.config import tiny_config
from distdiff_amd.weights import synthetic_weights
from distdiff_amd.engine import Engine
from distdiff_amd.scheduler import DDIMSchedule
def rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item()
cfg = tiny_config(max_batch=2)
eng = Engine(cfg, synthetic_weights(cfg, 0, 5), enable_grad=True, max_guidance_period=1)
s = DDIMSchedule(cfg.scheduler); ts = s.set_timesteps(10)
eng.set_schedule(ts, s.alphas_cumprod, s.final_alpha_cumprod)
g = torch.Generator().manual_seed(0)
B, L = 2, cfg.latent_size
eng.set_prompt(torch.randn(2 * B, cfg.text_len, cfg.unet.cross_attention_dim, generator=g).cuda())
prog = int(sys.argv[1]) if len(sys.argv) > 1 else 1
if prog == 1:
    x0 = torch.randn(B, 4, L, L, generator=g); gim = torch.randn(B, 3, 8 * L, 8 * L, generator=g)
    run = lambda sc: eng.decode_vjp(x0, sc * gim)
else:
    z = torch.randn(B, 4, L, L, generator=g); gg = torch.randn(2 * B, 4, L, L, generator=g)
    run = lambda sc: eng.unet_vjp(z, 5, sc * gg)
n = eng.debug_num_tensors(prog)
run(1.0); torch.cuda.synchronize()
g1 = [eng.debug_tensor(prog, i, grad=True) for i in range(n)]
run(-4.0); torch.cuda.synchronize()
g2 = [eng.debug_tensor(prog, i, grad=True) for i in range(n)]
for i in range(n - 1, -1, -1):
    r = rel(g2[i], -4 * g1[i])
    print(i, tuple(g1[i].shape), "%.3e" % r, "norm %.3e" % g1[i].norm().item())
