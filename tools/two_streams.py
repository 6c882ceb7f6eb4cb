"""Experiment: two engines of half the batch on two HIP streams (two host threads) against one engine of the full batch.
    python tools/two_streams.py [B_total]"""
import os, sys, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from distdiff_amd.config import sd15_config
from distdiff_amd.engine import Engine
from distdiff_amd.scheduler import DDIMSchedule
from distdiff_amd.weights import synthetic_weights

BT = int(sys.argv[1]) if len(sys.argv) > 1 else 16
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = 3
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)


def make(B):
    cfg = sd15_config(max_batch=B)
    w = synthetic_weights(cfg, seed=0, num_classes=100)
    eng = Engine(cfg, w, enable_grad=True, max_guidance_period=2, device="cuda:0")
    sched = DDIMSchedule(cfg.scheduler)
    ts = sched.set_timesteps(50)
    eng.set_schedule(ts, sched.alphas_cumprod, sched.final_alpha_cumprod, guidance_scale=7.5, gs=1.0, ls=1.0, rho=10.0, constraint_value=0.2, guidance_period=2)
    g = torch.Generator().manual_seed(3)
    D = cfg.guide.feature_dim
    eng.set_prototypes(torch.nn.functional.normalize(torch.randn(100, D, generator=g), dim=-1),
                       torch.nn.functional.normalize(torch.randn(100, 3, D, generator=g), dim=-1))
    L = cfg.latent_size
    data = dict(lat=(torch.randn(B, 4, L, L, generator=g) * 0.9).to(dev), noise=torch.randn(B, 4, L, L, generator=g).to(dev),
                e=torch.rand(B, 4, generator=g).to(dev), b=torch.randn(B, 4, generator=g).to(dev),
                targets=torch.randint(0, 100, (B,), generator=g).to(dev))
    eng.set_prompt(torch.randn(2 * B, cfg.text_len, cfg.unet.cross_attention_dim, generator=g).to(dev))
    return eng, data, len(ts)


def worker(eng, d, n_ts, stream, n):
    with torch.cuda.stream(stream):
        for _ in range(n):
            eng.expand(d["lat"], d["noise"], d["e"], d["b"], d["targets"], int(0.5 * n_ts), "transform_guidance", n_ts - 20, 2, want_image=True)
        stream.synchronize()


def run(engines, n):
    streams = [torch.cuda.Stream() for _ in engines]
    th = [threading.Thread(target=worker, args=(e, d, nts, s, n)) for (e, d, nts), s in zip(engines, streams)]
    torch.cuda.synchronize()
    t0 = time.time()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    return time.time() - t0


one = [make(BT)]
run(one, 1)
dt = run(one, steps)
print("1 stream  x B=%d: %.2f images/s" % (BT, steps * BT / dt), flush=True)
one[0][0].close(); del one
torch.cuda.empty_cache()
many = [make(BT // NS) for _ in range(NS)]
run(many, 1)
dt = run(many, steps)
print("%d streams x B=%d: %.2f images/s" % (NS, BT // NS, steps * BT / dt), flush=True)
