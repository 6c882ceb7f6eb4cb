"""Measurement of the stage before the loop on one MI355X: images/s of dd_vae_encode at 512x512 and
prompts/s of dd_text_encode (CLIP ViT-L/14 text tower, 77 tokens), HIP-event timed with inputs resident in HBM, with the
fp32 torch-CPU oracle timed beside it on a bounded sample.

    python tools/bench_encoders.py [--batch 16] [--iters 5]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--no_cpu", action="store_true")
    a = ap.parse_args()
    from distdiff_amd.config import sd15_config
    from distdiff_amd.engine import Engine
    from distdiff_amd.weights import synthetic_weights
    cfg = sd15_config(64, a.batch)
    w = synthetic_weights(cfg, seed=0, num_classes=10, encoders=True)
    eng = Engine(cfg, w, enable_grad=False, max_guidance_period=1)
    g = torch.Generator().manual_seed(0)
    x = (torch.rand(a.batch, 3, 512, 512, generator=g) * 2 - 1).cuda()
    n = torch.randn(a.batch, 4, 64, 64, generator=g).cuda()
    ids = torch.randint(0, cfg.text.vocab_size, (2 * a.batch, 77), generator=g).int().cuda()

    def timed(fn):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.iters

    eng.L.dd_flops_last(eng._h)
    ms_v = timed(lambda: eng.vae_encode(x, n))
    fl_v = eng.L.dd_flops_last(eng._h) / (a.iters + 1)
    ms_t = timed(lambda: eng.text_encode(ids))
    fl_t = eng.L.dd_flops_last(eng._h) / (a.iters + 1)
    out = {"vae_encode": {"images_per_s": a.batch / ms_v * 1e3, "ms": ms_v, "batch": a.batch, "tflop_per_image": fl_v / a.batch / 1e12,
                          "tflops": fl_v / ms_v / 1e9},
           "text_encode": {"prompts_per_s": 2 * a.batch / ms_t * 1e3, "ms": ms_t, "prompts": 2 * a.batch, "gflop_per_prompt": fl_t / (2 * a.batch) / 1e9,
                           "tflops": fl_t / ms_t / 1e9}}
    eng.close()
    if not a.no_cpu:
        from oracle import sd_oracle as O
        torch.set_num_threads(min(os.cpu_count() or 1, 64))
        xs = x[:1, :, :256, :256].cpu()
        with torch.no_grad():
            O.clip_text_encode(cfg, w["text"], ids[:2].cpu())
            t0 = time.perf_counter(); O.clip_text_encode(cfg, w["text"], ids[:8].cpu()); tt = time.perf_counter() - t0
            t0 = time.perf_counter(); O.vae_encode(cfg, w["vae"], xs, None); tv = time.perf_counter() - t0
        out["cpu_baseline"] = {"kind": "port", "cores": torch.get_num_threads(),
                               "vae_encode_images_per_s": 1.0 / (tv * 4.0), "vae_sample": "one 256x256 image, scaled x4 to 512x512",
                               "text_prompts_per_s": 8.0 / tt, "text_sample": "8 prompts"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
