"""Generates tests/golden/fullsize_loop_w_fixture.pt -- the fp32 CPU oracle's WHOLE configs[1] loop (the script of record: 25 executed
steps, transform guidance P = 2 + re-step, final decode; generate_data.py:1161-1234, expand_diff.sh:3-15) at BASELINE.json's full sizes
for two MORE weight draws (row "a" inputs of make_fullsize_loop_fixture.py):

  s1     synthetic_weights(cfg, seed=1): an independent draw of every UNet / VAE / guide tensor
  qk14   seed 0 with attn1.to_q / attn1.to_k x sqrt(2) in EVERY transformer block: self-attention scores x 2 -- the moderately non-flat
         draw: peaky softmaxes through the whole network, and still well conditioned (at the tiny config the oracle's own loop run in
         fp16 stays within 1.2 % of its fp32 run, in bf16 within 4.4 %: tests/test_oracle.py)
  qk2    the same with x 2 (scores x 4): ILL-conditioned -- the oracle's own fp16 execution of the tiny loop is 21 % away from its
         fp32 run, bf16 27 % -- kept as the reported case of what any reduced-precision run of such a network looks like
The oracle's own conditioning is recorded with every draw: eps of the first executed step with the input latents rounded to bf16
once, relative to the unrounded run ("cond_eps_bf16_input").

~8 min per draw on 8 cores, ~40 GB peak:   python tests/golden/make_fullsize_loop_w_fixture.py [s1|qk14|qk2|all]
"""
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fullsize_loop_w_fixture.pt")


def draw(cfg, name):
    """The weight draw `name`, shared with tests/test_fullsize_loop_gpu.py."""
    from distdiff_amd.weights import synthetic_weights
    if name == "s1":
        return synthetic_weights(cfg, seed=1, num_classes=100)
    w = synthetic_weights(cfg, seed=0, num_classes=100)
    u = w["unet"]
    n = 0
    gain = {"qk2": 2.0, "qk14": 2.0 ** 0.5}[name]
    for k in list(u.keys()):
        if k.endswith("attn1.to_q.weight") or k.endswith("attn1.to_k.weight"):
            u[k] = u[k] * gain
            n += 1
    assert n == 32, n
    return w


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    from make_fullsize_fixture import inputs
    from make_fullsize_loop_fixture import loop_inputs, run_loop
    from distdiff_amd.config import sd15_config
    from oracle import sd_oracle as O
    torch.set_num_threads(os.cpu_count() or 8)
    cfg = sd15_config(latent_size=64, max_batch=1)
    proto = inputs(cfg)
    fx = torch.load(OUT, weights_only=False) if os.path.exists(OUT) else {}
    a1 = O.SamplerArgs(guidance_type="transform_guidance", num_inference_steps=50, guidance_step=20, guidance_period=2, strength=0.5,
                       rho=10.0, constraint_value=0.2)
    d = loop_inputs(cfg, "a")
    for name in ("s1", "qk14", "qk2"):
        if which not in (name, "all"):
            continue
        w = draw(cfg, name)
        models = O.build_models(cfg, w)
        unet, vae, guide, sched = models
        # conditioning of the oracle itself on this draw: one bf16 rounding of the first step's input
        ts = sched.set_timesteps(50)
        si = O.start_index(0.5, len(ts))
        with torch.no_grad():
            z = sched.add_noise(d["latents"], d["noise"], ts[si])
            emb = torch.cat([d["neg"], d["pos"]])
            e0 = unet(torch.cat([z, z]), int(ts[si]), emb)[0]
            e1 = unet(torch.cat([z, z]).bfloat16().float(), int(ts[si]), emb)[0]
        cond = float((e1 - e0).norm() / e0.norm())
        print("[%s] conditioning: eps moves %.4f under one bf16 rounding of the latents" % (name, cond), flush=True)
        r = run_loop(O, a1, cfg, models, d, d["t100"], proto["Pc100"], proto["Pg100"], "c1/" + name)
        r["cond_eps_bf16_input"] = cond
        r["weights_checksum"] = float(sum(v.double().sum() for v in w["unet"].values()))
        fx["c1_" + name] = r
        torch.save(fx, OUT)
        del models, unet, vae, guide, w
    print("wrote", OUT, "%.1f MB" % (os.path.getsize(OUT) / 1e6))


if __name__ == "__main__":
    main()
