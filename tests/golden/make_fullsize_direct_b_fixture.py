"""Generates tests/golden/fullsize_direct_b_fixture.pt -- the fp32 CPU oracle's direct guidance (generate_data.py:735-767, BASELINE
configs[3]: StanfordCars sizes C = 196, K = 3) for the SECOND full-size input row ("b" = make_fullsize_p2_fixture.inputs2, class 33), so
that tests/test_fullsize_batch_gpu.py::test_direct_guidance_every_row compares every batch row with an oracle run of its own (row "a"
is in fullsize_fixture.pt).  One guided step at t = 381 from z0 = z (1 + e) + b, differentiated with torch.autograd; the guide is
evaluated AT the fp16-rounded decoded image that fullsize_p2_fixture.pt already stores for this row and step (image_1; straight-through
hook of the oracle, sd_oracle._guide_features image_at), which the test hands to the engine as well.

About 2 min on 8 cores:   python tests/golden/make_fullsize_direct_b_fixture.py
"""
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "fullsize_direct_b_fixture.pt")
TARGET_B = 33


def main():
    from make_fullsize_fixture import inputs
    from make_fullsize_p2_fixture import STEP_INDEX, inputs2
    from distdiff_amd.config import sd15_config
    from distdiff_amd.weights import synthetic_weights
    from oracle import sd_oracle as O
    torch.set_num_threads(os.cpu_count() or 8)
    cfg = sd15_config(latent_size=64, max_batch=1)
    w = synthetic_weights(cfg, seed=0, num_classes=100)
    unet, vae, guide, sched = O.build_models(cfg, w)
    ts = sched.set_timesteps(50)
    t = int(ts[STEP_INDEX])
    proto, d = inputs(cfg), inputs2(cfg)
    fx2 = torch.load(os.path.join(HERE, "fullsize_p2_fixture.pt"), weights_only=False)
    args = O.SamplerArgs(guidance_type="direct_guidance", num_inference_steps=50, guidance_step=20, guidance_period=2, strength=0.5,
                         rho=10.0, constraint_value=0.2)
    T0 = time.time()
    z0 = d["z"] * (1 + d["e"]) + d["b"]
    z_next, x0, score, g = O.direct_guidance(args, z0, torch.tensor([TARGET_B]), t, sched, unet, torch.cat([d["neg"], d["pos"]]), vae, guide,
                                            proto["Pc196"], proto["Pg196"], cfg.guide.input_size, image_at=fx2["b"]["image_1"].float())
    assert (x0 - fx2["b"]["x0_1"]).abs().max() < 1e-4            # same forward point as the stored image
    fx = {"t": t, "target": TARGET_B, "direct_score": score.clone(), "direct_gz": g.clone(), "direct_z_next": z_next.clone(), "x0": x0.clone(),
          "weights_checksum": float(sum(v.double().sum() for v in w["unet"].values()))}
    torch.save(fx, OUT)
    print("direct guidance row b: score %.5f, %.0f s; wrote %s %.2f MB" % (float(score), time.time() - T0, OUT, os.path.getsize(OUT) / 1e6))


if __name__ == "__main__":
    main()
