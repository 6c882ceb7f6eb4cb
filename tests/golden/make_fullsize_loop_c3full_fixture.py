"""Generates tests/golden/fullsize_loop_c3full_fixture.pt -- BASELINE.json configs[3]'s schedule IN FULL: the fp32 CPU oracle's whole
expansion loop at SD-1.5 / AutoencoderKL / ResNet-50 widths, 512x512, C = 196, K = 3, strength 1.0 = all 50 steps of the 50-step DDIM
schedule executed (40 plain steps, then direct guidance on each of the last 10, generate_data.py:1210-1216), final decode, uint8 --
for the two input rows of make_fullsize_loop_fixture.py.  (fullsize_loop_fixture.pt holds the same schedule shortened to strength 0.3 =
15 executed steps; this one closes the gap between "15 of 50 steps" and the schedule as BASELINE states it.)

About 12 min per row on 8 cores, ~40 GB peak:   python tests/golden/make_fullsize_loop_c3full_fixture.py
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
OUT = os.path.join(HERE, "fullsize_loop_c3full_fixture.pt")


def main():
    from make_fullsize_fixture import inputs
    from make_fullsize_loop_fixture import loop_inputs, run_loop
    from distdiff_amd.config import sd15_config
    from distdiff_amd.weights import synthetic_weights
    from oracle import sd_oracle as O
    torch.set_num_threads(os.cpu_count() or 8)
    cfg = sd15_config(latent_size=64, max_batch=1)
    w = synthetic_weights(cfg, seed=0, num_classes=100)
    models = O.build_models(cfg, w)
    proto = inputs(cfg)
    fx = torch.load(OUT, weights_only=False) if os.path.exists(OUT) else {}
    fx["weights_checksum"] = float(sum(v.double().sum() for v in w["unet"].values()))
    fx["strength"] = 1.0
    a3 = O.SamplerArgs(guidance_type="direct_guidance", num_inference_steps=50, guidance_step=10, guidance_period=10, strength=1.0,
                       rho=10.0, constraint_value=0.2)
    for tag in ("a", "b"):
        if "c3full_" + tag in fx:
            continue
        d = loop_inputs(cfg, tag)
        r = run_loop(O, a3, cfg, models, d, d["t196"], proto["Pc196"], proto["Pg196"], "c3full/" + tag)
        r["traj"] = r["traj"][::5].clone()              # every fifth state is enough for the growth diagnostic
        # (image16 stays: the PSNR of tests/test_fullsize_loop_gpu.py::_compare is taken against it)
        fx["c3full_" + tag] = r
        torch.save(fx, OUT)
    print("wrote", OUT, "%.1f MB" % (os.path.getsize(OUT) / 1e6))


if __name__ == "__main__":
    main()
