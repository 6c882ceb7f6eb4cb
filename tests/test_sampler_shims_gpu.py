"""The function-level drop-in (distdiff_amd/sampler.py): `denoise_one_step`, `transform_guidance`,
`direct_guidance` called with the REFERENCE's argument lists (generate_data.py:109, :687-689, :735-737) against
tests/golden/tiny_fixture.pt -- the outputs of the reference's own three functions (tests/golden/make_fixtures.py executes them
unchanged).  The CPU global RNG is seeded exactly as make_fixtures.py does, so the shim's `torch.rand` / `.normal_` draws for
(e, b) (:692-695) are the fixture's draw for draw.

Tolerances: tests/test_engine_gpu.py (bf16 UNet / VAE vs the fp32 reference run): z_prev 3 %, x0 4 %, transform latents 7 %,
direct latents 3 %, scores 0.5 %.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
FIX = os.path.join(os.path.dirname(__file__), "golden", "tiny_fixture.pt")


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    assert torch.isfinite(a).all()
    return ((a - b).norm() / (b.norm() + 1e-20)).item()


@pytest.fixture(scope="module")
def world(hip_lib):
    from distdiff_amd.config import tiny_config
    from distdiff_amd.engine import Engine
    from distdiff_amd.scheduler import DDIMSchedule
    from distdiff_amd.weights import synthetic_weights
    fx = torch.load(FIX, weights_only=False)
    cfg = tiny_config(max_batch=2)
    eng = Engine(cfg, synthetic_weights(cfg, seed=0, num_classes=5), enable_grad=True, max_guidance_period=2)
    sched = DDIMSchedule(cfg.scheduler)
    ts = sched.set_timesteps(fx["n_steps"])
    a = fx["args"]
    eng.set_schedule(ts, sched.alphas_cumprod, sched.final_alpha_cumprod, guidance_scale=a["guidance_scale"], gs=a["gs"], ls=a["ls"],
                     rho=a["rho"], constraint_value=a["constraint_value"], guidance_period=a["guidance_period"])
    eng.set_prototypes(fx["Pc"], fx["Pg"])
    yield eng, fx, sched
    eng.close()


def test_denoise_one_step_shim(world):
    from distdiff_amd import sampler
    eng, fx, sched = world
    embeds = torch.cat([fx["negative_embeds"], fx["prompt_embeds"]])            # generate_data.py:1184
    # reference call sites :1207 / :1218: denoise_one_step(latents, noise_scheduler, t, unet, prompt_embeds, class_labels)
    latents, x_0 = sampler.denoise_one_step(fx["z"], sched, int(fx["guide_timesteps"][0]), eng, embeds, None)
    assert rel(latents, fx["ref_denoise_z_prev"]) < 0.03
    assert rel(x_0, fx["ref_denoise_x0"]) < 0.04


def test_transform_guidance_shim_draws_e_b_like_the_reference(world):
    from distdiff_amd import sampler
    eng, fx, sched = world
    embeds = torch.cat([fx["negative_embeds"], fx["prompt_embeds"]])
    batch = {"targets": fx["targets"]}
    torch.manual_seed(1234)                                                     # make_fixtures.py: seed of the reference run
    # reference call site :1204-1206
    latents, score = sampler.transform_guidance(fx["z"].clone(), batch, fx["guide_timesteps"], sched, eng, embeds, None, None, None, None,
                                                torch.float32, None, fx["Pc"], fx["Pg"])
    assert abs(float(score) - float(fx["ref_transform_score"])) < 0.005 * abs(float(fx["ref_transform_score"]))
    assert rel(latents, fx["ref_transform_z"]) < 0.07
    assert float((latents.cpu() - fx["z"]).abs().max()) <= fx["args"]["constraint_value"] + 1e-5
    # the draws were the fixture's: the same call through the engine with the fixture's (e, b) gives the same latents bit for bit
    first = fx["timesteps"].tolist().index(fx["guide_timesteps"][0])
    z2, _, _ = eng.transform_guidance(fx["z"], fx["targets"], fx["e"], fx["b"], first, 2)
    assert torch.equal(latents.cpu(), z2.cpu())


def test_direct_guidance_shim(world):
    from distdiff_amd import sampler
    eng, fx, sched = world
    embeds = torch.cat([fx["negative_embeds"], fx["prompt_embeds"]])
    batch = {"targets": fx["targets"]}
    # reference call site :1211-1213
    latents, x_0, score = sampler.direct_guidance(fx["z"].clone(), batch, int(fx["guide_timesteps"][0]), sched, eng, embeds, None, None,
                                                  None, None, torch.float32, None, fx["Pc"], fx["Pg"])
    assert abs(float(score) - float(fx["ref_direct_score"])) < 0.005 * abs(float(fx["ref_direct_score"]))
    assert rel(latents, fx["ref_direct_z_next"]) < 0.03
    assert rel(x_0, fx["ref_direct_x0"]) < 0.04
