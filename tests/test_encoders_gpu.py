"""Parity of the stage before the loop through the production C ABI: dd_vae_encode replaces
`vae.encode(x).latent_dist.sample() * scaling_factor` (dataloader.py:808-809) and dd_text_encode replaces
`text_encoder(input_ids)[0]` (dataloader.py:633-646).

Tolerance (bf16 storage / MFMA inputs, fp32 accumulation, vs the fp32 oracle), relative L2 error: moments, latents and text
embeddings <= 3 %.  The text encoder is additionally checked against tests/golden/clip_fixture.pt, recorded from
transformers' own CLIPTextModel (tests/golden/make_clip_fixture.py).
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
CLIP_FIX = os.path.join(os.path.dirname(__file__), "golden", "clip_fixture.pt")


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    assert torch.isfinite(a).all()
    return ((a - b).norm() / (b.norm() + 1e-20)).item()


def _engine(cfg, w):
    from distdiff_amd.engine import Engine
    return Engine(cfg, w, enable_grad=False, max_guidance_period=1)


@pytest.mark.parametrize("act", ["quick_gelu", "gelu"])
def test_text_encoder_vs_transformers_fixture_and_oracle(hip_lib, act):
    from distdiff_amd.config import tiny_config
    from distdiff_amd.weights import synthetic_weights
    from oracle import sd_oracle as O
    cfg = tiny_config(max_batch=2)
    cfg.text.hidden_act = act
    w = synthetic_weights(cfg, seed=0, num_classes=5, encoders=True)
    case = [c for c in torch.load(CLIP_FIX, weights_only=False)["cases"] if c["hidden_act"] == act][0]
    eng = _engine(cfg, w)
    try:
        ids = case["input_ids"]
        out = eng.text_encode(ids)                       # n = 4 = 2 * max_batch
        assert out.shape == (4, cfg.text_len, cfg.unet.cross_attention_dim)
        assert rel(out, case["last_hidden_state"]) < 0.03
        assert rel(out, O.clip_text_encode(cfg, w["text"], ids)) < 0.03
        # fewer prompts than the built batch: rows are independent, the first n rows must not change
        out1 = eng.text_encode(ids[:1])
        assert torch.equal(out1, out[:1])
        with pytest.raises(RuntimeError):
            eng.text_encode(torch.zeros(5, cfg.text_len, dtype=torch.int32))
    finally:
        eng.close()


def test_vae_encoder_vs_oracle_tiny(hip_lib):
    from distdiff_amd.config import tiny_config
    from distdiff_amd.weights import synthetic_weights
    from oracle import sd_oracle as O
    cfg = tiny_config(max_batch=2)
    w = synthetic_weights(cfg, seed=0, num_classes=5, encoders=True)
    g = torch.Generator().manual_seed(21)
    S = 8 * cfg.latent_size
    x = torch.rand(2, 3, S, S, generator=g) * 2 - 1
    noise = torch.randn(2, 4, cfg.latent_size, cfg.latent_size, generator=g)
    with torch.no_grad():
        lat_ref, mom_ref = O.vae_encode(cfg, w["vae"], x, noise)
        mode_ref, _ = O.vae_encode(cfg, w["vae"], x, None)
    eng = _engine(cfg, w)
    try:
        lat, mom = eng.vae_encode(x, noise, return_moments=True)
        assert rel(mom, mom_ref) < 0.03
        assert rel(lat, lat_ref) < 0.03
        assert rel(eng.vae_encode(x), mode_ref) < 0.03
        # the sample is exactly mean + exp(logvar / 2) * noise of the engine's own moments
        C = cfg.vae.latent_channels
        again = (mom[:, :C] + torch.exp(0.5 * mom[:, C:]) * noise.cuda()) * cfg.vae.scaling_factor
        assert (lat - again).abs().max().item() < 1e-5
        # encode -> decode runs end to end on the same engine (the decoder slab is shared with the encoder)
        img = eng.decode(lat, denormalize=True)
        assert img.shape == (2, 3, S, S) and torch.isfinite(img).all()
        lat2 = eng.vae_encode(x, noise)
        assert torch.equal(lat2, lat)
    finally:
        eng.close()


def test_missing_encoder_weights_fail_loudly(hip_lib):
    from distdiff_amd.config import tiny_config
    from distdiff_amd.weights import synthetic_weights
    cfg = tiny_config(max_batch=2)
    eng = _engine(cfg, synthetic_weights(cfg, seed=0, num_classes=5))
    try:
        with pytest.raises(RuntimeError, match="encoder"):
            eng.vae_encode(torch.zeros(2, 3, 8 * cfg.latent_size, 8 * cfg.latent_size))
        with pytest.raises(RuntimeError, match="text encoder"):
            eng.text_encode(torch.zeros(1, cfg.text_len, dtype=torch.int32))
    finally:
        eng.close()


def test_sd15_shape_encoders_vs_oracle(hip_lib):
    """SD-1.x shapes: CLIP ViT-L/14 text tower on 77 tokens, AutoencoderKL encoder at 256x256 (the size the CPU oracle
    finishes in seconds) -- same architecture widths as the 512x512 script of record."""
    from distdiff_amd.config import sd15_config
    from distdiff_amd.weights import synthetic_guide, synthetic_text_encoder, synthetic_unet, synthetic_vae_decoder, synthetic_vae_encoder
    from oracle import sd_oracle as O
    cfg = sd15_config(latent_size=32, max_batch=1)
    vae = synthetic_vae_decoder(cfg, 0)
    vae.update(synthetic_vae_encoder(cfg, 0))
    w = {"unet": synthetic_unet(cfg, 0), "vae": vae, "guide": synthetic_guide(cfg, 0, 10), "text": synthetic_text_encoder(cfg, 0)}
    g = torch.Generator().manual_seed(22)
    x = torch.rand(1, 3, 256, 256, generator=g) * 2 - 1
    noise = torch.randn(1, 4, 32, 32, generator=g)
    ids = torch.randint(0, cfg.text.vocab_size, (2, 77), generator=g)
    ids[0, 0], ids[0, 6:] = 49406, 49407
    with torch.no_grad():
        lat_ref, mom_ref = O.vae_encode(cfg, w["vae"], x, noise)
        emb_ref = O.clip_text_encode(cfg, w["text"], ids)
    eng = _engine(cfg, w)
    try:
        lat, mom = eng.vae_encode(x, noise, return_moments=True)
        emb = eng.text_encode(ids)
        assert rel(mom, mom_ref) < 0.03
        assert rel(lat, lat_ref) < 0.03
        assert rel(emb, emb_ref) < 0.03
    finally:
        eng.close()
