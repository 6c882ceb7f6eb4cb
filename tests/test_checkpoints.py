"""The real-checkpoint path on the CPU (no GPU call): a Hugging Face-layout model directory and a reference-format guide checkpoint are
written with the real key names, then read back through the loaders the CLI uses (config.from_model_dir, weights.load_safetensors_dir,
model_utils.create_model(weight_path=...)).  Reference behaviour: generate_data.py:863-922 (from_pretrained of scheduler / tokenizer /
text_encoder / vae / unet) and model_utils.py:89-101 (`torch.load(path)['state_dict']`, optional `module.` prefix)."""
import json
import os

import pytest
import torch

from distdiff_amd.config import from_model_dir, tiny_config
from distdiff_amd.weights import load_guide_checkpoint, load_safetensors_dir, normalize_vae_keys, synthetic_weights


def write_model_dir(root, cfg, w, legacy_vae_names=True, scheduler=None):
    """A local `--pretrained_model_name_or_path`: unet/ vae/ text_encoder/ scheduler/ with config.json + safetensors."""
    from safetensors.torch import save_file
    u, v, t = cfg.unet, cfg.vae, cfg.text
    os.makedirs(os.path.join(root, "unet"))
    uc = {"in_channels": u.in_channels, "out_channels": u.out_channels, "block_out_channels": list(u.block_out_channels),
          "layers_per_block": u.layers_per_block, "cross_attention_dim": u.cross_attention_dim, "attention_head_dim": u.num_heads,
          "norm_num_groups": u.norm_num_groups, "norm_eps": u.norm_eps, "freq_shift": 0, "flip_sin_to_cos": True,
          "down_block_types": ["CrossAttnDownBlock2D" if a else "DownBlock2D" for a in u.down_attn],
          "up_block_types": ["CrossAttnUpBlock2D" if a else "UpBlock2D" for a in u.up_attn]}
    if u.transformer_depth:          # the SDXL-base unet/config.json fields
        uc.update({"attention_head_dim": list(u.level_heads), "transformer_layers_per_block": list(u.transformer_depth),
                   "use_linear_projection": True, "addition_embed_type": "text_time", "addition_time_embed_dim": u.add_time_dim,
                   "projection_class_embeddings_input_dim": u.add_text_dim + 6 * u.add_time_dim})
    json.dump(uc, open(os.path.join(root, "unet", "config.json"), "w"))
    save_file({k: x.contiguous() for k, x in w["unet"].items()}, os.path.join(root, "unet", "diffusion_pytorch_model.safetensors"))
    os.makedirs(os.path.join(root, "vae"))
    json.dump({"latent_channels": v.latent_channels, "block_out_channels": list(v.block_out_channels), "layers_per_block": v.layers_per_block,
               "norm_num_groups": v.norm_num_groups, "scaling_factor": v.scaling_factor}, open(os.path.join(root, "vae", "config.json"), "w"))
    vae = {}
    for k, x in w["vae"].items():
        if legacy_vae_names and ".attentions." in k:        # the names the published SD-1.x VAE checkpoints use, 1x1-conv shaped
            for new, old in (("to_q", "query"), ("to_k", "key"), ("to_v", "value"), ("to_out.0", "proj_attn")):
                k = k.replace(".%s." % new, ".%s." % old)
            if k.endswith(".weight") and x.dim() == 2:
                x = x[:, :, None, None]
        vae[k] = x.contiguous()
    save_file(vae, os.path.join(root, "vae", "diffusion_pytorch_model.safetensors"))
    os.makedirs(os.path.join(root, "text_encoder"))
    json.dump({"vocab_size": t.vocab_size, "hidden_size": t.hidden_size, "intermediate_size": t.intermediate_size,
               "num_hidden_layers": t.num_hidden_layers, "num_attention_heads": t.num_attention_heads,
               "max_position_embeddings": t.max_position_embeddings, "hidden_act": t.hidden_act, "layer_norm_eps": t.layer_norm_eps},
              open(os.path.join(root, "text_encoder", "config.json"), "w"))
    txt = {k: x.contiguous() for k, x in w["text"].items()}
    txt["text_model.embeddings.position_ids"] = torch.arange(t.max_position_embeddings)[None].float()    # present in real checkpoints
    save_file(txt, os.path.join(root, "text_encoder", "model.safetensors"))
    if cfg.text2 is not None:       # SDXL layout: text_encoder_2/ (CLIPTextModelWithProjection) + model_index.json
        t2 = cfg.text2
        os.makedirs(os.path.join(root, "text_encoder_2"))
        json.dump({"architectures": ["CLIPTextModelWithProjection"], "vocab_size": t2.vocab_size, "hidden_size": t2.hidden_size,
                   "intermediate_size": t2.intermediate_size, "num_hidden_layers": t2.num_hidden_layers,
                   "num_attention_heads": t2.num_attention_heads, "max_position_embeddings": t2.max_position_embeddings,
                   "hidden_act": t2.hidden_act, "layer_norm_eps": t2.layer_norm_eps, "projection_dim": t2.projection_dim,
                   "eos_token_id": 2}, open(os.path.join(root, "text_encoder_2", "config.json"), "w"))
        save_file({k: x.contiguous() for k, x in w["text2"].items()}, os.path.join(root, "text_encoder_2", "model.safetensors"))
        json.dump({"_class_name": "StableDiffusionXLPipeline", "force_zeros_for_empty_prompt": True},
                  open(os.path.join(root, "model_index.json"), "w"))
    os.makedirs(os.path.join(root, "scheduler"))
    sc = {"num_train_timesteps": 1000, "beta_start": 0.00085, "beta_end": 0.012, "beta_schedule": "scaled_linear", "steps_offset": 1,
          "set_alpha_to_one": False, "clip_sample": False, "prediction_type": "epsilon", "timestep_spacing": "leading"}
    sc.update(scheduler or {})
    json.dump(sc, open(os.path.join(root, "scheduler", "scheduler_config.json"), "w"))


def test_sdxl_model_dir_round_trip(tmp_path):
    """The SDXL-base layout (unet config with per-level heads / transformer depths / text_time conditioning, text_encoder_2/,
    model_index.json) -> from_model_dir reproduces the config the weights were made for."""
    from distdiff_amd.config import tiny_sdxl_config
    cfg = tiny_sdxl_config(max_batch=2)
    w = synthetic_weights(cfg, seed=0, num_classes=5, encoders=True)
    root = str(tmp_path / "sdxl-tiny")
    write_model_dir(root, cfg, w)
    got = from_model_dir(root, cfg.latent_size, 2)
    import dataclasses
    # (num_heads is not read when per-level head counts are given: the mid block takes the last level's)
    assert dataclasses.replace(got.unet, num_heads=cfg.unet.num_heads) == cfg.unet and got.text == cfg.text and got.text2 == cfg.text2
    assert got.text_hidden_layer == -2 and got.force_zeros_for_empty_prompt
    from distdiff_amd.weights import load_safetensors_dir
    sd = load_safetensors_dir(root, "text_encoder_2")
    assert torch.equal(sd["text_projection.weight"], w["text2"]["text_projection.weight"])


def test_model_dir_round_trip(tmp_path):
    cfg = tiny_config(max_batch=2)
    w = synthetic_weights(cfg, seed=0, num_classes=5, encoders=True)
    root = str(tmp_path / "sd-tiny")
    write_model_dir(root, cfg, w)
    got = from_model_dir(root, cfg.latent_size, 2)
    for part in ("unet", "vae", "text", "scheduler"):
        assert getattr(got, part) == getattr(cfg, part), part
    assert got.text_len == cfg.text_len
    for sub, model in (("unet", "unet"), ("vae", "vae"), ("text_encoder", "text")):
        sd = load_safetensors_dir(root, sub)
        sd = {k: v for k, v in sd.items() if "position_ids" not in k}
        assert set(sd) == set(w[model]), (sub, set(sd) ^ set(w[model]))
        assert all(torch.equal(sd[k], w[model][k]) for k in sd), sub       # legacy VAE names renamed, 1x1-conv weights squeezed
    # torch .bin pickles (older repos) are read too
    os.remove(os.path.join(root, "unet", "diffusion_pytorch_model.safetensors"))
    torch.save(w["unet"], os.path.join(root, "unet", "diffusion_pytorch_model.bin"))
    sd = load_safetensors_dir(root, "unet")
    assert all(torch.equal(sd[k], w["unet"][k]) for k in w["unet"])
    os.remove(os.path.join(root, "unet", "diffusion_pytorch_model.bin"))
    with pytest.raises(FileNotFoundError):
        load_safetensors_dir(root, "unet")


def test_vae_key_normalisation_is_idempotent():
    cfg = tiny_config()
    sd = synthetic_weights(cfg, seed=0, num_classes=2, encoders=True)["vae"]
    assert normalize_vae_keys(sd).keys() == sd.keys()
    assert all(torch.equal(normalize_vae_keys(sd)[k], sd[k]) for k in sd)


@pytest.mark.parametrize("bad", [{"prediction_type": "v_prediction"}, {"clip_sample": True}, {"timestep_spacing": "trailing"},
                                 {"beta_schedule": "linear"}])
def test_unsupported_scheduler_configs_are_refused(tmp_path, bad):
    """A v-prediction or clip_sample=true model would run with silently wrong results: from_model_dir refuses it."""
    cfg = tiny_config()
    root = str(tmp_path / "m")
    write_model_dir(root, cfg, synthetic_weights(cfg, seed=0, num_classes=2, encoders=True), scheduler=bad)
    with pytest.raises(NotImplementedError):
        from_model_dir(root)


@pytest.mark.parametrize("arch", ["resnet50", "resnext50", "wideresnet50"])
def test_guide_checkpoint_round_trip(tmp_path, arch):
    """train.py:201-207 writes {'epoch','state_dict','acc','best_acc','optimizer'} from an nn.DataParallel model (`module.` prefix);
    model_utils.py:89-101 strips the prefix.  Shapes are timm's resnet50 / resnext50_32x4d / wide_resnet50_2."""
    from distdiff_amd.config import guide_config, sd15_config
    from distdiff_amd.model_utils import create_model
    from distdiff_amd.weights import synthetic_guide
    cfg = sd15_config()
    cfg.guide = guide_config(arch)
    sd = synthetic_guide(cfg, seed=3, num_classes=7)
    shapes = {"resnet50": ((64, 64, 1, 1), (64, 64, 3, 3)), "resnext50": ((128, 64, 1, 1), (128, 4, 3, 3)),
              "wideresnet50": ((128, 64, 1, 1), (128, 128, 3, 3))}[arch]
    assert tuple(sd["layer1.0.conv1.weight"].shape) == shapes[0] and tuple(sd["layer1.0.conv2.weight"].shape) == shapes[1]
    assert tuple(sd["layer4.2.conv3.weight"].shape)[0] == 2048 and tuple(sd["fc.weight"].shape) == (7, 2048)
    path = str(tmp_path / "model_best.pth.tar")
    ck = {"epoch": 3, "state_dict": {"module." + k: v for k, v in sd.items()}, "acc": 0.5, "best_acc": 0.6, "optimizer": {}}
    ck["state_dict"]["module.bn1.num_batches_tracked"] = torch.tensor(10)
    torch.save(ck, path)
    got = load_guide_checkpoint(path)
    assert all(torch.equal(got[k], sd[k]) for k in sd)
    m = create_model(arch, num_classes=7, weight_path=path)
    assert all(torch.equal(m.state_dict()[k], sd[k]) for k in sd)
    with pytest.raises(NotImplementedError):
        create_model("efficientnet_b0")          # not one of the reference's five architectures (model_utils.py:47-87)


@pytest.mark.parametrize("arch,key,shape,dim", [("mobilenetv2", "blocks.1.0.conv_dw.weight", (96, 1, 3, 3), 1280),
                                                ("open_clip_vit_b32", "visual.transformer.resblocks.11.attn.in_proj_weight", (2304, 768), 512)])
def test_other_guide_families_shapes(arch, key, shape, dim):
    """timm mobilenetv2_100 (model_utils.py:64-71) and the open_clip ViT-B/32 image tower (:80-87): state-dict key names / shapes."""
    from distdiff_amd.config import guide_config
    from distdiff_amd.model_utils import create_model
    m = create_model(arch, num_classes=3)
    assert tuple(m.state_dict()[key].shape) == shape and guide_config(arch).feature_dim == dim


def test_dataset_listings(tmp_path):
    """caltech-101: ./data/caltech-101/train/<category> only (the held-out test/ split is never listed), BACKGROUND_Google and
    Faces_easy dropped, 100 classes, `_` -> ' ' (dataloader.py:272-315, :129); stanford_cars from the devkit .mat files with the year
    moved to the front of the class name, 196 classes ordered by label (dataloader.py:167-228)."""
    import numpy as np
    from scipy import io
    from distdiff_amd.datasets import load_train_listing
    root = tmp_path / "data"
    names = ["cls_%03d" % i for i in range(100)] + ["BACKGROUND_Google", "Faces_easy"]
    for split in ("train", "test"):
        for n in names:
            d = root / "caltech-101" / split / n
            d.mkdir(parents=True)
            for j in range(2):
                (d / ("%s_%d.jpg" % (split, j))).write_bytes(b"x")
    paths, labels, cls = load_train_listing("caltech-101", str(root))
    assert len(cls) == 100 and cls[0] == "cls 000" and len(paths) == 200 and labels == sorted(labels)
    assert all(os.sep + "train" + os.sep in p for p in paths) and not any("BACKGROUND" in p or "Faces_easy" in p for p in paths)
    # stanford_cars
    base = root / "stanford_cars"
    (base / "devkit").mkdir(parents=True)
    meta = np.empty((1, 196), dtype=object)
    for i in range(196):
        meta[0, i] = np.array(["Make%d Model Type %d" % (i, 1990 + i % 20)])
    io.savemat(str(base / "devkit" / "cars_meta.mat"), {"class_names": meta})
    ann = np.zeros((1, 400), dtype=[("fname", object), ("class", object)])
    for j in range(400):
        ann[0, j]["fname"] = np.array(["%05d.jpg" % (j + 1)])
        ann[0, j]["class"] = np.array([[(j * 5) % 196 + 1]])
    io.savemat(str(base / "devkit" / "cars_train_annos.mat"), {"annotations": ann})
    paths, labels, cls = load_train_listing("stanford_cars", str(root))
    assert len(cls) == 196 and cls[0] == "1990 Make0 Model Type" and cls[5] == "1995 Make5 Model Type"
    assert len(paths) == 400 and labels[:3] == [0, 5, 10] and paths[0].endswith(os.path.join("stanford_cars", "cars_train", "00001.jpg"))


def test_sdxl_unet_config_is_read_from_the_model_dir(tmp_path):
    """unet/config.json of an SDXL-style UNet (transformer_layers_per_block, per-level attention_head_dim, addition_embed_type text_time,
    use_linear_projection: BASELINE.json configs[4]) populates the engine config; other addition_embed_types are refused."""
    from distdiff_amd.config import tiny_sdxl_config
    cfg = tiny_sdxl_config(max_batch=2)
    w = synthetic_weights(cfg, seed=0, num_classes=2, encoders=True)
    assert tuple(w["unet"]["down_blocks.1.attentions.0.proj_in.weight"].shape) == (128, 128)              # nn.Linear, not a 1x1 conv
    assert "down_blocks.2.attentions.0.transformer_blocks.2.attn2.to_k.weight" in w["unet"] and "add_embedding.linear_1.weight" in w["unet"]
    assert tuple(w["unet"]["add_embedding.linear_1.weight"].shape) == (256, 24 + 6 * 8)
    root = str(tmp_path / "sdxl-tiny")
    write_model_dir(root, cfg, w)
    u = json.load(open(os.path.join(root, "unet", "config.json")))
    u.update({"attention_head_dim": [2, 2, 4], "transformer_layers_per_block": [1, 2, 3], "addition_embed_type": "text_time",
              "addition_time_embed_dim": 8, "projection_class_embeddings_input_dim": 24 + 6 * 8, "use_linear_projection": True})
    json.dump(u, open(os.path.join(root, "unet", "config.json"), "w"))
    got = from_model_dir(root, cfg.latent_size, 2)
    got.unet.num_heads = cfg.unet.num_heads          # the fallback head count is unused when every level has its own
    assert got.unet == cfg.unet
    u["addition_embed_type"] = "image"
    json.dump(u, open(os.path.join(root, "unet", "config.json"), "w"))
    with pytest.raises(NotImplementedError):
        from_model_dir(root)
